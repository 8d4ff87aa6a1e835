#!/bin/bash
# round 2, step g: rocSOLVER-backed eigen step (with the Jacobi kernel as cross-check), consumer load batches, batch sizes
export TMPDIR=/tmp
O=gpurun_out/r02_g; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -k "widths" > $O/pytest_w1.log 2>&1; echo "pytest widths (syevd) rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_w1.log | cut -c1-300 | tee -a $O/summary.txt
CMFREC_HIP_EIG=jacobi timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -k "lowrank" > $O/pytest_w2.log 2>&1; echo "pytest lowrank (jacobi) rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_w2.log | cut -c1-300 | tee -a $O/summary.txt
echo "== c5shard" | tee -a $O/summary.txt
timeout 900 python bench.py --workload c5shard --steps 2 --warmup 1 2>$O/c5.err | tail -1 | tee -a $O/summary.txt
for b in 32768 8192 131072; do
echo "== c3 batch $b" | tee -a $O/summary.txt
CMFREC_HIP_CHOL_BATCH=$b timeout 600 python bench.py --workload c3 --steps 5 --warmup 2 2>$O/c3_$b.err | tail -1 | cut -c1-200 | tee -a $O/summary.txt
done
R=$PWD
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_c5 -- python $R/bench.py --workload c5shard --steps 1 --warmup 1 > $R/$O/prof_c5.log 2>&1
cd $R; find $O/prof_c5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c5_kernel_stats.csv; head -12 $O/c5_kernel_stats.csv | cut -c1-200 | tee -a $O/summary.txt
