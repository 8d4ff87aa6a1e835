#!/bin/bash
# round 2, step d: low-rank path + E-less consumer
export TMPDIR=/tmp
O=gpurun_out/r02_d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -k "widths" > $O/pytest_widths.log 2>&1; echo "pytest widths rc=$?" | tee -a $O/summary.txt
tail -25 $O/pytest_widths.log | tee -a $O/summary.txt
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 -k "not widths" > $O/pytest_rest.log 2>&1; echo "pytest rest rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed" $O/pytest_rest.log | tee -a $O/summary.txt
echo "== c3" | tee -a $O/summary.txt
timeout 600 python bench.py --workload c3 --steps 5 --warmup 2 2>$O/c3.err | tail -1 | tee -a $O/summary.txt
echo "== c5shard" | tee -a $O/summary.txt
timeout 900 python bench.py --workload c5shard --steps 2 --warmup 1 2>$O/c5.err | tail -1 | tee -a $O/summary.txt
tail -3 $O/c5.err | tee -a $O/summary.txt
echo "== c5shard lowrank off" | tee -a $O/summary.txt
CMFREC_HIP_LOWRANK=0 timeout 900 python bench.py --workload c5shard --steps 1 --warmup 1 2>$O/c5off.err | tail -1 | tee -a $O/summary.txt
R=$PWD
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_c5 -- python $R/bench.py --workload c5shard --steps 1 --warmup 1 > $R/$O/prof_c5.log 2>&1
cd $R; find $O/prof_c5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c5_kernel_stats.csv; head -16 $O/c5_kernel_stats.csv | cut -c1-220 | tee -a $O/summary.txt
