#!/bin/bash
# round 2, step e: low-rank path fixes (eigen kernel on a side stream), 16-wave experiment for 17 tiles
export TMPDIR=/tmp
O=gpurun_out/r02_e; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -k "widths" > $O/pytest_widths.log 2>&1; echo "pytest widths rc=$?" | tee -a $O/summary.txt
tail -12 $O/pytest_widths.log | cut -c1-300 | tee -a $O/summary.txt
echo "== c5shard" | tee -a $O/summary.txt
timeout 900 python bench.py --workload c5shard --steps 2 --warmup 1 2>$O/c5.err | tail -1 | tee -a $O/summary.txt
echo "== c5shard rows17=16 waves" | tee -a $O/summary.txt
CMFREC_HIP_CHOL_ROWS17=1 timeout 900 python bench.py --workload c5shard --steps 2 --warmup 1 2>$O/c5w16.err | tail -1 | tee -a $O/summary.txt
R=$PWD
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_c5 -- python $R/bench.py --workload c5shard --steps 1 --warmup 1 > $R/$O/prof_c5.log 2>&1
cd $R; find $O/prof_c5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c5_kernel_stats.csv; head -10 $O/c5_kernel_stats.csv | cut -c1-200 | tee -a $O/summary.txt
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 -k "not widths" > $O/pytest_rest.log 2>&1; echo "pytest rest rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed" $O/pytest_rest.log | tee -a $O/summary.txt
