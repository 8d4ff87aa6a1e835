#!/bin/bash
# round 2, step b: sliced heavy rows in the wave-per-row Cholesky kernel
export TMPDIR=/tmp
O=gpurun_out/r02_b; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 -k "operators or golden or widths" > $O/pytest_ops.log 2>&1; echo "pytest ops rc=$?" | tee -a $O/summary.txt
tail -15 $O/pytest_ops.log | tee -a $O/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -k "not operators and not golden and not widths" > $O/pytest_rest.log 2>&1; echo "pytest rest rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed" $O/pytest_rest.log | tee -a $O/summary.txt
for v in default noslices; do
  unset CMFREC_HIP_CHOL
  [ $v = noslices ] && export CMFREC_HIP_CHOL=noslices
  echo "== c3 $v" | tee -a $O/summary.txt
  timeout 600 python bench.py --workload c3 --steps 5 --warmup 2 2>$O/c3_$v.err | tail -1 | tee -a $O/summary.txt
done
unset CMFREC_HIP_CHOL
echo "== k50 probe default" | tee -a $O/summary.txt
timeout 900 python tools/microbench/chol_k50_probe.py 2>&1 | tail -4 | tee -a $O/summary.txt
R=$PWD
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_c3 -- python $R/bench.py --workload c3 --steps 3 --warmup 1 > $R/$O/prof_c3.log 2>&1
cd $R; find $O/prof_c3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c3_kernel_stats.csv; head -12 $O/c3_kernel_stats.csv | cut -c1-200 | tee -a $O/summary.txt
