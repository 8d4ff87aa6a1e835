#!/bin/bash
# round 2, step j: grouped loads of the partials in the second Cholesky kernel
export TMPDIR=/tmp
O=gpurun_out/r02_j; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -k "widths or operators or golden or fullsize" > $O/pytest1.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest1.log | cut -c1-300 | tee -a $O/summary.txt
for sk in 0 1; do
echo "== c3 skip=$sk" | tee -a $O/summary.txt
CMFREC_HIP_WAVE_SKIP=$sk timeout 600 python bench.py --workload c3 --steps 5 --warmup 2 2>$O/c3_$sk.err | tail -1 | cut -c1-170 | tee -a $O/summary.txt
done
echo "== k50 probe" | tee -a $O/summary.txt
timeout 900 python tools/microbench/chol_k50_probe.py 2>&1 | tail -4 | tee -a $O/summary.txt
echo "== c5shard" | tee -a $O/summary.txt
timeout 900 python bench.py --workload c5shard --steps 2 --warmup 1 2>$O/c5.err | tail -1 | cut -c1-330 | tee -a $O/summary.txt
