#!/bin/bash
# round 2, step i: where the time of the second kernel (sum partials, factorise, solve) goes: phase skipping (results invalid)
export TMPDIR=/tmp
O=gpurun_out/r02_i; mkdir -p $O
for sk in 0 1 2 4 8 15 14; do
echo "== c3 skip=$sk" | tee -a $O/summary.txt
CMFREC_HIP_WAVE_SKIP=$sk timeout 600 python bench.py --workload c3 --steps 5 --warmup 2 2>$O/c3_$sk.err | tail -1 | cut -c1-170 | tee -a $O/summary.txt
done
