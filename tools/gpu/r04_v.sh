#!/bin/bash
# round 4, step v: plain closed-form rows with few entries on the low-rank kernel (no rotation): parity, then config 3
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_v; mkdir -p $R/$O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_config_widths.py tests/test_gpu_operators.py -m gpu -x -q -k "lowrank or large_k or c3 or c5 or weights_every" > $O/pytest.log 2>&1; tail -12 $O/pytest.log
for lr in 0 x; do
  if [ $lr = 0 ]; then export CMFREC_HIP_LOWRANK=0; else unset CMFREC_HIP_LOWRANK; fi
  echo "LOWRANK=$lr c3: $(timeout -k 10 600 python bench.py --workload c3 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-200)"
done | tee $O/c3.txt
