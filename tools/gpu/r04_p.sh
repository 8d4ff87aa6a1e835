#!/bin/bash
# round 4, step p: single-precision tiny kernel with its Gramian in registers (lib_greg float, -DCMF_TINY_GREG_F32=1) against the default;
# the default double build (register Gramian in the tiny kernel) through the operator / golden / poisoned-LDS tests
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_p; mkdir -p $R/$O; cd $R
timeout -k 10 1200 python -m pytest tests/test_gpu_operators.py tests/test_gpu_golden.py tests/test_gpu_poisoned_lds.py tests/test_gpu_switches.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for v in base greg base greg; do
  if [ $v = base ]; then unset CMFREC_HIP_LIBDIR; else export CMFREC_HIP_LIBDIR=$PWD/cmfrec_amd/lib_$v; fi
  echo "$v c4shard $(timeout -k 10 600 python bench.py --workload c4shard --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d.get("ms_per_iteration", d.get("ms_per_step")))')"
done | tee $O/greg_f32.txt
unset CMFREC_HIP_LIBDIR
timeout -k 10 600 python bench.py --workload c2 --no-cpu-baseline --no-scale-point --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("c2", d["ms_per_step"], d["roofline"]["frac"])' | tee $O/c2.txt
