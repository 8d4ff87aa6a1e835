#!/bin/bash
# round 4, step y: implicit rows of <= 16 entries on the one-row kernel by default: operator / switch / poisoned tests, C2 and c4shard
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_y; mkdir -p $R/$O; cd $R
timeout -k 10 1200 python -m pytest tests/test_gpu_operators.py tests/test_gpu_switches.py tests/test_gpu_poisoned_lds.py tests/test_gpu_golden.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2; do
timeout -k 10 600 python bench.py --workload c2 --no-cpu-baseline --no-scale-point --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("c2", d["ms_per_step"], r["frac"], r["iteration"]["frac_of_hbm_peak"])'
done | tee $O/c2.txt
timeout -k 10 600 python bench.py --workload c4shard --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("c4shard", d.get("ms_per_iteration"))' | tee -a $O/c2.txt
