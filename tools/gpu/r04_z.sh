#!/bin/bash
# round 4, step z: tiny bin as ONE launch, 16-slot tile for rows of <= 16 entries inside the one-row kernel (NE = 0): tests and A/B
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_z; mkdir -p $R/$O; cd $R
timeout -k 10 1500 python -m pytest tests/test_gpu_operators.py tests/test_gpu_switches.py tests/test_gpu_poisoned_lds.py tests/test_gpu_golden.py tests/test_gpu_fit.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
c2() { timeout -k 10 600 python bench.py --workload c2 --no-cpu-baseline --no-scale-point --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print(sys.argv[1], d["ms_per_step"], r["frac"], r["iteration"]["frac_of_hbm_peak"])' "$1"; }
c4() { timeout -k 10 600 python bench.py --workload c4shard --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], d.get("ms_per_iteration"))' "$1"; }
c1() { timeout -k 10 600 python bench.py --workload c1 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], d.get("ms_per_step"), d.get("ms_per_iteration"))' "$1"; }
sp() { timeout -k 10 900 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); sp=d["scale_point"]; print(sys.argv[1], sp["ms_per_step"], [(b["step"], b["bin"], b["inline_ms"]) for b in sp["per_bin_inline"] if b["bin"] in ("tiny", 5, "5")])' "$1"; }
{
for i in 1 2; do c2 "c2 default(mixed)"; CMFREC_HIP_TINY16=0 c2 "c2 tiny16=0"; done
for i in 1 2; do c4 "c4shard default(mixed)"; CMFREC_HIP_TINY16=0 c4 "c4shard ne4"; CMFREC_HIP_TINY2=1 c4 "c4shard tiny2"; done
for i in 1 2; do sp "C4 default(mixed)"; CMFREC_HIP_TINY16=0 sp "C4 ne4"; CMFREC_HIP_TINY2=1 sp "C4 tiny2"; done
for i in 1 2; do c1 "c1 default(mixed)"; CMFREC_HIP_TINY16=0 c1 "c1 ne4"; CMFREC_HIP_TINY2=1 c1 "c1 tiny2"; done
} | tee $O/ab.txt
