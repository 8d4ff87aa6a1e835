#!/bin/bash
# round 4, step zl: double precision: the entry weights of a tile through LDS as well (-DCMF_TILE_W_LDS=1: one store + eight broadcast reads instead
# of sixteen DPP moves, the Gramian product issued between store and reads): lib = default (pass vector through LDS), lib_pw = this
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_zl; mkdir -p $R/$O; cd $R
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
CMFREC_HIP_LIBDIR=$R/cmfrec_amd/lib_pw timeout -k 10 1200 python -m pytest tests/test_gpu_operators.py tests/test_gpu_switches.py tests/test_gpu_golden.py tests/test_gpu_config_widths.py -m gpu -q -x 2>&1 | grep -v "$F" | tail -4 | tee $O/pytest_pw.log
c2() { timeout -k 10 600 python bench.py --workload c2 --no-cpu-baseline --no-scale-point --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("c2", d["ms_per_step"], [(e["step"], round(e.get("inline_ms"),3)) for e in r["per_kernel"]])'; }
side() { timeout -k 10 600 python bench.py --workload $1 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], d.get("ms_per_iteration"), d.get("halfstep_ms"))' $1; }
{
for rep in 1 2 3; do for L in lib lib_pw; do export CMFREC_HIP_LIBDIR=$R/cmfrec_amd/$L; echo "$L $(c2)"; done; done
for L in lib lib_pw; do export CMFREC_HIP_LIBDIR=$R/cmfrec_amd/$L; echo "$L $(side c1)"; done
} 2>&1 | tee $O/ab.txt
