#!/bin/bash
# round 4, step zc: full-mask DPP permutations in the bound_ctrl form the compiler folds into the consuming 32-bit add (v_add_f32_dpp), and
# the xor-4 stage of the quad-uniform sums as one mirror move (lanes.hpp qxor4): lib = before, lib_fold = after.  Parity first.
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_zc; mkdir -p $R/$O; cd $R
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
CMFREC_HIP_LIBDIR=$R/cmfrec_amd/lib_fold timeout -k 10 1200 python -m pytest tests/test_gpu_operators.py tests/test_gpu_switches.py tests/test_gpu_golden.py tests/test_gpu_config_widths.py -m gpu -q -x 2>&1 | grep -v "$F" | tail -4 | tee $O/pytest_fold.log
c2() { timeout -k 10 600 python bench.py --workload c2 --no-cpu-baseline --no-scale-point --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("c2", d["ms_per_step"], [(e["step"], round(e.get("inline_ms"),3)) for e in r["per_kernel"]])'; }
side() { timeout -k 10 600 python bench.py --workload $1 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], d.get("ms_per_iteration"), d.get("halfstep_ms"))' $1; }
sp() { timeout -k 10 900 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); sp=d["scale_point"]; print("C4", sp["ms_per_step"], [(b["step"], b["bin"], round(b["inline_ms"],2)) for b in sp["per_bin_inline"]])'; }
{
for rep in 1 2; do for L in lib lib_fold; do export CMFREC_HIP_LIBDIR=$R/cmfrec_amd/$L; echo "$L $(c2)"; done; done
for L in lib lib_fold lib lib_fold; do export CMFREC_HIP_LIBDIR=$R/cmfrec_amd/$L; echo "$L $(side c4shard)"; done
for L in lib lib_fold; do export CMFREC_HIP_LIBDIR=$R/cmfrec_amd/$L; echo "$L $(sp)"; done
for L in lib lib_fold; do export CMFREC_HIP_LIBDIR=$R/cmfrec_amd/$L; echo "$L $(side c5shard)"; echo "$L $(side c3)"; echo "$L $(side c1)"; done
} 2>&1 | tee $O/ab.txt
