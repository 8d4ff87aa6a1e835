#!/bin/bash
# round 4, step zh: single precision: the two quotients of a CG step as numerator x v_rcp_f32 (cg_div) instead of the IEEE division sequence
# (tiny kernel 851 -> 801 vector instructions, no scratch any more).  Whole GPU suite, then config 4 / c4shard against lib_fold (= step zc).
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_zh; mkdir -p $R/$O; cd $R
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
timeout -k 10 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | tail -8 | tee $O/pytest_gpu.log
side() { timeout -k 10 600 python bench.py --workload $1 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], d.get("ms_per_iteration"), d.get("halfstep_ms"))' $1; }
sp() { timeout -k 10 900 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); sp=d["scale_point"]; print("C4", sp["ms_per_step"], [(b["step"], b["bin"], round(b["inline_ms"],2)) for b in sp["per_bin_inline"]])'; }
{
for L in lib_fold lib lib_fold lib; do export CMFREC_HIP_LIBDIR=$R/cmfrec_amd/$L; echo "$L $(sp)"; done
for L in lib_fold lib lib_fold lib; do export CMFREC_HIP_LIBDIR=$R/cmfrec_amd/$L; echo "$L $(side c4shard)"; done
for L in lib_fold lib; do export CMFREC_HIP_LIBDIR=$R/cmfrec_amd/$L; echo "$L $(side c5shard)"; done
} 2>&1 | tee $O/ab.txt
