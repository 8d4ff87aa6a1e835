#!/bin/bash
# round 4, step zd: single-precision slice kernel of the split rows with its per-slab vector work as packed instructions (B_j . a, the
# scaled operand, the weighted row sum: 22.5 -> 18 vector instructions per slab) on top of step zc: lib_fold = zc, lib_fold2 = this
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_zd; mkdir -p $R/$O; cd $R
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
CMFREC_HIP_LIBDIR=$R/cmfrec_amd/lib_fold2 timeout -k 10 1200 python -m pytest tests/test_gpu_operators.py tests/test_gpu_switches.py tests/test_gpu_golden.py tests/test_gpu_config_widths.py -m gpu -q -x 2>&1 | grep -v "$F" | tail -4 | tee $O/pytest_fold2.log
side() { timeout -k 10 600 python bench.py --workload $1 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], d.get("ms_per_iteration"), d.get("halfstep_ms"))' $1; }
sp() { timeout -k 10 900 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); sp=d["scale_point"]; print("C4", sp["ms_per_step"], [(b["step"], b["bin"], round(b["inline_ms"],2)) for b in sp["per_bin_inline"]])'; }
{
for L in lib_fold lib_fold2 lib_fold lib_fold2; do export CMFREC_HIP_LIBDIR=$R/cmfrec_amd/$L; echo "$L $(side c4shard)"; done
for L in lib_fold lib_fold2; do export CMFREC_HIP_LIBDIR=$R/cmfrec_amd/$L; echo "$L $(sp)"; done
} 2>&1 | tee $O/ab.txt
