#!/bin/bash
# round 4, step zm: single precision, the two parts of the pass vector through LDS separately: lib_p1 = replicated elements only (no ds_bpermute),
# lib_p2 = Gramian weights only (no broadcast moves); lib = default (cross-lane form in single precision).  c4shard + fp32 parity.
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_zm; mkdir -p $R/$O; cd $R
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
for L in lib_p1 lib_p2; do CMFREC_HIP_LIBDIR=$R/cmfrec_amd/$L timeout -k 10 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_config_widths.py -m gpu -q -x -k float32 2>&1 | grep -v "$F" | tail -2 | tee -a $O/pytest.log; done
side() { timeout -k 10 600 python bench.py --workload $1 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(sys.argv[1], d.get("ms_per_iteration"), d.get("halfstep_ms"))' $1; }
{
for rep in 1 2 3; do for L in lib lib_p1 lib_p2; do export CMFREC_HIP_LIBDIR=$R/cmfrec_amd/$L; echo "$L $(side c4shard)"; done; done
} 2>&1 | tee $O/ab.txt
