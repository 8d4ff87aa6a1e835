#!/bin/bash
# round 4, step zf: the build after the packed slice kernel was taken out again (no gain once its wait states were in place): whole GPU
# suite incl. the L1 new-rows tests (g22 + after-fit), smoke
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_zf; mkdir -p $R/$O; cd $R
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
timeout -k 10 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | tail -8 | tee $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -1 | tee $O/smoke.log
