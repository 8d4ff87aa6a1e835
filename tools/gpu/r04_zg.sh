#!/bin/bash
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_zg; mkdir -p $R/$O; cd $R
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
timeout -k 10 600 python -m pytest tests/test_gpu_fit.py -k "l1_after_fit" tests/test_gpu_golden.py -m gpu -q 2>&1 | grep -v "$F" | tail -30 | tee $O/pytest.log
