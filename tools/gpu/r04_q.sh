#!/bin/bash
# round 4, step q: explicit CG beyond 64 unknowns through the row's Gramian (gram_cg_wide_kernels.hpp): parity, then the config-3 shape under CG
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_q; mkdir -p $R/$O; cd $R
timeout -k 10 1500 python -m pytest tests/test_gpu_operators.py tests/test_gpu_fit.py tests/test_gpu_sweep.py tests/test_gpu_config_widths.py -m gpu -x -q -k "beyond_64 or observation_weights_every or block_cg or large_k or c3" > $O/pytest.log 2>&1; tail -12 $O/pytest.log
timeout -k 10 900 python tools/microbench/c3_block_cg.py 2>&1 | grep -v amdgpu | tee $O/c3_block_cg.txt
CMFREC_HIP_CG_KERNEL=generic timeout -k 10 900 python tools/microbench/c3_block_cg.py 2>&1 | grep -v amdgpu | head -1 | sed 's/^/generic kernel: /' | tee -a $O/c3_block_cg.txt
