#!/bin/bash
# round 4, step e: (1) the collective model behind CMFREC_HIP_DEVICES and the one-shard RCCL transport (tests), (2) one rank of
# BASELINE config 5 at its true share on one GPU (tools/microbench/c5_rank_of_n.py): smoke at 2 %, then the real size
export TMPDIR=/tmp
O=gpurun_out/r04_e; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_multidevice.py tests/test_gpu_c_caller.py -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest_multidevice.log
timeout -k 10 600 python tools/microbench/c5_rank_of_n.py 8 0.02 2>&1 | tail -5 | tee $O/c5_rank_smoke.txt
timeout -k 10 1500 python tools/microbench/c5_rank_of_n.py 8 1.0 2>&1 | tail -5 | tee $O/c5_rank_of_8.txt
