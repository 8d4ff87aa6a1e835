#!/bin/bash
# round 4, step l: golden + fit tests with NA_as_zero_U / _I and the zero-rows rule
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_l; mkdir -p $R/$O; cd $R
timeout -k 10 1200 python -m pytest tests/test_gpu_golden.py tests/test_gpu_fit.py tests/test_abi.py -m gpu -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
