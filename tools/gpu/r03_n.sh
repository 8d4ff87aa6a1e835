#!/bin/bash
# round 3: NA_as_zero_X and adjust_weight tests, then the whole GPU suite on the build with weights + NA_as_zero
export TMPDIR=/tmp
O=gpurun_out/r03_n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_fit.py -q -k "NA_as_zero or adjust_weight or observation" > $O/pytest_new.log 2>&1; tail -25 $O/pytest_new.log
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
