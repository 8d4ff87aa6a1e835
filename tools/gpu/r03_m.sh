#!/bin/bash
# round 3: observation weights -- the new GPU tests
export TMPDIR=/tmp
O=gpurun_out/r03_m; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_operators.py tests/test_gpu_golden.py -q -k "observation_weights or weights" > $O/pytest_weights.log 2>&1; tail -40 $O/pytest_weights.log
