#!/bin/bash
# round 2, step am: the poisoned-LDS parity cases against a single-precision library built from the same sources with
# gram_cg_kernel's two `mul` calls back under their `live ? ... : 0` (the state before the fix): they must fail.
export TMPDIR=/tmp
O=gpurun_out/r02_am; mkdir -p $O
CMFREC_HIP_LIBDIR=$PWD/cmfrec_amd/lib_prefix_r02 timeout -k 10 900 python -m pytest tests/test_gpu_poisoned_lds.py -m gpu -q -k float32 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | grep "passed\|failed\|^FAILED" | tee $O/pytest_poisoned_before_fix.log
CMFREC_HIP_LIBDIR=$PWD/cmfrec_amd/lib_prefix_r02 timeout -k 10 900 python -m pytest tests/test_gpu_operators.py -m gpu -q -k "float32 and very_heavy" 2>&1 | grep "passed\|failed\|^FAILED" | tee $O/pytest_unpoisoned_before_fix.log
