#!/bin/bash
# round 2, step an: the whole GPU suite with CMFREC_HIP_POISON_LDS=1 now also filling every freshly allocated device buffer with
# all-ones bytes (reads of unwritten global memory), and once without.
export TMPDIR=/tmp
O=gpurun_out/r02_an; mkdir -p $O
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl'
CMFREC_HIP_POISON_LDS=1 timeout -k 10 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | tail -12 | tee $O/pytest_gpu_poisoned.log
timeout -k 10 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | tail -3 | tee $O/pytest_gpu.log
CMFREC_HIP_POISON_LDS=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-300 | tee $O/bench_poisoned.json
