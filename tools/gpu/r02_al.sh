#!/bin/bash
# round 2, step al: the split rows that stayed at their start values (profiles/README.md, end) = gram_cg_kernel multiplying zeros
# of its matrix with LDS words it had not written.  The parity cases with the LDS poisoned in front of every launch, the flake
# hunter the same way, then the whole GPU suite on the fixed build.
export TMPDIR=/tmp
O=gpurun_out/r02_al; mkdir -p $O
H=tools/microbench/heavy_rows_flake.py
timeout -k 10 900 python -m pytest tests/test_gpu_poisoned_lds.py -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | tail -15 | tee $O/pytest_poisoned.log
( CMFREC_HIP_POISON_LDS=1 CMFREC_HIP_VH_GRAM_ASIDE=1 CMFREC_HIP_GRAM_SLICE_LEN=2048 timeout -k 10 300 python $H gram slice 7 0 100 | tail -3
  CMFREC_HIP_POISON_LDS=1 timeout -k 10 300 python $H gram 50 1 100 | tail -3 ) 2>&1 | tee $O/flake_poisoned.log
timeout -k 10 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | tail -4 | tee $O/pytest_gpu.log
timeout -k 10 400 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | cut -c1-400 | tee $O/bench_c2.json
