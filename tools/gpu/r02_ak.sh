#!/bin/bash
# round 2, step ak: is this box one on which the heavy-row case fails (profiles/README.md, end)?  If so, which knob makes it stop.
export TMPDIR=/tmp
O=gpurun_out/r02_ak; mkdir -p $O
H=tools/microbench/heavy_rows_flake.py
L=$O/flake_$(date +%H%M%S).log
run() { echo "== $*" >> $L; env "$@" timeout -k 10 200 python $H gram slice 7 0 150 2>&1 | tail -7 >> $L; }
run CMFREC_HIP_VH_GRAM_ASIDE=1 CMFREC_HIP_GRAM_SLICE_LEN=2048
if grep -q "iteration" $L; then
  run CMFREC_HIP_GRAM_SLICE_LEN=2048
  run CMFREC_HIP_VH_GRAM_ASIDE=1
  run CMFREC_HIP_VH_GRAM_ASIDE=1 CMFREC_HIP_GRAM_SLICE_LEN=2048 AMD_SERIALIZE_KERNEL=3
  run CMFREC_HIP_VH_GRAM_ASIDE=1 CMFREC_HIP_GRAM_SLICE_LEN=2048 GPU_MAX_HW_QUEUES=1
  run CMFREC_HIP_VH_GRAM_ASIDE=1 CMFREC_HIP_GRAM_SLICE_LEN=2048 CMFREC_HIP_CG_KERNEL=generic
  echo "== wave kernel, second stream" >> $L; CMFREC_HIP_VH_GRAM_ASIDE=1 CMFREC_HIP_GRAM_SLICE_LEN=2048 timeout -k 10 200 python $H gram 7 0 150 2>&1 | tail -7 >> $L
  echo "== streaming path, second stream" >> $L; timeout -k 10 200 python $H stream 7 0 150 2>&1 | tail -3 >> $L
  echo "== wave kernel k=50 implicit, second stream" >> $L; CMFREC_HIP_VH_GRAM_ASIDE=1 CMFREC_HIP_GRAM_SLICE_LEN=2048 timeout -k 10 200 python $H gram 50 1 150 2>&1 | tail -3 >> $L
fi
cat $L
