#!/bin/bash
# round 2, step ap: the GPU suite under the stream arrangements that the default single-GPU run does not take -- the Gramian
# path of few split rows on the second stream (CMFREC_HIP_VH_GRAM_ASIDE=1) and the bins of a launch alternating between two
# streams as the parts of a multi-GPU block do (CMFREC_HIP_BINS_ALT=1) -- each with the poison hook on.
export TMPDIR=/tmp
O=gpurun_out/r02_ap; mkdir -p $O
F='passed\|failed\|^FAILED'
CMFREC_HIP_POISON_LDS=1 CMFREC_HIP_VH_GRAM_ASIDE=1 timeout -k 10 1200 python -m pytest tests -m gpu -q 2>&1 | grep "$F" | tee $O/pytest_gram_aside.log
CMFREC_HIP_POISON_LDS=1 CMFREC_HIP_BINS_ALT=1 timeout -k 10 1200 python -m pytest tests -m gpu -q 2>&1 | grep "$F" | tee $O/pytest_bins_alt.log
CMFREC_HIP_BINS_ALT=1 CMFREC_HIP_VH_GRAM_ASIDE=1 timeout -k 10 1200 python -m pytest tests -m gpu -q 2>&1 | grep "$F" | tee $O/pytest_both_unpoisoned.log
