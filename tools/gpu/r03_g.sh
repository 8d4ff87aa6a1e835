#!/bin/bash
# round 3: calibrated stream balance; slice length of the split rows' Gramian path; tiny2 on / off
export TMPDIR=/tmp
O=gpurun_out/r03_g; mkdir -p $O
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], "ms", "frac", r["frac"], r["iteration"]["halfstep_ms"])
PY
}
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-scale-point --steps 20 --warmup 5 > $O/$name.json 2>$O/$name.err; summ $O/$name.json; }
run default X=1
run default_b X=1
run slice1024 CMFREC_HIP_GRAM_SLICE_LEN=1024
run slice512 CMFREC_HIP_GRAM_SLICE_LEN=512
run tiny2off CMFREC_HIP_TINY2=0
run par1 CMFREC_HIP_BINS_PAR=1
run par1_slice1024 CMFREC_HIP_BINS_PAR=1 CMFREC_HIP_GRAM_SLICE_LEN=1024
