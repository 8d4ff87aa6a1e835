#!/bin/bash
# round 3: knobs on the build of r03_b -- split-row threshold with the leaner double-precision Gramian kernel, bins alternating
# between two streams, first-generation kernels throughout
export TMPDIR=/tmp
O=gpurun_out/r03_c; mkdir -p $O
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], "ms", " | ".join("%s%s %.3f" % (k["step"], k["kernel"].split("(")[0][:14].strip().replace("cg_rows_",""), k["avg_ms"]) for k in r["per_kernel"]))
PY
}
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/$name.json 2>$O/$name.err; summ $O/$name.json; }
run base CMFREC_HIP_CG2=0
run vh513 CMFREC_HIP_CG2=0 CMFREC_HIP_VH_MIN=513
run vh385 CMFREC_HIP_CG2=0 CMFREC_HIP_VH_MIN=385
run vh769 CMFREC_HIP_CG2=0 CMFREC_HIP_VH_MIN=769
run alt CMFREC_HIP_CG2=0 CMFREC_HIP_BINS_ALT=1
run alt513 CMFREC_HIP_CG2=0 CMFREC_HIP_BINS_ALT=1 CMFREC_HIP_VH_MIN=513
run remv0 CMFREC_HIP_CG2=0 CMFREC_HIP_GRAM_REMV=0
run cg2_513 CMFREC_HIP_CG2=1 CMFREC_HIP_VH_MIN=513
