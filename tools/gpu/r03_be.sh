#!/bin/bash
# round 3: the tiny bin on a stream of its own with a share of the wave slots, launched first (CMFREC_HIP_TINY_PCT, experiment)
export TMPDIR=/tmp
O=gpurun_out/r03_be; mkdir -p $O
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], "ms", "A %.3f B %.3f |" % (r["iteration"]["halfstep_ms"]["A"], r["iteration"]["halfstep_ms"]["B"]), " | ".join("%s%s %.3f" % (k["step"], k["kernel"].split("(")[0][:14].strip().replace("cg_rows_",""), k["avg_ms"]) for k in r["per_kernel"]))
PY
}
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-scale-point --steps 10 --warmup 3 > $O/$name.json 2>$O/$name.err; summ $O/$name.json; }
run default X=1
run tiny25 CMFREC_HIP_TINY_PCT=25
run tiny50 CMFREC_HIP_TINY_PCT=50
run tiny12 CMFREC_HIP_TINY_PCT=12
run tiny100 CMFREC_HIP_TINY_PCT=100
run default2 X=1
run tiny25_b CMFREC_HIP_TINY_PCT=25
run par3_tiny25 CMFREC_HIP_TINY_PCT=25 CMFREC_HIP_BINS_PAR=3
