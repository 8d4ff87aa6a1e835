#!/bin/bash
# round 3: the split rows' slice kernel (matrix-pipe bound) at one / two wavefronts per SIMD, leaving the other slots to the row
# kernels (vector-ALU bound) for the whole of its run (CMFREC_HIP_GRAM_WGS, experiment)
export TMPDIR=/tmp
O=gpurun_out/r03_bh; mkdir -p $O
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], "ms", "A %.3f B %.3f |" % (r["iteration"]["halfstep_ms"]["A"], r["iteration"]["halfstep_ms"]["B"]), " | ".join("%s%s %.3f" % (k["step"], k["kernel"].split("(")[0][:14].strip().replace("cg_rows_",""), k["avg_ms"]) for k in r["per_kernel"]))
PY
}
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-scale-point --steps 10 --warmup 3 > $O/$name.json 2>$O/$name.err; summ $O/$name.json; }
run default X=1
run wgs1 CMFREC_HIP_GRAM_WGS=1
run wgs2 CMFREC_HIP_GRAM_WGS=2
run wgs3 CMFREC_HIP_GRAM_WGS=3
run par1_wgs1 CMFREC_HIP_GRAM_WGS=1 CMFREC_HIP_BINS_PAR=1
run par1_wgs2 CMFREC_HIP_GRAM_WGS=2 CMFREC_HIP_BINS_PAR=1
run par3_wgs1 CMFREC_HIP_GRAM_WGS=1 CMFREC_HIP_BINS_PAR=3
run default2 X=1
run wgs1_b CMFREC_HIP_GRAM_WGS=1
