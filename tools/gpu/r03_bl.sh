#!/bin/bash
# round 3: the slice kernel's translation unit under the back end's other scheduling strategies (in-line bins: the kernel's own time)
export TMPDIR=/tmp
O=gpurun_out/r03_bl; mkdir -p $O
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], "ms", "A %.3f B %.3f |" % (r["iteration"]["halfstep_ms"]["A"], r["iteration"]["halfstep_ms"]["B"]), " | ".join("%s%s %.3f" % (k["step"], k["kernel"].split("(")[0][:14].strip().replace("cg_rows_",""), k["avg_ms"]) for k in r["per_kernel"][:1]))
PY
}
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-scale-point --steps 10 --warmup 3 > $O/$name.json 2>$O/$name.err; summ $O/$name.json; }
run par1_max-ilp CMFREC_HIP_BINS_PAR=1
for v in iterative-ilp max-memory-clause iterative-minreg; do run par1_$v CMFREC_HIP_BINS_PAR=1 CMFREC_HIP_LIBDIR=$PWD/cmfrec_amd/lib_$v; done
run par1_max-ilp2 CMFREC_HIP_BINS_PAR=1
