#!/bin/bash
# round 3: the slice kernel with the weights of a group's slabs computed ahead of its products (lib_split_ilp / lib_split_def:
# ILP-first / default scheduler) against the committed build (slab by slab, ILP-first); bins in line = the kernel's own time
export TMPDIR=/tmp
O=gpurun_out/r03_bm; mkdir -p $O
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], "ms", "A %.3f B %.3f |" % (r["iteration"]["halfstep_ms"]["A"], r["iteration"]["halfstep_ms"]["B"]), " | ".join("%s%s %.3f" % (k["step"], k["kernel"].split("(")[0][:14].strip().replace("cg_rows_",""), k["avg_ms"]) for k in r["per_kernel"][:1]))
PY
}
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-scale-point --steps 10 --warmup 3 > $O/$name.json 2>$O/$name.err; summ $O/$name.json; }
for v in split_ilp split_def; do CMFREC_HIP_LIBDIR=$PWD/cmfrec_amd/lib_$v timeout 600 python -m pytest tests/test_gpu_operators.py -x -q -k "heavy or implicit" 2>&1 | tail -1; done
run par1_base CMFREC_HIP_BINS_PAR=1
for v in split_ilp split_def; do run par1_$v CMFREC_HIP_BINS_PAR=1 CMFREC_HIP_LIBDIR=$PWD/cmfrec_amd/lib_$v; done
run default_base X=1
for v in split_ilp split_def; do run default_$v CMFREC_HIP_LIBDIR=$PWD/cmfrec_amd/lib_$v; done
