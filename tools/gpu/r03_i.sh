#!/bin/bash
# round 3: the 33..64 bin with the next row's tile prefetched into LDS by DMA (second generation, one 8-wave workgroup per CU)
export TMPDIR=/tmp
O=gpurun_out/r03_i; mkdir -p $O
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], "ms", " | ".join("%s%s %.3f" % (k["step"], k["kernel"].split("(")[0][:14].strip().replace("cg_rows_",""), k["avg_ms"]) for k in r["per_kernel"]))
PY
}
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-scale-point --steps 10 --warmup 3 > $O/$name.json 2>$O/$name.err; summ $O/$name.json; }
timeout 600 python -m pytest tests/test_gpu_cg2.py -x -q > $O/pytest_cg2.log 2>&1; tail -3 $O/pytest_cg2.log
run par1 CMFREC_HIP_BINS_PAR=1
run par1_pf CMFREC_HIP_BINS_PAR=1 CMFREC_HIP_CG2_PF=1
run par1_cg2 CMFREC_HIP_BINS_PAR=1 CMFREC_HIP_CG2=1
run default X=1
run default_pf CMFREC_HIP_CG2_PF=1
