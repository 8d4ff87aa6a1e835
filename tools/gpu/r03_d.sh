#!/bin/bash
# round 3: bins side by side on 1 .. 6 streams, both kernel generations, split threshold 513 (new default), volatile Gramian reads
export TMPDIR=/tmp
O=gpurun_out/r03_d; mkdir -p $O
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], "ms", " | ".join("%s%s %.3f" % (k["step"], k["kernel"].split("(")[0][:14].strip().replace("cg_rows_",""), k["avg_ms"]) for k in r["per_kernel"]))
PY
}
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/$name.json 2>$O/$name.err; summ $O/$name.json; }
run par1_v1 CMFREC_HIP_CG2=0 CMFREC_HIP_BINS_PAR=1
run par2_v1 CMFREC_HIP_CG2=0 CMFREC_HIP_BINS_PAR=2
run par3_v1 CMFREC_HIP_CG2=0 CMFREC_HIP_BINS_PAR=3
run par4_v1 CMFREC_HIP_CG2=0 CMFREC_HIP_BINS_PAR=4
run par6_v1 CMFREC_HIP_CG2=0 CMFREC_HIP_BINS_PAR=6
run par1_v2 CMFREC_HIP_CG2=1 CMFREC_HIP_BINS_PAR=1
run par3_v2 CMFREC_HIP_CG2=1 CMFREC_HIP_BINS_PAR=3
run par6_v2 CMFREC_HIP_CG2=1 CMFREC_HIP_BINS_PAR=6
run par1_v1_1025 CMFREC_HIP_CG2=0 CMFREC_HIP_BINS_PAR=1 CMFREC_HIP_VH_MIN=1025
for w in c1 c3 c4shard; do
  for par in 1 4; do
    CMFREC_HIP_BINS_PAR=$par timeout 600 python bench.py --no-cpu-baseline --workload $w --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-260
  done
done
CMFREC_HIP_VH_MIN=1025 timeout 600 python bench.py --no-cpu-baseline --workload c1 --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-260
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log | head -1
CMFREC_HIP_BINS_PAR=4 timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_par4.log 2>&1; tail -3 $O/pytest_gpu_par4.log | head -1
