#!/bin/bash
# round 3: the same sources under other instruction-scheduling strategies of the compiler back end
# (cmfrec_amd/lib_ab: -mllvm -amdgpu-sched-strategy=max-ilp; cmfrec_amd/lib_ac: -mllvm -amdgpu-schedule-metric-bias=0)
export TMPDIR=/tmp
O=gpurun_out/r03_bj; mkdir -p $O
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], "ms", "A %.3f B %.3f |" % (r["iteration"]["halfstep_ms"]["A"], r["iteration"]["halfstep_ms"]["B"]), " | ".join("%s%s %.3f" % (k["step"], k["kernel"].split("(")[0][:14].strip().replace("cg_rows_",""), k["avg_ms"]) for k in r["per_kernel"]))
PY
}
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-scale-point --steps 10 --warmup 3 > $O/$name.json 2>$O/$name.err; summ $O/$name.json; }
run par1_base CMFREC_HIP_BINS_PAR=1
for v in ab ac; do [ -f cmfrec_amd/lib_$v/libcmfrec_hip_double.so ] && run par1_$v CMFREC_HIP_BINS_PAR=1 CMFREC_HIP_LIBDIR=$PWD/cmfrec_amd/lib_$v; done
run default_base X=1
for v in ab ac; do [ -f cmfrec_amd/lib_$v/libcmfrec_hip_double.so ] && run default_$v CMFREC_HIP_LIBDIR=$PWD/cmfrec_amd/lib_$v; done
run default_base2 X=1
