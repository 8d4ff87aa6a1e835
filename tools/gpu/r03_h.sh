#!/bin/bash
# round 3: how the register-tiled bins respond to the number of resident teams (bins in line, 100 / 75 / 50 / 25 % of the resident
# grid): latency- or bandwidth-bound?  Plus the shard-size test of configuration 5.
export TMPDIR=/tmp
O=gpurun_out/r03_h; mkdir -p $O
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], "ms", " | ".join("%s%s %.3f" % (k["step"], k["kernel"].split("(")[0][:14].strip().replace("cg_rows_",""), k["avg_ms"]) for k in r["per_kernel"]))
PY
}
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-scale-point --steps 10 --warmup 3 > $O/$name.json 2>$O/$name.err; summ $O/$name.json; }
for pct in 100 75 50 25; do run grid$pct CMFREC_HIP_BINS_PAR=1 CMFREC_HIP_CG_GRID_PCT=$pct; done
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -k c5 > $O/pytest_c5.log 2>&1; tail -5 $O/pytest_c5.log
