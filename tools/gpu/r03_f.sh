#!/bin/bash
# round 3: balanced bin-to-stream assignment, unrolled BtB kernel; the two-rank HIP-session test; the full default bench line
export TMPDIR=/tmp
O=gpurun_out/r03_f; mkdir -p $O
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], "ms", "frac", r["frac"], "|", " | ".join("%s%s %.3f" % (k["step"], k["kernel"].split("(")[0][:14].strip().replace("cg_rows_",""), k["avg_ms"]) for k in r["per_kernel"]))
PY
}
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-scale-point --steps 20 --warmup 5 > $O/$name.json 2>$O/$name.err; summ $O/$name.json; }
timeout 600 python -m pytest tests/test_gpu_two_ranks.py -x -q > $O/pytest_two_ranks.log 2>&1; tail -5 $O/pytest_two_ranks.log
run default X=1
run default_b X=1
run par1 CMFREC_HIP_BINS_PAR=1
run par3 CMFREC_HIP_BINS_PAR=3
python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err; tail -c 1500 $O/bench_full.json
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log
