#!/bin/bash
# round 3, first A/B of the second-generation CG kernels: GPU suite on the new default, bench with the first / second generation,
# instruction-cost microbenchmark
export TMPDIR=/tmp
O=gpurun_out/r03_b; mkdir -p $O
R=$GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_costs tools/microbench/valu_costs.hip 2>/dev/null && /tmp/valu_costs > $O/valu_costs.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print(sys.argv[1], d["ms_per_step"], "ms", " | ".join("%s %s %.3f" % (k["step"], k["kernel"].split("(")[0][:28].strip(), k["avg_ms"]) for k in r["per_kernel"]))
PY
}
for v in 0 1; do
  CMFREC_HIP_CG2=$v python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_cg2_$v.json 2>$O/bench_cg2_$v.err; summ $O/bench_cg2_$v.json
done
CMFREC_HIP_CG2=1 CMFREC_HIP_CG2_TINY=all python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_cg2_tinyall.json 2>$O/bench_cg2_tinyall.err; summ $O/bench_cg2_tinyall.json
for v in 0 1; do
  CMFREC_HIP_CG2=$v python bench.py --no-cpu-baseline --workload c4shard --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/bench_c4shard_cg2_$v.json; cut -c1-400 $O/bench_c4shard_cg2_$v.json
done
