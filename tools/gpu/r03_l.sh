#!/bin/bash
# round 3: observation weights -- the new GPU tests, the explicit side workloads (c1, c3) before / after the weight plumbing
export TMPDIR=/tmp
O=gpurun_out/r03_l; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_operators.py tests/test_gpu_golden.py -x -q -k "observation_weights or weights" > $O/pytest_weights.log 2>&1; tail -15 $O/pytest_weights.log
for w in c1 c3; do
  python bench.py --workload $w --no-cpu-baseline --steps 5 --warmup 2 > $O/$w.json 2>$O/$w.err; tail -c 600 $O/$w.json; echo
done
python bench.py --no-cpu-baseline --no-scale-point --steps 10 --warmup 3 > $O/c2.json 2>$O/c2.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_l/c2.json").read().strip().splitlines()[-1]); print("c2", d["ms_per_step"], d["roofline"]["frac"])
PY
