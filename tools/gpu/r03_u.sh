#!/bin/bash
# round 3: consumer of a batch on the second stream beside the producer of the next (CMFREC_HIP_GRAMK_PIPE), c5 parity cases
export TMPDIR=/tmp
O=gpurun_out/r03_u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_config_widths.py -x -q -k "c5" > $O/pytest_c5w.log 2>&1; tail -3 $O/pytest_c5w.log
for mode in pipe inline; do
  if [ $mode = inline ]; then export CMFREC_HIP_GRAMK_PIPE=0; else unset CMFREC_HIP_GRAMK_PIPE; fi
  timeout 1200 python bench.py --workload c5shard --no-cpu-baseline --steps 3 --warmup 1 > $O/c5shard_$mode.json 2>$O/c5shard_$mode.err
  python - <<PY
import json
d=json.loads(open("$O/c5shard_$mode.json").read().strip().splitlines()[-1]); print("$mode", d["ms_per_iteration"], d["item_step"], d["user_step_ms"])
PY
done
