#!/bin/bash
# round 3: consumer builds of the k_t = 257 item step (no gather code): 16 wavefronts x 1 row per CU against 8 wavefronts x 2 rows
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/r03_q; mkdir -p $R/$O
timeout 900 python -m pytest tests/test_gpu_config_widths.py -x -q -k "c5_width" > $O/pytest_c5w.log 2>&1; tail -3 $O/pytest_c5w.log
for cons in 16 8; do
  export CMFREC_HIP_GRAMK_CONS=$cons
  timeout 600 python -m pytest tests/test_gpu_config_widths.py -x -q -k "c5_width and default" > $O/pytest_c5w_$cons.log 2>&1; tail -1 $O/pytest_c5w_$cons.log
  cd /tmp; timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_$cons -o c5 -- python $R/bench.py --workload c5shard --no-cpu-baseline --steps 2 --warmup 1 > $R/$O/c5shard_$cons.json 2>$R/$O/c5shard_$cons.err
  cd $R; f=$(find $O/trace_$cons -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c5shard_${cons}_kernel_stats.csv && head -4 $f | cut -c1-160
  rm -rf $O/trace_$cons
  python - <<PY
import json
d=json.loads(open("$O/c5shard_$cons.json").read().strip().splitlines()[-1]); print("cons $cons", d["ms_per_iteration"], d["item_step"], d["user_step_ms"])
PY
done
