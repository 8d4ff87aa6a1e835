#!/bin/bash
# round 3: low-rank rows -- vector moves in the full groups, four wavefronts per SIMD for the shortest rows
export TMPDIR=/tmp
O=gpurun_out/r03_ab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_config_widths.py -x -q -k "c5 or lowrank" > $O/pytest_a.log 2>&1; tail -2 $O/pytest_a.log
R=$GRAFT_REPO_ROOT
cd /tmp; timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_c5 -o c5 -- python $R/bench.py --workload c5shard --no-cpu-baseline --steps 2 --warmup 1 > $R/$O/c5shard_prof.json 2>$R/$O/c5shard_prof.err
cd $R; f=$(find $O/trace_c5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c5shard_kernel_stats.csv && head -6 $f | cut -c1-180
rm -rf $O/trace_c5
python - <<PY
import json
d=json.loads(open("$O/c5shard_prof.json").read().strip().splitlines()[-1]); print("c5shard (under rocprof)", d["ms_per_iteration"], d["item_step"], d["user_step_ms"])
PY
