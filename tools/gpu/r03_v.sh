#!/bin/bash
# round 3: low-rank rows gathered once (resident for both sweeps) / prefetched; dense X; the new negative tests
export TMPDIR=/tmp
O=gpurun_out/r03_v; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_config_widths.py tests/test_gpu_golden.py tests/test_gpu_multidevice.py -x -q > $O/pytest_a.log 2>&1; tail -5 $O/pytest_a.log
timeout 900 python -m pytest tests/test_gpu_operators.py -x -q -k "coo_device or lowrank" > $O/pytest_b.log 2>&1; tail -3 $O/pytest_b.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -k "c5" > $O/pytest_c.log 2>&1; tail -3 $O/pytest_c.log
timeout 1200 python bench.py --workload c5shard --no-cpu-baseline --steps 3 --warmup 1 > $O/c5shard.json 2>$O/c5shard.err
python - <<PY
import json
d=json.loads(open("$O/c5shard.json").read().strip().splitlines()[-1]); print("c5shard", d["ms_per_iteration"], d["item_step"], d["user_step_ms"])
PY
R=$GRAFT_REPO_ROOT
cd /tmp; timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_c5 -o c5 -- python $R/bench.py --workload c5shard --no-cpu-baseline --steps 2 --warmup 1 > $R/$O/c5shard_prof.json 2>$R/$O/c5shard_prof.err
cd $R; f=$(find $O/trace_c5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c5shard_kernel_stats.csv && head -12 $f | cut -c1-180
rm -rf $O/trace_c5
