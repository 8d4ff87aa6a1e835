#!/bin/bash
# round 3: whole GPU suite after the switch clean-up, 1 / D table of the low-rank kernel, more wavefronts per SIMD for its longer rows
export TMPDIR=/tmp
O=gpurun_out/r03_y; mkdir -p $O
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl'
timeout -k 10 1800 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "$F" | tail -6 | tee $O/pytest_gpu.log
R=$GRAFT_REPO_ROOT
cd /tmp; timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_c5 -o c5 -- python $R/bench.py --workload c5shard --no-cpu-baseline --steps 2 --warmup 1 > $R/$O/c5shard_prof.json 2>$R/$O/c5shard_prof.err
cd $R; f=$(find $O/trace_c5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c5shard_kernel_stats.csv && head -9 $f | cut -c1-180
rm -rf $O/trace_c5
python - <<PY
import json
d=json.loads(open("$O/c5shard_prof.json").read().strip().splitlines()[-1]); print("c5shard (under rocprof)", d["ms_per_iteration"], d["item_step"], d["user_step_ms"])
PY
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('C2', d['ms_per_step'], r['frac'], r['traffic'], r['iteration'])"
