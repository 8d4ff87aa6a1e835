#!/bin/bash
# round 2, step s: c4shard per-bin breakdown and kernel trace; C2 re-check after the revert of the early gather
export TMPDIR=/tmp
O=gpurun_out/r02_s; mkdir -p $O
timeout 900 python bench.py --no-cpu-baseline --workload c4shard --steps 10 --warmup 3 2>/dev/null | tail -1 | tee $O/c4shard.json | cut -c1-1500
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/c2.json; python -c "
import json; d=json.load(open('$O/c2.json')); print('c2', d['ms_per_step'], d['roofline']['frac'], d['roofline']['iteration']['frac_of_hbm_peak'])"
R=$GRAFT_REPO_ROOT
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_c4 -o c4 -- python $R/bench.py --no-cpu-baseline --workload c4shard --steps 5 --warmup 2 > $R/$O/prof_c4.log 2>&1
cd $R; f=$(find $O/prof_c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 $f | cut -c1-200
