#!/bin/bash
# round 4, step i: eigen-decomposition chain behind the producer / consumer batches + prefetch of the other side's (EigCache):
# parity of the low-rank path, c5shard timing, timeline of the last iteration
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_i; mkdir -p $R/$O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "lowrank or c5 or collective" > $O/pytest_lowrank.log 2>&1; tail -3 $O/pytest_lowrank.log
for i in 1 2; do timeout -k 10 600 python bench.py --workload c5shard --no-cpu-baseline --steps 4 --warmup 2 2>/dev/null | tail -1 | cut -c1-700; done | tee $O/c5shard_bench.jsonl
cd /tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_c5shard -o c5 -- python $R/bench.py --workload c5shard --no-cpu-baseline --steps 2 --warmup 1 > $R/$O/c5shard_prof.json 2>$R/$O/c5shard_prof.err
cd $R
f=$(find $O/trace_c5shard -name "*kernel_trace.csv" | head -1)
python tools/kernel_timeline.py $f > $O/c5shard_timeline_all.txt
python - <<PY
lines=[l for l in open("$O/c5shard_timeline_all.txt") if ' ms  q' in l]
last=[i for i,l in enumerate(lines) if 'cmfhip::' in l and 'coo_' not in l][-1]
t_end=float(lines[last].split('+')[0])+5; sel=[l for l in lines if t_end-135 < float(l.split('+')[0]) < t_end]
open("$O/c5shard_timeline_last_iteration.txt","w").writelines(sel)
print(len(lines), len(sel))
PY
cp $(find $O/trace_c5shard -name "*kernel_stats.csv" | head -1) $O/c5shard_kernel_stats.csv
rm -rf $O/trace_c5shard $O/c5shard_timeline_all.txt
tail -1 $O/c5shard_prof.json | cut -c1-400
