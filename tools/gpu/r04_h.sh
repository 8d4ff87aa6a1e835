#!/bin/bash
# round 4, step h: where the time of the config-5 half-steps goes -- kernel traces of `bench.py --workload c5shard` and of one rank
# at its true share (tools/microbench/c5_rank_of_n.py), as timelines of the last iteration (tools/kernel_timeline.py)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_h; mkdir -p $R/$O
cd /tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_c5shard -o c5 -- python $R/bench.py --workload c5shard --no-cpu-baseline --steps 2 --warmup 1 > $R/$O/c5shard_prof.json 2>$R/$O/c5shard_prof.err
cd $R
f=$(find $O/trace_c5shard -name "*kernel_trace.csv" | head -1)
python tools/kernel_timeline.py $f > $O/c5shard_timeline_all.txt
# the last ~130 ms of dispatches = the last iteration
python - <<PY
lines=[l for l in open("$O/c5shard_timeline_all.txt") if ' ms  q' in l]
last=[i for i,l in enumerate(lines) if 'cmfhip::' in l and 'coo_' not in l][-1]
t_end=float(lines[last].split('+')[0])+5; sel=[l for l in lines if t_end-135 < float(l.split('+')[0]) < t_end]
open("$O/c5shard_timeline_last_iteration.txt","w").writelines(sel)
print(len(lines), len(sel))
PY
cp $(find $O/trace_c5shard -name "*kernel_stats.csv" | head -1) $O/c5shard_kernel_stats.csv
rm -rf $O/trace_c5shard $O/c5shard_timeline_all.txt
tail -1 $O/c5shard_prof.json | cut -c1-600
cd /tmp
timeout -k 10 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_rank -o rank -- env PYTHONPATH=$R python $R/tools/microbench/c5_rank_of_n.py 8 1.0 > $R/$O/c5_rank_prof.txt 2>$R/$O/c5_rank_prof.err
cd $R
f=$(find $O/trace_rank -name "*kernel_trace.csv" | head -1)
python tools/kernel_timeline.py $f > $O/rank_timeline_all.txt
python - <<PY
lines=[l for l in open("$O/rank_timeline_all.txt") if ' ms  q' in l]
# the timed iteration: the last 4 iterations are warm-up, repeat, timed x2 -> take the window of the last ~720 ms of session kernels before the checks
names=[i for i,l in enumerate(lines) if 'lowrank_rows_kernel' in l]
last=names[-1]; t_end=float(lines[last].split('+')[0])+50
sel=[l for l in lines if t_end-760 < float(l.split('+')[0]) < t_end]
open("$O/rank_timeline_last_iteration.txt","w").writelines(sel)
print(len(lines), len(sel))
PY
cp $(find $O/trace_rank -name "*kernel_stats.csv" | head -1) $O/rank_kernel_stats.csv
rm -rf $O/trace_rank $O/rank_timeline_all.txt
grep -v amdgpu $O/c5_rank_prof.txt | tail -2 | cut -c1-700
