#!/bin/bash
# round 3: kernel trace of the c5shard iteration with the producer / consumer item step
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/r03_p; mkdir -p $R/$O
cd /tmp; timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_c5 -o c5 -- python $R/bench.py --workload c5shard --no-cpu-baseline --steps 2 --warmup 1 > $R/$O/c5shard_prof.json 2>$R/$O/c5shard_prof.err
cd $R; f=$(find $O/trace_c5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c5shard_kernel_stats.csv && head -14 $f | cut -c1-200
rm -rf $O/trace_c5
tail -c 400 $O/c5shard_prof.json
