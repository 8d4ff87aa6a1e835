#!/bin/bash
# round 2, step w: kernel trace of c4shard with the split rows on the Gramian path
export TMPDIR=/tmp CMFREC_HIP_VH=gram
O=gpurun_out/r02_w; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_c4 -o c4 -- python $R/bench.py --no-cpu-baseline --workload c4shard --steps 5 --warmup 2 > $R/$O/prof_c4.log 2>&1
cd $R; f=$(find $O/prof_c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "gram|vh_" $f | cut -c1-200
