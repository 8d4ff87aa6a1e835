#!/bin/bash
# round 3: kernel statistics of the c3 side workload (k = 128 Cholesky + item side information, double precision) on the final build
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/r03_ad; mkdir -p $R/$O
cd /tmp; timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o c3 -- python $R/bench.py --workload c3 --no-cpu-baseline --steps 5 --warmup 1 > $R/$O/c3_prof.json 2>$R/$O/c3_prof.err
cd $R; f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c3_kernel_stats.csv && head -14 $f | cut -c1-200
rm -rf $O/trace; tail -c 500 $O/c3_prof.json
