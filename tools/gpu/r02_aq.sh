#!/bin/bash
# round 2, step aq: kernel traces of the side workloads on the final build (c4shard: one GPU's share of config 4, fp32;
# c3: Cholesky + side information; c5shard), `rocprofv3 --kernel-trace --stats`, summaries only.
export TMPDIR=/tmp
O=gpurun_out/r02_aq; mkdir -p $O
R=$GRAFT_REPO_ROOT
for w in c4shard c3 c5shard c1; do
  cd /tmp; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_$w -o $w -- python $R/bench.py --no-cpu-baseline --workload $w --steps 5 --warmup 2 > $R/$O/bench_$w.json 2>$R/$O/bench_$w.err; echo "$w rc=$?"
  cd $R; f=$(find $O/trace_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${w}_kernel_stats.csv && head -8 $f | cut -c1-150
  rm -rf $O/trace_$w
done
