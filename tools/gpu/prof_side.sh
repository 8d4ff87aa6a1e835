#!/bin/bash
# tools/gpu/prof_side.sh <workload> ...: rocprofv3 kernel statistics of bench.py's side workloads (c3, c5shard, ...), one csv each
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/${TAG:-prof_side}; mkdir -p $O
for w in "$@"; do
  cd /tmp; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_$w -o $w -- python $R/bench.py --no-cpu-baseline --workload $w --steps 10 --warmup 3 > $R/$O/bench_$w.json 2>$R/$O/bench_$w.err; echo "trace $w rc=$?"
  cd $R; f=$(find $O/trace_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$w.csv
  rm -rf $O/trace_$w
  tail -1 $O/bench_$w.json | cut -c1-400
  python - <<PY
import csv
rows = list(csv.DictReader(open("$O/kernel_stats_$w.csv")))
for r in rows[:14]:
    print(r["Name"][:110], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us", r["Percentage"])
PY
done
