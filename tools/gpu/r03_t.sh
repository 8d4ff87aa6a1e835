#!/bin/bash
# round 3: kernel trace of the c5shard iteration with the four-wavefront consumer
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/r03_t; mkdir -p $R/$O
cd /tmp; timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_c5 -o c5 -- python $R/bench.py --workload c5shard --no-cpu-baseline --steps 2 --warmup 1 > $R/$O/c5shard_prof.json 2>$R/$O/c5shard_prof.err
cd $R; f=$(find $O/trace_c5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c5shard_kernel_stats.csv && head -14 $f | cut -c1-200
g=$(find $O/trace_c5 -name "*kernel_trace.csv" | head -1); [ -n "$g" ] && python - <<PY
import csv
rows=list(csv.DictReader(open("$g")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
# the last iteration: print kernels longer than 0.3 ms with stream/queue
out=open("$O/c5shard_timeline.txt","w")
for r in rows:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
    if d>0.3: out.write(f'{(int(r["Start_Timestamp"])-t0)/1e6:10.2f} +{d:8.2f} ms  q{r.get("Queue_Id","?")}  {r["Kernel_Name"][:90]}\n')
PY
rm -rf $O/trace_c5
tail -c 400 $O/c5shard_prof.json
