#!/bin/bash
# round 2, step af: kernel trace of the c5 shard (which kernels carry its 272 ms item step)
export TMPDIR=/tmp
O=gpurun_out/r02_af; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp; timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o c5 -- python $R/bench.py --no-cpu-baseline --workload c5shard --steps 2 --warmup 1 > $R/$O/prof.log 2>&1
cd $R; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c5shard_kernel_stats.csv && head -14 $f | cut -c1-230
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r02_af/prof/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
# last iteration: print kernels > 1 ms in order
big=[(r['Kernel_Name'][:70], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6, r['Grid_Size_X'], r['Workgroup_Size_X']) for r in rows]
for b in big[-400:]:
    if b[1] > 1.0: print(b)
PY
rm -rf $O/prof
