#!/bin/bash
# round 2, step ah: durations of the 17-tile Cholesky launches of the c5 shard with the factorisation skipped (debug build)
export TMPDIR=/tmp
O=gpurun_out/r02_ah; mkdir -p $O
R=$GRAFT_REPO_ROOT
for m in 0 4; do
cd /tmp; CMFREC_HIP_CHOL_SKIP=$m timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof$m -o c5 -- python $R/bench.py --no-cpu-baseline --workload c5shard --steps 1 --warmup 1 > $R/$O/prof$m.log 2>&1
cd $R; echo "== skip $m"; python - <<PY
import csv,glob
f=glob.glob('gpurun_out/r02_ah/prof$m/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
t0=min(int(r['Start_Timestamp']) for r in rows)
for r in rows:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
    if d>3.0: print("%9.2f ms  start %9.2f  %s grid %s" % (d, (int(r['Start_Timestamp'])-t0)/1e6, r['Kernel_Name'][:60], r['Grid_Size_X']))
PY
rm -rf $O/prof$m
done
