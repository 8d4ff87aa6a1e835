#!/bin/bash
# round 2, step ai: which clock does the single-precision Gramian kernel run at?  (-DCMF_CG_DEBUG build: ticks per wave, with the
# launch durations of the same run from rocprofv3)
export TMPDIR=/tmp
O=gpurun_out/r02_ai; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp; CMFREC_HIP_GRAM_TICKS=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o c4 -- python $R/bench.py --no-cpu-baseline --workload c4shard --steps 3 --warmup 1 > $R/$O/run.log 2>&1
cd $R; grep "gram_wave:" $O/run.log | tail -4
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r02_ai/prof/**/*kernel_trace.csv', recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'gram_wave' in r['Kernel_Name']]
for r in rows[-4:]: print("gram_wave launch %.3f ms grid %s" % ((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6, r['Grid_Size_X']))
PY
rm -rf $O/prof
