#!/bin/bash
# round 2, step y: where to cut between the 8-wave team kernel and the Gramian path in double precision (C2, C1)
export TMPDIR=/tmp
O=gpurun_out/r02_y; mkdir -p $O
for e in "X=1" "CMFREC_HIP_VH_MIN=513" "CMFREC_HIP_VH_MIN=257"; do
  echo "== c2 $e" | tee -a $O/summary.txt
  env $e timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/c2.json
  python -c "
import json; d=json.load(open('$O/c2.json')); r=d['roofline']; print(d['ms_per_step'], r['frac'], r['iteration']['frac_of_hbm_peak'], r['iteration']['halfstep_ms']); print(' '.join('%s:%s=%.3f' % (e['step'], e['kernel'][:14], e['avg_ms']) for e in r['per_kernel']))" | tee -a $O/summary.txt
done
for e in "X=1" "CMFREC_HIP_VH=gram"; do
echo "== c1 $e" | tee -a $O/summary.txt
env $e timeout 600 python bench.py --no-cpu-baseline --workload c1 --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-200 | tee -a $O/summary.txt
done
