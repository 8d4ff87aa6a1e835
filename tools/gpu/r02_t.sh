#!/bin/bash
# round 2, step t: c4shard with the split rows forced onto the single-gather (Gramian on the matrix cores) path
export TMPDIR=/tmp
O=gpurun_out/r02_t; mkdir -p $O
for e in "X=1" "CMFREC_HIP_VH=gram" "CMFREC_HIP_VH_MIN=2049" "CMFREC_HIP_VH_MIN=513"; do
  echo "== $e" | tee -a $O/summary.txt
  env $e timeout 900 python bench.py --no-cpu-baseline --workload c4shard --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/tmp.json
  python -c "
import json; d=json.load(open('$O/tmp.json')); b=[v for k,v in d.items() if k.startswith('bins')][0]
print(d['ms_per_iteration'], {k: round(v,3) for k,v in d['halfstep_ms'].items()}, ' '.join('%s=%.3f' % (k, v['ms']) for k, v in b.items()))" | tee -a $O/summary.txt
done
