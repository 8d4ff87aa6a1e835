#!/bin/bash
# tools/gpu/pair32.sh: single precision, the <= 32-entry bin one row / two rows per wavefront (CMFREC_HIP_PAIR), c4shard alternating
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/${TAG:-pair32}; mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --no-scale-point"
for rep in 1 2 3; do for x in 0 1; do
  CMFREC_HIP_PAIR=$x $B --workload c4shard --steps 20 --warmup 3 2>/dev/null | python -c "
import sys, json
l = [x for x in sys.stdin if x.startswith('{')]
d = json.loads(l[-1]) if l else {}
key = [k for k in d if k.startswith('bins')]
print('c4shard pair=$x', d.get('ms_per_iteration'), d.get('halfstep_ms'), {k: v['ms'] for k, v in d[key[0]].items()} if key else None)" | tee -a $O/lines.txt
done; done
