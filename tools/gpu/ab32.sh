#!/bin/bash
# tools/gpu/ab32.sh <libdirA> <libdirB> ...: the single-precision workloads (c4shard twice, config 4 per bin) on several builds, one box
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/${TAG:-ab32}; mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --no-scale-point"
for rep in 1 2; do for L in "$@"; do
  CMFREC_HIP_LIBDIR=$R/cmfrec_amd/$L $B --workload c4shard --steps 20 --warmup 3 2>/dev/null | python -c "
import sys, json
l = [x for x in sys.stdin if x.startswith('{')]
d = json.loads(l[-1]) if l else {}
key = [k for k in d if k.startswith('bins')]
print('c4shard $L', d.get('ms_per_iteration'), d.get('halfstep_ms'), {k: v['ms'] for k, v in d[key[0]].items()} if key else None)" | tee -a $O/lines.txt
done; done
for L in "$@"; do
  CMFREC_HIP_LIBDIR=$R/cmfrec_amd/$L python $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
l = [x for x in sys.stdin if x.startswith('{')]
d = json.loads(l[-1]) if l else {}
sp = d.get('scale_point') or {}
print('c2 $L', d.get('ms_per_step'), '| c4', sp.get('ms_per_step'), (sp.get('inline') or {}).get('halfstep_ms'), [(e['step'], e['bin'], e['inline_ms']) for e in sp.get('per_bin_inline', [])])" | tee -a $O/lines.txt
done
