#!/bin/bash
# round 2, step o: phase skipping on the C2 CG kernels (debug build, -DCMF_CG_DEBUG): which component does the time follow?
export TMPDIR=/tmp
O=gpurun_out/r02_o; mkdir -p $O
run() { echo "== skip=$1" | tee -a $O/summary.txt; CMFREC_HIP_CG_SKIP=$1 timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/tmp.json; python - <<'PY' | tee -a gpurun_out/r02_o/summary.txt
import json
d=json.load(open('gpurun_out/r02_o/tmp.json'))
r=d.get('roofline',{})
it=r.get('iteration') or {}
print(d.get('ms_per_step'), {k: round(v, 3) for k, v in it.get('halfstep_ms', {}).items()}, ' '.join('%s%s=%.3f' % (e['step'], e['kernel'].split(' ')[0][-14:], e['avg_ms']) for e in r.get('per_kernel', [])))
PY
}
for m in 0 1 2 4 8 6 14 15 3 7; do run $m; done
