#!/bin/bash
# round 4, step b: where the lock-step kernel loses its time -- the same kernel without its MFMAs (results wrong: timing only),
# and with eight instead of sixteen wavefronts per workgroup (two workgroups per CU, half-filled MFMA columns)
export TMPDIR=/tmp
O=gpurun_out/r04_b; mkdir -p $O
for v in base nomfma nw8 nw8nomfma; do
  if [ $v = base ]; then unset CMFREC_HIP_LIBDIR; else export CMFREC_HIP_LIBDIR=$PWD/cmfrec_amd/lib_$v; fi
  for mm in 0 1 all; do
    [ $v != base ] && [ $mm = 0 ] && continue
    CMFREC_HIP_BINS_PAR=1 CMFREC_HIP_MM=$mm timeout -k 10 600 python bench.py --no-cpu-baseline --no-scale-point --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_${v}_mm$mm.json
    python - <<PY
import json
d=json.loads(open('$O/bench_${v}_mm$mm.json').read())
pk=d['roofline'].get('per_kernel',[])
print('$v mm=$mm ms/iter', d['ms_per_step'], ' '.join('%s:tiny=%.3f' % (p['step'], p['avg_ms']) for p in pk if 'tiny' in p['kernel']))
PY
  done
done 2>&1 | tee $O/summary.txt
