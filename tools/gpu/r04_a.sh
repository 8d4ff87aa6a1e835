#!/bin/bash
# round 4, step a: lock-step short-row kernel with the Gramian product on the matrix pipe (cg_mm_kernels.hpp):
# parity of the implicit operator cases with it, then C2 with CMFREC_HIP_MM = 0 / 1 / all (in line and on two streams)
export TMPDIR=/tmp
O=gpurun_out/r04_a; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_golden.py -x -q -m gpu -k "implicit or slot or two_rows or golden" 2>&1 | tail -5 > $O/pytest_mm1.log
cat $O/pytest_mm1.log
CMFREC_HIP_MM=all timeout -k 10 900 python -m pytest tests/test_gpu_operators.py -x -q -m gpu -k "implicit or slot or two_rows" 2>&1 | tail -5 > $O/pytest_mmall.log
cat $O/pytest_mmall.log
for par in 1 2; do
for mm in 0 1 all; do
  CMFREC_HIP_BINS_PAR=$par CMFREC_HIP_MM=$mm timeout -k 10 600 python bench.py --no-cpu-baseline --no-scale-point --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_par${par}_mm$mm.json
  python - <<PY
import json
d=json.loads(open('$O/bench_par${par}_mm$mm.json').read())
pk=d['roofline'].get('per_kernel',[])
print('par=$par mm=$mm ms/iter', d['ms_per_step'], ' '.join('%s:%s=%.3f' % (p['step'], p['kernel'].split(' ')[0][:22], p['avg_ms']) for p in pk))
PY
done; done
