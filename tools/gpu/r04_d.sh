#!/bin/bash
# round 4, step d: baseline of the round on today's box -- default bench line (now with the in-line per-bin legs for C2 and the
# config-4 scale point), side workloads, the new C-caller and dense-X multi-device tests
export TMPDIR=/tmp
O=gpurun_out/r04_d; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_c_caller.py tests/test_gpu_multidevice.py -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest_new.log
timeout -k 10 900 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_d/bench.json').read())
print('C2 ms/iter', d['ms_per_step'], 'frac', d['roofline']['frac'], 'inline', d['roofline'].get('inline'))
for e in d['roofline']['per_kernel']: print('  ', e['step'], e['kernel'][:28], e['avg_ms'], e.get('inline_ms'), e.get('inline_frac'))
sp=d['scale_point']; print('C4 one GPU ms/iter', sp.get('ms_per_step'), sp.get('iteration_frac_of_hbm_peak'), sp.get('inline'))
for e in sp.get('per_bin_inline') or []: print('  ', e['step'], e['kernel'][:28], e['rows'], e['nnz'], e['inline_ms'], e['frac'])
print('cpu', d['cpu_baseline'])
PY
for w in c4shard c3 c5shard c1; do
  timeout -k 10 900 python bench.py --no-cpu-baseline --workload $w --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/side_$w.json
  python -c "import json; d=json.loads(open('$O/side_$w.json').read()); print('$w', d.get('ms_per_iteration', d.get('ms_per_step')), d.get('halfstep_ms'), str(d.get('roofline',{}).get('per_kernel'))[:600])"
done
