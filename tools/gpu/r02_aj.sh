#!/bin/bash
# round 2, step aj: two rows per wavefront for the rows of <= 16 entries, and the Gramian path of few split rows in line
# (short slices) instead of on the second stream.  Flake hunt of the one parity case that has failed, both ways; parity; benches.
export TMPDIR=/tmp
O=gpurun_out/r02_aj; mkdir -p $O
H=tools/microbench/heavy_rows_flake.py
( echo "== second stream (round-2 behaviour so far)"
  CMFREC_HIP_VH_GRAM_ASIDE=1 timeout -k 10 300 python $H gram slice 7 0 300 | tail -4
  CMFREC_HIP_VH_GRAM_ASIDE=1 timeout -k 10 300 python $H gram 7 0 300 | tail -2
  echo "== in line"
  timeout -k 10 300 python $H gram slice 7 0 300 | tail -4
  timeout -k 10 300 python $H gram 7 0 300 | tail -2
  timeout -k 10 300 python $H gram 50 1 300 | tail -2 ) > $O/flake.log 2>&1
cat $O/flake.log
timeout -k 10 1500 python -m pytest tests/test_gpu_operators.py tests/test_gpu_config_widths.py -m gpu -q 2>&1 | tail -8 | tee $O/pytest.log
for w in c2 c4shard c1; do
  timeout -k 10 400 python bench.py --no-cpu-baseline --workload $w --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/bench_$w.json
  CMFREC_HIP_TINY2=0 timeout -k 10 400 python bench.py --no-cpu-baseline --workload $w --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/bench_${w}_tiny2off.json
done
CMFREC_HIP_VH_GRAM_ASIDE=1 timeout -k 10 400 python bench.py --no-cpu-baseline --workload c2 --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/bench_c2_aside.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02_aj/bench_*.json')):
    try:
        j=json.loads(open(f).read()); print(f.split('/')[-1], j['ms_per_step'], j.get('roofline',{}).get('frac'), j.get('config',{}).get('per_kernel_ms'))
    except Exception as e: print(f, 'unreadable', e)
PY
