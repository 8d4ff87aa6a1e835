#!/bin/bash
# round 2, step a: wave-per-row Cholesky kernel -- parity + side benches (run through gpurun from the repo root)
export TMPDIR=/tmp
O=gpurun_out/r02_a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 -k "operators or golden" > $O/pytest_ops.log 2>&1; echo "pytest ops rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest_ops.log | tee -a $O/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -k "not operators and not golden" > $O/pytest_rest.log 2>&1; echo "pytest rest rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest_rest.log | tee -a $O/summary.txt
for v in default rows rows9; do
  unset CMFREC_HIP_CHOL CMFREC_HIP_CHOL_ROWS9
  [ $v = rows ] && export CMFREC_HIP_CHOL=rows
  [ $v = rows9 ] && export CMFREC_HIP_CHOL=rows CMFREC_HIP_CHOL_ROWS9=1
  echo "== c3 $v" | tee -a $O/summary.txt
  timeout 600 python bench.py --workload c3 --steps 5 --warmup 2 2>$O/c3_$v.err | tail -1 | tee -a $O/summary.txt
done
unset CMFREC_HIP_CHOL CMFREC_HIP_CHOL_ROWS9
echo "== k50 probe default" | tee -a $O/summary.txt
timeout 900 python tools/microbench/chol_k50_probe.py 2>&1 | tail -4 | tee -a $O/summary.txt
echo "== k50 probe rows" | tee -a $O/summary.txt
CMFREC_HIP_CHOL=rows timeout 900 python tools/microbench/chol_k50_probe.py 2>&1 | tail -4 | tee -a $O/summary.txt
echo "== c5shard" | tee -a $O/summary.txt
timeout 900 python bench.py --workload c5shard --steps 2 --warmup 1 2>$O/c5.err | tail -1 | tee -a $O/summary.txt
echo "== bench c2" | tee -a $O/summary.txt
timeout 600 python bench.py --no-cpu-baseline 2>$O/c2.err | tail -1 > $O/bench_c2.json; python - <<'PY' | tee -a gpurun_out/r02_a/summary.txt
import json
d=json.load(open('gpurun_out/r02_a/bench_c2.json'))
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['iteration'])
PY
