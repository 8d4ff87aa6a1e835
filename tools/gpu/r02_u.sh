#!/bin/bash
# round 2, step u: the whole of BASELINE config 4 on ONE GPU (what `bench.py --gpus 1` of the scaling run does): per-bin times
export TMPDIR=/tmp
O=gpurun_out/r02_u; mkdir -p $O
timeout 1500 python bench.py --force-dist --no-cpu-baseline --steps 3 --warmup 1 > $O/c4_n1.log 2>&1; echo "rc=$?"
tail -1 $O/c4_n1.log > $O/c4_n1.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02_u/c4_n1.json'))
print(d['ms_per_step'], d['value'], d['config'].get('gen_seconds'), d['config'].get('setup_seconds'))
for e in d['roofline']['per_kernel_rank0']: print(e)
PY
