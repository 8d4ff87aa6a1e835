#!/bin/bash
# round 5, second call: the GPU suite on the rebuilt libraries (g23, full-size reference parity tests), smoke, and the default bench
# line with parity_vs_reference
export TMPDIR=/tmp
O=gpurun_out/r05_b; mkdir -p $O
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
timeout -k 10 900 python -m pytest tests/test_gpu_reference_fullsize.py tests/test_gpu_golden.py -m gpu -q -s -k "reference or fullsize or eight or c1_ or c2_" 2>&1 | grep -v "$F" | tail -15 | tee $O/pytest_new.log
timeout -k 10 1800 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "$F" | tail -5 | tee $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -1 | tee $O/smoke.log
timeout -k 10 900 python bench.py --no-scale-point > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"
python - <<PY
import json
l=[x for x in open('$O/bench.json') if x.startswith('{')][-1]
d=json.loads(l)
print(d['ms_per_step'], d['roofline']['frac'], json.dumps(d.get('parity_vs_reference')), json.dumps(d['cpu_baseline'])[:300])
PY
