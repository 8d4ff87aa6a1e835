#!/bin/bash
# round 2, step v: one-wavefront-per-slice Gramian kernel for the split rows: parity, then c4shard / C4 on one GPU / C2 with
# the split rows on the Gramian path (two kernels) against the streaming path
export TMPDIR=/tmp
O=gpurun_out/r02_v; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_operators.py -m gpu -q -x --timeout 600 -k "very_heavy" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.log | cut -c1-300 | tee -a $O/summary.txt
for e in "CMFREC_HIP_VH=stream" "CMFREC_HIP_VH=gram" "CMFREC_HIP_VH=gram CMFREC_HIP_GRAM_KERNEL=slice" "CMFREC_HIP_VH=gram CMFREC_HIP_VH_MIN=513" "CMFREC_HIP_VH=gram CMFREC_HIP_VH_MIN=257"; do
  echo "== c4shard $e" | tee -a $O/summary.txt
  env $e timeout 900 python bench.py --no-cpu-baseline --workload c4shard --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/tmp.json
  python -c "
import json; d=json.load(open('$O/tmp.json')); b=[v for k,v in d.items() if k.startswith('bins')][0]
print(d['ms_per_iteration'], {k: round(v,3) for k,v in d['halfstep_ms'].items()}, ' '.join('%s=%.3f' % (k, v['ms']) for k, v in b.items()))" | tee -a $O/summary.txt
done
for e in "CMFREC_HIP_VH=stream" "CMFREC_HIP_VH=gram" "CMFREC_HIP_VH=gram CMFREC_HIP_VH_MIN=513"; do
  echo "== C4 on one GPU $e" | tee -a $O/summary.txt
  env $e timeout 1500 python bench.py --force-dist --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/tmp.json
  python -c "
import json; d=json.load(open('$O/tmp.json'))
print(d['ms_per_step'], ' '.join('%s%s=%.2f' % (e['step'], e['kernel'].split(' ')[0][-10:], e['avg_ms']) for e in d['roofline']['per_kernel_rank0']))" | tee -a $O/summary.txt
done
for e in "CMFREC_HIP_VH=stream" "CMFREC_HIP_VH=gram"; do
  echo "== c2 $e" | tee -a $O/summary.txt
  env $e timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/tmp.json
  python -c "
import json; d=json.load(open('$O/tmp.json')); print(d['ms_per_step'], d['roofline']['iteration']['halfstep_ms'])" | tee -a $O/summary.txt
done
