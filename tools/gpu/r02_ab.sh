#!/bin/bash
# round 2, step ab: gram_cg with the slice sums batched across the thread's elements: parity, then C2 / c4shard / config 4
export TMPDIR=/tmp
O=gpurun_out/r02_ab; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "very_heavy or c4 or c2 or fullsize" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed" $O/pytest.log | tail -2 | tee -a $O/summary.txt
echo "== c2" | tee -a $O/summary.txt
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/c2.json
python -c "
import json; d=json.load(open('$O/c2.json')); r=d['roofline']; print(d['ms_per_step'], r['frac'], r['traffic'], r['alg_bytes_per_launch'], r['iteration']['frac_of_hbm_peak'], r['iteration']['halfstep_ms']); print(' '.join('%s:%s=%.3f' % (e['step'], e['kernel'][:14], e['avg_ms']) for e in r['per_kernel']))" | tee -a $O/summary.txt
echo "== c4shard" | tee -a $O/summary.txt
timeout 900 python bench.py --no-cpu-baseline --workload c4shard --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/tmp.json
python -c "
import json; d=json.load(open('$O/tmp.json')); b=[v for k,v in d.items() if k.startswith('bins')][0]
print(d['ms_per_iteration'], {k: round(v,3) for k,v in d['halfstep_ms'].items()}, ' '.join('%s=%.3f' % (k, v['ms']) for k, v in b.items()))" | tee -a $O/summary.txt
echo "== C4 on one GPU" | tee -a $O/summary.txt
timeout 1500 python bench.py --force-dist --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 > $O/c4_n1.json
python -c "
import json; d=json.load(open('$O/c4_n1.json'))
print(d['ms_per_step'], d['value'], ' '.join('%s%s=%.2f' % (e['step'], e['kernel'].split(' ')[0][-10:], e['avg_ms']) for e in d['roofline']['per_kernel_rank0']))" | tee -a $O/summary.txt
