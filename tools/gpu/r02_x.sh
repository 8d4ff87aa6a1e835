#!/bin/bash
# round 2, step x: Gramian path by default for the split rows (one wavefront per slice), faster gram_cg: full GPU suite + benches
export TMPDIR=/tmp
O=gpurun_out/r02_x; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 -k "not fullsize and not multidevice" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.log | cut -c1-300 | tee -a $O/summary.txt
echo "== c2" | tee -a $O/summary.txt
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/c2.json
python -c "
import json; d=json.load(open('$O/c2.json')); r=d['roofline']; print(d['ms_per_step'], r['frac'], r['kernel'], r['iteration']['frac_of_hbm_peak'], r['iteration']['halfstep_ms']); print(' '.join('%s:%s=%.3f' % (e['step'], e['kernel'][:14], e['avg_ms']) for e in r['per_kernel']))" | tee -a $O/summary.txt
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
echo "== c1" | tee -a $O/summary.txt
timeout 600 python bench.py --no-cpu-baseline --workload c1 --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-200 | tee -a $O/summary.txt
