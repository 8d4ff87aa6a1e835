#!/bin/bash
# round 2, step ac: Gramian kernel with the two live columns of k = 50's last block on the VALU (6 instead of 10 MFMAs per slab)
export TMPDIR=/tmp
O=gpurun_out/r02_ac; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 -k "very_heavy or c2 or fullsize or fit or session" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed" $O/pytest.log | tail -2 | tee -a $O/summary.txt
for e in "X=1" "CMFREC_HIP_VH_MIN=513"; do
echo "== c2 $e" | tee -a $O/summary.txt
env $e timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/c2.json
python -c "
import json; d=json.load(open('$O/c2.json')); r=d['roofline']; print(d['ms_per_step'], r['frac'], r['kernel'][:40], r['iteration']['frac_of_hbm_peak'], r['iteration']['halfstep_ms']); print(' '.join('%s:%s=%.3f' % (e['step'], e['kernel'][:14], e['avg_ms']) for e in r['per_kernel']))" | tee -a $O/summary.txt
done
