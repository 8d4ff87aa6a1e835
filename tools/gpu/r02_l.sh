#!/bin/bash
# round 2, step l: new full-size property tests; CG headline knobs (bins on two streams, split-row threshold / mode)
export TMPDIR=/tmp
O=gpurun_out/r02_l; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 -k "fullsize" > $O/pytest_full.log 2>&1; echo "pytest fullsize rc=$?" | tee -a $O/summary.txt
tail -6 $O/pytest_full.log | cut -c1-300 | tee -a $O/summary.txt
run() { echo "== c2 $1" | tee -a $O/summary.txt; env $1 timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/c2_tmp.json; python - <<'PY' | tee -a gpurun_out/r02_l/summary.txt
import json
d=json.load(open('gpurun_out/r02_l/c2_tmp.json'))
print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['iteration']['frac_of_hbm_peak'], d['roofline']['iteration']['halfstep_ms'])
PY
}
run "X=1"
run "CMFREC_HIP_BINS_ALT=1"
run "CMFREC_HIP_VH_MIN=513"
run "CMFREC_HIP_VH=gram"
run "CMFREC_HIP_VH_MIN=513 CMFREC_HIP_VH=gram"
run "CMFREC_HIP_VH_MIN=2049"
