#!/bin/bash
# round 2, step p: gather of the next row issued behind the last tile product of the current one
# row, packed single-precision FMAs with an interleaved Gramian, resolved split-row work items)
export TMPDIR=/tmp
O=gpurun_out/r02_p; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 -k "not fullsize and not multidevice" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -6 $O/pytest.log | cut -c1-300 | tee -a $O/summary.txt
run() { echo "== $1 $2" | tee -a $O/summary.txt; env $1 timeout 600 python bench.py --no-cpu-baseline $2 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/tmp.json; python - <<'PY' | tee -a gpurun_out/r02_p/summary.txt
import json
d=json.load(open('gpurun_out/r02_p/tmp.json'))
r=d.get('roofline',{})
print(d.get('ms_per_step', d.get('ms_per_iteration')), r.get('frac'), (r.get('iteration') or {}).get('frac_of_hbm_peak'), (r.get('iteration') or {}).get('halfstep_ms', d.get('halfstep_ms')))
PY
}
run "X=1" ""
run "X=1" "--workload c4shard"
run "X=1" "--workload c1"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o c2 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 2 > $GRAFT_REPO_ROOT/$O/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -24 $f | cut -c1-200 | tee -a $O/summary.txt
