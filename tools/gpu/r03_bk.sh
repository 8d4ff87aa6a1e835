#!/bin/bash
# round 3: the slice kernel in a translation unit of its own, double-precision object with -mllvm -amdgpu-sched-strategy=max-ilp
export TMPDIR=/tmp
O=gpurun_out/r03_bk; mkdir -p $O
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], "ms", "A %.3f B %.3f |" % (r["iteration"]["halfstep_ms"]["A"], r["iteration"]["halfstep_ms"]["B"]), " | ".join("%s%s %.3f" % (k["step"], k["kernel"].split("(")[0][:14].strip().replace("cg_rows_",""), k["avg_ms"]) for k in r["per_kernel"]))
PY
}
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-scale-point --steps 10 --warmup 3 > $O/$name.json 2>$O/$name.err; summ $O/$name.json; }
timeout 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_poisoned_lds.py tests/test_gpu_config_widths.py tests/test_gpu_switches.py -x -q 2>&1 | tail -30 > $O/pytest_ops.log; tail -2 $O/pytest_ops.log
run par1 CMFREC_HIP_BINS_PAR=1
run default X=1
run default2 X=1
cd /tmp; CMFREC_HIP_BINS_PAR=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o c2 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-scale-point --steps 10 --warmup 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_inline.csv; rm -rf $O/trace
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r03_bk/kernel_stats_inline.csv')):
    n=r['Name']
    if 'cmfhip' in n and 'double' in n:
        print('%-95s %5s %10.1f us avg'%(n[:95], r['Calls'], float(r['AverageNs'])/1e3))
PY
