#!/bin/bash
# round 2, step h: multi-device C ABI test, whole GPU suite, c3 profile + SQ counters of the two-kernel Cholesky
export TMPDIR=/tmp
O=gpurun_out/r02_h; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -k "multidevice" > $O/pytest_md.log 2>&1; echo "pytest multidevice rc=$?" | tee -a $O/summary.txt
tail -8 $O/pytest_md.log | cut -c1-300 | tee -a $O/summary.txt
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 -k "not multidevice" > $O/pytest_all.log 2>&1; echo "pytest all rc=$?" | tee -a $O/summary.txt
grep -E "passed|failed" $O/pytest_all.log | tee -a $O/summary.txt
R=$PWD
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_c3 -- python $R/bench.py --workload c3 --steps 3 --warmup 1 > $R/$O/prof_c3.log 2>&1
cd $R; find $O/prof_c3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c3_kernel_stats.csv; head -6 $O/c3_kernel_stats.csv | cut -c1-220 | tee -a $O/summary.txt
cd /tmp; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $R/$O/pmc_c3 -- python $R/bench.py --workload c3 --steps 2 --warmup 1 > $R/$O/pmc_c3.log 2>&1
cd $R; python - <<'PY' | tee -a gpurun_out/r02_h/summary.txt
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
for f in glob.glob('gpurun_out/r02_h/pmc_c3/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:90]
        if 'chol_wave' not in k: continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
for k,v in agg.items():
    print(k, 'launches', len(cnt[k]))
    for c,x in sorted(v.items()): print('   %-28s %.4g' % (c, x/len(cnt[k])))
PY
