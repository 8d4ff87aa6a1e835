#!/bin/bash
# round 3: matrix-pipe counters of the c5 shard's kernels (producer / consumer pair, low-rank rows, GEMMs)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/r03_aa; mkdir -p $R/$O
B="python $R/bench.py --workload c5shard --no-cpu-baseline --steps 1 --warmup 1"
cd /tmp; timeout -k 10 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $R/$O/pmc_mfma -- $B > $R/$O/pmc_mfma.log 2>&1; echo "pmc rc=$?"
cd $R; python - <<'PY'
import csv, glob, collections, json, os
O = "gpurun_out/r03_aa"
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(O + "/pmc_mfma/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if "cmfhip" not in name: continue
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], name)
        if key not in seen: seen.add(key); calls[name] += 1
out = {}
for name, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
    busy = c.get("SQ_BUSY_CYCLES", 0.0); mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    out[name] = dict(dispatches=calls[name], **{k: v for k, v in c.items()}, mfma_busy_over_busy=(mf / busy if busy else None))
    print("%-90s disp %3d  MFMA_BUSY/BUSY %.3f" % (name[:90], calls[name], mf / busy if busy else float("nan")))
json.dump(out, open(O + "/c5shard_mfma_counters.json", "w"), indent=1)
PY
rm -rf $O/pmc_mfma
