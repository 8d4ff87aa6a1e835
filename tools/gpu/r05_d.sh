#!/bin/bash
# round 5, fourth call: pair kernel v2 (lane-linear Gramian layout read with ds_read_b128; CMFREC_HIP_PAIR=2: two launches, the rows
# of <= 16 entries with the Gramian in registers) against one row per wavefront (PAIR=0); cache-policy bits on the gather.
export TMPDIR=/tmp
O=gpurun_out/r05_d; mkdir -p $O
R=$GRAFT_REPO_ROOT
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
timeout -k 10 1200 python -m pytest tests/test_gpu_operators.py tests/test_gpu_switches.py tests/test_gpu_distributed.py tests/test_gpu_multidevice.py -m gpu -q -x 2>&1 | grep -v "$F" | tail -8 | tee $O/pytest_targeted.log
cat > /tmp/line.py <<'PY'
import sys, json
tag = sys.argv[1]
l = [x for x in sys.stdin if x.startswith('{')]
if not l:
    print(tag, "no line"); sys.exit()
d = json.loads(l[-1])
rf = d.get("roofline") or {}
def short(e):
    return (e["step"], e["kernel"].split(" (")[0][:22], e.get("inline_ms", e.get("avg_ms")))
bins = [short(e) for e in rf.get("per_kernel", [])]
pb = [short(e) for e in (rf.get("per_bin_inline") or [])]
sel = lambda L: [b for b in L if "tiny" in b[1] or "pair" in b[1]]
print(tag, d.get("ms_per_step"), rf.get("frac"), sel(bins) or "", sel(pb) or "", (rf.get("inline") or {}).get("halfstep_ms"))
PY
B="python $R/bench.py --no-cpu-baseline --no-scale-point"
for rep in 1 2; do
for pair in 0 1 2; do
  CMFREC_HIP_PAIR=$pair $B --steps 40 --warmup 5 2>/dev/null | python /tmp/line.py "c2 pair=$pair" | tee -a $O/lines.txt
done
done
for pair in 0 1 2; do
  CMFREC_HIP_PAIR=$pair $B --workload c4shard --steps 20 --warmup 3 2>/dev/null | tail -1 | cut -c1-400 | sed "s/^/c4shard pair=$pair /" | tee -a $O/lines.txt
  CMFREC_HIP_PAIR=$pair $B --workload c1 --steps 40 --warmup 5 2>/dev/null | tail -1 | cut -c1-300 | sed "s/^/c1 pair=$pair /" | tee -a $O/lines.txt
done
./tools/microbench/gather_policy 2>&1 | tee $O/gather_policy.txt
