#!/bin/bash
# round 5, third call: the two-rows-per-wavefront kernel of the <= 32-entry bin (cg_pair_kernels.hpp).  Targeted parity tests, then
# A/B on the same box: CMFREC_HIP_PAIR=0 (one row per wavefront, round 4's kernel) against the default, C2 / C1 / c4shard, bins in
# line and side by side; then where the wavefronts' time goes (instrumented build).
export TMPDIR=/tmp
O=gpurun_out/r05_c; mkdir -p $O
R=$GRAFT_REPO_ROOT
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
timeout -k 10 1200 python -m pytest tests/test_gpu_operators.py tests/test_gpu_golden.py tests/test_gpu_switches.py -m gpu -q -x 2>&1 | grep -v "$F" | tail -8 | tee $O/pytest_targeted.log
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
print(tag, d.get("ms_per_step"), rf.get("frac"), sel(bins) or "", sel(pb) or "")
PY
B="python $R/bench.py --no-cpu-baseline --no-scale-point"
for rep in 1 2; do
for pair in 0 1; do
  CMFREC_HIP_PAIR=$pair $B --steps 40 --warmup 5 2>/dev/null | python /tmp/line.py "c2 pair=$pair" | tee -a $O/lines.txt
  CMFREC_HIP_PAIR=$pair $B --workload c4shard --steps 20 --warmup 3 2>/dev/null | python /tmp/line.py "c4shard pair=$pair" | tee -a $O/lines.txt
done
done
for pair in 0 1; do
  CMFREC_HIP_PAIR=$pair $B --workload c1 --steps 40 --warmup 5 2>/dev/null | python /tmp/line.py "c1 pair=$pair" | tee -a $O/lines.txt
done
CMFREC_HIP_LIBDIR=$R/cmfrec_amd/lib_ticks CMFREC_HIP_BINS_PAR=1 timeout 600 python tools/microbench/cg_ticks.py 10 2>&1 | grep -v "$F" | tee $O/cg_ticks_inline.txt
CMFREC_HIP_LIBDIR=$R/cmfrec_amd/lib_ticks timeout 600 python tools/microbench/cg_ticks.py 10 2>&1 | grep -v "$F" | tee $O/cg_ticks_side_by_side.txt
