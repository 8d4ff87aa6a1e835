#!/bin/bash
# round 5, fifth call: the tile follows the row (cg_rows_kernel with NT = 5..8 entries per lane group by row length) against the
# 64-entry tile everywhere (a -DCMF_CG_NT_MIN_S=9 build in cmfrec_amd/lib_nt8), same box, alternating; pair-kernel modes again.
export TMPDIR=/tmp
O=gpurun_out/r05_e; mkdir -p $O
R=$GRAFT_REPO_ROOT
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
timeout -k 10 1500 python -m pytest tests/test_gpu_operators.py tests/test_gpu_golden.py tests/test_gpu_switches.py tests/test_gpu_config_widths.py -m gpu -q -x 2>&1 | grep -v "$F" | tail -8 | tee $O/pytest_targeted.log
cat > /tmp/line.py <<'PY'
import sys, json
tag = sys.argv[1]
l = [x for x in sys.stdin if x.startswith('{')]
if not l:
    print(tag, "no line"); sys.exit()
d = json.loads(l[-1])
rf = d.get("roofline") or {}
def short(e):
    return (e["step"], e["kernel"].split(" (")[0][-14:], e.get("inline_ms", e.get("avg_ms")))
bins = [short(e) for e in rf.get("per_kernel", [])]
print(tag, d.get("ms_per_step"), rf.get("frac"), bins, (rf.get("inline") or {}).get("halfstep_ms"))
PY
B="python $R/bench.py --no-cpu-baseline --no-scale-point"
for rep in 1 2; do
  CMFREC_HIP_LIBDIR=$R/cmfrec_amd/lib_nt8 $B --steps 40 --warmup 5 2>/dev/null | python /tmp/line.py "c2 nt8" | tee -a $O/lines.txt
  $B --steps 40 --warmup 5 2>/dev/null | python /tmp/line.py "c2 nt-by-row" | tee -a $O/lines.txt
done
for pair in 0 2; do
  CMFREC_HIP_PAIR=$pair $B --steps 40 --warmup 5 2>/dev/null | python /tmp/line.py "c2 nt-by-row pair=$pair" | tee -a $O/lines.txt
done
for w in c4shard c1 c3; do
  CMFREC_HIP_LIBDIR=$R/cmfrec_amd/lib_nt8 $B --workload $w --steps 20 --warmup 3 2>/dev/null | tail -1 | cut -c1-330 | sed "s/^/$w nt8 /" | tee -a $O/lines.txt
  $B --workload $w --steps 20 --warmup 3 2>/dev/null | tail -1 | cut -c1-330 | sed "s/^/$w nt-by-row /" | tee -a $O/lines.txt
done
python $R/bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/bench_default.json
cut -c1-600 $O/bench_default.json
