#!/bin/bash
# round 4, step z2: explicit model, rows <= 16: two-rows kernel / one-row 16-slot / one-row 32-slot on config 4 (one GPU, fp32) and c1 (fp64 cg)
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_z2; mkdir -p $R/$O; cd $R
sp() { timeout -k 10 900 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); sp=d["scale_point"]; print(sys.argv[1], sp["ms_per_step"], [(b["step"], b["bin"], b["inline_ms"]) for b in sp["per_bin_inline"] if b["bin"] in ("tiny", 5, "5")])' "$1"; }
{
for i in 1 2; do sp "C4 default(tiny2)"; CMFREC_HIP_TINY2=0 sp "C4 tiny2=0 ne2"; CMFREC_HIP_TINY2=0 CMFREC_HIP_TINY16=0 sp "C4 tiny2=0 ne4"; done
} | tee $O/ab.txt
