#!/bin/bash
# round 4, step w: where the split rows begin in double precision (CMFREC_HIP_VH_MIN): the Gramian path runs at 0.57 of the HBM peak,
# the eight-wave bin below it at 0.37-0.41 -- does the boundary belong lower than 513?
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_w; mkdir -p $R/$O; cd $R
for vm in 513 385 321 257 513 257; do
  echo "VH=gram VH_MIN=$vm $(CMFREC_HIP_VH=gram CMFREC_HIP_VH_MIN=$vm timeout -k 10 600 python bench.py --workload c2 --no-cpu-baseline --no-scale-point --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("c2", d["ms_per_step"], r["frac"], d["roofline"]["iteration"]["halfstep_ms"], [(e["step"], e["kernel"][:14], e.get("inline_ms")) for e in r["per_kernel"]])')"
done | tee $O/vh_min_gram.txt
