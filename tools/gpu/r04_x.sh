#!/bin/bash
# round 4, step x: rows of <= 16 entries on the one-row kernel (its Gramian now in registers) against two rows per wavefront (LDS-bound)
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_x; mkdir -p $R/$O; cd $R
for t2 in 1 0 1 0; do
  echo "TINY2=$t2 $(CMFREC_HIP_TINY2=$t2 timeout -k 10 600 python bench.py --workload c2 --no-cpu-baseline --no-scale-point --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("c2", d["ms_per_step"], [(e["step"], e.get("inline_ms")) for e in r["per_kernel"] if "tiny" in e["kernel"]])')"
done | tee $O/tiny2.txt
for t2 in 1 0; do
  echo "TINY2=$t2 c4shard $(CMFREC_HIP_TINY2=$t2 timeout -k 10 600 python bench.py --workload c4shard --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d.get("ms_per_iteration"), d.get("halfstep_ms"))')"
done | tee -a $O/tiny2.txt
