#!/bin/bash
# round 4, step j: how many workgroup slots the producer / consumer launches have to leave open for the eigen chain to run beside them
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_j; mkdir -p $R/$O; cd $R
for open in 24 32 40 48 64 80 96; do
  echo "open=$open $(CMFREC_HIP_GK_OPEN=$open timeout -k 10 600 python bench.py --workload c5shard --no-cpu-baseline --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["ms_per_iteration"], d["halfstep_ms"])')"
done | tee $O/gk_open2.txt
