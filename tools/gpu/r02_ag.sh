#!/bin/bash
# round 2, step ag: phase skipping on the workgroup-per-row Cholesky kernel at 17 tiles (-DCMF_CHOL_DEBUG build): c5 shard
export TMPDIR=/tmp
O=gpurun_out/r02_ag; mkdir -p $O
for m in 0 1 2 4 8 16 5 13 29 31; do
  echo "== skip=$m" | tee -a $O/summary.txt
  CMFREC_HIP_CHOL_SKIP=$m timeout 500 python bench.py --no-cpu-baseline --workload c5shard --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_iteration'], d['halfstep_ms'])" | tee -a $O/summary.txt
done
CMFREC_HIP_CHOL_ROWS17=c32 timeout 500 python bench.py --no-cpu-baseline --workload c5shard --steps 2 --warmup 1 2>/dev/null | tail -1 | cut -c100-330 | tee -a $O/summary.txt
