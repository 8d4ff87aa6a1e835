#!/bin/bash
# round 3: the side workloads (config 1, 3, the config-4 and config-5 shards) with the WHOLE library under the ILP-first scheduler
# (cmfrec_amd/lib_ab) against the committed build
export TMPDIR=/tmp
O=gpurun_out/r03_bn; mkdir -p $O
for w in c1 c3 c4shard c5shard; do
  for v in base ab; do
    if [ $v = ab ]; then export CMFREC_HIP_LIBDIR=$PWD/cmfrec_amd/lib_ab; else unset CMFREC_HIP_LIBDIR; fi
    timeout -k 10 900 python bench.py --no-cpu-baseline --workload $w --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/${w}_$v.json
    python -c "import json,sys; d=json.loads(open('$O/${w}_$v.json').read()); print('$w $v', d.get('ms_per_iteration'), d.get('halfstep_ms'))"
  done
done
