#!/bin/bash
# round 3: phase timers inside the consumer of the k_t = 257 item step (debug build, cmfrec_amd/lib_dbg)
export TMPDIR=/tmp
O=gpurun_out/r03_r; mkdir -p $O
export CMFREC_HIP_LIBDIR=$GRAFT_REPO_ROOT/cmfrec_amd/lib_dbg
CMFREC_HIP_CHOL_TICKS=1 timeout 900 python bench.py --workload c5shard --no-cpu-baseline --steps 1 --warmup 1 > $O/c5shard_ticks.json 2> $O/c5shard_ticks.err
grep "ticks/row" $O/c5shard_ticks.err | tail -12
tail -c 300 $O/c5shard_ticks.json
