#!/bin/bash
# round 3: four-wavefront consumer of the k_t = 257 item step (gramk_consumer_kernel) -- parity, then the c5 shard
export TMPDIR=/tmp
O=gpurun_out/r03_s; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_config_widths.py -x -q -k "c5" > $O/pytest_c5w.log 2>&1; tail -8 $O/pytest_c5w.log
for mode in on off; do
  if [ $mode = off ]; then export CMFREC_HIP_GRAMK=0; else unset CMFREC_HIP_GRAMK; fi
  timeout 1200 python bench.py --workload c5shard --no-cpu-baseline --steps 3 --warmup 1 > $O/c5shard_$mode.json 2>$O/c5shard_$mode.err
  tail -c 900 $O/c5shard_$mode.json; echo
done
