#!/bin/bash
# round 4, step k: whole GPU suite on the build with the eigen chain beside the batches, own potrs, NA_as_zero_X + side information
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_k; mkdir -p $R/$O; cd $R
timeout -k 10 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout -k 10 600 python bench.py --workload c5shard --no-cpu-baseline --steps 4 --warmup 2 2>/dev/null | tail -1 | cut -c1-600 | tee $O/c5shard.json
