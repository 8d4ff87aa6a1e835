#!/bin/bash
# round 4, step m: default bench line (C2 + scale point with per-bin legs) on the current build
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_m; mkdir -p $R/$O; cd $R
timeout -k 10 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 6000 $O/bench_default.json
