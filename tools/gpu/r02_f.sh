#!/bin/bash
# round 2, step f: eigen-solver timing, device-COO distributed engine, C4 path on one rank
export TMPDIR=/tmp
O=gpurun_out/r02_f; mkdir -p $O
echo "== eig bench n=256 p=512" | tee -a $O/summary.txt
timeout 300 tools/microbench/eig_bench 256 512 2>&1 | tee -a $O/summary.txt
echo "== eig bench n=128 p=64" | tee -a $O/summary.txt
timeout 300 tools/microbench/eig_bench 128 64 2>&1 | tee -a $O/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -k "distributed or large_k or widths" > $O/pytest_dist.log 2>&1; echo "pytest dist rc=$?" | tee -a $O/summary.txt
tail -12 $O/pytest_dist.log | cut -c1-300 | tee -a $O/summary.txt
echo "== bench --force-dist --scale 0.125 (C4 path, one rank)" | tee -a $O/summary.txt
timeout 900 python bench.py --force-dist --scale 0.125 --steps 5 --warmup 2 2>$O/c4dist.err | tail -1 | cut -c1-1500 | tee -a $O/summary.txt
tail -5 $O/c4dist.err | tee -a $O/summary.txt
echo "== c4shard" | tee -a $O/summary.txt
timeout 900 python bench.py --workload c4shard --steps 5 --warmup 2 2>$O/c4.err | tail -1 | tee -a $O/summary.txt
echo "== c5shard" | tee -a $O/summary.txt
timeout 900 python bench.py --workload c5shard --steps 2 --warmup 1 2>$O/c5.err | tail -1 | tee -a $O/summary.txt
