#!/bin/bash
# round 2, end-of-round measurement, part 4: side workloads again after the 17-tile Cholesky changes (the C2 line, its trace and
# counter passes of parts 2-3 are unaffected: no CG / Gramian kernel changed)
export TMPDIR=/tmp
O=gpurun_out/r02_final; mkdir -p $O
rm -f $O/bench_side.jsonl
for w in c1 c3 c4shard c5shard fit; do
  timeout -k 10 900 python bench.py --no-cpu-baseline --workload $w --steps 10 --warmup 3 2>/dev/null | tail -1 >> $O/bench_side.jsonl
done
timeout -k 10 900 python bench.py --no-cpu-baseline --workload c1 --implicit-features --steps 10 --warmup 3 2>/dev/null | tail -1 >> $O/bench_side.jsonl
cut -c1-220 $O/bench_side.jsonl
