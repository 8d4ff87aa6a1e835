#!/bin/bash
# round 2, step z: full-size / multi-device tests and the side workloads after the Gramian-path default
export TMPDIR=/tmp
O=gpurun_out/r02_z; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 -k "fullsize or multidevice or very_heavy or distributed" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -4 $O/pytest.log | cut -c1-300 | tee -a $O/summary.txt
for w in c1 c2 c3 c5shard fit; do
  echo "== $w" | tee -a $O/summary.txt
  a="--workload $w"; [ $w = c2 ] && a=""
  timeout 900 python bench.py --no-cpu-baseline $a --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-420 | tee -a $O/summary.txt
done
