#!/bin/bash
# round 2, step r: shader clock / power while the C2 iteration runs (is the additive load + compute time a clock effect?)
export TMPDIR=/tmp
O=gpurun_out/r02_r; mkdir -p $O
rocm-smi --showclocks --showpower > $O/idle.txt 2>&1
(timeout 300 python bench.py --no-cpu-baseline --steps 1500 --warmup 5 > $O/bench.json 2>/dev/null) &
BP=$!
sleep 45
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' ' >> $O/load.txt; echo >> $O/load.txt
  sleep 0.4
done
wait $BP
tail -1 $O/bench.json | cut -c1-200
grep -E "sclk|Power" $O/idle.txt | head -4
cat $O/load.txt | cut -c1-300
