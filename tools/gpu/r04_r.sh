#!/bin/bash
# round 4, step r: where the time of C1 + implicit features goes (kernel stats), and the plain C1 / C3 lines on the current build
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_r; mkdir -p $R/$O; cd $R
for w in "c1" "c1 --implicit-features" "c3" "c3 --implicit-features"; do
  echo "$w: $(timeout -k 10 600 python bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-500)"
done | tee $O/side_workloads.txt
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_c1if -o c1if -- python $R/bench.py --workload c1 --implicit-features --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2>&1
cd $R
cp $(find $O/trace_c1if -name "*kernel_stats.csv" | head -1) $O/c1_implicit_features_kernel_stats.csv; rm -rf $O/trace_c1if
head -25 $O/c1_implicit_features_kernel_stats.csv | cut -c1-220
