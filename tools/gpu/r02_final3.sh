#!/bin/bash
# round 2, end-of-round measurement, part 3: the bench line and the kernel trace again on the final kernels (the counter passes
# of part 2 stay valid: gram_cg only changed the order of its loads)
export TMPDIR=/tmp
O=gpurun_out/r02_final; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp; CMFREC_HIP_VH_INLINE=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o c2 -- python $R/bench.py --no-cpu-baseline > $R/$O/bench_under_rocprof.json 2>$R/$O/bench_under_rocprof.err; echo "trace rc=$?"
cd $R; f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv && head -8 $f | cut -c1-160
rm -rf $O/trace
timeout -k 10 900 python bench.py > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"; tail -1 $O/bench.json | cut -c1-3000
rm -f $O/bench_side.jsonl
for w in c1 c3 c4shard c5shard fit; do
  timeout -k 10 900 python bench.py --no-cpu-baseline --workload $w --steps 10 --warmup 3 2>/dev/null | tail -1 >> $O/bench_side.jsonl
done
timeout -k 10 900 python bench.py --no-cpu-baseline --workload c1 --implicit-features --steps 10 --warmup 3 2>/dev/null | tail -1 >> $O/bench_side.jsonl
timeout -k 10 1500 python bench.py --force-dist --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/bench_c4_one_gpu.json
cut -c1-200 $O/bench_side.jsonl; cut -c1-300 $O/bench_c4_one_gpu.json
