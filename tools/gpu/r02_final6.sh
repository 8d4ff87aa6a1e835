#!/bin/bash
# round 2, end-of-round measurement, part 6: the bench line and the trace again after the report fixes (the in-line Gramian
# launch of the user step has a duration of its own; the tiny bin names both of its kernels); parity suite on the same build.
export TMPDIR=/tmp
O=gpurun_out/r02_final; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl" | tail -3 | tee $O/pytest_gpu.log
cd /tmp; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o c2 -- python $R/bench.py --no-cpu-baseline > $R/$O/bench_under_rocprof.json 2>$R/$O/bench_under_rocprof.err; echo "trace rc=$?"
cd $R; f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv && head -10 $f | cut -c1-160
rm -rf $O/trace
timeout -k 10 900 python bench.py > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"; tail -1 $O/bench.json | cut -c1-1500
