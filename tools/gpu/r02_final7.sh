#!/bin/bash
# round 2, end-of-round measurement, part 7 (after the gram_cg fix: the source hash of the counter passes must match again):
# whole GPU suite, once more with the LDS poisoned; counter passes, trace, bench line, side workloads.
export TMPDIR=/tmp
O=gpurun_out/r02_final; mkdir -p $O
R=$GRAFT_REPO_ROOT
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl'
timeout -k 10 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | tail -3 | tee $O/pytest_gpu.log
CMFREC_HIP_POISON_LDS=1 timeout -k 10 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | tail -6 | tee $O/pytest_gpu_poisoned.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -1 | tee $O/smoke.log
B="python $R/bench.py --no-cpu-baseline --steps 4 --warmup 1"
pass() { name=$1; shift; rm -rf $R/$O/pmc_$name; cd /tmp; timeout -k 10 400 rocprofv3 --pmc "$@" --output-format csv -d $R/$O/pmc_$name -- $B > $R/$O/pmc_$name.log 2>&1; echo "pmc $name rc=$?"; cd $R; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
pass busy SQ_BUSY_CYCLES SQ_WAVE_CYCLES
pass tcc TCC_HIT_sum TCC_MISS_sum
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_valu $O/pmc_busy $O/pmc_tcc --calibration profiles/fetch_calibration.json --round r02_final -o $O/pmc_summary.json 2>&1 | tail -2
cp $O/pmc_summary.json profiles/pmc_latest.json      # read by bench.py below (roofline.traffic)
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_valu $O/pmc_busy $O/pmc_tcc
cd /tmp; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o c2 -- python $R/bench.py --no-cpu-baseline > $R/$O/bench_under_rocprof.json 2>$R/$O/bench_under_rocprof.err; echo "trace rc=$?"
cd $R; f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv
rm -rf $O/trace
timeout -k 10 900 python bench.py > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"; tail -1 $O/bench.json | cut -c1-1200
rm -f $O/bench_side.jsonl
for w in c1 c3 c4shard c5shard fit; do
  timeout -k 10 900 python bench.py --no-cpu-baseline --workload $w --steps 10 --warmup 3 2>/dev/null | tail -1 >> $O/bench_side.jsonl
done
timeout -k 10 900 python bench.py --no-cpu-baseline --workload c1 --implicit-features --steps 10 --warmup 3 2>/dev/null | tail -1 >> $O/bench_side.jsonl
timeout -k 10 1500 python bench.py --force-dist --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/bench_c4_one_gpu.json
cut -c1-200 $O/bench_side.jsonl; cut -c1-300 $O/bench_c4_one_gpu.json
