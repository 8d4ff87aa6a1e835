#!/bin/bash
# round 4, end-of-round measurement: GPU suite (plain / poisoned LDS), smoke, counter passes with the bins in line (a dispatch's counters
# are its own only then; matrix-pipe counters included), kernel traces of the default run and of the in-line run, the default bench line,
# side workloads.
export TMPDIR=/tmp
O=gpurun_out/r04_final; mkdir -p $O
R=$GRAFT_REPO_ROOT
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
if [ "$1" != "nosuite" ]; then
timeout -k 10 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | tail -3 | tee $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -1 | tee $O/smoke.log
fi
B="python $R/bench.py --no-cpu-baseline --no-scale-point --steps 4 --warmup 1"
pass() { name=$1; shift; rm -rf $R/$O/pmc_$name; cd /tmp; CMFREC_HIP_BINS_PAR=1 timeout -k 10 400 rocprofv3 --pmc "$@" --output-format csv -d $R/$O/pmc_$name -- $B > $R/$O/pmc_$name.log 2>&1; echo "pmc $name rc=$?"; cd $R; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
pass busy SQ_BUSY_CYCLES SQ_WAVE_CYCLES
pass tcc TCC_HIT_sum TCC_MISS_sum
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64
pass mfmai SQ_INSTS_MFMA
pass lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_valu $O/pmc_busy $O/pmc_tcc $O/pmc_mfma $O/pmc_mfmai $O/pmc_lds --calibration profiles/fetch_calibration.json --round r04_final -o $O/pmc_summary.json 2>&1 | tail -2
cp $O/pmc_summary.json profiles/pmc_latest.json      # read by bench.py below (roofline.traffic)
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_valu $O/pmc_busy $O/pmc_tcc $O/pmc_mfma $O/pmc_mfmai $O/pmc_lds
for mode in default inline; do
  if [ $mode = inline ]; then export CMFREC_HIP_BINS_PAR=1; else unset CMFREC_HIP_BINS_PAR; fi
  cd /tmp; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_$mode -o c2 -- python $R/bench.py --no-cpu-baseline --no-scale-point > $R/$O/bench_under_rocprof_$mode.json 2>$R/$O/bench_under_rocprof_$mode.err; echo "trace $mode rc=$?"
  cd $R; f=$(find $O/trace_$mode -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$mode.csv
  rm -rf $O/trace_$mode
done
unset CMFREC_HIP_BINS_PAR
timeout -k 10 900 python bench.py > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"; tail -1 $O/bench.json | cut -c1-1500
rm -f $O/bench_side.jsonl
for w in c1 c3 c4shard c5shard fit; do
  timeout -k 10 900 python bench.py --no-cpu-baseline --workload $w --steps 10 --warmup 3 2>/dev/null | tail -1 >> $O/bench_side.jsonl
done
timeout -k 10 900 python bench.py --no-cpu-baseline --workload c1 --implicit-features --steps 10 --warmup 3 2>/dev/null | tail -1 >> $O/bench_side.jsonl
timeout -k 10 900 python tools/microbench/c3_block_cg.py 2>&1 | grep -v "$F" > $O/c3_shape_cg.txt
cut -c1-260 $O/bench_side.jsonl; cat $O/c3_shape_cg.txt
CMFREC_HIP_POISON_LDS=1 timeout -k 10 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | tail -3 | tee $O/pytest_gpu_poisoned.log
