#!/bin/bash
# (First attempt: its counter pass put FETCH_SIZE and WRITE_SIZE into one rocprofv3 run, which this part cannot collect together;
# rocprofv3 aborted and then hung until the call's limit.  Only the calibration of step 1 comes from this script; the rest of the
# end-of-round data was taken by r02_final2.sh / r02_final3.sh / r02_final4.sh.)
# round 2, end-of-round measurement: FETCH_SIZE calibration (three access patterns), counter passes and kernel trace of the
# default `python bench.py`, the bench line itself (with the CPU baseline), the side workloads and BASELINE config 4 on one GPU
export TMPDIR=/tmp
O=gpurun_out/r02_final; mkdir -p $O
R=$GRAFT_REPO_ROOT
# 1. calibration
cd /tmp; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_calib -- $R/tools/microbench/fetch_calib > $R/$O/fetch_calib.out 2>$R/$O/fetch_calib.err
cd $R; python tools/fetch_calibration.py $O/pmc_calib $O/fetch_calib.out -o $O/fetch_calibration.json > $O/fetch_calibration.log 2>&1; tail -3 $O/fetch_calibration.log
# 2. counter passes (each in its own run; kernel trace separately)
B="python $R/bench.py --no-cpu-baseline --steps 4 --warmup 1"
cd /tmp; rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $R/$O/pmc_mem -- $B > $R/$O/pmc_mem.log 2>&1
cd /tmp; rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $R/$O/pmc_sq -- $B > $R/$O/pmc_sq.log 2>&1
cd /tmp; rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/$O/pmc_tcc -- $B > $R/$O/pmc_tcc.log 2>&1
cd /tmp; rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/$O/pmc_mfma -- $B > $R/$O/pmc_mfma.log 2>&1
cd $R; python tools/pmc_summary.py $O/pmc_mem $O/pmc_sq $O/pmc_tcc $O/pmc_mfma --calibration $O/fetch_calibration.json --round r02_final -o $O/pmc_summary.json 2>&1 | tail -2
cp $O/pmc_summary.json profiles/pmc_latest.json      # read by bench.py below (roofline.traffic)
# 3. kernel trace of the default command
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o c2 -- python $R/bench.py > $R/$O/bench_under_rocprof.json 2>$R/$O/bench_under_rocprof.err
cd $R; f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats.csv && head -14 $f | cut -c1-160
rm -rf $O/trace/*kernel_trace.csv
# 4. the bench line and the side workloads
python bench.py > $O/bench.json 2>$O/bench.err; tail -1 $O/bench.json | cut -c1-900
for w in c1 c3 c4shard c5shard fit; do
  timeout 900 python bench.py --no-cpu-baseline --workload $w --steps 10 --warmup 3 2>/dev/null | tail -1 >> $O/bench_side.jsonl
done
timeout 900 python bench.py --no-cpu-baseline --workload c1 --implicit-features --steps 10 --warmup 3 2>/dev/null | tail -1 >> $O/bench_side.jsonl
timeout 1500 python bench.py --force-dist --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/bench_c4_one_gpu.json
cut -c1-300 $O/bench_side.jsonl; cut -c1-400 $O/bench_c4_one_gpu.json
du -sh $O
