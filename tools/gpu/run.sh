#!/bin/bash
# tools/gpu/run.sh <mode> [args] -- what a `gpurun` call of this repository runs (one script, round 5; the one-shot scripts of the
# earlier rounds are in the history: `git log -- tools/gpu`).  Output under gpurun_out/<TAG>/ (TAG: environment, default = mode).
#   tests [pytest args]        the GPU suite (or a selection: `tests tests/test_gpu_golden.py -k NA_as_zero`)
#   poisoned                   the GPU suite with NaN patterns in LDS and fresh buffers (CMFREC_HIP_POISON_LDS=1)
#   bench [bench.py args]      one bench line, stderr kept
#   ab <libdirA> <libdirB>     the same workloads on two builds, alternating on this box (C2 twice, c4shard, c1); libdir relative
#                              to cmfrec_amd/ (`lib` = the default build, `lib_nt8` = an experiment build made with
#                              `make -C cmfrec_amd/csrc OUTDIR=../lib_nt8 EXTRA=-D...`)
#   env <VAR> <v1> <v2> ...    the C2 line (and c4shard) once per value of an environment switch
#   sidestats <workload> ...   rocprofv3 kernel statistics of side workloads (c3, c5shard, c1, c4shard)
#   sideab <workload> <libdirA> <libdirB>   a side workload on two builds, alternating
#   sideenv <workload> <VAR> <v1> <v2> ...  a side workload once per value of an environment switch, alternating
#   final [nosuite]            the end-of-round set: suite, smoke, counter passes with the bins in line, kernel statistics of the
#                              default and the in-line run, the default bench line, side workloads, the --force-dist launch paths on
#                              one rank, the suite again with poisoned LDS
export TMPDIR=/tmp
MODE=${1:-tests}; shift
TAG=${TAG:-$MODE}
O=gpurun_out/$TAG; mkdir -p $O
R=$GRAFT_REPO_ROOT
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
B="python $R/bench.py --no-cpu-baseline --no-scale-point"
cat > /tmp/line.py <<'PY'
import sys, json
tag = sys.argv[1]
l = [x for x in sys.stdin if x.startswith('{')]
if not l:
    print(tag, "no line"); sys.exit()
d = json.loads(l[-1])
if "roofline" not in d:
    print(tag, d.get("ms_per_iteration"), d.get("frac_of_hbm_peak"), d.get("halfstep_ms")); sys.exit()
rf = d["roofline"]
short = lambda e: (e["step"], e["kernel"].split(" (")[0][-14:], e.get("inline_ms", e.get("avg_ms")))
print(tag, d.get("ms_per_step"), rf.get("frac"), [short(e) for e in rf.get("per_kernel", [])], (rf.get("inline") or {}).get("halfstep_ms"))
PY
suite() { timeout -k 10 2400 python -m pytest "${@:-tests}" -m gpu -q -x 2>&1 | grep -v "$F" | tail -12; }
case $MODE in
tests) suite "$@" | tee $O/pytest.log ;;
poisoned) CMFREC_HIP_POISON_LDS=1 suite "$@" | tee $O/pytest_poisoned.log ;;
bench) python $R/bench.py "$@" 2>$O/bench.err | tail -1 > $O/bench.json; grep -v "$F" $O/bench.err | tail -5; cut -c1-1200 $O/bench.json ;;
ab)
  for rep in 1 2; do for L in "$1" "$2"; do
    CMFREC_HIP_LIBDIR=$R/cmfrec_amd/$L $B --steps 40 --warmup 5 2>$O/err_$L.txt | python /tmp/line.py "c2 $L" | tee -a $O/lines.txt
  done; done
  for w in c4shard c1; do for L in "$1" "$2"; do
    CMFREC_HIP_LIBDIR=$R/cmfrec_amd/$L $B --workload $w --steps 20 --warmup 3 2>/dev/null | python /tmp/line.py "$w $L" | tee -a $O/lines.txt
  done; done
  # config 4 on one GPU (the scale point of the default line), per bin in line
  for L in "$1" "$2"; do
    CMFREC_HIP_LIBDIR=$R/cmfrec_amd/$L python $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
l = [x for x in sys.stdin if x.startswith('{')]
sp = (json.loads(l[-1]).get('scale_point') or {}) if l else {}
print('c4 $L', sp.get('ms_per_step'), (sp.get('inline') or {}).get('halfstep_ms'), [(e['step'], e['bin'], e['inline_ms']) for e in sp.get('per_bin_inline', [])])" | tee -a $O/lines.txt
  done ;;
env)
  V=$1; shift
  for rep in 1 2; do for x in "$@"; do
    env $V=$x $B --steps 40 --warmup 5 2>$O/err_$x.txt | python /tmp/line.py "c2 $V=$x" | tee -a $O/lines.txt
  done; done
  for x in "$@"; do env $V=$x $B --workload c4shard --steps 20 --warmup 3 2>/dev/null | python /tmp/line.py "c4shard $V=$x" | tee -a $O/lines.txt; done ;;
sideenv)
  # sideenv <workload> <VAR> <v1> <v2> ...: a side workload once per value of an environment switch, alternating three times
  w=$1; V=$2; shift; shift
  for rep in 1 2 3; do for x in "$@"; do
    env $V=$x $B --workload $w --steps 10 --warmup 3 2>/dev/null | python /tmp/line.py "$w $V=$x" | tee -a $O/lines.txt
  done; done ;;
sidestats)
  # rocprofv3 kernel statistics of bench.py's side workloads (c3, c5shard, ...), one csv each
  for w in "$@"; do
    cd /tmp; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_$w -o $w -- python $R/bench.py --no-cpu-baseline --workload $w --steps 10 --warmup 3 > $R/$O/bench_$w.json 2>$R/$O/bench_$w.err; echo "trace $w rc=$?"
    cd $R; f=$(find $O/trace_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$w.csv
    rm -rf $O/trace_$w; tail -1 $O/bench_$w.json | cut -c1-400
  done ;;
sideab)
  # sideab <workload> <libdirA> <libdirB>: a side workload on two builds, alternating on this box
  w=$1; shift
  for rep in 1 2 3; do for L in "$@"; do
    CMFREC_HIP_LIBDIR=$R/cmfrec_amd/$L $B --workload $w --steps 10 --warmup 3 2>/dev/null | python /tmp/line.py "$w $L" | tee -a $O/lines.txt
  done; done ;;
final)
  if [ "$1" != "nosuite" ]; then
    suite | tee $O/pytest_gpu.log
    python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -1 | tee $O/smoke.log
  fi
  BP="$B --steps 4 --warmup 1"
  pass() { name=$1; shift; rm -rf $R/$O/pmc_$name; cd /tmp; CMFREC_HIP_BINS_PAR=1 timeout -k 10 400 rocprofv3 --pmc "$@" --output-format csv -d $R/$O/pmc_$name -- $BP > $R/$O/pmc_$name.log 2>&1; echo "pmc $name rc=$?"; cd $R; }
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  pass valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
  pass busy SQ_BUSY_CYCLES SQ_WAVE_CYCLES
  pass tcc TCC_HIT_sum TCC_MISS_sum
  pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64
  pass mfmai SQ_INSTS_MFMA
  pass lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
  python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_valu $O/pmc_busy $O/pmc_tcc $O/pmc_mfma $O/pmc_mfmai $O/pmc_lds --calibration profiles/fetch_calibration.json --round ${ROUND:-r06}_final -o $O/pmc_summary.json 2>&1 | tail -2
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
  # the multi-rank launch paths on ONE rank (sharded engine + RCCL collectives / direct placement; config 4 and config 5)
  for ag in collective p2p; do
    timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --force-dist --allgather $ag --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | cut -c1-400 | sed "s/^/force-dist $ag /" | tee -a $O/force_dist.txt
  done
  timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29572 bench.py --gpus 1 --force-dist --workload c5 --scale 0.125 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | grep "^{" | tail -1 | cut -c1-400 | sed "s/^/force-dist c5 (one rank, an eighth of the problem: the launch path only) /" | tee -a $O/force_dist.txt
  CMFREC_HIP_POISON_LDS=1 suite | tail -3 | tee $O/pytest_gpu_poisoned.log ;;
*) echo "unknown mode $MODE"; exit 2 ;;
esac
