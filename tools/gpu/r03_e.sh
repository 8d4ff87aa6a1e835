#!/bin/bash
# round 3: new defaults (first generation + two streams, per-shard split boundary), second generation incl. 6-slot tiles,
# issue-level counters of the row kernels with the bins in line
export TMPDIR=/tmp
O=gpurun_out/r03_e; mkdir -p $O
R=$GRAFT_REPO_ROOT
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], "ms", " | ".join("%s%s %.3f" % (k["step"], k["kernel"].split("(")[0][:14].strip().replace("cg_rows_",""), k["avg_ms"]) for k in r["per_kernel"]))
PY
}
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/$name.json 2>$O/$name.err; summ $O/$name.json; }
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log
run default X=1
run par1 CMFREC_HIP_BINS_PAR=1
run cg2_par1 CMFREC_HIP_CG2=1 CMFREC_HIP_BINS_PAR=1
run cg2_nt6_par1 CMFREC_HIP_CG2=1 CMFREC_HIP_CG2_NT6=1 CMFREC_HIP_BINS_PAR=1
run cg2_nt6 CMFREC_HIP_CG2=1 CMFREC_HIP_CG2_NT6=1
timeout 600 python bench.py --no-cpu-baseline --workload c1 --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-200
# counters (bins in line)
rocprofv3 -L > $O/counters_list.txt 2>&1
B="python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1"
cd /tmp; CMFREC_HIP_BINS_PAR=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $R/$O/pmc_a -- $B > $R/$O/pmc_a.log 2>&1
cd /tmp; CMFREC_HIP_BINS_PAR=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $R/$O/pmc_b -- $B > $R/$O/pmc_b.log 2>&1
cd $R; python tools/pmc_summary.py $O/pmc_a $O/pmc_b --round r03_e -o $O/pmc_issue.json 2>&1 | tail -2
rm -rf $O/pmc_a $O/pmc_b
