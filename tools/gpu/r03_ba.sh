#!/bin/bash
# round 3: replicate through a wave-private LDS line (cmfrec_amd/lib) against the ds_bpermute version (cmfrec_amd/lib_ab,
# -DCMF_REP_LDS=0); both carry the Gramian kernel with the last column block of k = 50 on the vector ALU.
export TMPDIR=/tmp
O=gpurun_out/r03_ba; mkdir -p $O
summ() { python - "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print(sys.argv[1].split("/")[-1], d["ms_per_step"], "ms", " | ".join("%s%s %.3f" % (k["step"], k["kernel"].split("(")[0][:14].strip().replace("cg_rows_",""), k["avg_ms"]) for k in r["per_kernel"]))
PY
}
run() { name=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-scale-point --steps 10 --warmup 3 > $O/$name.json 2>$O/$name.err; summ $O/$name.json; }
timeout 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_poisoned_lds.py -x -q 2>&1 | tail -2 | tee $O/pytest_ops.log
run par1_lds CMFREC_HIP_BINS_PAR=1
run par1_shfl CMFREC_HIP_BINS_PAR=1 CMFREC_HIP_LIBDIR=$PWD/cmfrec_amd/lib_ab
run default_lds X=1
run default_shfl CMFREC_HIP_LIBDIR=$PWD/cmfrec_amd/lib_ab
run default_lds2 X=1
run default_shfl2 CMFREC_HIP_LIBDIR=$PWD/cmfrec_amd/lib_ab
cd /tmp; CMFREC_HIP_BINS_PAR=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o c2 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-scale-point --steps 10 --warmup 2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_inline.csv; rm -rf $O/trace
grep "cmfhip" $O/kernel_stats_inline.csv | awk -F'","' '{printf "%-90s %s calls %.1f us\n", substr($1,2,90), $2, $4/1000}' | head -14
