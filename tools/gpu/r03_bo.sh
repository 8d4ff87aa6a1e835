#!/bin/bash
# round 3: workgroups per CU of the factor matrices' Gramian kernel (first stage), bins in line under rocprofv3
export TMPDIR=/tmp
O=gpurun_out/r03_bo; mkdir -p $O
for gb in 1 2 4 8; do
  cd /tmp; CMFREC_HIP_GRAM_BLOCKS=$gb CMFREC_HIP_BINS_PAR=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace$gb -o c2 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-scale-point --steps 10 --warmup 2 > $GRAFT_REPO_ROOT/$O/bench$gb.json 2>/dev/null
  cd $GRAFT_REPO_ROOT; f=$(find $O/trace$gb -name "*kernel_stats.csv" | head -1)
  python - $f $gb $O/bench$gb.json <<'PY'
import csv,sys,json
d=json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
out="blocks/CU %s: %.4f ms/iter" % (sys.argv[2], d["ms_per_step"])
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'gram_mfma_partial_kernel<double' in n or 'gram_reduce_kernel<double' in n:
        out += " | %s %.1f us" % (n.split('::')[1].split('<')[0], float(r['AverageNs'])/1e3)
print(out)
PY
  rm -rf $O/trace$gb
done
