#!/bin/bash
# round 2, step m: block systems (side information / implicit features under CG) on the tiled CG kernels
export TMPDIR=/tmp
O=gpurun_out/r02_m; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 -k "not fullsize and not multidevice" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -6 $O/pytest.log | cut -c1-300 | tee -a $O/summary.txt
for v in tiled generic; do
  unset CMFREC_HIP_BLOCK_CG_GENERIC; [ $v = generic ] && export CMFREC_HIP_BLOCK_CG_GENERIC=1
  echo "== c1 + implicit features ($v)" | tee -a $O/summary.txt
  timeout 600 python bench.py --workload c1 --implicit-features --steps 5 --warmup 2 2>/dev/null | tail -1 | cut -c1-200 | tee -a $O/summary.txt
  echo "== c3 block CG ($v)" | tee -a $O/summary.txt
  timeout 600 python tools/microbench/c3_block_cg.py 2>&1 | tail -3 | tee -a $O/summary.txt
done
unset CMFREC_HIP_BLOCK_CG_GENERIC
echo "== c1 plain" | tee -a $O/summary.txt
timeout 600 python bench.py --workload c1 --steps 5 --warmup 2 2>/dev/null | tail -1 | cut -c1-200 | tee -a $O/summary.txt
