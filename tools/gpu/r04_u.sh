#!/bin/bash
# round 4, step u: double precision, rows of 257..384 entries on six-wave teams (CMFREC_HIP_HEAVY_SPLIT=0: eight as before)
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_u; mkdir -p $R/$O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_config_widths.py tests/test_gpu_golden.py tests/test_gpu_switches.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for hs in 0 1 0 1; do
  echo "HEAVY_SPLIT=$hs $(CMFREC_HIP_HEAVY_SPLIT=$hs timeout -k 10 600 python bench.py --workload c2 --no-cpu-baseline --no-scale-point --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("c2", d["ms_per_step"], r["frac"], [(e["step"], e.get("inline_ms")) for e in r["per_kernel"] if "W=8" in e["kernel"]])')"
done | tee $O/c2_w6.txt
