#!/bin/bash
# round 4, step n: single precision, rows of 257..512 entries on four-wave teams with two resident tiles (CMFREC_HIP_HEAVY_SPLIT=0: as before)
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_n; mkdir -p $R/$O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_config_widths.py tests/test_gpu_operators.py -m gpu -x -q -k "float32 or c4 or single" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for hs in 0 1; do
  echo "HEAVY_SPLIT=$hs c4shard: $(CMFREC_HIP_HEAVY_SPLIT=$hs timeout -k 10 600 python bench.py --workload c4shard --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d.get("ms_per_iteration", d.get("ms_per_step")), json.dumps(d.get("per_kernel") or d.get("bins") or "")[:900])')"
done | tee $O/c4shard.txt
for hs in 0 1; do
  echo "HEAVY_SPLIT=$hs scale point: $(CMFREC_HIP_HEAVY_SPLIT=$hs timeout -k 10 900 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); sp=d["scale_point"]; print(sp["ms_per_step"], [(b["step"], b["bin"], b["inline_ms"], b["frac"]) for b in sp["per_bin_inline"]])')"
done | tee $O/scale_point.txt
