#!/bin/bash
# round 4, step t: slice kernel with compile-time column offsets (k = 64) and swapped slab buffers: parity of the split-row paths, then timing
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_t; mkdir -p $R/$O; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_operators.py tests/test_gpu_config_widths.py tests/test_gpu_golden.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for i in 1 2; do
  echo "c4shard $(timeout -k 10 600 python bench.py --workload c4shard --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d.get("ms_per_iteration"), d.get("halfstep_ms"))')"
done | tee $O/c4shard.txt
timeout -k 10 600 python bench.py --workload c2 --no-cpu-baseline --no-scale-point --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("c2", d["ms_per_step"], r["frac"], [(e["step"], e["kernel"][:12], e.get("inline_ms")) for e in r["per_kernel"]])' | tee $O/c2.txt
timeout -k 10 900 python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); sp=d["scale_point"]; print("scale point", sp["ms_per_step"], [(b["step"], b["bin"], b["inline_ms"], b["frac"]) for b in sp["per_bin_inline"]])' | tee $O/scale_point.txt
