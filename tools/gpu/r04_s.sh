#!/bin/bash
# round 4, step s: whole GPU suite (plain and with poisoned LDS) + smoke + side workloads on the current build
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_s; mkdir -p $R/$O; cd $R
timeout -k 10 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
CMFREC_HIP_POISON_LDS=1 timeout -k 10 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_poisoned.log 2>&1; tail -4 $O/pytest_gpu_poisoned.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
for w in "c1" "c1 --implicit-features" "c3" "c4shard" "c5shard"; do
  echo "$w: $(timeout -k 10 600 python bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-420)"
done | tee $O/side_workloads.txt
