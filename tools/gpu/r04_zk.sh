#!/bin/bash
# round 4, step zk: default build with the pass vector through LDS in double precision: whole GPU suite (plain), smoke, C2 bench line
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_zk; mkdir -p $R/$O; cd $R
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
timeout -k 10 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | tail -6 | tee $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -1 | tee $O/smoke.log
timeout -k 10 600 python bench.py --workload c2 --no-cpu-baseline --no-scale-point --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("c2", d["ms_per_step"], [(e["step"], round(e.get("inline_ms"),3)) for e in r["per_kernel"]])' | tee $O/c2.txt
