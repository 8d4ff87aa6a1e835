#!/bin/bash
# round 4, step o: the tiny kernel with its Gramian elements in registers (lib_greg, -DCMF_TINY_GREG) against the default build, C2
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_o; mkdir -p $R/$O; cd $R
for v in base greg base greg; do
  if [ $v = base ]; then unset CMFREC_HIP_LIBDIR; else export CMFREC_HIP_LIBDIR=$PWD/cmfrec_amd/lib_$v; fi
  echo "$v $(timeout -k 10 600 python bench.py --workload c2 --no-cpu-baseline --no-scale-point --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print(d["ms_per_step"], [(e["step"], e["kernel"][:22], e.get("inline_ms")) for e in r["per_kernel"] if "tiny" in e["kernel"] or "W=1" in e["kernel"]])')"
done | tee $O/greg.txt
