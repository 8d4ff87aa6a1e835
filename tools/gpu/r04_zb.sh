#!/bin/bash
# round 4, step zb: Gramian reads from LDS as single ds_read_b64 (CMF_LDS_B64=1: 2 LDS cycles each, 256 B/clk) instead of the paired
# ds_read2_b64 the compiler forms (8 cycles, 128 B/clk); lib_b64 = that alone, lib_b64g = that + the tiny kernel's Gramian back in LDS
# (four wavefronts per SIMD); both also with two rows per wavefront forced for the short implicit rows (CMFREC_HIP_TINY2=1)
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_zb; mkdir -p $R/$O; cd $R
c2() { timeout -k 10 600 python bench.py --workload c2 --no-cpu-baseline --no-scale-point --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("c2", d["ms_per_step"], [(e["step"], round(e.get("inline_ms"),3)) for e in r["per_kernel"]])'; }
c4() { timeout -k 10 600 python bench.py --workload c4shard --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("c4shard", d.get("ms_per_step"), d.get("ms_per_iteration"), d.get("halfstep_ms"))'; }
{
for rep in 1 2; do
for L in lib lib_b64 lib_b64g; do
  export CMFREC_HIP_LIBDIR=$R/cmfrec_amd/$L
  echo "$L          $(c2)"
  echo "$L TINY2=1  $(CMFREC_HIP_TINY2=1 c2)"
done
done
for L in lib lib_b64 lib_b64g; do
  export CMFREC_HIP_LIBDIR=$R/cmfrec_amd/$L
  echo "$L          $(c4)"
  echo "$L TINY2=1  $(CMFREC_HIP_TINY2=1 c4)"
done
} 2>&1 | tee $O/ab.txt
