export TMPDIR=/tmp; O=gpurun_out/r06_n; mkdir -p $O
one() { python bench.py --no-cpu-baseline --workload c1 --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"c1 $1\", d[\"ms_per_iteration\"], d[\"halfstep_ms\"], d.get(\"frac_of_hbm_peak\"))" | tee -a $O/c1_sweep.txt; }
for rep in 1 2; do
one "default"
CMFREC_HIP_VH=gram one "VH=gram"
CMFREC_HIP_VH=gram CMFREC_HIP_VH_MIN=513 one "VH=gram VH_MIN=513"
CMFREC_HIP_VH=gram CMFREC_HIP_VH_MIN=385 one "VH=gram VH_MIN=385"
CMFREC_HIP_VH=gram CMFREC_HIP_VH_MIN=257 one "VH=gram VH_MIN=257"
CMFREC_HIP_VH_MIN=513 one "VH_MIN=513 (stream)"
done
