# tools/gpu/c3ab.sh -- the factorisation kernels of the eight-block rows side by side on this box (TAG = output directory under gpurun_out/):
# the C3-width tests, then bench.py --workload c3 with CMFREC_HIP_CHOL_WG = 0 (one wavefront per row) / 4 / 2 (wavefronts per row of the
# workgroup kernel), alternating, then the phases of the workgroup kernel left out one by one (CMFREC_HIP_WAVE_SKIP: results are wrong)
export TMPDIR=/tmp; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_config_widths.py tests/test_gpu_operators.py -q -x -k "c3 or cholesky or chol" 2>&1 | tail -4 > $O/pytest_c3.log
one() { python bench.py --no-cpu-baseline --workload c3 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"c3 $1\", d[\"ms_per_iteration\"], d[\"halfstep_ms\"], d.get(\"executed_frac_of_fp64_peak_78.6\"))" | tee -a $O/c3_ab.txt; }
for rep in 1 2; do for v in 0 4 2; do CMFREC_HIP_CHOL_WG=$v one "WG=$v"; done; done
for v in 2 4 8; do CMFREC_HIP_CHOL_WG=${SKIPWG:-2} CMFREC_HIP_WAVE_SKIP=$v one "WG=${SKIPWG:-2} skip=$v"; done
cat $O/pytest_c3.log
