#!/bin/bash
# tools/gpu/c3ov.sh: config 3 -- the rank-k kernel on fewer CUs (does it slow down?), then the two kernels of consecutive batches
# side by side for several splits of the CUs (CMFREC_HIP_WAVE_OVERLAP)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/${TAG:-c3ov}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_config_widths.py -m gpu -q -x -k "lowrank or c3_width" 2>&1 | tail -4 | tee $O/pytest.log
line() { python -c "
import sys, json
l = [x for x in sys.stdin if x.startswith('{')]
d = json.loads(l[-1]) if l else {}
print('c3 $1', d.get('ms_per_iteration'), d.get('halfstep_ms'))" | tee -a $O/lines.txt; }
B="python $R/bench.py --no-cpu-baseline --workload c3 --steps 10 --warmup 3"
for p in 0 192 128; do CMFREC_HIP_WAVE_OVERLAP=0 CMFREC_HIP_WAVE_PROD_CUS=$p $B 2>/dev/null | line "sequential, rank-k kernel on $p CUs (0 = all)"; done
for rep in 1 2; do for ov in 0 64 96 112 128 144 160; do
  CMFREC_HIP_WAVE_OVERLAP=$ov $B 2>/dev/null | line "overlap=$ov"
done; done
for b in 4096 16384; do CMFREC_HIP_WAVE_OVERLAP=112 CMFREC_HIP_WAVE_BATCH=$b $B 2>/dev/null | line "overlap=112 batch=$b"; done
