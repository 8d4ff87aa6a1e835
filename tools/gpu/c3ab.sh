#!/bin/bash
# tools/gpu/c3ab.sh <libdirA> <libdirB>: config 3 (closed form, k = 128, double) on two builds, alternating on one box; first the tests of
# the closed-form widths
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/${TAG:-c3ab}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_config_widths.py -m gpu -q -x -k "lowrank or c3_width" 2>&1 | tail -4 | tee $O/pytest.log
for rep in 1 2 3; do for L in "$@"; do
  CMFREC_HIP_LIBDIR=$R/cmfrec_amd/$L python $R/bench.py --no-cpu-baseline --workload c3 --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
l = [x for x in sys.stdin if x.startswith('{')]
d = json.loads(l[-1]) if l else {}
print('c3 $L', d.get('ms_per_iteration'), d.get('halfstep_ms'))" | tee -a $O/lines.txt
done; done
