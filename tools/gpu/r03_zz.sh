#!/bin/bash
# round 3, final check of the session: GPU suite, smoke, the default bench line.
export TMPDIR=/tmp
O=gpurun_out/r03_zz; mkdir -p $O
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl'
timeout -k 10 1700 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | tail -4 | tee $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -1 | tee $O/smoke.log
timeout -k 10 900 python bench.py > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"; tail -1 $O/bench.json | cut -c1-600
