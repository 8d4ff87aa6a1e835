#!/bin/bash
# round 4, step zo: the final commit on one more box: GPU suite, smoke, default bench line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_zo; mkdir -p $R/$O; cd $R
F='^RCCL\|^HIP ver\|^ROCm ver\|^Hostname\|^Librccl\|amdgpu.ids'
timeout -k 10 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "$F" | tail -3 | tee $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "$F" | tail -1 | tee $O/smoke.log
timeout -k 10 900 python bench.py > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"; tail -1 $O/bench.json | cut -c1-400
