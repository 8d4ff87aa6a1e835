#!/bin/bash
# round 2, step k: packed initial matrices in the second Cholesky kernel; FETCH_SIZE calibration; new distributed tests
export TMPDIR=/tmp
O=gpurun_out/r02_k; mkdir -p $O
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -k "widths or distributed or golden" > $O/pytest1.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -6 $O/pytest1.log | cut -c1-300 | tee -a $O/summary.txt
echo "== c3" | tee -a $O/summary.txt
timeout 600 python bench.py --workload c3 --steps 5 --warmup 2 2>$O/c3.err | tail -1 | cut -c1-170 | tee -a $O/summary.txt
echo "== fetch calibration" | tee -a $O/summary.txt
cd /tmp; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/pmc_calib -- $R/tools/microbench/fetch_calib > $R/$O/fetch_calib.out 2>$R/$O/fetch_calib.err
cd $R; python tools/fetch_calibration.py $O/pmc_calib $O/fetch_calib.out -o $O/fetch_calibration.json 2>&1 | tail -30 | tee -a $O/summary.txt
