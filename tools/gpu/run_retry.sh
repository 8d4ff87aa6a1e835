#!/bin/bash
# tools/gpu/run_retry.sh <timeout_s> <script>: gpurun with retries while no GPU slot / box is free (exit code 3 = nothing charged)
t=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $t -- "bash $*"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
