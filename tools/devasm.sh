#!/bin/bash
# tools/devasm.sh <file.hip> [extra flags]: device-only compile of one translation unit to /tmp/dis/<name>.s and a table of
# registers / scratch / occupancy per kernel (what to look at before spending a GPU call on a new kernel)
set -e
src=$1; shift
name=$(basename "$src" .hip)
mkdir -p /tmp/dis
cd "$(dirname "$src")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DROCBLAS_NO_DEPRECATED_WARNINGS "$@" --cuda-device-only -S "$(basename "$src")" -o /tmp/dis/$name.s 2>&1 | grep -v "argument unused" | cut -c1-240 | head -20
awk '/^_ZN6cmfhip.*:$/ {n=$1} /^; (NumVgprs|NumAgprs|ScratchSize|Occupancy):/ {printf "%s %s  ", $2, $3} /^; Occupancy/ {print substr(n, 11, 60)}' /tmp/dis/$name.s
