#!/bin/bash
# Register / scratch / LDS use of every kernel in a .hip file of the library (hipcc -Rpass-analysis=kernel-resource-usage).
#   tools/resources.sh [file.hip] [extra hipcc flags...]       default: sparse_align.hip
cd "$(dirname "$0")/../dsdtm_amd/csrc" || exit 1
f=${1:-sparse_align.hip}; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -x hip -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[.*/,"",name)}
       /     VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /Occupancy/ {o=$(NF-1)}
       /LDS Size/ {l=$(NF-1); printf "%-90s VGPRs %3s AGPRs %3s scratch %4s B/lane occupancy %s LDS %s\n", name, v, a, s, o, l}' | sort
