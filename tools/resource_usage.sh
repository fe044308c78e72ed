#!/bin/bash
# Per-kernel register / scratch / LDS usage of one translation unit (hipcc remarks), one line per kernel.
#   tools/resource_usage.sh f64 [extra hipcc flags]      tools/resource_usage.sh f32
set -e
unit=${1:-f32}; shift || true
cd "$(dirname "$0")/../raytracer_amd/csrc"
contract=on; [ "$unit" = f64 ] && contract=off
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -ffp-contract=$contract \
  -Rpass-analysis=kernel-resource-usage "$@" -c rt_$unit.hip -o /tmp/rt_${unit}_ru.o 2>&1 |
awk '/Function Name:/ {for(i=1;i<=NF;i++) if($i=="Name:") name=$(i+1)} / VGPRs:/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /ScratchSize/ {s=$(NF-1)} /TotalSGPRs/ {sg=$(NF-1)} /Occupancy/ {o=$(NF-1)} /LDS Size/ {l=$(NF-1); cmd="c++filt -p " name; cmd | getline d; close(cmd); printf "%-90s vgpr %3s agpr %3s sgpr %3s scratch %4s occ %s lds %s\n", substr(d,1,90), v, a, sg, s, o, l}'
