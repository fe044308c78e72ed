#!/bin/bash
# Static resource usage of the f64 (or f32) kernels under extra flags: tools/f64_meta.sh <f32|f64> <name-substring> [flags...]
unit=$1; pat=$2; shift; shift
cd "$(dirname "$0")/../raytracer_amd/csrc"
c=off; [ $unit = f32 ] && c=on
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -ffp-contract=$c -S --cuda-device-only rt_$unit.hip -o /tmp/meta_$unit.s "$@" 2>/dev/null || exit 1
awk -v pat="$pat" '/\.name:/{f=index($0,pat)>0; if(f)print} f&&/(vgpr_count|private_segment_fixed_size|vgpr_spill|sgpr_spill)/{printf "%s ", $0} f&&/\.wavefront_size/{print ""; f=0}' /tmp/meta_$unit.s
