#!/bin/bash
# Builds an A/B variant of librt_hip.so with extra -D flags: tools/build_variant.sh <name> [flags...]
# -> raytracer_amd/lib/librt_hip_<name>.so ; select it with RT_HIP_LIBNAME=librt_hip_<name>.so
set -e
name=$1; shift
cd "$(dirname "$0")/../raytracer_amd/csrc"
tmp=../lib/obj/var_$name; mkdir -p $tmp
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -fno-slp-vectorize $*"
/opt/rocm/bin/hipcc $F -ffp-contract=${F32_CONTRACT:-on} -c rt_f32.hip -o $tmp/rt_f32.o &
/opt/rocm/bin/hipcc $F -ffp-contract=off -c rt_f64.hip -o $tmp/rt_f64.o &
/opt/rocm/bin/hipcc $F -c rt_api.hip -o $tmp/rt_api.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/librt_hip_$name.so $tmp/rt_api.o $tmp/rt_f32.o $tmp/rt_f64.o
echo built ../lib/librt_hip_$name.so
