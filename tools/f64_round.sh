#!/bin/bash
# f64 kernel A/B on the GPU box: parity of both f64 walks, then timings of the variants named on the command line.
# usage: tools/f64_round.sh <tag> [spp] [lib ...]
set -o pipefail
tag=$1; spp=${2:-256}; shift; shift
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_goldens.py tests/test_mixed_scenes.py -m gpu -q -x -k "f64" --timeout 300 --timeout-method thread > $out/pytest.log 2>&1; rc=$?
tail -5 $out/pytest.log
[ $rc -ne 0 ] && exit $rc
echo "reference-order walk (RT_F64_NO_LEAVES=1):"
RT_F64_NO_LEAVES=1 timeout -k 10 120 python3 tools/f64_ab.py $spp || exit $?
for lib in librt_hip.so "$@"; do
  RT_HIP_LIBNAME=$lib timeout -k 10 120 python3 tools/f64_ab.py $spp || exit $?
done
