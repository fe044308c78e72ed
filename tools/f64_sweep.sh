#!/bin/bash
# f64 kernel on Cornell 1024^2: parity subset, then kernel time at 256 spp and at the metric's 4096 spp over unit sizes.
set -o pipefail
out=gpurun_out/${1:-f64sweep}; mkdir -p $out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_goldens.py tests/test_mixed_scenes.py -m gpu -q -x -k "f64" --timeout 300 --timeout-method thread > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python3 tools/f64_ab.py 256 || exit $?
for u in ${UNITS:-16 32 64 128 256}; do
  echo -n "4096 spp, unit $u: "; RT_UNIT_SAMPLES=$u timeout -k 10 200 python3 tools/f64_ab.py 4096 || exit $?
done
echo -n "f32 headline: "; timeout -k 10 200 python3 bench.py --precision f32 --no-cpu-baseline --no-peer-leg --no-other-configs --steps 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value'],1),'Mrays/s', round(d['ms_per_step'],1),'ms/step')"
