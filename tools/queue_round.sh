#!/bin/bash
# The queue kernel (RT_QUEUE=1) through the GPU parity tests, then A/B timings against the persistent kernel.
set -o pipefail
tag=${1:-q1}; out=gpurun_out/$tag; mkdir -p $out
RT_QUEUE=1 timeout -k 10 900 python -u -m pytest tests -m gpu -q -x --timeout 600 --timeout-method thread --deselect tests/test_gpu_level3.py > $out/pytest.log 2>&1; rc=$?
tail -8 $out/pytest.log
[ $rc -ne 0 ] && exit $rc
for spp in 256 1024; do
  timeout -k 10 120 python3 tools/f64_ab.py $spp || exit 1
  RT_QUEUE=1 timeout -k 10 120 python3 tools/f64_ab.py $spp || exit 1
done
for q in 0 1; do
  echo -n "f32 RT_QUEUE=$q: "; env RT_QUEUE=$q timeout -k 10 200 python3 bench.py --no-cpu-baseline --precision f32 --steps 3 --warmup 1 --no-other-configs --no-peer-leg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value'],1),'Mrays/s', round(d['ms_per_step'],1),'ms/step', 'rays/sample',round(d['rays_per_sample'],3))" || exit 1
done
