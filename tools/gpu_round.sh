#!/bin/bash
# One GPU-box round: the WIP test (if any), the GPU suite, a bench line.  Every GPU step under its own time limit, chained.
# usage: tools/gpu_round.sh <tag> [pytest args...]
set -o pipefail
tag=${1:-r02}; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
if [ -f tests/wip_no_alloc.py ]; then
  timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -v --timeout 120 --timeout-method thread > $out/wip.log 2>&1; rc=$?
  tail -15 $out/wip.log
  [ $rc -eq 124 -o $rc -eq 137 -o $rc -eq 134 -o $rc -eq 139 ] && exit $rc
fi
timeout -k 10 1100 python -u -m pytest tests -m gpu -q -rf --timeout 900 --timeout-method thread "$@" > $out/pytest.log 2>&1; rc=$?
tail -40 $out/pytest.log
[ $rc -eq 124 -o $rc -eq 137 -o $rc -eq 134 -o $rc -eq 139 ] && exit $rc
timeout -k 10 400 python -u bench.py > $out/bench.json 2> $out/bench.err; rc=$?
tail -c 3000 $out/bench.json; tail -5 $out/bench.err
exit $rc
