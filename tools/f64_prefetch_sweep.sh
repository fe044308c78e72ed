#!/bin/bash
# f64 Cornell timing under RT_UNIT_PREFETCH / RT_UNIT_SAMPLES settings: tools/f64_prefetch_sweep.sh <spp> "<pf us>" ...
spp=$1; shift
for cfg in "$@"; do set -- $cfg
  echo -n "prefetch $1 unit $2: "; RT_UNIT_PREFETCH=$1 RT_UNIT_SAMPLES=$2 timeout -k 10 120 python3 tools/f64_ab.py $spp || exit $?
done
