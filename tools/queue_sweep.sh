#!/bin/bash
# f64 queue kernel: unit-size sweep at a given spp.  usage: tools/queue_sweep.sh <spp> <unit> ...
spp=$1; shift
for u in "$@"; do echo -n "unit $u: "; RT_QUEUE=1 RT_UNIT_SAMPLES=$u timeout -k 10 120 python3 tools/f64_ab.py $spp || exit 1; done
