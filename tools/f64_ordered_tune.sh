#!/bin/bash
# f64 ordered walk in the persistent kernel: refill threshold x leaf-phase threshold on C1 / C3 / C4 (kernel ms, best of 3)
for re in 2 4 6 7 8; do for ls in 0 1 2; do
  echo "== refill_eighths $re leaf_shift $ls"
  ONLY_ORDERED=1 RT_REFILL_EIGHTHS=$re RT_LEAF_SHIFT=$ls timeout -k 10 120 python3 tools/f64_big_ab.py C1 C3 C4 2>&1 | grep -v amdgpu || exit 1
done; done
