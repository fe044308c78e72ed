set -o pipefail
for tw in 98304 196608 393216 786432 1572864; do RT_TARGET_WAVES=$tw timeout -k 10 100 python -u tools/configs_ab.py C3,C1big || exit 1; done
for tw in 15000 20000 30000 40000; do RT_TARGET_WAVES=$tw RT_REFILL_EIGHTHS=4 RT_LEAF_SHIFT=1 timeout -k 10 100 python -u tools/configs_ab.py C1 || exit 1; done
