#!/bin/bash
# Runs on the GPU box (through gpurun).  Kernel-trace stats of the default bench command, then
# separate PMC passes (never combined with tracing).  Output under gpurun_out/prof_<tag>/.
# usage: tools/profile.sh <tag> [bench args...]
set -o pipefail
tag=${1:-r01}; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
BENCH="$PWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-peer-leg --no-other-configs $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python3 $BENCH > $out/trace.log 2>&1 || { echo "trace failed"; tail -5 $out/trace.log; exit 1; }
# PMC passes on a short run (counters are per-dispatch; rates per ray are spp-independent)
SHORT="$PWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp 256 $*"
pass() { # name counters...
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -o pmc -- python3 $SHORT > $out/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -3 $out/pmc_$name.log; return 1; }
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY &&
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM &&
pass tcc1 TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum &&
pass tcc2 TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum &&
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
echo "profile done: $out"
ls $out $out/trace | head -40
