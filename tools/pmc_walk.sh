#!/bin/bash
# PMC passes over tools/prof_walk.py (one fast-path render).  usage: tools/pmc_walk.sh <tag> <scene> <n> <W> <spp>
export TMPDIR=/tmp
tag=$1; shift
out=$PWD/gpurun_out/pmcw_$tag; rm -rf $out; mkdir -p $out
pass() { name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o pmc -- python3 $PWD/tools/prof_walk.py $ARGS > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $out/$name.log; return 1; }
}
ARGS="$*"
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY &&
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_INSTS_VALU_TRANS &&
pass tcc1 TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum &&
pass tcc2 TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum &&
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
python3 - $out <<'PY'
import csv,glob,sys
from collections import defaultdict
out=sys.argv[1]
acc=defaultdict(float)
for f in glob.glob(out+"/**/*counter_collection.csv",recursive=True):
    p=f.split("/")[-3] if "pmc_" in f else f
    for r in csv.DictReader(open(f)):
        if "render" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]=acc.get(r["Counter_Name"],0)+float(r["Counter_Value"])
for k in sorted(acc): print("%-32s %.5g"%(k,acc[k]))
g=acc
print("lane utilisation %.3f"%(g["SQ_THREAD_CYCLES_VALU"]/(g["SQ_ACTIVE_INST_VALU"]*64)))
print("wait_any/wave_cycles %.3f  valu_active/wave_cycles %.3f"%(g["SQ_WAIT_ANY"]/g["SQ_WAVE_CYCLES"], g["SQ_ACTIVE_INST_VALU"]/g["SQ_WAVE_CYCLES"]))
print("L2 hit %.3f  L1 hit %.3f"%(g["TCC_HIT_sum"]/g["TCC_REQ_sum"], 1-g["TCP_TCC_READ_REQ_sum"]/g["TCP_TOTAL_CACHE_ACCESSES_sum"]))
print("EA read requests %.4g (x64B = %.4g GB)"%(g["TCC_EA0_RDREQ_sum"], g["TCC_EA0_RDREQ_sum"]*64/1e9))
PY
grep kernel_ms $out/*.log
