#!/bin/bash
# SQ counters + phase stamps of the f64 kernel on Cornell 1024^2 (whichever walk the library dispatches).
# usage: tools/pmc_f64.sh [spp]   (RT_F64_NO_LEAVES=1 in the environment: the reference-order walk)
export TMPDIR=/tmp
spp=${1:-256}
out=$PWD/gpurun_out/pmc64; rm -rf $out; mkdir -p $out
P="python3 $PWD/tools/prof_walk.py 6 0 1024 $spp 1.0 f64"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --output-format csv -d $out -o pmc -- $P > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_SMEM --output-format csv -d $out/b -o pmc -- $P > $out/logb.txt 2>&1 || { tail -5 $out/logb.txt; exit 1; }
python3 - "$out" "$spp" <<'PY'
import csv,glob,sys
from collections import defaultdict
out=sys.argv[1]; spp=int(sys.argv[2])
acc={}
for f in glob.glob(out+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "render" in r["Kernel_Name"]: acc[r["Counter_Name"]]=acc.get(r["Counter_Name"],0)+float(r["Counter_Value"]); name=r["Kernel_Name"][:70]
print(name)
rays=1024*1024*spp*5.13
for k in sorted(acc): print("%-28s %.4g  per 64 rays %.1f"%(k,acc[k],acc[k]*64/rays))
print("lane utilisation %.3f"%(acc["SQ_THREAD_CYCLES_VALU"]/(acc["SQ_ACTIVE_INST_VALU"]*64)))
print("wait_any / wave_cycles %.3f" % (acc["SQ_WAIT_ANY"]/acc["SQ_WAVE_CYCLES"]))
PY
grep kernel_ms $out/log.txt
if [ -f raytracer_amd/lib/librt_hip_stamps.so ]; then
  RT_HIP_LIBNAME=librt_hip_stamps.so RT_PRINT_STAMPS=1 timeout -k 10 120 $P
fi
