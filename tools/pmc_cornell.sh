#!/bin/bash
# SQ counters of the Cornell fast-path kernel (whichever the library dispatches), per ray.
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmcc; rm -rf $out; mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --output-format csv -d $out -o pmc -- python3 $PWD/tools/prof_walk.py 6 0 1024 256 1.0 > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $out/b -o pmc -- python3 $PWD/tools/prof_walk.py 6 0 1024 256 1.0 > $out/logb.txt 2>&1 || { tail -5 $out/logb.txt; exit 1; }
python3 - "$out" <<'PY'
import csv,glob,sys
from collections import defaultdict
out=sys.argv[1]
acc=defaultdict(float)
for f in glob.glob(out+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "render" in r["Kernel_Name"]: acc[r["Counter_Name"]]=acc.get(r["Counter_Name"],0)+float(r["Counter_Value"]); name=r["Kernel_Name"][:60]
print(name)
rays=1024*1024*256*2.4307
for k in sorted(acc): print("%-28s %.4g  per 64 rays %.1f"%(k,acc[k],acc[k]*64/rays))
print("lane utilisation %.3f"%(acc["SQ_THREAD_CYCLES_VALU"]/(acc["SQ_ACTIVE_INST_VALU"]*64)))
PY
grep kernel_ms $out/log.txt
