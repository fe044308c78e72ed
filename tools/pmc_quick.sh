#!/bin/bash
# One SQ counter pass on a short run; prints per-ray VALU instruction and lane-utilisation figures.
# usage: tools/pmc_quick.sh [bench args]   (env RT_HIP_LIBNAME selects a variant build)
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmcq; rm -rf $out; mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --output-format csv -d $out -o pmc -- python3 $PWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --spp 256 "$@" > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
python3 - "$out" <<'PY'
import csv,glob,sys,json
from collections import defaultdict
out=sys.argv[1]
f=glob.glob(out+"/**/*counter_collection.csv",recursive=True)[0]
acc=defaultdict(float)
for r in csv.DictReader(open(f)):
    if "render_kernel" in r["Kernel_Name"] and "Lb0ELb1" not in r["Kernel_Name"] and "false" in r["Kernel_Name"].split("<")[1]:
        acc[r["Counter_Name"]]+=float(r["Counter_Value"])
line=[l for l in open(out+"/log.txt") if l.startswith("{")][-1]
d=json.loads(line)
rays=d["value"]*1e6*d["ms_per_step"]*1e-3
print("rays",rays,"Mrays/s",round(d["value"],1))
print("VALU wave-instr per 64 rays: %.0f"%(acc["SQ_INSTS_VALU"]*64/rays), " SALU: %.0f"%(acc["SQ_INSTS_SALU"]*64/rays))
print("lane utilisation: %.3f"%(acc["SQ_THREAD_CYCLES_VALU"]/(acc["SQ_ACTIVE_INST_VALU"]*64)))
print("effective VALU lane-instr per ray: %.0f"%(acc["SQ_THREAD_CYCLES_VALU"]/acc["SQ_ACTIVE_INST_VALU"]*acc["SQ_INSTS_VALU"]/rays))
print("wait_any / wave_cycles: %.3f"%(acc["SQ_WAIT_ANY"]/acc["SQ_WAVE_CYCLES"]), " active_valu/wave_cycles: %.3f"%(acc["SQ_ACTIVE_INST_VALU"]/acc["SQ_WAVE_CYCLES"]))
print("active_inst_valu (quad-cycles) per SIMD / busy: %.3f"%(acc["SQ_ACTIVE_INST_VALU"]*4/1024/(acc["SQ_BUSY_CYCLES"]/32)))
PY
