#!/bin/bash
# One SQ pass of the Cornell kernel per setting: VALU / SALU / LDS / VMEM wave-instructions per 64 rays, lane utilisation.
# usage: tools/pmc_cornell_ab.sh <f32|f64> <spp> "<ENV=..> <ENV=..>" ...   (each argument: environment of one run)
export TMPDIR=/tmp
prec=$1; spp=$2; shift; shift
rps=5.134; [ $prec = f32 ] && rps=2.431
i=0
for envs in "$@"; do
  i=$((i+1)); out=$PWD/gpurun_out/pmcc_$i; rm -rf $out; mkdir -p $out
  P="python3 $PWD/tools/prof_walk.py 6 0 1024 $spp 1.0 $prec"
  env $envs TMPDIR=/tmp true || exit 1
  ( export $envs; timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $out -o pmc -- $P > $out/log.txt 2>&1 ) || { tail -5 $out/log.txt; exit 1; }
  python3 - "$out" "$spp" "$envs" "$rps" <<'PY'
import csv,glob,sys
out=sys.argv[1]; spp=int(sys.argv[2]); rps=float(sys.argv[4])
acc={}
for f in glob.glob(out+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "render" in r["Kernel_Name"]: acc[r["Counter_Name"]]=acc.get(r["Counter_Name"],0)+float(r["Counter_Value"])
rays=1024*1024*spp*rps
ms=[l for l in open(out+"/log.txt") if "kernel_ms" in l][-1].split()[-1]
print(sys.argv[3], "ms", ms, " per 64 rays: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.1f"%tuple(acc[k]*64/rays for k in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_INSTS_VMEM_RD")),
      " lane util %.3f wait %.3f"%(acc["SQ_THREAD_CYCLES_VALU"]/(acc["SQ_ACTIVE_INST_VALU"]*64), acc["SQ_WAIT_ANY"]/acc["SQ_WAVE_CYCLES"]), flush=True)
PY
done
