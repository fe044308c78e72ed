#!/bin/bash
# L2<->fabric traffic of the default bench workload (full 4096-spp launch), two TCC passes.
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmct; rm -rf $out; mkdir -p $out
B="$PWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-f64-leg --no-other-configs $*"
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/a -o pmc -- python3 $B > $out/a.log 2>&1 || { tail -3 $out/a.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum --output-format csv -d $out/b -o pmc -- python3 $B > $out/b.log 2>&1 || { tail -3 $out/b.log; exit 1; }
python3 - $out <<'PY'
import csv,glob,sys
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(float)); n=defaultdict(int)
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "render" in r["Kernel_Name"]:
            k=r["Kernel_Name"].split("(")[0].replace("void rt::","")
            acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k in acc:
    print(k)
    for c,v in sorted(acc[k].items()): print("   %-26s %.5g"%(c,v))
PY
