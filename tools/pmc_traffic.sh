#!/bin/bash
# Memory-side traffic of ONE launch of a bench workload's dominant kernel: FETCH_SIZE and WRITE_SIZE in separate --pmc
# passes (never combined with tracing), written as profiles-ready JSON: gpurun_out/traffic/<tag>_traffic.json.
# usage: tools/pmc_traffic.sh <tag> <workload key as bench.py prints it> [bench args selecting that workload...]
export TMPDIR=/tmp
tag=$1; key=$2; shift 2
out=$PWD/gpurun_out/traffic/$tag; rm -rf $out; mkdir -p $out
B="$PWD/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-peer-leg --no-other-configs $*"
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o pmc -- python3 $B > $out/fetch.log 2>&1 || { tail -3 $out/fetch.log; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o pmc -- python3 $B > $out/write.log 2>&1 || { tail -3 $out/write.log; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/l2 -o pmc -- python3 $B > $out/l2.log 2>&1 || { tail -3 $out/l2.log; exit 1; }
python3 - $out "$tag" "$key" "$*" <<'PY'
import csv, glob, json, sys
out, tag, key, args = sys.argv[1:5]
def first_render_dispatch(sub):
    """counter values of the FIRST render dispatch of the pass (the timed launch; bench.py's counted pass comes later)"""
    rows = []
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if "render" in r["Kernel_Name"]]
    if not rows:
        raise SystemExit(f"no render dispatch in pass {sub}")
    d0 = min(int(r["Dispatch_Id"]) for r in rows)
    vals = {}
    for r in rows:
        if int(r["Dispatch_Id"]) == d0:
            vals[r["Counter_Name"]] = vals.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return rows[0]["Kernel_Name"].split("(")[0].replace("void rt::", ""), vals
kernel, fetch = first_render_dispatch("fetch")
_, write = first_render_dispatch("write")
_, l2 = first_render_dispatch("l2")
fetch_b, write_b = fetch["FETCH_SIZE"] * 1024.0, write["WRITE_SIZE"] * 1024.0   # both counters are in KiB
rec = {
    "workload": key, "kernel": kernel, "bench_args": args,
    "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
    "bytes_per_launch": fetch_b + write_b,
    "l2_hit_rate": l2["TCC_HIT_sum"] / max(1.0, l2["TCC_HIT_sum"] + l2["TCC_MISS_sum"]),
    "atomic_requests": l2["TCC_EA0_ATOMIC_sum"],
    "note": "FETCH_SIZE + WRITE_SIZE (KiB x 1024) of the first render dispatch, one --pmc pass each "
            "(MI355X_MICROARCH.md, HBM): memory-side requests of the L2s, Infinity-Cache hits included. The read side is "
            "left as counted: the guide's x2 correction is for wide coalesced streams (16 B per lane); this kernel's "
            "reads are scattered 64-byte node and primitive lines, an access width the guide calls uncalibrated. "
            "WRITE_SIZE is exact for the float atomics that make up the writes.",
}
path = f"{out}/../{tag}_traffic.json"
json.dump(rec, open(path, "w"), indent=1)
print(json.dumps(rec))
PY
