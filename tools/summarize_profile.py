#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (rocprofv3 CSVs) into profiles/<tag>_summary.md + copies the
kernel-stats CSV.  usage: tools/summarize_profile.py <tag>"""
import csv
import glob
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
lines = [f"# rocprofv3 summary `{tag}`", ""]

stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
    lines += ["## kernel-trace --stats (command: `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline`)", "",
              "| kernel | calls | total ms | avg ms | % |", "|---|---|---|---|---|"]
    for r in csv.DictReader(open(stats[0])):
        name = r["Name"][:90]
        lines.append(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | "
                     f"{float(r['AverageNs'])/1e6:.4f} | {float(r['Percentage']):.2f} |")
    lines.append("")
trace = glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True)
if trace:
    rows = list(csv.DictReader(open(trace[0])))
    seen = set()
    for r in rows:  # one entry per render kernel (first dispatch of each)
        kn = r["Kernel_Name"].split("(")[0].replace("void rt::", "")
        if "render" not in kn or kn in seen:
            continue
        seen.add(kn)
        lines += [f"## `{kn}` dispatch resources (kernel trace; VGPR_Count as rocprofv3 reports it, half the compiler's "
                  f"per-lane count for these wave64 kernels: 40 = 80 VGPRs)", ""]
        for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size",
                  "Grid_Size"):
            if k in r:
                lines.append(f"* {k}: {r[k]}")
        lines.append("")

for pmc in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(pmc):
        continue
    files = glob.glob(os.path.join(pmc, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(files[0])):
        if "render" not in r["Kernel_Name"]:
            continue
        kn = r["Kernel_Name"].split("(")[0].replace("void rt::", "")
        acc[kn][r["Counter_Name"]] += float(r["Counter_Value"])
        n[kn][r["Counter_Name"]] += 1
    for kn in sorted(acc):
        lines += [f"## PMC pass `{os.path.basename(pmc)}` — `{kn}`, summed over its dispatches "
                  f"(template args: <R, STATS, EARLY_EXIT, TRAV>; STATS=true is bench.py's counted pass)", "",
                  "| counter | sum | dispatches | per dispatch |", "|---|---|---|---|"]
        for k in sorted(acc[kn]):
            lines.append(f"| {k} | {acc[kn][k]:.6g} | {n[kn][k]} | {acc[kn][k]/max(1,n[kn][k]):.6g} |")
        lines.append("")
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines))
print("\n".join(lines))
