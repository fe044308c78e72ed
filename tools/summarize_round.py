#!/usr/bin/env python3
"""Writes profiles/<tag>_summary.md (+ <tag>_kernel_stats.csv, <tag>_*_traffic.json) from what tools/profile_round.sh
left under gpurun_out/: the kernel trace of the default bench command, the traffic passes, the SQ/TCC/TCP passes of the
BVH-bound configs.  usage: tools/summarize_round.py <tag>"""
import csv
import glob
import json
import os
import re
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
L = [f"# rocprofv3 summary `{tag}` (final build of the round)", "",
     "Command of the kernel trace: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline`",
     f"(the default bench: Cornell 1024^2 x 4096 spp on the f64 kernel = the headline, its f32 peer leg, and the C1 / C3 / C4 / C4-hbm lines with their f64 samples), `tools/profile_round.sh {tag}`;",
     f"this file is written by `tools/summarize_round.py {tag}`.", ""]


def short(name, n=110):
    return name[:n]


stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
    L += ["## kernel-trace --stats", "", "| kernel | calls | total ms | avg ms | % |", "|---|---|---|---|---|"]
    for r in csv.DictReader(open(stats[0])):
        L.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | "
                 f"{float(r['AverageNs']) / 1e6:.4f} | {float(r['Percentage']):.2f} |")
    L.append("")

trace = glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True)
if trace:
    L += ["## per-dispatch durations of the render kernels (kernel trace), in launch order", "",
          "| kernel | grid | LDS B | VGPR (rocprof units) | scratch B | ms |", "|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(trace[0])):
        m = re.search(r"rt::(render_\w+<[^>]*>)", r["Kernel_Name"])
        if not m:
            continue
        grid = f"{int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))} x {r['Grid_Size_Y']} workgroups"
        L.append(f"| `{m.group(1)}` | {grid} | {r['LDS_Block_Size']} | {r['VGPR_Count']} | {r['Scratch_Size']} | "
                 f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:.3f} |")
    L.append("")

traffic = sorted(glob.glob(os.path.join(root, "gpurun_out", "traffic", f"{tag}_*_traffic.json")))
if traffic:
    L += ["## traffic passes (`tools/pmc_traffic.sh`: FETCH_SIZE, WRITE_SIZE, TCC hit/miss/atomic — one `--pmc` run each)", ""]
    for f in traffic:
        shutil.copy(f, os.path.join(dst, os.path.basename(f)))
        d = json.load(open(f))
        L.append(f"* `{d['workload']}` — `{d['kernel']}`: fetch {d['fetch_bytes_per_launch'] / 1e6:.1f} MB, write "
                 f"{d['write_bytes_per_launch'] / 1e6:.1f} MB per launch, L2 hit {d['l2_hit_rate']:.3f}, "
                 f"{d['atomic_requests']:.3g} atomic requests  (`profiles/{os.path.basename(f)}`)")
    L.append("")

walks = [("c1", "C1 random_scene 400x225 x 100 spp"), ("c3", "C3 final_scene 1024^2 x 512 spp"),
         ("c4", "C4 bvh_stress 10^6 spheres 3840x2160 x 4 spp"),
         ("f64", "f64 kernel (the headline's), Cornell 1024^2 x 1024 spp")]
have = [(k, t) for k, t in walks if os.path.exists(os.path.join(src, f"walk_{k}.txt"))]
if have:
    L += ["## SQ / TCC / TCP passes of the BVH-bound configs and of the f64 kernel (`tools/pmc_walk.sh`, one render each, five `--pmc` runs)", ""]
    for k, title in have:
        body = []
        for line in open(os.path.join(src, f"walk_{k}.txt")):
            line = line.rstrip()
            if re.match(r"^(SQ_|TCC_|TCP_|GRBM_|lane util|wait_any|L2 hit|EA read)", line):
                body.append(line)
            elif "kernel_ms" in line and not any("kernel_ms" in b for b in body):
                body.append(line.split(":", 1)[-1])
        kern = "render_queue_kernel<double, ..., TRAV_LEAVES>" if k == "f64" else "render_units_kernel"
        L += [f"### {title} — `{kern}`", "", "```"] + body + ["```", ""]

open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(L) + "\n")
print("\n".join(L[:40]))
