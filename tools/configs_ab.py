"""BASELINE configs C1 / C3 / C4 (bounded spp) on the library named by RT_HIP_LIBNAME: Mrays/s from kernel_ms, best of 3.
usage: configs_ab.py [C1,C3,C4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as R
CFG = {"C1": ("random", 0, 16 / 9, 0, 400, 100), "C3": ("final", 8, 1.0, 0, 1024, 512), "C4": ("stress1M", 7, 16 / 9, 1000000, 3840, 4),
       "C1big": ("random1920", 0, 16 / 9, 0, 1920, 256)}
which = (sys.argv[1] if len(sys.argv) > 1 else "C1,C3,C4").split(",")
out = []
for c in which:
    name, sid, asp, n, W, spp = CFG[c]
    ds = R.DeviceScene(R.HostScene(sid, asp, n=n), 0)
    h = R.image_height(W, asp)
    ds.render(W, h, 1, 50)
    rays = ds.render(W, h, spp, 50, stats=True)[1]["rays"]
    import torch
    acc = torch.zeros((h, W, 3), dtype=torch.float32, device="cuda:0")
    best = min(ds.render_device(acc.data_ptr(), W, h, spp, 50)["kernel_ms"] for _ in range(3))
    out.append(f"{c} {rays / best / 1e3:8.1f} ({best:.2f} ms)")
tags = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("RT_") and k != "RT_HIP_LIBNAME")
print(os.environ.get("RT_HIP_LIBNAME", "librt_hip.so"), tags, " | ".join(out), "Mrays/s", flush=True)
