"""Fast-path throughput on the BVH-bound scenes for the library named by RT_HIP_LIBNAME.
usage: walk_ab.py [W] [spp]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as R
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
out = []
for name, sid, asp, n, depth in (("random", 0, 16 / 9, 0, 50), ("final", 8, 1.0, 0, 50),
                                 ("stress20k", 7, 16 / 9, 20000, 50), ("stress1M", 7, 16 / 9, 1000000, 50)):
    ds = R.DeviceScene(R.HostScene(sid, asp, n=n), 0)
    h = R.image_height(W, asp)
    ds.render(W, h, 1, depth)
    _, st = ds.render(W, h, spp, depth, stats=True)
    best = min(ds.render(W, h, spp, depth)[1]["kernel_ms"] for _ in range(2))
    out.append(f"{name} {st['rays'] / best / 1e3:7.1f}")
print(os.environ.get("RT_HIP_LIBNAME", "librt_hip.so"), " | ".join(out), "Mrays/s", flush=True)
