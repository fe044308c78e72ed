"""One fast-path render of a BVH-bound scene, for rocprofv3 (no oracle, no reference-order pass).
usage: prof_walk.py <scene id> <n> <W> <spp> [aspect] [f64]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as R
sid, n, W, spp = (int(x) for x in sys.argv[1:5])
asp = float(sys.argv[5]) if len(sys.argv) > 5 else 16 / 9
ds = R.DeviceScene(R.HostScene(sid, asp, n=n), 0)
h = R.image_height(W, asp)
f64 = len(sys.argv) > 6 and sys.argv[6] == "f64"
_, st = ds.render(W, h, spp, 50, stats=False, f64=f64)
print(f"scene {sid} n={n} {W}x{h} {spp}spp kernel_ms {st['kernel_ms']:.2f}", flush=True)
