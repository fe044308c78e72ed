"""Phase stamps (s_memtime) of the refill kernel: RT_HIP_LIBNAME=librt_hip_stamps.so RT_PRINT_STAMPS=1"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as R
for name, sid, asp, n, spp in (("random", 0, 16 / 9, 0, 32), ("final", 8, 1.0, 0, 32), ("stress1M", 7, 16 / 9, 1000000, 4)):
    ds = R.DeviceScene(R.HostScene(sid, asp, n=n), 0)
    h = R.image_height(1920, asp)
    print(name, flush=True)
    sys.stderr.flush()
    _, st = ds.render(1920, h, spp, 50, stats=True)
    print(f"   kernel_ms {st['kernel_ms']:.1f} rays {st['rays']} boxes/ray {st['nodes_visited']/st['rays']:.1f}", flush=True)
