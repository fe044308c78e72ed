"""End-to-end wall time of rt_render (host accumulator) vs its kernel time, small jobs.  usage: e2e_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as R
for name, sid, asp, n, W, spp, depth in (("C1 random_scene 400x225x100", 0, 16 / 9, 0, 400, 100, 50), ("simple_light 1200x800x500 d5 (reference default)", 3, 1.5, 0, 1200, 500, 5),
                                         ("cornell 256x256x64", 6, 1.0, 0, 256, 64, 50)):
    t0 = time.perf_counter()
    hs = R.HostScene(sid, asp, n=n)
    t1 = time.perf_counter()
    ds = R.DeviceScene(hs, 0)
    t2 = time.perf_counter()
    h = R.image_height(W, asp)
    ds.render(W, h, 1, depth)
    best = (1e9, 0)
    for _ in range(5):
        t3 = time.perf_counter()
        _, st = ds.render(W, h, spp, depth)
        t4 = time.perf_counter()
        best = min(best, (1e3 * (t4 - t3), st["kernel_ms"]))
    print(f"{name}: host build {1e3 * (t1 - t0):.1f} ms, rt_scene_create {1e3 * (t2 - t1):.1f} ms, rt_render wall {best[0]:.2f} ms of which kernels {best[1]:.2f} ms", flush=True)
