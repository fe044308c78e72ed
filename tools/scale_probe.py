"""Scene-size sweep of bvh_stress: host build time, device upload time and render time per sphere count.
usage: python tools/scale_probe.py <n> [<n> ...]   (appends to gpurun_out/scale_probe.log)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as R
log = open("gpurun_out/scale_probe.log", "a")
def P(*a):
    print(*a, flush=True); print(*a, file=log, flush=True)
w, h = 960, 540
for n in [int(x) for x in sys.argv[1:]]:
    t = time.time(); hs = R.HostScene(7, 16/9, n=n); t1 = time.time() - t
    t = time.time(); ds = R.DeviceScene(hs, 0); t2 = time.time() - t
    for spp in (1, 8):
        t = time.time(); acc, st = ds.render(w, h, spp, 50, stats=True); t3 = time.time() - t
        P(f"n={n} host {t1:.2f}s upload {t2:.2f}s spp={spp} render {t3:.2f}s kernel {st['kernel_ms']:.1f}ms rays {st['rays']} Mrays/s {st['rays']/st['kernel_ms']/1e3:.1f} nodes/ray {st['nodes_visited']/st['rays']:.1f} tests/ray {st['sphere_tests']/st['rays']:.1f}")
