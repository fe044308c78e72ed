"""Ordered (near-child-first) walk vs the reference's left-first walk on the BVH-bound scenes:
box tests per ray, leaf tests per ray, kernel time.  Usage: walk_compare.py [W] [spp]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as R
log = open("gpurun_out/walk_compare.log", "a")
def P(*a):
    print(*a, flush=True); print(*a, file=log, flush=True)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 640
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for name, sid, asp, n, depth in (("random_scene", 0, 16 / 9, 0, 50), ("final_scene", 8, 1.0, 0, 50),
                                 ("bvh_stress_20k", 7, 16 / 9, 20000, 50), ("bvh_stress_1M", 7, 16 / 9, 1000000, 50)):
    hs = R.HostScene(sid, asp, n=n)
    ds = R.DeviceScene(hs, 0)
    h = R.image_height(W, asp)
    for force in (False, True):
        ds.render(W, h, 1, depth, force_bvh=force)
        acc, st = ds.render(W, h, spp, depth, stats=True, force_bvh=force)
        _, st2 = ds.render(W, h, spp, depth, stats=False, force_bvh=force)
        P(f"{name:16s} {'reference' if force else 'fast     '} rays {st['rays']:>11d} box/ray {st['nodes_visited']/st['rays']:7.1f} "
          f"sph/ray {st['sphere_tests']/st['rays']:6.1f} kernel {st2['kernel_ms']:9.1f} ms  {st['rays']/st2['kernel_ms']/1e3:8.1f} Mrays/s sum {acc.sum():.6g}")
