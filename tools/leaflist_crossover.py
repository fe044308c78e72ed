"""Leaf list vs ordered walk on scenes of 13-33 leaves (bvh_stress with a handful of spheres).
Compare the default build (RT_LIN_MAX_LEAVES 32: leaf list) with `tools/build_variant.sh lin16 -DRT_LIN_MAX_LEAVES=16`
(ordered BVH walk above 16 leaves): measured +21 % at 21 leaves, +4 % at 27, +9 % at 32 for the leaf list."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as R
out = []
for n in (12, 15, 20, 26, 31):
    hs = R.HostScene(7, 1.5, n=n)
    ds = R.DeviceScene(hs, 0)
    ds.render(1280, 853, 4, 12)
    _, st = ds.render(1280, 853, 64, 12, stats=True)
    best = min(ds.render(1280, 853, 64, 12)[1]["kernel_ms"] for _ in range(2))
    out.append(f"n={n} ({hs.n_objects} objects, nodes/ray {st['nodes_visited'] / st['rays']:.1f}): {st['rays'] / best / 1e3:7.1f}")
print(os.environ.get("RT_HIP_LIBNAME", "librt_hip.so"), " | ".join(out), "Mrays/s", flush=True)
