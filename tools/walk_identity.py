"""Are the production walk's IMAGES bit-identical to the reference-order walk's (same f32 leaf code, explicit fused forms)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, raytracer_amd as R
for name, sid, asp, n, depth in (("random_scene", 0, 16 / 9, 0, 50), ("final_scene", 8, 1.0, 0, 50), ("bvh_stress_20k", 7, 16 / 9, 20000, 50),
                                 ("bvh_stress_1M", 7, 16 / 9, 1000000, 50), ("cornell_box", 6, 1.0, 0, 50), ("small_scene", 5, 1.5, 0, 5), ("cube_scene", 4, 1.5, 0, 5), ("mixed 3/70", 9, 1.5, 70, 12)):
    hs = R.HostScene(sid, asp, n=n, scene_seed=3 if sid == 9 else 5, texels=R.load_earthmap() if sid in (2, 4, 8) else None)
    ds = R.DeviceScene(hs, 0)
    h = R.image_height(128, asp)
    a, sa = ds.render(128, h, 16, depth, seed=5, stats=True)
    b, sb = ds.render(128, h, 16, depth, seed=5, stats=True, force_bvh=True)
    same = np.isclose(a, b, rtol=1e-5, atol=1e-6, equal_nan=True).all(axis=-1)
    print(f"{name:16s} rays {sa['rays']} vs {sb['rays']}  pixels equal to rounding {same.mean():.5f}  bit-identical {np.array_equal(a, b, equal_nan=True)}", flush=True)
