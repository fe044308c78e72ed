"""Debug helper for tests/test_mixed_scenes.py (uses the oracle, so it lives under tests/; not collected by pytest):
f64 audit kernel vs oracle on one generated scene, per depth, listing the primary objects of deviating pixels.
usage: python tests/debug_mixed.py <seed> <n>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, raytracer_amd as R, oracle_ffi as O
seed, n = int(sys.argv[1]), int(sys.argv[2])
W, A = 64, 1.5
hs = R.HostScene(9, A, n=n, scene_seed=seed); ds = R.DeviceScene(hs, 0); os_ = O.OracleScene(9, A, n=n, scene_seed=seed)
h = R.image_height(W, A)
oid, ot, on, ouv, off = os_.primary_hits(W, h, seed=3, time=0.0)
draws = O.scene_draws(seed, 0, 12 * n)
shape = (draws[0::12] * 14).astype(int); mat = (draws[1::12] * 9).astype(int)
names = {0: "ground"}
for i in range(n): names[i + 1] = f"obj{i} shape{shape[i]} mat{mat[i]}"
for d in (1, 2, 3, 12):
    g, gst = ds.render(W, h, 8, d, seed=11, sample_offset=5, stats=True, f64=True)
    o, ocnt, _ = os_.render(W, h, 8, d, seed=11, sample_offset=5)
    print("depth", d, "finite agree", (np.isfinite(o) == np.isfinite(g)).mean(), "nonfinite g", (~np.isfinite(g)).sum(), "o", (~np.isfinite(o)).sum())
    for k in ("rays", "nodes_visited", "sphere_tests", "cube_tests", "closest_hits", "diffuse_vertices", "specular_vertices", "emitter_hits", "misses", "light_probes", "medium_probes"):
        if gst[k] != ocnt[k]: print("   ", k, gst[k], ocnt[k])
    fin = np.isfinite(o) & np.isfinite(g)
    err = np.where(fin, np.abs(g - o) / (1 + np.abs(o)), 1.0).max(axis=-1)
    bad = err > 1e-9
    print("   bad pixels", bad.sum())
    for obj in np.unique(oid[bad]):
        m = bad & (oid == obj)
        print("      primary object", obj, names.get(int(obj), "light/other"), "bad", m.sum(), "of", (oid == obj).sum())
