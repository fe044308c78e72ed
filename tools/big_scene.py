"""Scale check: bvh_stress with n spheres (default 8 M) — build times, memory, product walk vs reference-order walk.
usage: big_scene.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as R
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000000
t0 = time.perf_counter()
hs = R.HostScene(7, 16 / 9, n=n)
t1 = time.perf_counter()
ds = R.DeviceScene(hs, 0)
t2 = time.perf_counter()
print(f"n={n}: host build {t1 - t0:.1f} s, rt_scene_create {t2 - t1:.1f} s, nodes {hs.desc.n_nodes}, height {hs.desc.bvh_height}", flush=True)
w, h = 640, 360
a, sa = ds.render(w, h, 2, 50, seed=3, stats=True)
print(f"product walk: {sa['rays']} rays, {sa['kernel_ms']:.1f} ms, boxes/ray {sa['nodes_visited'] / sa['rays']:.1f}", flush=True)
b, sb = ds.render(w, h, 2, 50, seed=3, stats=True, force_bvh=True)
print(f"reference-order walk: {sb['rays']} rays, {sb['kernel_ms']:.1f} ms, boxes/ray {sb['nodes_visited'] / sb['rays']:.1f}", flush=True)
fin = np.isfinite(a) & np.isfinite(b)
print("equal ray counts:", sa["rays"] == sb["rays"], " max rel diff of pixel sums:", float(np.max(np.abs(a - b)[fin] / (1e-6 + np.abs(b[fin])))), flush=True)
ids_f, *_ = ds.primary_hits(w, h, fast=True)
ids_r, *_ = ds.primary_hits(w, h, f64=True)
print("primary ids equal (product walk vs f64 reference walk):", bool(np.array_equal(ids_f, ids_r)), flush=True)
