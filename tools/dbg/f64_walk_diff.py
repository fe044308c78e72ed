"""Where do the default f64 walk (leaf list / ordered walk) and the reference-order f64 walk part ways on a generated
scene?  usage: f64_walk_diff.py <seed> <n> [spp] [depth]   (mixed_scene, id 9, as tests/sweep_mixed.py builds it)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_ffi as O
import raytracer_amd as R

seed, n = int(sys.argv[1]), int(sys.argv[2])
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 4
max_depth = int(sys.argv[4]) if len(sys.argv) > 4 else 10
W, A = 64, 1.5
h = R.image_height(W, A)
tex = R.load_earthmap() if seed % 2 == 0 else None
hs = R.HostScene(9, A, n=n, scene_seed=seed, texels=tex)
ds = R.DeviceScene(hs, 0)
os_ = O.OracleScene(9, A, n=n, scene_seed=seed, texels=tex)
d = hs.desc
print(f"seed {seed} n {n}: nodes {d.n_nodes} spheres {d.n_spheres} mspheres {d.n_mspheres} cubes {d.n_cubes} complex {d.n_complex} "
      f"ops {d.n_ops} lights {d.n_lights}")
for depth in sorted({1, 2, 3, max_depth}):
    o, ocnt, _ = os_.render(W, h, spp, depth, seed=seed)
    a, sa = ds.render(W, h, spp, depth, seed=seed, stats=True, f64=True)
    b, sb = ds.render(W, h, spp, depth, seed=seed, stats=True, f64=True, force_bvh=True)
    fin = np.isfinite(a) & np.isfinite(b) & np.isfinite(o)
    ea = np.where(fin, np.abs(a - o) / (1 + np.abs(o)), 0)
    eb = np.where(fin, np.abs(b - o) / (1 + np.abs(o)), 0)
    bad = (ea > 1e-9).any(-1)
    print(f"depth {depth}: default vs oracle {int((ea > 1e-9).sum())} channels off, ref-order vs oracle {int((eb > 1e-9).sum())}; "
          f"rays {sa['rays']} / {sb['rays']} / {ocnt['rays']}; closest {sa['closest_hits']} / {sb['closest_hits']}; "
          f"medium visits {sa['medium_visits']} / {sb['medium_visits']}; nodes {sa['nodes_visited']} / {sb['nodes_visited']}")
    ys, xs = np.nonzero(bad)
    for y, x in list(zip(ys, xs))[:6]:
        print(f"   pixel ({x},{y}): default {a[y, x]} ref-order {b[y, x]} oracle {o[y, x]}")
# primary hits of both walks
for time in (0.0, 0.45):
    gid, gt, *_ = ds.primary_hits(W, h, seed=seed, time=time, f64=True)
    oid, ot, *_ = os_.primary_hits(W, h, seed=seed, time=time)
    print(f"primary f64 ids differ at time {time}: {(gid != oid).sum()}")
