"""For generated scenes that tests/sweep_mixed.py flags with "f32/f64 totals": is there a DIELECTRIC cube (the reference's
Cube::hit knife-edge), and does the difference sit on its pixels?
usage: classify_dev.py <seed>:<n> [<seed>:<n> ...]   (on the GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import raytracer_amd as R  # noqa: E402

W, A = 64, 1.5
h = R.image_height(W, A)
earth = R.load_earthmap()


def arr(p, n):
    return np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, dtype=np.int64)


for spec in sys.argv[1:]:
    seed, n = (int(x) for x in spec.split(":"))
    hs = R.HostScene(9, A, n=n, scene_seed=seed, texels=earth if seed % 2 == 0 else None)
    ds = R.DeviceScene(hs, 0)
    d = hs.desc
    mt = arr(d.mat_tag, d.n_materials)
    cube_mat, cube_obj = arr(d.cube_material, d.n_cubes), arr(d.cube_object, d.n_cubes)
    xb, xo = arr(d.complex_body, d.n_complex), arr(d.complex_object, d.n_complex)
    glass = set()
    for i in range(d.n_cubes):
        if cube_mat[i] < len(mt) and mt[cube_mat[i]] == 1:
            glass.add(int(cube_obj[i]))
    x, _ = ds.render(W, h, 512, 3, seed=seed)
    y, _ = ds.render(W, h, 512, 3, seed=seed, f64=True)
    m = np.isfinite(x).all(-1) & np.isfinite(y).all(-1)
    cx, cy = np.clip(x, 0, 2048), np.clip(y, 0, 2048)
    diff = (cx - cy).sum(-1)
    diff[~m] = 0
    pid = ds.primary_hits(W, h, seed=seed, time=0.0, f64=True)[0]
    per = {int(i): float(diff[pid == i].sum()) for i in np.unique(pid)}
    top = sorted(per.items(), key=lambda kv: -abs(kv[1]))[:3]
    on_glass = sum(v for k, v in per.items() if k in glass)
    print(f"seed {seed} n {n}: totals {cx[m].sum() / cy[m].sum():.4f}; dielectric cubes {sorted(glass)}; difference by primary "
          f"object {[(k if k != 0xFFFFFFFF else -1, round(v)) for k, v in top]}; on glass cubes {on_glass:.0f} of {diff.sum():.0f}", flush=True)
