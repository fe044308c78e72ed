"""Wide sweep of generated scenes (mixed_scene) through every parity level; prints only deviations.
Not collected by pytest (it uses the oracle, so it lives under tests/).
usage: python tests/sweep_mixed.py <first_seed> <last_seed>   (on the GPU box)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, raytracer_amd as R, oracle_ffi as O
W, A = 64, 1.5
h = R.image_height(W, A)
earth = R.load_earthmap()
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2]) + 1):
    for n in (5, 12, 40, 150):
        tex = earth if seed % 2 == 0 else None
        hs = R.HostScene(9, A, n=n, scene_seed=seed, texels=tex); ds = R.DeviceScene(hs, 0)
        os_ = O.OracleScene(9, A, n=n, scene_seed=seed, texels=tex)
        msgs = []
        for time in (0.0, 0.45):
            oid, ot, on, ouv, off = os_.primary_hits(W, h, seed=seed, time=time)
            gid, gt, gn, guv, gff = ds.primary_hits(W, h, seed=seed, time=time, f64=True)
            hit = oid != 0xFFFFFFFF
            if not np.array_equal(gid, oid): msgs.append(f"f64 ids differ {(gid != oid).sum()}")
            elif not (np.allclose(gt[hit], ot[hit], rtol=1e-12, atol=0) and np.allclose(gn[hit], on[hit], rtol=0, atol=1e-12) and np.allclose(guv[hit], ouv[hit], rtol=0, atol=1e-12) and np.array_equal(gff[hit], off[hit])):
                msgs.append("f64 t/n/uv/ff differ")
            f = ds.primary_hits(W, h, seed=seed, time=time, f64=False)
            a = ds.primary_hits(W, h, seed=seed, time=time, fast=True)
            if (f[0] != oid).mean() > 0.005: msgs.append(f"f32 ids differ {(f[0] != oid).sum()}")
            ok = (f[0] == oid) & hit
            rel = np.abs(f[1][ok] - ot[ok]) / np.abs(ot[ok])
            if rel.size and (np.quantile(rel, 0.995) > 1e-5 or rel.max() > 1e-3): msgs.append(f"f32 t rel p99.5 {np.quantile(rel, 0.995):.2g} max {rel.max():.2g}")
            fh = f[0] != 0xFFFFFFFF
            if not (np.array_equal(a[0], f[0]) and np.array_equal(a[1][fh], f[1][fh]) and np.array_equal(a[2][fh], f[2][fh])): msgs.append(f"fast walk differs: ids {(a[0] != f[0]).sum()} t {(a[1][fh] != f[1][fh]).sum()}")
        o, ocnt, _ = os_.render(W, h, 4, 10, seed=seed)
        for force_bvh in (True, False):  # the reference-order f64 walk, then the default one (leaf list / ordered walk)
            tag = "f64 ref-order" if force_bvh else "f64 default"
            g, gst = ds.render(W, h, 4, 10, seed=seed, stats=True, f64=True, force_bvh=force_bvh)
            if (np.isfinite(o) == np.isfinite(g)).mean() < 1.0: msgs.append(f"{tag}: finite masks differ {(np.isfinite(o) != np.isfinite(g)).sum()}")
            fin = np.isfinite(o) & np.isfinite(g)
            err = np.abs(g - o)[fin] / (1 + np.abs(o[fin]))
            if (err > 1e-9).mean() > 1e-3: msgs.append(f"{tag}: image differs in {(err > 1e-9).mean():.4f}")
            if gst["rays"] != ocnt["rays"] and abs(gst["rays"] - ocnt["rays"]) > 1e-3 * ocnt["rays"]: msgs.append(f"{tag}: rays {gst['rays']} vs {ocnt['rays']}")
        for d in (3,):
            x, _ = ds.render(W, h, 64, d, seed=seed)
            y, _ = ds.render(W, h, 64, d, seed=seed, f64=True)
            z, _ = ds.render(W, h, 64, d, seed=seed, force_bvh=True)
            m = np.isfinite(x).all(-1) & np.isfinite(y).all(-1) & np.isfinite(z).all(-1)
            tx, ty, tz = (float(np.clip(v[m], 0, 256).sum()) for v in (x, y, z))
            if abs(tx / ty - 1) > 0.01: msgs.append(f"f32/f64 totals {tx / ty:.4f}")
            if abs(tx / tz - 1) > 0.005: msgs.append(f"f32 walks totals {tx / tz:.4f}")
        if msgs:
            bad += 1
            print(f"seed {seed} n {n}:", "; ".join(msgs), flush=True)
    if seed % 10 == 0: print("... seed", seed, flush=True)
print("deviating scenes:", bad)
