"""Wide sweep of the SEEDED reference scenes (random_scene 0, random_scene_stdrng 14, bvh_stress 7 at n = 3000, final_scene
8) over scene seeds and BVH-axis seeds, through every parity level; prints only deviations.  The collected tests run
these scenes at the default seeds (5, 7); this looks for rare geometry the defaults do not have.
Not collected by pytest (it uses the oracle, so it lives under tests/).
usage: python tests/sweep_scenes.py <first_seed> <last_seed>   (on the GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_ffi as O  # noqa: E402
import raytracer_amd as R  # noqa: E402

W = 64
earth = R.load_earthmap()
bad = 0
total = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2]) + 1):
    for sid, asp, n in ((0, 16 / 9, 0), (14, 16 / 9, 0), (7, 16 / 9, 3000), (8, 1.0, 0)):
        h = R.image_height(W, asp)
        tex = earth if sid == 8 else None
        hs = R.HostScene(sid, asp, scene_seed=seed, bvh_seed=seed + 100, n=n, texels=tex)
        ds = R.DeviceScene(hs, 0)
        os_ = O.OracleScene(sid, asp, scene_seed=seed, bvh_seed=seed + 100, n=n, texels=tex)
        total += 1
        msgs = []
        for time in (0.0, 0.6):
            oid, ot, on, ouv, off = os_.primary_hits(W, h, seed=seed, time=time)
            gid, gt, gn, guv, gff = ds.primary_hits(W, h, seed=seed, time=time, f64=True)
            hit = oid != 0xFFFFFFFF
            if not np.array_equal(gid, oid):
                msgs.append(f"f64 ids differ {(gid != oid).sum()}")
            elif not (np.allclose(gt[hit], ot[hit], rtol=1e-12, atol=0) and np.allclose(gn[hit], on[hit], rtol=0, atol=1e-12)
                      and np.array_equal(gff[hit], off[hit])):
                msgs.append("f64 t/n/ff differ")
            f = ds.primary_hits(W, h, seed=seed, time=time)
            a = ds.primary_hits(W, h, seed=seed, time=time, fast=True)
            if not np.array_equal(f[0], oid):
                msgs.append(f"f32 ids differ {(f[0] != oid).sum()}")
            ok = (f[0] == oid) & hit
            rel = np.abs(f[1][ok] - ot[ok]) / np.abs(ot[ok])
            if rel.size and rel.max() > 1e-5:
                msgs.append(f"f32 t rel max {rel.max():.2g}")
            fh = f[0] != 0xFFFFFFFF
            if not (np.array_equal(a[0], f[0]) and np.array_equal(a[1][fh], f[1][fh])):
                msgs.append(f"fast walk differs: ids {(a[0] != f[0]).sum()}")
        o, ocnt, _ = os_.render(W, h, 4, 12, seed=seed)
        # both f64 walks: the reference-order walk (every work counter is the oracle's), then the default (ordered walk of
        # the SAH tree / leaf list: same paths, so the same image and outcome counters; traversal counters differ)
        for force_bvh in (True, False):
            tag = "f64 ref-order" if force_bvh else "f64 default"
            g, gst = ds.render(W, h, 4, 12, seed=seed, stats=True, f64=True, force_bvh=force_bvh)
            if (np.isfinite(o) != np.isfinite(g)).any():
                msgs.append(f"{tag}: finite masks differ {(np.isfinite(o) != np.isfinite(g)).sum()}")
            fin = np.isfinite(o) & np.isfinite(g)
            err = np.abs(g - o)[fin] / (1 + np.abs(o[fin]))
            if (err > 1e-9).mean() > 1e-3:
                msgs.append(f"{tag}: image differs in {(err > 1e-9).mean():.4f}")
            for k in ("rays", "closest_hits", "misses") + (("nodes_visited", "sphere_tests") if force_bvh else ()):
                if abs(gst[k] - ocnt[k]) > 1e-3 * max(1, ocnt[k]):
                    msgs.append(f"{tag}: counter {k}: {gst[k]} vs {ocnt[k]}")
        x, sx = ds.render(W, h, 64, 4, seed=seed, stats=True)
        y, _ = ds.render(W, h, 64, 4, seed=seed, f64=True)
        z, sz = ds.render(W, h, 64, 4, seed=seed, stats=True, force_bvh=True)
        m = np.isfinite(x).all(-1) & np.isfinite(y).all(-1) & np.isfinite(z).all(-1)
        tx, ty, tz = (float(np.clip(v[m], 0, 256).sum()) for v in (x, y, z))
        if abs(tx / ty - 1) > 0.005:
            msgs.append(f"f32/f64 totals {tx / ty:.4f}")
        if sx["rays"] != sz["rays"] or abs(tx / tz - 1) > 1e-4:
            msgs.append(f"f32 walks: rays {sx['rays']} vs {sz['rays']}, totals {tx / tz:.6f}")
        if msgs:
            bad += 1
            print(f"seed {seed} scene {sid}:", "; ".join(msgs), flush=True)
    if seed % 5 == 0:
        print("... seed", seed, flush=True)
print("deviating scenes:", bad, "of", total)
