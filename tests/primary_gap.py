"""How far is the f32 product walk's primary hit from the oracle's (north_star levels 1-2)?  Per scene: differing ids,
max / p99.9 relative t error, max normal error; details of the worst pixels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

import oracle_ffi as O
import raytracer_amd as R
from conftest import SCENES

widths = [int(a) for a in sys.argv[1:]] or [64, 256]
for name, (sid, asp, n, depth) in SCENES.items():
    tex = R.load_earthmap() if sid in (2, 4, 8, 11) else None
    hs = R.HostScene(sid, asp, n=n)
    ds = R.DeviceScene(hs, 0)
    orc = O.OracleScene(sid, asp, n=n, texels=tex)
    for W in widths:
        h = R.image_height(W, asp)
        for time in (0.25,):
            for fast in (False, True):
                gid, gt, gn, guv, gff = ds.primary_hits(W, h, seed=3, time=time, fast=fast)
                oid, ot, on, ouv, off = orc.primary_hits(W, h, seed=3, time=time)
                diff = gid != oid
                hit = (~diff) & (oid != 0xFFFFFFFF)
                rel = np.abs(gt[hit] - ot[hit]) / np.abs(ot[hit])
                dn = np.abs(gn[hit] - on[hit]).max(axis=-1)
                duv = np.abs(guv[hit] - ouv[hit]).max(axis=-1) if hit.any() else np.zeros(1)
                print(f"{name:20s} W={W:4d} fast={int(fast)} redecided {ds.redecided:5d} ids differ {diff.sum():5d}/{diff.size}  rel_t max {rel.max() if rel.size else 0:.3g} "
                      f"p99.9 {np.quantile(rel, 0.999) if rel.size else 0:.3g} n>1e-5: {(rel > 1e-5).sum()}  dn max {dn.max() if dn.size else 0:.3g} "
                      f"n>1e-5: {(dn > 1e-5).sum()}  duv max {duv.max():.3g} ff differ {(gff[hit] != off[hit]).sum()}", flush=True)
                if diff.any() and not fast:
                    ys, xs = np.nonzero(diff)
                    for y, x in list(zip(ys, xs))[:4]:
                        print(f"      px ({x},{y}): gpu id {gid[y, x]:#x} t {gt[y, x]:.9g}   oracle id {oid[y, x]:#x} t {ot[y, x]:.12g}")
                if rel.size and rel.max() > 1e-5 and not fast:
                    idx = np.argsort(rel)[-3:]
                    ys, xs = np.nonzero(hit)
                    for k in idx:
                        y, x = ys[k], xs[k]
                        print(f"      worst t px ({x},{y}): id {oid[y, x]:#x} gpu t {gt[y, x]:.9g} oracle t {ot[y, x]:.12g} rel {rel[k]:.3g}")
