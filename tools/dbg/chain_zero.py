import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import raytracer_amd as R
from desc_util import chain_desc
pre, seq = sys.argv[1], sys.argv[2]
if pre != "none":
    hs0 = R.HostScene(int(pre), 1.5)
    ds0 = R.DeviceScene(hs0, 0)
    for kw in ({}, {"f64": True}):
        ds0.render(50, 37, 16, 8, seed=4, stats=True, **kw)
    del ds0
view = R.HostScene(5, 1.5)
d, keep = chain_desc(64)
ds = R.DeviceScene(types.SimpleNamespace(desc=d, camera=view.camera), 0)
w, h = 96, 64
for ch in seq:
    if ch == "f":
        ds.primary_hits(w, h, fast=True); continue
    if ch == "d":
        ds.primary_hits(w, h, f64=True); continue
    kw = {"a": {"force_bvh": True}, "b": {"f64": True}, "c": {}, "p": None}[ch]
    if kw is None:
        ds.primary_hits(w, h); continue
    img, st = ds.render(w, h, 4, 6, seed=2, stats=True, **kw)
    print(pre, seq, ch, "sum", float(np.nansum(img)), "rays", st["rays"], "nodes", st["nodes_visited"], flush=True)
