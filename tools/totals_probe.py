import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, raytracer_amd as R
earth = R.load_earthmap(); W = 64
for seed in (5175, 5197, 5209, 5350, 5100):
    hs = R.HostScene(8, 1.0, scene_seed=seed, bvh_seed=seed + 100, texels=earth); ds = R.DeviceScene(hs, 0)
    h = R.image_height(W, 1.0)
    x, sx = ds.render(W, h, 64, 4, seed=seed, stats=True)
    y, _ = ds.render(W, h, 64, 4, seed=seed, f64=True)
    m = np.isfinite(x).all(-1) & np.isfinite(y).all(-1)
    tx, ty = (float(np.clip(v[m], 0, 256).sum()) for v in (x, y))
    print(os.environ.get("RT_HIP_LIBNAME"), seed, f"{tx/ty:.5f}", sx["rays"], float(x[m].sum()), float(y[m].sum()), flush=True)
