"""Generated scenes at the FULL path depth (50): the f64 kernel against the oracle — NaN masks, images, ray counts; prints
only deviations.  Not collected by pytest.  usage: python tests/sweep_deep.py <first_seed> <last_seed>   (on the GPU box)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, raytracer_amd as R, oracle_ffi as O
W, A = 48, 1.5
h = R.image_height(W, A)
earth = R.load_earthmap()
bad = 0; total = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2]) + 1):
    for n in (12, 150):
        tex = earth if seed % 2 == 0 else None
        hs = R.HostScene(9, A, n=n, scene_seed=seed, texels=tex); ds = R.DeviceScene(hs, 0)
        os_ = O.OracleScene(9, A, n=n, scene_seed=seed, texels=tex)
        g, gst = ds.render(W, h, 3, 50, seed=seed + 7, stats=True, f64=True)
        o, ocnt, _ = os_.render(W, h, 3, 50, seed=seed + 7)
        total += 1
        msgs = []
        nm = (np.isfinite(o) != np.isfinite(g)).any(-1).sum()
        if nm: msgs.append(f"finite masks differ in {nm} px")
        fin = np.isfinite(o) & np.isfinite(g)
        err = np.abs(g - o)[fin] / (1 + np.abs(o[fin]))
        if (err > 1e-9).mean() > 3e-3: msgs.append(f"image differs in {(err > 1e-9).mean():.4f}")
        if abs(gst["rays"] - ocnt["rays"]) > 2e-3 * ocnt["rays"]: msgs.append(f"rays {gst['rays']} vs {ocnt['rays']}")
        if msgs:
            bad += 1; print(f"seed {seed} n {n}:", "; ".join(msgs), flush=True)
print("deviating:", bad, "of", total)
