"""For a generated scene where the f64 kernel's NaN pixels differ from the oracle's (tests/sweep_mixed.py: "finite masks
differ"): which sample, and from which path depth on.  usage: nan_probe.py <seed> <n> [width spp depth render_seed]   (on the GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_ffi as O  # noqa: E402
import raytracer_amd as R  # noqa: E402

seed, n = int(sys.argv[1]), int(sys.argv[2])
W, SPP, DEPTH, RSEED = (int(x) for x in sys.argv[3:7]) if len(sys.argv) > 6 else (64, 4, 10, seed)
A = 1.5
h = R.image_height(W, A)
tex = R.load_earthmap() if seed % 2 == 0 else None
hs = R.HostScene(9, A, n=n, scene_seed=seed, texels=tex)
ds = R.DeviceScene(hs, 0)
os_ = O.OracleScene(9, A, n=n, scene_seed=seed, texels=tex)
g, gst = ds.render(W, h, SPP, DEPTH, seed=RSEED, stats=True, f64=True)
o, ocnt, _ = os_.render(W, h, SPP, DEPTH, seed=RSEED)
bad = np.argwhere((np.isfinite(o) != np.isfinite(g)).any(-1))
print("pixels whose finiteness differs (y, x):", bad.tolist())
print("counters that differ:", {k: (gst[k], ocnt[k]) for k in gst if k in ocnt and gst[k] != ocnt[k] and "visits" not in k})
for py, px in bad:
    for k in range(SPP):
        shown = 0
        for d in range(1, DEPTH + 1):
            gg = ds.render(W, h, 1, d, seed=RSEED, sample_offset=k, f64=True)[0][py, px]
            oo = os_.render(W, h, 1, d, seed=RSEED, sample_offset=k)[0][py, px]
            if not np.allclose(gg, oo, rtol=1e-9, atol=1e-12, equal_nan=True) and shown < 3:
                shown += 1
                print(f"pixel ({px}, {py}) sample {k} depth {d}: gpu {gg} oracle {oo}")
