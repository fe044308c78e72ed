import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as R
hs = R.HostScene(8, 1.0); ds = R.DeviceScene(hs, 0)
W, spp = 64, 1024
for label, kw in (("f32 fast", {}), ("f32 refwalk", {"force_bvh": True}), ("f64", {"f64": True})):
    a, st = ds.render(W, W, spp, 50, seed=21, stats=True, **kw)
    m = a.astype(np.float64) / spp
    fin = np.isfinite(m)
    print(f"{label:12s} rays/sample {st['rays']/(W*W*spp):.3f} clipped mean {np.clip(m,0,1)[fin].mean():.4f} median {np.median(m[fin]):.4f} "
          f"p90 {np.quantile(m[fin],0.9):.4f} max {m[fin].max():.3g} nonfinite px {(~fin).sum()} diffuse {st['diffuse_vertices']} spec {st['specular_vertices']} "
          f"emit {st['emitter_hits']} miss {st['misses']} medium_probes {st['medium_probes']} closest {st['closest_hits']} nf {st['nonfinite_samples']}")
