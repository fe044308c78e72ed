"""The f32 reference-order kernel on the reference's default job (simple_light 1200x800, 500 spp, depth 5) and on forced walks:
kernel_ms under the environment in effect.  usage: [RT_QUEUE_REF32=1] [RT_UNIT_SAMPLES=n] ref32_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as R
earth = R.load_earthmap()
out = []
for name, sid, asp, w, spp, depth, force in (("simple_light", 3, 1.5, 1200, 500, 5, False), ("cornell forced", 6, 1.0, 1024, 256, 50, True),
                                             ("random forced", 0, 16 / 9, 400, 100, 50, True), ("simple_light small", 3, 1.5, 400, 100, 5, False),
                                             ("cornell 4K forced", 6, 16 / 9, 3840, 4, 50, True), ("cornell 256 forced", 6, 1.0, 256, 256, 50, True),
                                             ("final forced", 8, 1.0, 512, 16, 50, True)):
    ds = R.DeviceScene(R.HostScene(sid, asp, texels=earth), 0)
    h = R.image_height(w, asp)
    acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
    ds.render_device(acc.data_ptr(), w, h, 2, depth, force_bvh=force)
    st = ds.render_device(acc.data_ptr(), w, h, spp, depth, seed=3, force_bvh=force, stats=True)
    ms = min(ds.render_device(acc.data_ptr(), w, h, spp, depth, seed=3, force_bvh=force)["kernel_ms"] for _ in range(3))
    out.append(f"{name} {ms:.1f} ms rays {st['rays']} sum {float(acc.sum()):.6g}")
print({k: v for k, v in os.environ.items() if k.startswith("RT_")}, " | ".join(out), flush=True)
