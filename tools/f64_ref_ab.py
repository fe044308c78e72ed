"""f64 reference-order walk (scenes whose boxes clip, extension scenes, RT_FLAG_FORCE_BVH): kernel_ms under the library /
environment in effect.  usage: [RT_QUEUE=0|1] f64_ref_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as R
earth = R.load_earthmap()
out = []
for name, sid, asp, w, spp, depth, force in (("simple_light", 3, 1.5, 1200, 64, 5, False), ("cornell forced", 6, 1.0, 1024, 64, 50, True),
                                             ("cornell_area", 12, 1.0, 512, 64, 50, False), ("random forced", 0, 16 / 9, 400, 100, 50, True)):
    ds = R.DeviceScene(R.HostScene(sid, asp, texels=earth), 0)
    h = R.image_height(w, asp)
    acc = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
    ds.render_device(acc.data_ptr(), w, h, 2, depth, f64=True, force_bvh=force)
    ms = min(ds.render_device(acc.data_ptr(), w, h, spp, depth, seed=3, f64=True, force_bvh=force)["kernel_ms"] for _ in range(2))
    out.append(f"{name} {ms:.1f} ms")
print("RT_QUEUE=" + os.environ.get("RT_QUEUE", "-"), " | ".join(out), flush=True)
