"""f32 reference-order kernel timing (library from RT_HIP_LIBNAME): the reference's default job (simple_light 1200x800, 500 spp,
depth 5) and random_scene forced onto the reference-order walk; kernel_ms, best of 3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as R
for name, sid, asp, w, spp, depth, kw in (("simple_light default job", 3, 1.5, 1200, 500, 5, {}), ("random_scene forced BVH", 0, 16 / 9, 400, 100, 50, {"force_bvh": True}),
                                          ("final_scene forced BVH", 8, 1.0, 256, 64, 50, {"force_bvh": True})):
    ds = R.DeviceScene(R.HostScene(sid, asp), 0)
    h = R.image_height(w, asp)
    acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
    ds.render_device(acc.data_ptr(), w, h, 4, depth, **kw)
    best = min(ds.render_device(acc.data_ptr(), w, h, spp, depth, **kw)["kernel_ms"] for _ in range(3))
    print(os.environ.get("RT_HIP_LIBNAME", "librt_hip.so"), name, f"{best:.2f} ms", flush=True)
