"""f64 audit kernel timing on Cornell 1024^2 (library from RT_HIP_LIBNAME): kernel_ms, best of 2.  usage: f64_ab.py [spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as R
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ds = R.DeviceScene(R.HostScene("cornell_box", 1.0), 0)
acc = torch.zeros((1024, 1024, 3), dtype=torch.float64, device="cuda:0")
ds.render_device(acc.data_ptr(), 1024, 1024, 4, 50, f64=True)
best = min(ds.render_device(acc.data_ptr(), 1024, 1024, spp, 50, f64=True)["kernel_ms"] for _ in range(2 if spp <= 1024 else 1))
print(os.environ.get("RT_HIP_LIBNAME", "librt_hip.so"), f"{best:.1f} ms  {1024 * 1024 * spp / best / 1e3:.1f} Msamples/s", flush=True)
