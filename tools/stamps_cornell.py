"""Phase stamps (s_memtime) of the persistent leaf-list kernel on Cornell:
RT_HIP_LIBNAME=librt_hip_stamps.so RT_PRINT_STAMPS=1 python tools/stamps_cornell.py
stamp i = wave-time between stamp i-1 and stamp i, summed over waves (see render_persist_kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as R
ds = R.DeviceScene(R.HostScene(6, 1.0), 0)
for spp in (256, 1024):
    sys.stderr.flush()
    _, st = ds.render(1024, 1024, spp, 50)
    print(f"spp {spp} kernel_ms {st['kernel_ms']:.1f}", flush=True)
