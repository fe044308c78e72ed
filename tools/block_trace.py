"""Where and when the refill kernel's waves ran (diagnostic build: tools/build_variant.sh trace -DRT_BLOCK_TRACE).
usage: RT_HIP_LIBNAME=librt_hip_trace.so python tools/block_trace.py <scene id> <n> <W> <spp> [aspect]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as R
from raytracer_amd import _ffi as F
sid, n, W, spp = (int(x) for x in sys.argv[1:5])
asp = float(sys.argv[5]) if len(sys.argv) > 5 else 16 / 9
ds = R.DeviceScene(R.HostScene(sid, asp, n=n), 0)
h = R.image_height(W, asp)
ds.render(W, h, spp, 50)
_, st = ds.render(W, h, spp, 50)
N = 1 << 20
buf = np.zeros((N, 4), dtype=np.uint32)
F.hip.rt_debug_block_trace.argtypes = [C.c_void_p, C.c_uint32]
assert F.hip.rt_debug_block_trace(buf.ctypes.data, N) == 0
used = buf[:, 1] != 0
t = buf[used]
t0 = t[:, 0].min()
start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0   # microseconds
xcc = t[:, 2] & 15
cu = (t[:, 3] >> 8) & 15
se = (t[:, 3] >> 13) & 7
print(f"kernel_ms {st['kernel_ms']:.2f}; {used.sum()} waves; span {end.max() / 1e3:.2f} ms; wave duration mean {np.mean(end - start) / 1e3:.3f} ms, "
      f"p50 {np.median(end - start) / 1e3:.3f}, p99 {np.quantile(end - start, 0.99) / 1e3:.3f}, max {(end - start).max() / 1e3:.3f}")
ids = np.nonzero(used)[0] // 4
print("block id % 8 == XCC id for", np.mean((ids % 8) == xcc), "of waves; XCC ids seen:", np.unique(xcc))
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"  XCC {x}: waves {m.sum():7d}  busy wave-ms {np.sum((end - start)[m]) / 1e3:10.1f}  first start {start[m].min() / 1e3:7.2f}  last end {end[m].max() / 1e3:7.2f} ms")
# occupancy over time: resident waves in 40 slices
edges = np.linspace(0, end.max(), 41)
occ = [np.sum(np.clip(np.minimum(end, b) - np.maximum(start, a), 0, None)) / (b - a) for a, b in zip(edges[:-1], edges[1:])]
print("resident waves per 1/40 of the span:", " ".join(f"{o:.0f}" for o in occ))
# within a block: spread of its 4 waves' end times relative to the block's duration
blk = {}
order = np.nonzero(used)[0]
b_id = order // 4
bs = np.array([start[b_id == b].min() for b in np.unique(b_id)[:2000]])
be = np.array([end[b_id == b].max() for b in np.unique(b_id)[:2000]])
wsum = np.array([np.sum((end - start)[b_id == b]) for b in np.unique(b_id)[:2000]])
print(f"first 2000 blocks: wave-time / (4 x block duration) = {np.mean(wsum / (4 * (be - bs))):.3f}")
