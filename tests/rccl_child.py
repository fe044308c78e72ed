"""Child process of tests/test_gpu_rccl.py (not collected by pytest: no test_ prefix): a world-size-1 `nccl` (= RCCL)
process group on cuda:0, created the way bench.py creates its group, then raytracer_amd.distributed.render_sharded over
a real DeviceScene — the f32 product kernel and the f64 kernel — against the plain render of the same samples, and a
strong-scaling shard check: the sum of render_sharded's shards for world sizes 1, 2, 3 (rendered one after another by
this one rank, the all-reduce standing in for itself) equals one render."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as R  # noqa: E402
from raytracer_amd.distributed import allreduce_accum, render_sharded, shard_samples  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
w = h = 96
spp, depth = 48, 12
ds = R.DeviceScene(R.HostScene("cornell_box", 1.0), 0)
stream = torch.cuda.current_stream().cuda_stream
for f64 in (False, True):
    dt = torch.float64 if f64 else torch.float32

    def render_fn(accum, offset, count):
        ds.render_device(accum.data_ptr(), w, h, count, depth, seed=3, sample_offset=offset, stream=stream, f64=f64,
                         want_stats=False)

    merged = render_sharded(render_fn, w, h, spp, base_offset=7, device=dev, dtype=dt)  # one all_reduce over RCCL
    torch.cuda.synchronize()
    plain, _ = ds.render(w, h, spp, depth, seed=3, sample_offset=7, f64=f64)
    tol = dict(rtol=1e-11, atol=1e-11) if f64 else dict(rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(merged.cpu().numpy(), plain, **tol)
    # strong scaling: the shards of a G-rank job, rendered here one by one and summed by the collective's own add
    for world in (2, 3):
        total = torch.zeros((h, w, 3), dtype=dt, device=dev)
        for g in range(world):
            off, cnt = shard_samples(spp, g, world, base_offset=7)
            part = torch.zeros_like(total)
            if cnt:
                render_fn(part, off, cnt)
            allreduce_accum(part)  # world size 1: goes through RCCL, changes nothing
            total += part
        torch.cuda.synchronize()
        np.testing.assert_allclose(total.cpu().numpy(), plain, **(dict(rtol=1e-11, atol=1e-11) if f64 else dict(rtol=5e-5, atol=2e-5)))
dist.barrier()
dist.destroy_process_group()
print("RCCL_CHILD_OK backend", "nccl", flush=True)
