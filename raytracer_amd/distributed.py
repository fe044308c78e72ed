"""Multi-GPU rendering: one process per GPU, samples sharded by index, one RCCL sum all-reduce.

Samples of a pixel are i.i.d. (main.rs:160-166), so rank g of G renders the global sample
indices [offset_g, offset_g + count_g) of EVERY pixel with the Philox counter
(pixel, global sample index): the union of the streams does not depend on G.  The scene is
replicated per GPU; the only exchange is a sum all-reduce of the float32 radiance
accumulators (1024^2: 12.6 MB, 3840x2160: 99.5 MB) over RCCL/xGMI.  Float sum order differs
with G, so images across GPU counts agree to f32 rounding, not bit-exactly.
"""
import torch
import torch.distributed as dist


def shard_samples(total_spp, rank, world_size, base_offset=0):
    """Contiguous split of `total_spp` global sample indices: returns (offset, count) of `rank`:
    [rank*total/world, (rank+1)*total/world) in integer division — the rule rt_multi_shard (include/rt.h) applies
    to the devices of one process, so a job renders the same shards whichever way it is spread."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank {rank} of {world_size}")
    lo = int(total_spp) * rank // int(world_size)
    hi = int(total_spp) * (rank + 1) // int(world_size)
    return base_offset + lo, hi - lo


def allreduce_accum(accum):
    """Sum the per-rank accumulators in place (RCCL on GPUs, gloo on CPU).  No-op without a group (a group of one
    rank does go through the collective: that is how the RCCL path is rehearsed on a one-GPU box)."""
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(accum, op=dist.ReduceOp.SUM)
    return accum


def render_sharded(render_fn, width, height, total_spp, rank=None, world_size=None, base_offset=0,
                   device=None, dtype=torch.float32):
    """Render this rank's share and merge.

    render_fn(accum_tensor, sample_offset, spp) must write the SUM of `spp` samples starting at global
    sample index `sample_offset` into accum_tensor ([h][w][3]).  On GPUs that is
    DeviceScene.render_device on accum.data_ptr(); the CPU tests pass an oracle-backed callable to
    exercise the sharding and the collective under gloo.
    Returns the merged accumulator (identical on every rank)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    offset, count = shard_samples(total_spp, rank, world_size, base_offset)
    accum = torch.zeros((height, width, 3), dtype=dtype, device=device)
    if count > 0:
        render_fn(accum, offset, count)
    return allreduce_accum(accum)
