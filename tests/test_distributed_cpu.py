"""Multi-process path on the CPU (gloo, world_size 2): the sample sharding and the sum all-reduce of
raytracer_amd.distributed, with the oracle standing in for the GPU renderer (tests may use it)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_samples_partitions_the_index_range():
    from raytracer_amd.distributed import shard_samples
    for total in (0, 1, 7, 8, 4096, 65536):
        for world in (1, 2, 3, 8):
            got = [shard_samples(total, r, world, base_offset=100) for r in range(world)]
            assert sum(c for _, c in got) == total
            pos = 100
            for off, cnt in got:
                assert off == pos
                pos += cnt
            assert max(c for _, c in got) - min(c for _, c in got) <= 1
    with pytest.raises(ValueError):
        shard_samples(8, 2, 2)


def test_rt_multi_shard_is_the_same_partition():
    """rt_multi_shard (include/rt.h; host arithmetic, no GPU): the in-process multi-device path and the
    process-per-GPU path cut [0, spp) the same way, also when spp is not a multiple of the device count."""
    import raytracer_amd as R
    from raytracer_amd.distributed import shard_samples
    for total in (0, 1, 5, 7, 8, 100, 4095, 4096, 2**32 - 1):
        for world in (1, 2, 3, 5, 8, 64):
            got = [R.multi_shard(total, g, world) for g in range(world)]
            assert got == [shard_samples(total, g, world) for g in range(world)]
            assert got[0][0] == 0 and sum(c for _, c in got) == total
            assert all(got[g][0] + got[g][1] == got[g + 1][0] for g in range(world - 1))
            assert max(c for _, c in got) - min(c for _, c in got) <= 1
    assert R.multi_shard(10, 0, 3) == (0, 3) and R.multi_shard(10, 2, 3) == (6, 4)
    assert R.multi_shard(10, 0, 0) == (0, 0)   # n_dev == 0: empty, not a division by zero


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_ffi as O
    from raytracer_amd.distributed import render_sharded
    w, h, spp, depth = 24, 24, 9, 10
    sc = O.OracleScene(6, 1.0)

    def render_fn(accum, offset, count):
        a, _, _ = sc.render(w, h, count, depth, seed=4, sample_offset=offset, threads=2)
        accum.copy_(torch.from_numpy(a))

    merged = render_sharded(render_fn, w, h, spp, base_offset=3, dtype=torch.float64)
    if rank == 0:
        np.save(out_path, merged.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "merged.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    merged = np.load(out)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O
    sc = O.OracleScene(6, 1.0)
    single, _, _ = sc.render(24, 24, 9, 10, seed=4, sample_offset=3, threads=2)
    # the union of the per-rank Philox streams equals the single-rank stream; only the order of the
    # final additions differs
    np.testing.assert_allclose(merged, single, rtol=1e-12, atol=1e-12)
