"""rt_render_device_noalloc (and rt_render_device with stats == NULL) under HIP graph capture — the caller the
no-allocation entry point exists for: one render is captured into a graph and replayed several times; every replay must
give the plain call's image.  (Found with this test: with hipMemsetAsync clearing the accumulator, replays after the
first lost the sums of the units that finished before the memset node did; the entries now clear with a kernel.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("noalloc", [True, False])
@pytest.mark.parametrize("sid,asp", [(6, 1.0), (0, 16 / 9), (3, 1.5)])
def test_render_can_be_captured_and_replayed(sid, asp, noalloc):
    import raytracer_amd as R
    hs = R.HostScene(sid, asp)
    ds = R.DeviceScene(hs, 0)
    w = 96
    h = R.image_height(w, asp)
    ref = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
    ds.render_device(ref.data_ptr(), w, h, 8, 50, seed=3, noalloc=noalloc, want_stats=False)
    torch.cuda.synchronize()
    acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
    ds.render_device(acc.data_ptr(), w, h, 8, 50, seed=3, noalloc=noalloc, want_stats=False)  # warm-up outside capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g.capture_begin()
        ds.render_device(acc.data_ptr(), w, h, 8, 50, seed=3, noalloc=noalloc, want_stats=False, stream=s.cuda_stream)
        g.capture_end()
    torch.cuda.synchronize()
    for _ in range(4):
        acc.fill_(-1.0)
        g.replay()
        torch.cuda.synchronize()
        np.testing.assert_allclose(acc.cpu().numpy(), ref.cpu().numpy(), rtol=3e-5, atol=1e-6)
