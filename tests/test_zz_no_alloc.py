"""rt_render_device_noalloc (include/rt.h): the variant of rt_render_device whose kernels' scratch — per launch, the
work counters and the persistent kernels' unit counter — lives in a file-scope __device__ array instead of memory
the library allocates.  Checked against rt_render_device on the same inputs, through the C ABI, on two calls in a
row, the second with a smaller input; then the limit (RT_NOALLOC_MAX_LAUNCHES launches per call) and a scene on the
other kernel families.

Bar: integer counters (rays, hits, vertices ...) bit-exact; the f32 radiance sums to f32 summation order
(rtol 2e-5, atol 1e-5: the kernels add a pixel's sample units with float atomics, whose order is not fixed from
launch to launch — the same tolerance test_sample_ranges_compose uses between two runs of ONE entry point).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = dict(rtol=2e-5, atol=1e-5)
COUNTERS = ("rays", "light_probes", "medium_probes", "closest_hits", "diffuse_vertices", "specular_vertices",
            "emitter_hits", "misses", "launches")


def _both(ds, torch, w, h, spp, depth, **kw):
    """(noalloc image, noalloc stats, rt_render_device image, stats) for one input."""
    out = []
    for noalloc in (True, False):
        acc = torch.full((h, w, 3), 7.0, dtype=torch.float32, device="cuda:0")  # must be overwritten, not added to
        st = ds.render_device(acc.data_ptr(), w, h, spp, depth, noalloc=noalloc, **kw)
        torch.cuda.synchronize()
        out += [acc.cpu().numpy(), st]
    return out


@pytest.mark.parametrize("sid,asp", [(6, 1.0), (0, 16.0 / 9.0), (3, 1.5)])  # leaf-list/persistent, refill, reference walk
def test_noalloc_equals_render_device_twice_in_a_row(sid, asp):
    import torch
    import raytracer_amd as R
    hs = R.HostScene(sid, asp)
    ds = R.DeviceScene(hs, 0)
    # first call: 3 launches (40 spp at 16 per launch) -> 3 scratch slots
    a, sa, b, sb = _both(ds, torch, 72, 56, 40, 12, seed=5, sample_offset=9, spp_per_launch=16, stats=True)
    assert sa["launches"] == 3
    for k in COUNTERS:
        assert sa[k] == sb[k], (k, sa[k], sb[k])
    np.testing.assert_allclose(a, b, **TOL)
    assert np.isfinite(a).all() and a.mean() > 0
    # second call, smaller input (fewer pixels, fewer samples, ONE launch): the scratch still holds the first
    # call's counters and a spent unit counter; the call must clear what it uses
    a2, sa2, b2, sb2 = _both(ds, torch, 33, 18, 7, 12, seed=6, sample_offset=0, stats=True)
    assert sa2["launches"] == 1
    for k in COUNTERS:
        assert sa2[k] == sb2[k], (k, sa2[k], sb2[k])
    np.testing.assert_allclose(a2, b2, **TOL)
    # and against the host-buffer entry (rt_render) for good measure
    c2, sc2 = ds.render(33, 18, 7, 12, seed=6, sample_offset=0, stats=True)
    np.testing.assert_allclose(a2, c2, **TOL)
    assert sa2["rays"] == sc2["rays"]


def test_noalloc_without_stats_is_asynchronous_and_accumulates():
    import torch
    import raytracer_amd as R
    hs = R.HostScene(6, 1.0)
    ds = R.DeviceScene(hs, 0)
    w = h = 40
    acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ds.render_device(acc.data_ptr(), w, h, 8, 6, seed=3, sample_offset=0, noalloc=True, want_stats=False,
                         stream=s.cuda_stream)
        ds.render_device(acc.data_ptr(), w, h, 8, 6, seed=3, sample_offset=8, accumulate=True, noalloc=True,
                         want_stats=False, stream=s.cuda_stream)
    s.synchronize()
    both, _ = ds.render(w, h, 16, 6, seed=3)
    np.testing.assert_allclose(acc.cpu().numpy(), both, **TOL)


def test_noalloc_refuses_more_launches_than_it_has_slots():
    import torch
    import raytracer_amd as R
    from raytracer_amd import _ffi as F
    hs = R.HostScene(6, 1.0)
    ds = R.DeviceScene(hs, 0)
    acc = torch.zeros((8, 8, 3), dtype=torch.float32, device="cuda:0")
    with pytest.raises(F.RtError, match="static scratch holds 63") as e:
        ds.render_device(acc.data_ptr(), 8, 8, 64, 4, spp_per_launch=1, noalloc=True)
    assert e.value.code == F.RT_ERR_BAD_ARG
    st = ds.render_device(acc.data_ptr(), 8, 8, 63, 4, spp_per_launch=1, noalloc=True)  # exactly the limit
    assert st["launches"] == 63
    ref = ds.render_device(acc.data_ptr(), 8, 8, 63, 4, spp_per_launch=1)
    assert st["rays"] == ref["rays"]
