"""The wavefront form (RT_FLAG_WAVEFRONT: path pool in HBM, TRACE + SHADE kernels per bounce, device/wavefront.hpp)
against the megakernel (render_refill_kernel) on the BVH-bound scenes: SURVEY 8(f)1's "real wavefront queue".
Both run the same leaf tests, hit records and shade() on Philox streams keyed by (pixel, sample, slot), so they must
trace exactly the same rays — every work counter equal — and produce the same image up to the f32 order in which a
pixel's samples are summed (float atomics in both)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COUNTERS = ("rays", "closest_hits", "misses", "diffuse_vertices", "specular_vertices", "emitter_hits", "light_probes",
            "medium_probes", "nonfinite_samples")
CASES = [  # name, scene, aspect, n, width, spp, depth
    ("random_scene", 0, 16.0 / 9.0, 0, 200, 24, 50),
    ("final_scene", 8, 1.0, 0, 128, 16, 50),        # media: keyed free-path draws in the trace kernel
    ("bvh_stress", 7, 16.0 / 9.0, 20000, 131, 7, 50),  # ragged tiles, odd spp
    ("random_scene depth 1", 0, 16.0 / 9.0, 0, 64, 4, 1),
    ("random_scene one sample", 0, 16.0 / 9.0, 0, 40, 1, 50),  # fewer jobs than a pool: slots retire at once
]


@pytest.mark.parametrize("name,sid,asp,n,w,spp,depth", CASES)
def test_wavefront_renders_the_megakernels_paths(name, sid, asp, n, w, spp, depth):
    import raytracer_amd as R
    ds = R.DeviceScene(R.HostScene(sid, asp, n=n), 0)
    h = R.image_height(w, asp)
    a, sa = ds.render(w, h, spp, depth, seed=5, sample_offset=3, stats=True)
    b, sb = ds.render(w, h, spp, depth, seed=5, sample_offset=3, stats=True, wavefront=True)
    assert sb["launches"] > sa["launches"]  # it really took the other path: 1 + 2 launches per iteration
    for k in COUNTERS:
        assert sa[k] == sb[k], (k, sa[k], sb[k])
    same = np.isclose(a, b, rtol=1e-4, atol=1e-5).all(axis=-1)
    assert same.mean() >= 0.999, same.mean()
    # twice in a row on one scene (the pool and its counters are reused), accumulating
    import torch
    acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
    ds.render_device(acc.data_ptr(), w, h, spp, depth, seed=5, sample_offset=3, wavefront=True)
    ds.render_device(acc.data_ptr(), w, h, spp, depth, seed=6, sample_offset=3, wavefront=True, accumulate=True)
    torch.cuda.synchronize()
    c, _ = ds.render(w, h, spp, depth, seed=6, sample_offset=3)
    np.testing.assert_allclose(acc.cpu().numpy(), a + c, rtol=2e-4, atol=2e-5)


def test_wavefront_flag_is_ignored_where_no_tree_is_walked():
    """Leaf-list scenes (Cornell) and scenes whose boxes clip (simple_light) have no ordered walk: the flag changes nothing."""
    import raytracer_amd as R
    for sid, asp in ((6, 1.0), (3, 1.5)):
        ds = R.DeviceScene(R.HostScene(sid, asp), 0)
        a, sa = ds.render(48, 32, 8, 6, seed=2, stats=True)
        b, sb = ds.render(48, 32, 8, 6, seed=2, stats=True, wavefront=True)
        assert sa["launches"] == sb["launches"] and sa["rays"] == sb["rays"]
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-5)
