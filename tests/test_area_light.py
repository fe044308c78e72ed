"""EXTENSION: a Cube as an importance-sampled area light — RT_LIGHT_CUBE (pdf_value / random over the
box's surface) and RT_MAT_AREA_LIGHT (emits on every hit; the reference's DiffuseLight on a Cube is black from
outside, hittable.rs:234-245 + material.rs:120).

PARITY UNPINNED: the reference has no such light and no output to compare with.  What pins it instead:
  * a closed form — Lambert's edge sum for the irradiance of a point from a uniformly emitting polygon — which the
    CPU oracle and both GPU kernels must reproduce on scene 13 (direct light only, max_depth 2);
  * the f64 audit kernel following the oracle path-for-path, the f32 kernel inside Monte-Carlo noise, exact primary
    hits, the golden fixture: the shared tests in test_gpu_parity.py / test_goldens.py / test_host_bvh.py, which
    iterate conftest.SCENES and so include scenes 12 (cornell_box_area) and 13 (area_light_plate)."""
import ctypes as C

import numpy as np
import pytest

from area_light_util import FLOOR_ALBEDO, LIGHT_RADIANCE, floor_radiance

W = H = 48
SPP = 2048


def _floor_points(cam, ids, t):
    """world position behind every pixel centre that sees the floor (object 0)"""
    o, ll, hz, vt = cam[0:3], cam[3:6], cam[6:9], cam[9:12]
    pts = {}
    for y in range(H):
        for x in range(W):
            if ids[y, x] != 0:
                continue
            s, tt = (x + 0.5) / (W - 1), (H - y - 1 + 0.5) / (H - 1)
            pts[(y, x)] = o + t[y, x] * (ll + s * hz + tt * vt - o)
    return pts


def _check_against_closed_form(img_mean, pts, label):
    """per-pixel mean radiance vs rho/pi * E(p): the image-wide ratio (a bias test: thousands of pixels x SPP samples)
    and every pixel within Monte-Carlo noise plus the footprint of a pixel on the floor"""
    got = np.array([img_mean[y, x] for (y, x) in pts])
    want = np.array([floor_radiance(p) for p in pts.values()])
    ratio = got.sum(axis=0) / want.sum(axis=0)
    print(label, "pixels", len(pts), "sum ratio per channel", ratio)
    assert np.all(np.abs(ratio - 1.0) <= 0.01), (label, ratio)
    lit = want[:, 0] >= 0.1 * want[:, 0].max()  # (far, dim floor pixels are all relative noise at this sample count)
    rel = (np.abs(got - want) / want)[lit]
    assert lit.sum() > 100 and np.quantile(rel, 0.95) <= 0.08 and rel.max() <= 0.25, (label, np.quantile(rel, 0.95), rel.max())
    # the three channels carry the light's colour exactly (one path, three products)
    np.testing.assert_allclose(got[:, 1] / got[:, 0], LIGHT_RADIANCE[1] / LIGHT_RADIANCE[0], rtol=1e-4)


def test_oracle_matches_lamberts_formula():
    import oracle_ffi as O
    sc = O.OracleScene(13, 1.0)
    ids, t, _, _, _ = sc.primary_hits(W, H)
    pts = _floor_points(sc.camera_array(), ids, t)
    assert len(pts) > 500
    acc, cnt, _ = sc.render(W, H, SPP, 2, seed=3)
    _check_against_closed_form(acc / SPP, pts, "oracle")
    assert cnt["light_probes"] == cnt["diffuse_vertices"] and cnt["emitter_hits"] > 0


def test_host_flattens_the_extension_tags():
    import raytracer_amd as R
    from raytracer_amd import _ffi as F
    for sid, box in ((12, (213.0, 553.0, 227.0, 343.0, 554.0, 332.0)), (13, (-1.0, 3.0, -1.5, 1.0, 3.25, 1.5))):
        hs = R.HostScene(sid, 1.0)  # (owns the arrays the descriptor points into)
        d = hs.desc
        assert d.lights_kind == 1 and d.n_lights == 1 and d.light_tag[0] == 2  # RT_LIGHTS_SINGLE, RT_LIGHT_CUBE
        assert tuple(d.light_box[k] for k in range(6)) == box
        assert 5 in [d.mat_tag[i] for i in range(d.n_materials)]  # RT_MAT_AREA_LIGHT
    # a reference scene's lights stay what they were
    hs = R.HostScene(6, 1.0)
    d = hs.desc
    assert d.light_tag[0] == 0 and 5 not in [d.mat_tag[i] for i in range(d.n_materials)]


def test_bad_light_boxes_are_refused():
    import raytracer_amd as R
    from raytracer_amd import _ffi as F
    hs = R.HostScene(13, 1.0)
    out = C.c_void_p()
    d = F.RtSceneDesc.from_buffer_copy(hs.desc)
    d.light_box = None
    assert F.hip.rt_scene_create(0, C.byref(d), C.byref(out)) == F.RT_ERR_BAD_ARG and b"light_box" in F.hip.rt_last_error()
    bad = (C.c_double * 6)(-1.0, 3.0, -1.5, 1.0, 3.0, 1.5)  # zero thickness
    d.light_box = C.cast(bad, C.POINTER(C.c_double))
    assert F.hip.rt_scene_create(0, C.byref(d), C.byref(out)) == F.RT_ERR_BAD_ARG and b"p0 < p1" in F.hip.rt_last_error()
    d = F.RtSceneDesc.from_buffer_copy(hs.desc)
    tags = (C.c_uint8 * 1)(3)
    d.light_tag = C.cast(tags, C.POINTER(C.c_uint8))
    assert F.hip.rt_scene_create(0, C.byref(d), C.byref(out)) == F.RT_ERR_BAD_ARG and b"light tag" in F.hip.rt_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("f64", [True, False])
def test_gpu_matches_lamberts_formula(f64):
    import raytracer_amd as R
    hs = R.HostScene(13, 1.0)
    ds = R.DeviceScene(hs, 0)
    ids, t, _, _, _ = ds.primary_hits(W, H, f64=True)
    pts = _floor_points(hs.camera_array(), ids, t)
    assert len(pts) > 500
    acc, st = ds.render(W, H, SPP, 2, seed=5, stats=True, f64=f64)
    _check_against_closed_form(acc.astype(np.float64) / SPP, pts, "f64 kernel" if f64 else "f32 kernel")
    assert st["light_probes"] == st["diffuse_vertices"] and st["emitter_hits"] > 0 and st["nonfinite_samples"] == 0


@pytest.mark.gpu
def test_gpu_cornell_with_the_box_light_is_lit_from_the_ceiling():
    """scene 12 end to end on the product entry point: the floor under the panel is bright, the panel itself shows its
    radiance (15, clipped by the tonemap), and no sample is lost"""
    import raytracer_amd as R
    ds = R.DeviceScene(R.HostScene(12, 1.0), 0)
    acc, st = ds.render(64, 64, 256, 50, seed=2, stats=True)
    img = acc / 256
    assert st["nonfinite_samples"] <= 4
    ids, _, _, _, _ = ds.primary_hits(64, 64)
    panel = ids == 5
    # (pixels on the panel's edge see it through part of their jitter footprint only)
    assert panel.sum() >= 4 and np.allclose(img[panel].max(), 15.0, rtol=1e-3) and img[panel].mean() >= 10.0
    floor = ids == 2
    assert img[floor].mean() > 0.05  # (direct light alone gives 0.73 / pi * 15 * 13650 / 553^2 = 0.16 under the panel)
