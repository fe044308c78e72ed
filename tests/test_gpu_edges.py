"""Edge cases of the render entry points on the GPU, for every kernel family: the leaf-list kernel
(cornell_box), the ordered-walk refill kernel (random_scene), the reference-order f32 walk
(simple_light) and the f64 audit kernel.  The reference's own edge behaviour is cited per test."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FAMILIES = [("cornell_box", 6, 1.0, {}), ("random_scene", 0, 16.0 / 9.0, {}), ("simple_light", 3, 1.5, {}),
            ("small_scene_f64", 5, 1.5, {"f64": True})]


def _scene(sid, asp):
    import raytracer_amd as R
    hs = R.HostScene(sid, asp)
    return R, hs, R.DeviceScene(hs, 0)


@pytest.mark.parametrize("name,sid,asp,kw", FAMILIES)
def test_sample_ranges_compose(name, sid, asp, kw):
    """Samples are keyed by their GLOBAL index (main.rs:160-166 made reproducible): rendering [0,5) and
    [5,16) separately, in one call, or in launches of 3 gives the same sums up to f32 summation order —
    what sample sharding across GPUs relies on."""
    R, hs, ds = _scene(sid, asp)
    w, h = 50, 37  # not multiples of the 16x16 tile
    whole, sw = ds.render(w, h, 16, 8, seed=4, sample_offset=100, stats=True, **kw)
    a, sa = ds.render(w, h, 5, 8, seed=4, sample_offset=100, stats=True, **kw)
    b, sb = ds.render(w, h, 11, 8, seed=4, sample_offset=105, stats=True, **kw)
    c, sc = ds.render(w, h, 16, 8, seed=4, sample_offset=100, spp_per_launch=3, stats=True, **kw)
    assert sa["rays"] + sb["rays"] == sw["rays"] == sc["rays"]
    assert sc["launches"] == 6
    tol = dict(rtol=1e-12, atol=1e-12) if kw.get("f64") else dict(rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(a + b, whole, **tol)
    np.testing.assert_allclose(c, whole, **tol)
    # more launches than a call has counter slots (16): a slot is used again, its unit counter re-zeroed in between,
    # and the work counters keep adding up
    d, sd = ds.render(w, h, 16, 8, seed=4, sample_offset=100, spp_per_launch=1 if not kw.get("f64") else 1, stats=True, **kw)
    assert sd["launches"] == 16 + 0 and sd["rays"] == sw["rays"]
    e, se = ds.render(w, h, 40, 8, seed=4, sample_offset=100, spp_per_launch=2, stats=True, **kw)
    f, sf = ds.render(w, h, 40, 8, seed=4, sample_offset=100, stats=True, **kw)
    assert se["launches"] == 20 and se["rays"] == sf["rays"] and se["closest_hits"] == sf["closest_hits"]
    np.testing.assert_allclose(d, whole, **tol)
    np.testing.assert_allclose(e, f, **(dict(rtol=1e-12, atol=1e-12) if kw.get("f64") else dict(rtol=5e-5, atol=2e-5)))


@pytest.mark.parametrize("name,sid,asp,kw", FAMILIES)
def test_depth_zero_one_and_no_samples(name, sid, asp, kw):
    R, hs, ds = _scene(sid, asp)
    w, h = 33, 18
    # main.rs:73-75: depth 0 -> black whatever the scene
    z, sz = ds.render(w, h, 4, 0, seed=1, stats=True, **kw)
    assert not z.any() and sz["rays"] == 0
    # depth 1: one ray per sample; only emitted light and the background can arrive
    o, so = ds.render(w, h, 4, 1, seed=1, stats=True, **kw)
    assert so["rays"] == w * h * 4
    assert so["closest_hits"] + so["misses"] == so["rays"]
    # no samples: the accumulator is cleared
    n, sn = ds.render(w, h, 0, 5, seed=1, stats=True, **kw)
    assert not n.any() and sn["rays"] == 0


def test_render_device_accumulates_and_tonemaps_like_the_host():
    import torch
    import raytracer_amd as R
    from raytracer_amd import _ffi as F
    hs = R.HostScene(6, 1.0)
    ds = R.DeviceScene(hs, 0)
    w = h = 40
    acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
    ds.render_device(acc.data_ptr(), w, h, 8, 6, seed=3, sample_offset=0)
    ds.render_device(acc.data_ptr(), w, h, 8, 6, seed=3, sample_offset=8, accumulate=True)
    torch.cuda.synchronize()
    both, _ = ds.render(w, h, 16, 6, seed=3)
    np.testing.assert_allclose(acc.cpu().numpy(), both, rtol=2e-5, atol=1e-5)
    rgb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
    F.check_hip(F.hip.rt_tonemap_device(0, acc.data_ptr(), w * h, 16, rgb.data_ptr(), None))
    torch.cuda.synchronize()
    assert np.array_equal(rgb.cpu().numpy(), R.tonemap(acc.cpu().numpy(), 16))


def test_f64_accumulator_tonemaps_on_the_device_like_the_oracle():
    """rt_tonemap_f64_device on the f64 kernel's accumulator == the oracle's image of the same render, byte for byte
    (main.rs:167-168, 51-63 in f64)."""
    import torch
    import oracle_ffi as O
    import raytracer_amd as R
    from raytracer_amd import _ffi as F
    ds = R.DeviceScene(R.HostScene(6, 1.0), 0)
    w = h = 48
    acc = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
    ds.render_device(acc.data_ptr(), w, h, 16, 8, seed=3, f64=True)
    rgb = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
    F.check_hip(F.hip.rt_tonemap_f64_device(0, acc.data_ptr(), w * h, 16, rgb.data_ptr(), None))
    torch.cuda.synchronize()
    host = acc.cpu().numpy()
    assert np.array_equal(rgb.cpu().numpy(), R.tonemap(host, 16))
    o, _, _ = O.OracleScene(6, 1.0).render(w, h, 16, 8, seed=3)
    assert (rgb.cpu().numpy() != O.tonemap(o, 16)).mean() <= 1e-3  # (a sum 1e-16 apart may straddle a step)


def test_arguments_are_validated_not_trusted():
    """The reference panics (cam.rs:69 empty shutter range) or divides by zero (main.rs:161-163, w-1);
    the library returns an error code and a message instead."""
    import raytracer_amd as R
    from raytracer_amd import _ffi as F
    hs = R.HostScene(6, 1.0)
    ds = R.DeviceScene(hs, 0)
    with pytest.raises(RuntimeError, match="2x2"):
        ds.render(1, 8, 1, 1)
    cam = F.RtCamera()
    C.memmove(C.byref(cam), C.byref(hs.camera), C.sizeof(cam))
    cam.time_close = cam.time_open
    with pytest.raises(RuntimeError, match="shutter"):
        ds.render(8, 8, 1, 1, camera=cam)
    with pytest.raises(RuntimeError, match="32 bits"):
        ds.render(8, 8, 16, 1, sample_offset=0xFFFFFFF8)
    tiny, st = ds.render(2, 2, 3, 4, stats=True)  # the smallest legal image
    assert tiny.shape == (2, 2, 3) and st["rays"] >= 12
    # a tile origin is packed as x | y << 16 by the persistent kernels: wider or taller than 65535 is refused, not
    # rendered into the wrong pixels; 65535 itself renders (every kernel family packs origins the same way)
    for fam in ({}, {"f64": True}):
        with pytest.raises(RuntimeError, match="65535"):
            ds.render(65537, 2, 1, 2, **fam)
        with pytest.raises(RuntimeError, match="65535"):
            ds.render(2, 65536, 1, 2, **fam)
    wide, sw = ds.render(65535, 2, 1, 2, stats=True)
    assert sw["rays"] >= 2 * 65535 and np.isfinite(wide).all() and wide[:, -8:].sum() >= 0


def test_non_sphere_light_follows_hittable_rs_498_513():
    """Q8: a light that is not a Sphere has pdf_value 0 and random() = (1, 0, 0).  None of the nine scenes
    has one, so simple_light's second light is re-tagged RT_LIGHT_OTHER in a copy of the description; the
    f32 and f64 kernels implement the arm independently (shared code aside) and must agree with each other at
    equal seed, and the image must change (the arm is live)."""
    import raytracer_amd as R
    from raytracer_amd import _ffi as F
    hs = R.HostScene(3, 1.5)
    d = F.RtSceneDesc()
    C.memmove(C.byref(d), C.byref(hs.desc), C.sizeof(d))
    tags = (C.c_uint8 * 2)(hs.desc.light_tag[0], 1)  # RT_LIGHT_SPHERE, RT_LIGHT_OTHER
    d.light_tag = C.cast(tags, type(d.light_tag))

    class Mutated(R.DeviceScene):
        def __init__(self):
            self.host = hs
            self.device = 0
            h = C.c_void_p()
            F.check_hip(F.hip.rt_scene_create(0, C.byref(d), C.byref(h)))
            self._h = h

    ds, ref = Mutated(), R.DeviceScene(hs, 0)
    w, h = 48, 32
    a, _ = ds.render(w, h, 64, 3, seed=2)
    b, _ = ds.render(w, h, 64, 3, seed=2, f64=True)
    c, _ = ref.render(w, h, 64, 3, seed=2)
    # (1,0,0) below the horizon with pdf 0 gives 0/0 (Q10): the f64 kernel lets it poison the pixel like the
    # reference, the f32 kernel drops the sample -- compare the pixels no such sample fell into
    ok = np.isfinite(b).all(axis=-1)
    assert 0.2 < ok.mean() < 1.0
    ta, tb = float(a[ok].sum()), float(b[ok].sum())
    assert abs(ta / tb - 1.0) <= 0.01, (ta, tb)
    same = np.isclose(a[ok], b[ok], rtol=1e-3, atol=1e-4).all(axis=-1)
    assert same.mean() >= 0.75  # (64 samples per pixel: one f32-vs-f64 branch flip anywhere makes a pixel differ)
    assert not np.allclose(c, a, rtol=1e-3, atol=1e-3)  # the arm is live: the estimate changed


def _arr(ptr, n, dtype):
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype).copy() if n else np.zeros(0, dtype)


def test_cornell_smoke_exercises_the_leaf_list_medium_arm():
    """No reference scene puts a constant medium into a small scene, so the leaf-list kernels' medium arm
    (traverse_linear<MEDIA = true>, cube boundaries, hittable.rs:319-363) would go untested: turn the Cornell
    box's two blocks into smoke (Shirley's cornell_smoke: ConstantMedium(Translate(Rotate(Cube))), density 0.01,
    black and white Isotropic) in a copy of the description.  Leaf-list f32, reference-order f32 and the f64
    audit kernel must agree at equal seed."""
    import raytracer_amd as R
    from raytracer_amd import _ffi as F
    hs = R.HostScene(6, 1.0)
    src = hs.desc
    assert src.n_complex == 2 and src.n_ops == 4
    d = F.RtSceneDesc()
    C.memmove(C.byref(d), C.byref(src), C.sizeof(d))
    keep = []  # numpy arrays the description points into

    def put(field, a, ptype):
        a = np.ascontiguousarray(a)
        keep.append(a)
        setattr(d, field, a.ctypes.data_as(ptype))

    op_tag = _arr(src.op_tag, 4, np.uint8)
    op_data = _arr(src.op_data, 16, np.float64).reshape(4, 4)
    med = lambda density: np.array([[-1.0 / density, 0, 0, 0]])
    put("op_tag", np.concatenate([[2], op_tag[0:2], [2], op_tag[2:4]]).astype(np.uint8), F._u8p)
    put("op_data", np.concatenate([med(0.01), op_data[0:2], med(0.01), op_data[2:4]]).astype(np.float64), F._f64p)
    d.n_ops = 6
    put("complex_op_first", np.array([0, 3], dtype=np.uint32), F._u32p)
    put("complex_op_count", np.array([3, 3], dtype=np.uint32), F._u32p)
    nm, nt = src.n_materials, src.n_textures
    put("mat_tag", np.concatenate([_arr(src.mat_tag, nm, np.uint8), [4, 4]]).astype(np.uint8), F._u8p)  # RT_MAT_ISOTROPIC
    put("mat_texture", np.concatenate([_arr(src.mat_texture, nm, np.uint32), [nt, nt + 1]]).astype(np.uint32), F._u32p)
    put("mat_params", np.concatenate([_arr(src.mat_params, 4 * nm, np.float64), np.zeros(8)]), F._f64p)
    d.n_materials = nm + 2
    put("complex_material", np.array([nm, nm + 1], dtype=np.uint32), F._u32p)
    put("tex_tag", np.concatenate([_arr(src.tex_tag, nt, np.uint8), [0, 0]]).astype(np.uint8), F._u8p)
    put("tex_color", np.concatenate([_arr(src.tex_color, 3 * nt, np.float64), [0, 0, 0, 1, 1, 1]]), F._f64p)
    for f, ty, pt in (("tex_odd", np.uint32, F._u32p), ("tex_even", np.uint32, F._u32p),
                      ("tex_texel_offset", np.uint64, F._u64p), ("tex_width", np.uint32, F._u32p),
                      ("tex_height", np.uint32, F._u32p)):
        put(f, np.concatenate([_arr(getattr(src, f), nt, ty), [0, 0]]).astype(ty), pt)
    d.n_textures = nt + 2

    class Smoke(R.DeviceScene):
        def __init__(self):
            self.host = hs
            self.device = 0
            h = C.c_void_p()
            F.check_hip(F.hip.rt_scene_create(0, C.byref(d), C.byref(h)))
            self._h = h

    ds = Smoke()
    w = h = 64
    a, sa = ds.render(w, h, 64, 4, seed=8, stats=True)                    # leaf list + medium arm
    b, sb = ds.render(w, h, 64, 4, seed=8, stats=True, force_bvh=True)    # reference-order walk, f32
    c, sc = ds.render(w, h, 64, 4, seed=8, stats=True, f64=True)          # audit
    assert sa["nodes_visited"] == 0 and sb["nodes_visited"] > 0
    assert sa["medium_visits"] > 0 and sa["medium_probes"] > sa["medium_visits"]
    for x, y in ((a, b), (a, c)):
        assert abs(float(x.sum()) / float(y[np.isfinite(y)].sum()) - 1.0) <= 0.01
    plain, _ = R.DeviceScene(hs, 0).render(w, h, 64, 4, seed=8)
    assert not np.allclose(plain, a, rtol=1e-2, atol=1e-2)  # smoke is not a solid block


def test_shutter_outside_the_moving_spheres_box_window_falls_back():
    """new_bvh boxes the moving spheres over t in [0, 1] (main.rs:125); the ordered walk culls by those boxes,
    so a camera whose shutter leaves that window must take the reference-order walk (which reproduces
    whatever the reference's pruning does there)."""
    import raytracer_amd as R
    from raytracer_amd import _ffi as F
    hs = R.HostScene(0, 16.0 / 9.0)
    ds = R.DeviceScene(hs, 0)
    cam = F.RtCamera()
    C.memmove(C.byref(cam), C.byref(hs.camera), C.sizeof(cam))
    cam.time_open, cam.time_close = 0.5, 1.5
    w, h = 64, 36
    a, sa = ds.render(w, h, 8, 10, seed=3, stats=True, camera=cam)
    b, sb = ds.render(w, h, 8, 10, seed=3, stats=True, camera=cam, force_bvh=True)
    assert sa["nodes_visited"] == sb["nodes_visited"]
    np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)  # same kernel; float atomics of the sample splits reorder
    inside, si = ds.render(w, h, 8, 10, seed=3, stats=True)  # shutter 0..1: the ordered walk
    assert si["nodes_visited"] < sb["nodes_visited"]


def test_full_size_cornell_properties():
    """BASELINE config 2's own size (1024x1024), where the CPU oracle would need minutes: size-independent
    properties instead.  Sample ranges compose (what the 8-GPU sharding relies on), two seeds estimate the same
    image (mean within 0.1 %, per-pixel difference consistent with Monte-Carlo noise), and the work per sample
    is the scene's 2.43 rays."""
    import raytracer_amd as R
    hs = R.HostScene(6, 1.0)
    ds = R.DeviceScene(hs, 0)
    w = h = 1024
    whole, sw = ds.render(w, h, 512, 50, seed=1, stats=True)
    a, sa = ds.render(w, h, 200, 50, seed=1, stats=True)
    b, sb = ds.render(w, h, 312, 50, seed=1, sample_offset=200, stats=True)
    assert sa["rays"] + sb["rays"] == sw["rays"]
    np.testing.assert_allclose(a + b, whole, rtol=1e-4, atol=1e-3)
    assert abs(sw["rays"] / (w * h * 512) - 2.431) < 0.01
    other, _ = ds.render(w, h, 512, 50, seed=2)
    assert abs(float(other.sum()) / float(whole.sum()) - 1.0) < 1e-3
    d = (np.clip(other / 512, 0, 1) - np.clip(whole / 512, 0, 1))
    assert 0.01 < np.sqrt(np.mean(d * d)) < 0.06  # two independent 512-spp estimates: noise, not bias, not identity


def test_scenes_come_and_go_without_leaking_device_memory():
    import torch
    import raytracer_amd as R
    hs = R.HostScene(7, 16.0 / 9.0, n=20000)
    ds = R.DeviceScene(hs, 0)
    ds.render(32, 18, 1, 2)
    ds.close()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(0)
    for _ in range(20):
        ds = R.DeviceScene(hs, 0)
        ds.render(32, 18, 1, 2)
        ds.primary_hits(32, 18)
        ds.close()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(0)
    assert free0 - free1 < 8 << 20, f"{(free0 - free1) >> 20} MiB lost over 20 create/destroy cycles"
    # two scenes alive at once render independently
    a, b = R.DeviceScene(R.HostScene(6, 1.0), 0), R.DeviceScene(R.HostScene(5, 1.5), 0)
    ia, _ = a.render(40, 40, 4, 5, seed=1)
    ib, _ = b.render(40, 26, 4, 5, seed=1)
    ia2, _ = a.render(40, 40, 4, 5, seed=1)
    np.testing.assert_allclose(ia, ia2, rtol=1e-5, atol=1e-6)
    assert ib.shape == (26, 40, 3)


@pytest.mark.parametrize("size", [(2, 2), (37, 23), (3, 130)])
def test_empty_world_and_ragged_tiny_images(size):
    """Hittable::Empty as the whole world (new_bvh of no objects, hittable.rs:171, 413): every ray misses, every
    sample is the background (main.rs:104) — through every kernel family, on image sizes that fill no tile."""
    import types
    import ctypes as C
    import raytracer_amd as R
    from raytracer_amd import _ffi as F
    w, h = size
    d = F.RtSceneDesc()
    d.abi_version = F.RT_ABI_VERSION
    d.root = 0xFFFFFFFF
    d.lights_kind = 0
    bg = (0.5, 0.7, 1.0)
    for k in range(3):
        d.background[k] = bg[k]
    view = R.HostScene(5, w / h)  # only for its camera (kept alive: the camera lives inside the host scene)
    ds = R.DeviceScene(types.SimpleNamespace(desc=d, camera=view.camera), 0)
    for kw in ({}, {"f64": True}, {"force_bvh": True}):
        a, st = ds.render(w, h, 5, 7, seed=3, stats=True, **kw)
        assert a.shape == (h, w, 3)
        np.testing.assert_allclose(a, np.broadcast_to(np.array(bg) * 5, a.shape), rtol=1e-6)
        assert st["rays"] == w * h * 5 and st["misses"] == w * h * 5 and st["closest_hits"] == 0
    for kw in ({}, {"f64": True}, {"fast": True}):
        ids = ds.primary_hits(w, h, **kw)[0]
        assert (ids == 0xFFFFFFFF).all()


def test_deepest_accepted_tree_renders_and_a_deeper_one_is_refused():
    """RT_MAX_BVH_HEIGHT: the reference-order kernels keep 4 B per tree level and lane in LDS.  A chain tree of
    exactly that height renders on every kernel family that walks it (f32 reference-order walk, f64 audit,
    and whatever the fast path picks) and hits the same spheres; one level more is RT_ERR_STACK_DEPTH at
    rt_scene_create, not an opaque launch failure later."""
    import types
    import raytracer_amd as R
    from desc_util import chain_desc
    from raytracer_amd import _ffi as F
    view = R.HostScene(5, 1.5)  # camera (13,2,3) -> origin sees the row of spheres
    d, keep = chain_desc(64)   # 63 nodes on the root-to-leaf path: the limit
    ds = R.DeviceScene(types.SimpleNamespace(desc=d, camera=view.camera), 0)
    w, h = 96, 64
    ids = {}
    for tag, kw in (("f32", {}), ("f64", {"f64": True}), ("fast", {"fast": True})):
        ids[tag] = ds.primary_hits(w, h, **kw)[0]
    assert np.array_equal(ids["f32"], ids["f64"]) and np.array_equal(ids["f32"], ids["fast"])
    assert len(np.unique(ids["f32"])) > 20  # most of the 64 spheres are seen
    a, sa = ds.render(w, h, 4, 6, seed=2, stats=True, force_bvh=True)
    b, sb = ds.render(w, h, 4, 6, seed=2, stats=True, f64=True)
    c, sc = ds.render(w, h, 4, 6, seed=2, stats=True)
    assert sa["nodes_visited"] > 0 and sa["rays"] == sc["rays"]
    assert abs(float(a.sum()) / float(b.sum()) - 1.0) < 1e-3
    np.testing.assert_allclose(a, c, rtol=1e-4, atol=1e-5)
    d2, keep2 = chain_desc(65)  # 64 nodes deep
    with pytest.raises(F.RtError) as e:
        R.DeviceScene(types.SimpleNamespace(desc=d2, camera=view.camera), 0)
    assert e.value.code == F.RT_ERR_STACK_DEPTH


def test_f32_drops_the_samples_that_poison_f64_pixels():
    """Q9/Q10: a light-sampled direction with pdf 0 makes the sample NaN; the reference (and the f64
    audit kernel) add it to the pixel, which quantises to 0 (main.rs:59-61, 165); the f32 product kernels drop and
    count it (rt.h, rt_render_device).  Scene: simple_light with its second light re-tagged RT_LIGHT_OTHER
    (random() = (1,0,0), pdf_value = 0: hittable.rs:498, 513), which makes such samples common.  At equal seed the
    two kernels follow the same paths almost everywhere, so the f32 kernel must report dropped samples, and the
    pixels the f64 kernel poisoned must be (nearly all) pixels where f32 dropped something: checked through the
    per-pixel deficit of the f32 image against a render that cannot drop (depth-limited sums are positive)."""
    import raytracer_amd as R
    from raytracer_amd import _ffi as F
    hs = R.HostScene(3, 1.5)
    d = F.RtSceneDesc()
    C.memmove(C.byref(d), C.byref(hs.desc), C.sizeof(d))
    tags = (C.c_uint8 * 2)(hs.desc.light_tag[0], 1)
    d.light_tag = C.cast(tags, type(d.light_tag))

    class Mutated(R.DeviceScene):
        def __init__(self):
            self.host = hs
            self.device = 0
            h = C.c_void_p()
            F.check_hip(F.hip.rt_scene_create(0, C.byref(d), C.byref(h)))
            self._h = h

    ds = Mutated()
    w, h, spp = 48, 32, 16
    f32, s32 = ds.render(w, h, spp, 3, seed=2, stats=True)
    f64, s64 = ds.render(w, h, spp, 3, seed=2, stats=True, f64=True)
    assert np.isfinite(f32).all()                       # f32: never poisoned
    bad64 = ~np.isfinite(f64).all(axis=-1)              # f64: pixels the reference would paint black
    assert s32["nonfinite_samples"] > 0 and s64["nonfinite_samples"] > 0
    assert bad64.sum() > 0
    # the f64 kernel counts poisoned SAMPLES too.  At equal seed the f32 kernel sees a SUBSET of its events: it also
    # stops a path whose throughput has reached exactly 0 (a light-sampled direction below the horizon:
    # scattering_pdf 0), and the reference walks such a path on, where a later vertex with pdf_value 0 turns the
    # 0 into 0 * inf = NaN.  Those samples contribute 0 in f32 and poison the pixel in the reference.
    assert 0.3 * s64["nonfinite_samples"] <= s32["nonfinite_samples"] <= 1.02 * s64["nonfinite_samples"] + 4
    # tone-mapped: the reference's pixel is 0 where poisoned, the f32 one is a finite colour
    u8_64 = R.tonemap(np.nan_to_num(f64, nan=np.nan).astype(np.float32), spp)
    assert (u8_64[bad64] == 0).all()


@pytest.mark.parametrize("size", [(2, 2), (37, 23), (3, 130), (130, 3), (49, 33)])
@pytest.mark.parametrize("spp", [1, 3, 11])
def test_units_kernel_on_ragged_images_and_odd_sample_counts(size, spp):
    """render_units_kernel (persistent waves, 8x8-tile x sample-chunk units, two open units per wave): images that fill
    no 16x16 block (units whose tile lies partly or wholly outside the image) and sample counts that are no multiple of
    the unit size must render exactly the reference-order walk's paths — equal ray counts, pixel sums equal to f32
    summation order — and add on top of an existing accumulator."""
    import torch
    import raytracer_amd as R
    w, h = size
    hs = R.HostScene(7, w / h, n=3000)
    ds = R.DeviceScene(hs, 0)
    a, sa = ds.render(w, h, spp, 50, seed=4, sample_offset=2, stats=True)
    b, sb = ds.render(w, h, spp, 50, seed=4, sample_offset=2, stats=True, force_bvh=True)
    assert sa["rays"] == sb["rays"] and sa["nonfinite_samples"] == sb["nonfinite_samples"]
    fin = np.isfinite(a) & np.isfinite(b)
    np.testing.assert_allclose(a[fin], b[fin], rtol=3e-5, atol=1e-6)
    acc = torch.full((h, w, 3), 2.0, dtype=torch.float32, device="cuda:0")
    ds.render_device(acc.data_ptr(), w, h, spp, 50, seed=4, sample_offset=2, accumulate=True)
    np.testing.assert_allclose(acc.cpu().numpy()[fin], (a + 2.0)[fin], rtol=3e-5, atol=1e-5)


@pytest.mark.parametrize("size", [(2, 2), (37, 23), (130, 3)])
def test_f64_persistent_kernel_on_ragged_images_accumulates_on_the_device(size):
    """The f64 audit kernel runs in the persistent form (units of an 8x8 tile x a few samples, f64 atomics): images that
    fill no tile, sample counts that are no multiple of the unit size, and adding onto an existing device accumulator
    must give the oracle's sums."""
    import torch
    import oracle_ffi as O
    import raytracer_amd as R
    w, h = size
    hs = R.HostScene(6, w / h)
    ds = R.DeviceScene(hs, 0)
    o, ocnt, _ = O.OracleScene(6, w / h).render(w, h, 11, 50, seed=4, sample_offset=2)
    acc = torch.full((h, w, 3), 2.0, dtype=torch.float64, device="cuda:0")
    s1 = ds.render_device(acc.data_ptr(), w, h, 7, 50, seed=4, sample_offset=2, accumulate=True, f64=True, stats=True)
    s2 = ds.render_device(acc.data_ptr(), w, h, 4, 50, seed=4, sample_offset=9, accumulate=True, f64=True, stats=True)
    torch.cuda.synchronize()
    assert s1["rays"] + s2["rays"] == ocnt["rays"]
    got = acc.cpu().numpy() - 2.0
    fin = np.isfinite(o)
    assert (np.isfinite(got) == fin).all()
    np.testing.assert_allclose(got[fin], o[fin], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("size", [(2, 2), (37, 23), (130, 3), (49, 33)])
@pytest.mark.parametrize("spp", [1, 3, 11])
def test_f64_ordered_walk_on_ragged_images_and_odd_sample_counts(size, spp):
    """The f64 kernel on a BVH-bound scene (ordered walk as a resumable state machine inside the persistent kernel: lanes
    whose walk is over go on to their vertex while their neighbours keep walking): ragged images, sample counts that are
    no multiple of the unit size, depths 1 and 50 — the oracle's sums and ray count, and equal to the reference-order f64
    walk of the same tree, also on top of an existing accumulator."""
    import torch
    import oracle_ffi as O
    import raytracer_amd as R
    w, h = size
    hs = R.HostScene(7, w / h, n=3000)
    ds = R.DeviceScene(hs, 0)
    orc = O.OracleScene(7, w / h, n=3000)
    for depth in (1, 50):
        o, ocnt, _ = orc.render(w, h, spp, depth, seed=4, sample_offset=2)
        a, sa = ds.render(w, h, spp, depth, seed=4, sample_offset=2, stats=True, f64=True)
        b, sb = ds.render(w, h, spp, depth, seed=4, sample_offset=2, stats=True, f64=True, force_bvh=True)
        assert sa["rays"] == sb["rays"] == ocnt["rays"]
        fin = np.isfinite(o)
        assert (np.isfinite(a) == fin).all() and (np.isfinite(b) == fin).all()
        np.testing.assert_allclose(a[fin], o[fin], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(a[fin], b[fin], rtol=1e-12, atol=1e-12)
    acc = torch.full((h, w, 3), 2.0, dtype=torch.float64, device="cuda:0")
    ds.render_device(acc.data_ptr(), w, h, spp, 50, seed=4, sample_offset=2, accumulate=True, f64=True)
    np.testing.assert_allclose(acc.cpu().numpy()[fin], (a + 2.0)[fin], rtol=1e-12, atol=1e-12)


def test_back_face_of_an_emitter_still_multiplies_its_zero_up_the_path():
    """material.rs:118-126 + main.rs:93-97: a light hit from behind emits Color::new(), and every caller up the
    recursion still multiplies that zero into attenuation * scattering_pdf / pdf_value — NaN where a vertex had
    pdf_value 0 (Q9).  Found by tests/sweep_mixed.py (generated scene, seed 1181, 150 objects: sample 2 of pixel
    (48, 38) ends on the back of a light behind such a vertex): the f64 kernel's NaN pixels are the oracle's."""
    import oracle_ffi as O
    import raytracer_amd as R
    w, asp, seed = 64, 1.5, 1181
    h = R.image_height(w, asp)
    hs = R.HostScene(9, asp, n=150, scene_seed=seed)
    ds = R.DeviceScene(hs, 0)
    g, _ = ds.render(w, h, 4, 10, seed=seed, f64=True)
    o, _, _ = O.OracleScene(9, asp, n=150, scene_seed=seed).render(w, h, 4, 10, seed=seed)
    assert np.isnan(o[38, 48]).all() and np.isnan(g[38, 48]).all()
    assert (np.isfinite(o) == np.isfinite(g)).all()
    # f32: a non-finite sample is dropped (or the path has stopped at zero throughput before): the image stays finite
    f, sf = ds.render(w, h, 4, 10, seed=seed, stats=True)
    assert np.isfinite(f).all()


def test_two_scenes_on_two_streams_overlap_without_interference():
    """include/rt.h: "use one rt_scene per stream".  Two scenes (a leaf-list one and a BVH-bound one), each with its own
    stream, enqueued back to back without any host wait in between (stats == NULL keeps the calls asynchronous), several
    rounds: each image equals the one the scene renders alone."""
    import torch
    import raytracer_amd as R
    a = R.DeviceScene(R.HostScene(6, 1.0), 0)
    b = R.DeviceScene(R.HostScene(0, 16 / 9), 0)
    w = 160
    ha, hb = R.image_height(w, 1.0), R.image_height(w, 16 / 9)
    ref_a = torch.zeros((ha, w, 3), dtype=torch.float32, device="cuda:0")
    ref_b = torch.zeros((hb, w, 3), dtype=torch.float32, device="cuda:0")
    a.render_device(ref_a.data_ptr(), w, ha, 64, 50, seed=7, want_stats=False)
    b.render_device(ref_b.data_ptr(), w, hb, 64, 50, seed=7, want_stats=False)
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    out_a, out_b = torch.zeros_like(ref_a), torch.zeros_like(ref_b)
    for _ in range(3):
        out_a.fill_(-3.0)
        out_b.fill_(-3.0)
        torch.cuda.synchronize()
        a.render_device(out_a.data_ptr(), w, ha, 64, 50, seed=7, want_stats=False, stream=sa.cuda_stream)
        b.render_device(out_b.data_ptr(), w, hb, 64, 50, seed=7, want_stats=False, stream=sb.cuda_stream)
        a.render_device(out_a.data_ptr(), w, ha, 64, 50, seed=7, sample_offset=64, accumulate=True, want_stats=False,
                        stream=sa.cuda_stream)
        torch.cuda.synchronize()
        np.testing.assert_allclose(out_b.cpu().numpy(), ref_b.cpu().numpy(), rtol=3e-5, atol=1e-5)
        both = torch.zeros_like(ref_a)
        a.render_device(both.data_ptr(), w, ha, 128, 50, seed=7, want_stats=False)
        torch.cuda.synchronize()
        np.testing.assert_allclose(out_a.cpu().numpy(), both.cpu().numpy(), rtol=3e-5, atol=1e-4)


@pytest.mark.parametrize("sid,asp,kw", [(6, 1.0, {}), (0, 16 / 9, {}), (6, 1.0, {"f64": True})])
def test_one_scene_on_two_streams_overlaps_without_interference(sid, asp, kw):
    """include/rt.h "STREAMS AND THREADS": every render call takes a counter block of its own (the persistent kernels'
    unit counter lives there), so two calls on ONE rt_scene enqueued on two streams without a host wait in between —
    they overlap on the GPU — each render every unit exactly once: both images equal the scene's lone render of the
    same samples (persistent leaf-list kernel, units kernel, f64 kernel)."""
    import torch
    import raytracer_amd as R
    ds = R.DeviceScene(R.HostScene(sid, asp), 0)
    w = 192
    h = R.image_height(w, asp)
    dt = torch.float64 if kw else torch.float32
    spp = 96 if not kw else 32
    refs = []
    for off in (0, 1000):
        r = torch.zeros((h, w, 3), dtype=dt, device="cuda:0")
        ds.render_device(r.data_ptr(), w, h, spp, 50, seed=9, sample_offset=off, want_stats=False, **kw)
        torch.cuda.synchronize()
        refs.append(r.cpu().numpy())
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    o1 = torch.zeros((h, w, 3), dtype=dt, device="cuda:0")
    o2 = torch.zeros_like(o1)
    for _ in range(3):
        o1.fill_(-5.0)
        o2.fill_(-5.0)
        torch.cuda.synchronize()
        ds.render_device(o1.data_ptr(), w, h, spp, 50, seed=9, sample_offset=0, want_stats=False, stream=s1.cuda_stream, **kw)
        ds.render_device(o2.data_ptr(), w, h, spp, 50, seed=9, sample_offset=1000, want_stats=False, stream=s2.cuda_stream, **kw)
        torch.cuda.synchronize()
        tol = dict(rtol=1e-11, atol=1e-11) if kw else dict(rtol=3e-5, atol=1e-4)
        np.testing.assert_allclose(o1.cpu().numpy(), refs[0], **tol)
        np.testing.assert_allclose(o2.cpu().numpy(), refs[1], **tol)
