"""rt_scene_create must answer ANY descriptor with a status, never with a crash: a valid descriptor (scenes that use
every array: wrappers, media, textures incl. an image and a Perlin table, several lights, the box light) is copied,
one array at a time is filled with hostile values (NaN, infinities, huge and wrapped indices, zeros), and the call
must return RT_ERR_BAD_ARG / RT_ERR_UNSUPPORTED / RT_ERR_STACK_DEPTH — or accept the scene (RT_OK with a GPU,
RT_ERR_HIP without one: validation comes before the first device call).  Runs without a GPU."""
import ctypes as C

import numpy as np
import pytest

# field -> (count field or constant, elements per item, numpy dtype)
ARRAYS = {
    "node_box": ("n_nodes", 6, np.float64), "node_left": ("n_nodes", 1, np.uint32), "node_right": ("n_nodes", 1, np.uint32),
    "sphere": ("n_spheres", 4, np.float64), "sphere_material": ("n_spheres", 1, np.uint32), "sphere_object": ("n_spheres", 1, np.uint32),
    "msphere": ("n_mspheres", 9, np.float64), "msphere_material": ("n_mspheres", 1, np.uint32), "msphere_object": ("n_mspheres", 1, np.uint32),
    "cube": ("n_cubes", 6, np.float64), "cube_material": ("n_cubes", 1, np.uint32), "cube_object": ("n_cubes", 1, np.uint32),
    "complex_body": ("n_complex", 1, np.uint32), "complex_op_first": ("n_complex", 1, np.uint32),
    "complex_op_count": ("n_complex", 1, np.uint32), "complex_material": ("n_complex", 1, np.uint32),
    "complex_object": ("n_complex", 1, np.uint32),
    "op_tag": ("n_ops", 1, np.uint8), "op_data": ("n_ops", 4, np.float64),
    "mat_tag": ("n_materials", 1, np.uint8), "mat_texture": ("n_materials", 1, np.uint32), "mat_params": ("n_materials", 4, np.float64),
    "tex_tag": ("n_textures", 1, np.uint8), "tex_color": ("n_textures", 3, np.float64), "tex_odd": ("n_textures", 1, np.uint32),
    "tex_even": ("n_textures", 1, np.uint32), "tex_texel_offset": ("n_textures", 1, np.uint64),
    "tex_width": ("n_textures", 1, np.uint32), "tex_height": ("n_textures", 1, np.uint32),
    "light_tag": ("n_lights", 1, np.uint8), "light_sphere": ("n_lights", 4, np.float64), "light_box": ("n_lights", 6, np.float64),
}
SCALARS = ["bvh_height", "n_nodes", "root", "n_spheres", "n_mspheres", "n_cubes", "n_complex", "n_ops", "n_materials",
           "n_textures", "n_texel_bytes", "lights_kind", "n_lights"]
HOSTILE_F = [np.nan, np.inf, -np.inf, 0.0, -0.0, 1e308, -1e308, 5e-324, -1.0]
HOSTILE_U = [0, 1, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 0xFFFFFFFE, 0x80000001, 0x3FFFFFFF, 0xB0000000, 0x8FFFFFFF]
OK_CODES = None  # filled in below


RENDER = {"cam": None}  # set by the GPU test: accepted descriptors are also rendered (tiny image, both precisions)


def _call(F, d):
    out = C.c_void_p()
    rc = F.hip.rt_scene_create(0, C.byref(d), C.byref(out))
    assert rc in (F.RT_OK, F.RT_ERR_BAD_ARG, F.RT_ERR_HIP, F.RT_ERR_UNSUPPORTED, F.RT_ERR_STACK_DEPTH), rc
    if rc == F.RT_OK:
        if RENDER["cam"] is not None:
            p = F.RtParams()
            p.width, p.height, p.spp, p.max_depth, p.seed = 8, 8, 2, 6, 1
            st = F.RtStats()
            acc = np.zeros((8, 8, 3), dtype=np.float32)
            acc64 = np.zeros((8, 8, 3), dtype=np.float64)
            ids = np.zeros((8, 8), dtype=np.uint32)
            t = np.zeros((8, 8), dtype=np.float32)
            nrm = np.zeros((8, 8, 3), dtype=np.float32)
            uv = np.zeros((8, 8, 2), dtype=np.float32)
            ff = np.zeros((8, 8), dtype=np.uint8)
            codes = (F.RT_OK, F.RT_ERR_BAD_ARG, F.RT_ERR_UNSUPPORTED)
            assert F.hip.rt_render(out, C.byref(RENDER["cam"]), C.byref(p), acc.ctypes.data_as(C.c_void_p), C.byref(st)) in codes
            assert F.hip.rt_render_f64(out, C.byref(RENDER["cam"]), C.byref(p), acc64.ctypes.data_as(C.c_void_p), C.byref(st)) in codes
            assert F.hip.rt_primary_hits(out, C.byref(RENDER["cam"]), 8, 8, 1, 0.5, *(a.ctypes.data_as(C.c_void_p) for a in (ids, t, nrm, uv, ff))) in codes
        F.hip.rt_scene_destroy(out)
    else:
        assert len(F.hip.rt_last_error()) > 0
    return rc


def _array(d, name):
    cnt, per, dt = ARRAYS[name]
    n = int(getattr(d, cnt)) * per
    ptr = getattr(d, name)
    if n == 0 or not ptr:
        return None
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dt).copy() if np.dtype(dt).itemsize == C.sizeof(ptr._type_) else None


@pytest.mark.parametrize("scene", [(9, 1.5, 0, 3), (9, 1.5, 0, 6), (8, 1.0, 0, 5), (11, 1.0, 0, 5), (12, 1.0, 0, 5), (0, 1.5, 0, 5), (3, 1.5, 0, 5)])
def test_hostile_descriptors_get_a_status(scene, earthmap):
    import raytracer_amd as R
    from raytracer_amd import _ffi as F
    sid, asp, n, seed = scene
    hs = R.HostScene(sid, asp, n=n, scene_seed=seed)
    base = _call(F, F.RtSceneDesc.from_buffer_copy(hs.desc))
    assert base in (F.RT_OK, F.RT_ERR_HIP)  # the unmodified descriptor is valid
    rng = np.random.default_rng(1000 * sid + seed)
    refused = accepted = 0
    keep = []
    for name in ARRAYS:
        src = _array(hs.desc, name)
        if src is None:
            continue
        hostile = HOSTILE_F if src.dtype == np.float64 else HOSTILE_U
        for trial in range(12):
            a = src.copy()
            k = 1 if trial < 6 else max(1, len(a) // 3)
            idx = rng.choice(len(a), size=min(k, len(a)), replace=False)
            vals = rng.choice(hostile, size=len(idx))
            if a.dtype == np.uint8:
                vals = np.asarray(vals, dtype=np.uint64) & 0xFF
            a[idx] = np.asarray(vals).astype(a.dtype, copy=False) if a.dtype != np.float64 else vals
            d = F.RtSceneDesc.from_buffer_copy(hs.desc)
            keep.append(a)
            setattr(d, name, a.ctypes.data_as(type(getattr(d, name))))
            rc = _call(F, d)
            refused += rc not in (F.RT_OK, F.RT_ERR_HIP)
            accepted += rc in (F.RT_OK, F.RT_ERR_HIP)
    # counts and scalar fields: shrink, zero, off by one (never larger than the arrays that back them)
    for name in SCALARS:
        v = int(getattr(hs.desc, name))
        for nv in {0, max(0, v - 1), v // 2, 1} | ({v + 1, 0xFFFFFFFF, 0x80000000} if name in ("root", "bvh_height", "lights_kind") else set()):
            d = F.RtSceneDesc.from_buffer_copy(hs.desc)
            setattr(d, name, nv)
            rc = _call(F, d)
            refused += rc not in (F.RT_OK, F.RT_ERR_HIP)
    # null pointers one at a time
    for name in list(ARRAYS) + ["texels"]:
        d = F.RtSceneDesc.from_buffer_copy(hs.desc)
        setattr(d, name, None)
        _call(F, d)
    print(f"scene {sid}: {refused} refused, {accepted} accepted")
    assert refused > 50


@pytest.mark.gpu
@pytest.mark.parametrize("scene", [(9, 1.5, 0, 3), (8, 1.0, 0, 5), (12, 1.0, 0, 5), (0, 1.5, 0, 5)])
def test_hostile_descriptors_that_are_accepted_also_render(scene, earthmap):
    """the same sweep on the GPU, and every descriptor that rt_scene_create accepts is rendered (8x8, f32 and f64
    kernels, audited primary hits): NaN / infinite geometry and wrong-but-in-range indices may give any image, but the
    kernels must come back"""
    import raytracer_amd as R
    hs = R.HostScene(scene[0], scene[1], n=scene[2], scene_seed=scene[3])
    RENDER["cam"] = hs.camera
    try:
        test_hostile_descriptors_get_a_status(scene, earthmap)
    finally:
        RENDER["cam"] = None
