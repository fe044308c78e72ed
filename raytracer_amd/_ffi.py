"""ctypes bindings of the two native libraries (no torch types cross this boundary).

librt_hip.so  — include/rt.h: the gfx950 kernels behind the C ABI (the product).
librt_host.so — include/rt_host.h: C++17 host layer (scenes, flat BVH, flatten, camera).

There is no CPU fallback: if either library is missing, importing this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

RT_ABI_VERSION = 3
RT_REF_LEAF = 0x80000000
RT_REF_NONE = 0xFFFFFFFF
RT_LEAF_SPHERE, RT_LEAF_MSPHERE, RT_LEAF_CUBE, RT_LEAF_COMPLEX = 0, 1, 2, 3
RT_FLAG_STATS = 1
RT_FLAG_FORCE_BVH = 2
RT_FLAG_WAVEFRONT = 4
RT_OK, RT_ERR_BAD_ARG, RT_ERR_HIP, RT_ERR_UNSUPPORTED, RT_ERR_STACK_DEPTH = 0, 1, 2, 3, 4

_p = C.c_void_p
_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)
_f64p = C.POINTER(C.c_double)
_u64p = C.POINTER(C.c_uint64)


class RtSceneDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("bvh_height", C.c_uint32),
        ("n_nodes", C.c_uint32), ("root", C.c_uint32),
        ("node_box", _f64p), ("node_left", _u32p), ("node_right", _u32p),
        ("n_spheres", C.c_uint32), ("sphere", _f64p), ("sphere_material", _u32p), ("sphere_object", _u32p),
        ("n_mspheres", C.c_uint32), ("msphere", _f64p), ("msphere_material", _u32p), ("msphere_object", _u32p),
        ("n_cubes", C.c_uint32), ("cube", _f64p), ("cube_material", _u32p), ("cube_object", _u32p),
        ("n_complex", C.c_uint32), ("complex_body", _u32p), ("complex_op_first", _u32p),
        ("complex_op_count", _u32p), ("complex_material", _u32p), ("complex_object", _u32p),
        ("n_ops", C.c_uint32), ("op_tag", _u8p), ("op_data", _f64p),
        ("n_materials", C.c_uint32), ("mat_tag", _u8p), ("mat_texture", _u32p), ("mat_params", _f64p),
        ("n_textures", C.c_uint32), ("tex_tag", _u8p), ("tex_color", _f64p), ("tex_odd", _u32p),
        ("tex_even", _u32p), ("tex_texel_offset", _u64p), ("tex_width", _u32p), ("tex_height", _u32p),
        ("n_texel_bytes", C.c_uint64), ("texels", _u8p),
        ("lights_kind", C.c_uint32), ("n_lights", C.c_uint32), ("light_tag", _u8p), ("light_sphere", _f64p), ("light_box", _f64p),
        ("background", C.c_double * 3),
    ]


class RtCamera(C.Structure):
    _fields_ = [
        ("origin", C.c_double * 3), ("lower_left", C.c_double * 3), ("horizontal", C.c_double * 3),
        ("vertical", C.c_double * 3), ("u", C.c_double * 3), ("v", C.c_double * 3),
        ("lens_radius", C.c_double), ("time_open", C.c_double), ("time_close", C.c_double),
    ]


class RtParams(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("max_depth", C.c_uint32),
        ("seed", C.c_uint64), ("sample_offset", C.c_uint64), ("spp_per_launch", C.c_uint32), ("flags", C.c_uint32),
    ]


STAT_FIELDS = [
    "rays", "light_probes", "medium_probes", "nodes_visited", "sphere_tests", "sphere_hits", "cube_tests",
    "cube_hits", "rotate_visits", "translate_visits", "medium_visits", "closest_hits", "diffuse_vertices",
    "specular_vertices", "emitter_hits", "misses", "nonfinite_samples",
]


class RtStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in STAT_FIELDS] + [
        ("launches", C.c_uint32), ("reserved", C.c_uint32), ("kernel_ms", C.c_double)]

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n in STAT_FIELDS}
        d["launches"] = int(self.launches)
        d["kernel_ms"] = float(self.kernel_ms)
        return d


class RthSceneOpts(C.Structure):
    _fields_ = [
        ("scene_id", C.c_int32), ("rotate_bbox", C.c_int32), ("aspect_ratio", C.c_double),
        ("scene_seed", C.c_uint64), ("bvh_seed", C.c_uint64), ("n", C.c_uint32),
        ("tex_width", C.c_uint32), ("tex_height", C.c_uint32), ("texels", _p),
    ]


class RthRenderOpts(C.Structure):
    _fields_ = [
        ("scene", RthSceneOpts), ("image_width", C.c_uint32), ("samples_per_pixel", C.c_uint32),
        ("max_depth", C.c_uint32), ("device", C.c_int32), ("seed", C.c_uint64),
        ("precision", C.c_uint32), ("reserved", C.c_uint32),
    ]


# every symbol include/rt.h and include/rt_host.h declare (tests check the export lists against the headers)
RT_SYMBOLS = [
    "rt_scene_create", "rt_scene_destroy", "rt_primary_hits", "rt_primary_hits_f64", "rt_primary_hits_fast", "rt_primary_redecided", "rt_render",
    "rt_render_f64", "rt_render_device", "rt_render_f64_device", "rt_render_device_noalloc", "rt_render_multi", "rt_multi_create", "rt_multi_render", "rt_multi_render_f64", "rt_multi_destroy", "rt_multi_shard", "rt_debug_build_tree", "rt_tonemap", "rt_tonemap_device", "rt_tonemap_f64", "rt_tonemap_f64_device", "rt_kat_sphere_uv", "rt_kat_rotate",
    "rt_kat_philox", "rt_kat_libm", "rt_device_count", "rt_abi_version", "rt_last_error",
]
RTH_SYMBOLS = [
    "rth_scene_build", "rth_scene_free", "rth_scene_desc", "rth_scene_camera", "rth_scene_object_count",
    "rth_scene_bvh_calls", "rth_camera_new", "rth_render", "rth_image_height", "rth_write_png", "rth_rotate",
    "rth_bvh_axis", "rth_texture_from_rgb8_file", "rth_texture_from_image_file", "rth_texture_free", "rth_kat_chacha", "rth_kat_chacha_quarter",
    "rth_kat_stdrng",
    "rth_last_error",
]


def _load(name):
    path = os.path.join(LIB_DIR, name)
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C raytracer_amd/csrc`). raytracer_amd has no CPU fallback.")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


# RT_HIP_LIBNAME: developer switch for A/B-ing kernel builds (tools/build_variant.sh); default = the product
hip = _load(os.environ.get("RT_HIP_LIBNAME", "librt_hip.so"))
host = _load("librt_host.so")

hip.rt_last_error.restype = C.c_char_p
hip.rt_abi_version.restype = C.c_uint32
hip.rt_device_count.restype = C.c_int
hip.rt_scene_create.argtypes = [C.c_int, C.POINTER(RtSceneDesc), C.POINTER(_p)]
hip.rt_scene_destroy.argtypes = [_p]
hip.rt_scene_destroy.restype = None
hip.rt_primary_hits.argtypes = [_p, C.POINTER(RtCamera), C.c_uint32, C.c_uint32, C.c_uint64, C.c_double,
                                _p, _p, _p, _p, _p]
hip.rt_primary_hits_f64.argtypes = hip.rt_primary_hits.argtypes
hip.rt_primary_hits_fast.argtypes = hip.rt_primary_hits.argtypes
hip.rt_primary_redecided.argtypes = [_p]
hip.rt_primary_redecided.restype = C.c_uint64
hip.rt_render.argtypes = [_p, C.POINTER(RtCamera), C.POINTER(RtParams), _p, C.POINTER(RtStats)]
hip.rt_render_f64.argtypes = hip.rt_render.argtypes
hip.rt_render_device.argtypes = [_p, C.POINTER(RtCamera), C.POINTER(RtParams), _p, C.c_int, _p, C.POINTER(RtStats)]
hip.rt_render_f64_device.argtypes = hip.rt_render_device.argtypes
hip.rt_render_device_noalloc.argtypes = hip.rt_render_device.argtypes
hip.rt_render_multi.argtypes = [C.c_int, C.POINTER(C.c_int), _p, C.POINTER(RtCamera), C.POINTER(RtParams), _p,
                                C.POINTER(RtStats)]
hip.rt_multi_create.argtypes = [C.c_int, C.POINTER(C.c_int), _p, C.POINTER(_p)]
hip.rt_multi_render.argtypes = [_p, C.POINTER(RtCamera), C.POINTER(RtParams), _p, C.POINTER(RtStats)]
hip.rt_multi_render_f64.argtypes = [_p, C.POINTER(RtCamera), C.POINTER(RtParams), _p, C.POINTER(RtStats)]
hip.rt_multi_destroy.argtypes = [_p]
hip.rt_multi_destroy.restype = None
hip.rt_multi_shard.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, _u64p, _u64p]
hip.rt_multi_shard.restype = None
hip.rt_debug_build_tree.argtypes = [C.c_uint32, _p, _p, C.c_uint32, _p, C.POINTER(C.c_uint32)]
hip.rt_tonemap.argtypes = [_p, C.c_uint64, C.c_uint64, _p]
hip.rt_tonemap.restype = None
hip.rt_tonemap_device.argtypes = [C.c_int, _p, C.c_uint64, C.c_uint64, _p, _p]
if hasattr(hip, "rt_tonemap_f64") or "RT_HIP_LIBNAME" not in os.environ:  # (an older A/B build may lack them)
    hip.rt_tonemap_f64.argtypes = [_p, C.c_uint64, C.c_uint64, _p]
    hip.rt_tonemap_f64.restype = None
    hip.rt_tonemap_f64_device.argtypes = [C.c_int, _p, C.c_uint64, C.c_uint64, _p, _p]
hip.rt_kat_sphere_uv.argtypes = [C.c_int, C.c_uint32, _p, _p, C.c_int]
hip.rt_kat_rotate.argtypes = [C.c_int, C.c_uint32, _p, _p, _p, _p, C.c_int]
hip.rt_kat_philox.argtypes = [C.c_int, C.c_uint32, _p, _p, C.c_uint32, _p]
hip.rt_kat_libm.argtypes = [C.c_int, C.c_uint32, C.c_uint32, _p, _p, _p]

host.rth_last_error.restype = C.c_char_p
host.rth_scene_build.argtypes = [C.POINTER(RthSceneOpts), C.POINTER(_p)]
host.rth_scene_free.argtypes = [_p]
host.rth_scene_free.restype = None
host.rth_scene_desc.argtypes = [_p]
host.rth_scene_desc.restype = C.POINTER(RtSceneDesc)
host.rth_scene_camera.argtypes = [_p]
host.rth_scene_camera.restype = C.POINTER(RtCamera)
host.rth_scene_object_count.argtypes = [_p]
host.rth_scene_object_count.restype = C.c_uint32
host.rth_scene_bvh_calls.argtypes = [_p]
host.rth_scene_bvh_calls.restype = C.c_uint32
host.rth_camera_new.argtypes = [_p, _p, _p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                C.c_double, C.POINTER(RtCamera)]
host.rth_render.argtypes = [C.POINTER(RthRenderOpts), _p, _p, _u32p, C.POINTER(RtStats)]
host.rth_image_height.argtypes = [C.c_uint32, C.c_double]
host.rth_image_height.restype = C.c_uint32
host.rth_write_png.argtypes = [C.c_char_p, _p, C.c_uint32, C.c_uint32]
host.rth_texture_from_rgb8_file.argtypes = [C.c_char_p, _u32p, _u32p, C.POINTER(_u8p)]
host.rth_texture_from_image_file.argtypes = host.rth_texture_from_rgb8_file.argtypes
host.rth_texture_free.argtypes = [_u8p]
host.rth_texture_free.restype = None
host.rth_rotate.argtypes = [C.c_double, _p, _p, _p]
host.rth_rotate.restype = None
host.rth_bvh_axis.argtypes = [C.c_uint64, C.c_uint32]
host.rth_bvh_axis.restype = C.c_int
host.rth_kat_chacha.argtypes = [_p, C.c_uint32, _p]
host.rth_kat_chacha.restype = None
host.rth_kat_chacha_quarter.argtypes = [_p]
host.rth_kat_chacha_quarter.restype = None
host.rth_kat_stdrng.argtypes = [C.c_uint64, C.c_uint32, _p, _p, _p, C.c_double, C.c_double]
host.rth_kat_stdrng.restype = None


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[rt_status {code}] {msg}")
        self.code = code


def check_hip(rc):
    if rc != 0:
        raise RtError(rc, hip.rt_last_error().decode())


def check_host(rc):
    if rc != 0:
        raise RtError(rc, host.rth_last_error().decode())
