"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module. The product package (raytracer_amd/) never does.
"""
import ctypes as C
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB_PATH = os.path.join(_ROOT, "oracle", "liboracle.so")

NO_OBJECT = 0xFFFFFFFF
LEAF_BIT = 0x80000000


class OrcCamera(C.Structure):
    _fields_ = [
        ("origin", C.c_double * 3),
        ("lower_left", C.c_double * 3),
        ("horizontal", C.c_double * 3),
        ("vertical", C.c_double * 3),
        ("u", C.c_double * 3),
        ("v", C.c_double * 3),
        ("lens_radius", C.c_double),
        ("time_open", C.c_double),
        ("time_close", C.c_double),
    ]


class OrcInfo(C.Structure):
    _fields_ = [
        ("n_objects", C.c_uint32),
        ("n_nodes", C.c_uint32),
        ("has_lights", C.c_uint32),
        ("n_lights", C.c_uint32),
        ("rotate_bbox_mode", C.c_uint32),
        ("root_is_leaf", C.c_uint32),
        ("background", C.c_double * 3),
        ("camera", OrcCamera),
    ]


COUNTER_FIELDS = [
    "rays", "light_probes", "medium_probes", "nodes_visited", "sphere_tests", "sphere_hits",
    "cube_tests", "cube_hits", "rotate_visits", "translate_visits", "medium_visits",
    "closest_hits", "diffuse_vertices", "specular_vertices", "emitter_hits", "misses",
    "nonfinite_samples",
]


class OrcCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_FIELDS]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_FIELDS}


_libs = {}


def lib(name="liboracle.so"):
    """liboracle.so: the checker (counts its work, -O2, no fp contraction).  liboracle_nocount.so: the same
    sources at -O3 without the work counters, for bench.py's timed CPU baseline only."""
    if name not in _libs:
        path = name if os.path.isabs(name) else os.path.join(_ROOT, "oracle", name)  # (absolute: a build made elsewhere)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.orc_last_error.restype = C.c_char_p
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.c_int, C.c_double, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int,
                                       C.c_void_p, C.c_uint32, C.c_uint32]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_info.argtypes = [C.c_void_p, C.POINTER(OrcInfo)]
        L.orc_bvh_preorder.restype = C.c_uint32
        L.orc_bvh_preorder.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_primary_hits.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_double,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_render.restype = C.c_double
        L.orc_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                 C.c_uint64, C.c_uint32, C.c_void_p, C.POINTER(OrcCounters)]
        L.orc_tonemap.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
        L.orc_get_sphere_uv.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_rotate.argtypes = [C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_philox.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_bvh_axis.restype = C.c_int
        L.orc_bvh_axis.argtypes = [C.c_uint64, C.c_uint32]
        L.orc_scene_draws.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_path_draws.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        _libs[name] = L
    return _libs[name]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleScene:
    def __init__(self, scene_id, aspect_ratio, scene_seed=5, bvh_seed=7, n=0, rotate_bbox=-1, texels=None,
                 library="liboracle.so"):
        self._lib = L = lib(library)
        tw = th = 0
        tp = None
        if texels is not None:
            texels = np.ascontiguousarray(texels, dtype=np.uint8)
            th, tw = texels.shape[:2]
            tp = _ptr(texels)
        self._h = L.orc_scene_create(scene_id, aspect_ratio, scene_seed, bvh_seed, n, rotate_bbox, tp, tw, th)
        if not self._h:
            raise RuntimeError("orc_scene_create: " + L.orc_last_error().decode())
        self.info = OrcInfo()
        L.orc_scene_info(self._h, C.byref(self.info))

    def close(self):
        if self._h:
            self._lib.orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def camera_array(self):
        c = self.info.camera
        return np.array(list(c.origin) + list(c.lower_left) + list(c.horizontal) + list(c.vertical)
                        + list(c.u) + list(c.v) + [c.lens_radius, c.time_open, c.time_close])

    def bvh_preorder(self):
        n = self.info.n_nodes
        boxes = np.zeros((max(n, 1), 6), dtype=np.float64)
        left = np.zeros(max(n, 1), dtype=np.uint32)
        right = np.zeros(max(n, 1), dtype=np.uint32)
        root = self._lib.orc_bvh_preorder(self._h, _ptr(boxes), _ptr(left), _ptr(right))
        return root, boxes[:n], left[:n], right[:n]

    def primary_hits(self, w, h, seed=1, time=0.0):
        ids = np.zeros((h, w), dtype=np.uint32)
        t = np.zeros((h, w), dtype=np.float64)
        nrm = np.zeros((h, w, 3), dtype=np.float64)
        uv = np.zeros((h, w, 2), dtype=np.float64)
        ff = np.zeros((h, w), dtype=np.uint8)
        self._lib.orc_primary_hits(self._h, w, h, seed, time, _ptr(ids), _ptr(t), _ptr(nrm), _ptr(uv), _ptr(ff))
        return ids, t, nrm, uv, ff

    def render(self, w, h, spp, max_depth, seed=1, sample_offset=0, threads=None):
        if threads is None:
            threads = os.cpu_count() or 1
        accum = np.zeros((h, w, 3), dtype=np.float64)
        cnt = OrcCounters()
        secs = self._lib.orc_render(self._h, w, h, spp, max_depth, seed, sample_offset, threads, _ptr(accum),
                                C.byref(cnt))
        return accum, cnt.as_dict(), secs


def tonemap(accum, spp):
    accum = np.ascontiguousarray(accum, dtype=np.float64)
    out = np.zeros(accum.shape, dtype=np.uint8)
    lib().orc_tonemap(_ptr(accum), accum.size // 3, spp, _ptr(out))
    return out


def get_sphere_uv(p):
    p = np.asarray(p, dtype=np.float64)
    uv = np.zeros(2)
    lib().orc_get_sphere_uv(_ptr(p), _ptr(uv))
    return uv


def rotate(angle_rad, axis, v):
    axis = np.asarray(axis, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    out = np.zeros(3)
    lib().orc_rotate(angle_rad, _ptr(axis), _ptr(v), _ptr(out))
    return out


def philox(ctr, key, rounds=10):
    ctr = np.asarray(ctr, dtype=np.uint32)
    key = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox(_ptr(ctr), _ptr(key), rounds, _ptr(out))
    return out


def scene_draws(seed, stream, n):
    out = np.zeros(n)
    lib().orc_scene_draws(seed, stream, n, _ptr(out))
    return out


def path_draws(seed, pixel, sample, slot, n):
    out = np.zeros(n)
    lib().orc_path_draws(seed, pixel, sample, slot, n, _ptr(out))
    return out
