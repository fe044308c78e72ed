"""Python face of the host layer: the reference's `Opts` / `main()` surface (main.rs:28-49,
107-177) over the native libraries.  numpy in, numpy out; torch is only used by
raytracer_amd.distributed for device buffers and the RCCL all-reduce."""
import ctypes as C
import os

import numpy as np

from . import _ffi as F

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")

SCENE_IDS = {
    "random_scene": 0, "two_spheres": 1, "earth": 2, "simple_light": 3, "cube_scene": 4, "small_scene": 5,
    "cornell_box": 6, "bvh_stress": 7, "final_scene": 8, "mixed_scene": 9, "perlin_spheres": 10, "final_scene_perlin": 11,
    "cornell_box_area": 12, "area_light_plate": 13, "random_scene_stdrng": 14, "area_light_list": 15,
    "final_scene_instanced": 16, "bvh_stress_sparse": 17,
}
_NEEDS_EARTHMAP = (2, 4, 8, 11, 16)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def load_earthmap():
    """assets/earthmap.jpg of the reference, decoded once to RGB8 (texture.rs:29-30 decodes with the
    `image` crate; here the decode was done offline, see tests/golden/make_fixtures.py)."""
    with np.load(os.path.join(_ASSETS, "earthmap_rgb8.npz")) as z:
        return np.ascontiguousarray(z["rgb8"])


def write_earthmap_ppm(path):
    """The earth texture as a binary PPM (P6): what the `raytracer` command line reads with --earthmap
    (default assets/earthmap.ppm) where the reference opens assets/earthmap.jpg (scenes.rs:126)."""
    rgb8 = load_earthmap()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (rgb8.shape[1], rgb8.shape[0]))
        f.write(rgb8.tobytes())


def texture_from_file(path):
    """Texture::new_image (texture.rs:25-31) through the host library: a baseline JPEG (what the reference opens:
    assets/earthmap.jpg) or a binary PPM (P6, maxval 255), as RGB8 [h][w][3].  A missing, malformed or unsupported
    (progressive, CMYK ...) file raises RtError(RT_ERR_BAD_ARG) where the reference panics."""
    w, h, px = C.c_uint32(), C.c_uint32(), F._u8p()
    F.check_host(F.host.rth_texture_from_image_file(os.fsencode(path), C.byref(w), C.byref(h), C.byref(px)))
    try:
        return np.ctypeslib.as_array(px, shape=(h.value, w.value, 3)).copy()
    finally:
        F.host.rth_texture_free(px)


def write_png(path, rgb8):
    """image.save(path) of main.rs:176 for an RGB8 image [h][w][3] (8-bit RGB PNG; row filters + fixed-Huffman deflate)."""
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    if rgb8.ndim != 3 or rgb8.shape[2] != 3:
        raise ValueError("rgb8 must be [h][w][3]")
    F.check_host(F.host.rth_write_png(os.fsencode(path), _ptr(rgb8), rgb8.shape[1], rgb8.shape[0]))


CLI = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "raytracer")  # main.rs:28-49, 107-178


def image_height(image_width, aspect_ratio):
    """main.rs:112: `(image_width as f64 / aspect_ratio) as u32` (truncation)."""
    return int(F.host.rth_image_height(int(image_width), float(aspect_ratio)))


class HostScene:
    """main.rs:117-150 — scene function, world BVH, lights fold, camera — flattened (host memory)."""

    def __init__(self, scene, aspect_ratio, scene_seed=5, bvh_seed=7, n=0, rotate_bbox=-1, texels=None):
        scene_id = SCENE_IDS[scene] if isinstance(scene, str) else int(scene)
        if texels is None and scene_id in _NEEDS_EARTHMAP:
            texels = load_earthmap()
        o = F.RthSceneOpts()
        o.scene_id = scene_id
        o.rotate_bbox = rotate_bbox
        o.aspect_ratio = aspect_ratio
        o.scene_seed = scene_seed
        o.bvh_seed = bvh_seed
        o.n = n
        self._texels = None
        if texels is not None:
            self._texels = np.ascontiguousarray(texels, dtype=np.uint8)
            o.tex_height, o.tex_width = self._texels.shape[:2]
            o.texels = _ptr(self._texels)
        self.opts = o
        h = C.c_void_p()
        F.check_host(F.host.rth_scene_build(C.byref(o), C.byref(h)))
        self._h = h
        self.desc = F.host.rth_scene_desc(h).contents
        # a COPY: the camera is plain data, and a reference into the host scene would dangle in
        # `HostScene(...).camera` once the temporary is collected (desc, below, points into the scene's arrays
        # and is only valid while this object lives)
        self.camera = F.RtCamera.from_buffer_copy(F.host.rth_scene_camera(h).contents)
        self.n_objects = int(F.host.rth_scene_object_count(h))
        self.bvh_calls = int(F.host.rth_scene_bvh_calls(h))
        self.scene_id = scene_id
        self.aspect_ratio = aspect_ratio

    def close(self):
        if getattr(self, "_h", None):
            F.host.rth_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # numpy copies of the flat arrays (for tests and inspection)
    def _arr(self, ptr, shape, dtype):
        n = int(np.prod(shape))
        if n == 0:
            return np.zeros(shape, dtype=dtype)
        return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).reshape(shape).copy()

    def nodes(self):
        d = self.desc
        return (self._arr(d.node_box, (d.n_nodes, 6), np.float64), self._arr(d.node_left, (d.n_nodes,), np.uint32),
                self._arr(d.node_right, (d.n_nodes,), np.uint32), int(d.root))

    def leaf_object(self, ref):
        """world-level object id behind a typed leaf ref"""
        d = self.desc
        kind, idx = (ref >> 28) & 7, ref & 0x0FFFFFFF
        arr = [d.sphere_object, d.msphere_object, d.cube_object, d.complex_object][kind]
        return int(arr[idx])

    def camera_array(self):
        c = self.camera
        return np.array(list(c.origin) + list(c.lower_left) + list(c.horizontal) + list(c.vertical) + list(c.u)
                        + list(c.v) + [c.lens_radius, c.time_open, c.time_close])


class DeviceScene:
    """An rt_scene: the flat scene resident in one GPU's HBM."""

    def __init__(self, host_scene, device=0):
        self.host = host_scene
        self.device = device
        h = C.c_void_p()
        F.check_hip(F.hip.rt_scene_create(device, C.byref(host_scene.desc), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            F.hip.rt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def primary_hits(self, width, height, seed=1, time=0.0, f64=False, camera=None, fast=False):
        dt = np.float64 if f64 else np.float32
        ids = np.zeros((height, width), dtype=np.uint32)
        t = np.zeros((height, width), dtype=dt)
        nrm = np.zeros((height, width, 3), dtype=dt)
        uv = np.zeros((height, width, 2), dtype=dt)
        ff = np.zeros((height, width), dtype=np.uint8)
        fn = F.hip.rt_primary_hits_f64 if f64 else (F.hip.rt_primary_hits_fast if fast else F.hip.rt_primary_hits)
        cam = camera if camera is not None else self.host.camera
        F.check_hip(fn(self._h, C.byref(cam), width, height, seed, time, _ptr(ids), _ptr(t), _ptr(nrm), _ptr(uv),
                       _ptr(ff)))
        self.redecided = int(F.hip.rt_primary_redecided(self._h))  # f32: pixels the audit handed to the f64 kernel
        return ids, t, nrm, uv, ff

    def _params(self, width, height, spp, max_depth, seed, sample_offset, spp_per_launch, stats, force_bvh=False,
                wavefront=False):
        p = F.RtParams()
        p.width, p.height, p.spp, p.max_depth = width, height, spp, max_depth
        p.seed, p.sample_offset = seed, sample_offset
        p.spp_per_launch = spp_per_launch
        p.flags = ((F.RT_FLAG_STATS if stats else 0) | (F.RT_FLAG_FORCE_BVH if force_bvh else 0)
                   | (F.RT_FLAG_WAVEFRONT if wavefront else 0))
        return p

    def render(self, width, height, spp, max_depth, seed=1, sample_offset=0, spp_per_launch=0, stats=False,
               f64=False, camera=None, force_bvh=False, wavefront=False):
        """The render closure main.rs:154-167 for every pixel: returns (accum SUM [h][w][3], stats dict)."""
        p = self._params(width, height, spp, max_depth, seed, sample_offset, spp_per_launch, stats, force_bvh, wavefront)
        accum = np.zeros((height, width, 3), dtype=np.float64 if f64 else np.float32)
        st = F.RtStats()
        fn = F.hip.rt_render_f64 if f64 else F.hip.rt_render
        cam = camera if camera is not None else self.host.camera
        F.check_hip(fn(self._h, C.byref(cam), C.byref(p), _ptr(accum), C.byref(st)))
        return accum, st.as_dict()

    def render_device(self, accum_ptr, width, height, spp, max_depth, seed=1, sample_offset=0, spp_per_launch=0,
                      accumulate=False, stream=None, want_stats=True, stats=False, camera=None, force_bvh=False,
                      f64=False, noalloc=False, wavefront=False):
        """Same, into DEVICE memory (accum_ptr = raw device address of a float32 [h][w][3] buffer; float64 with
        f64=True: rt_render_f64_device).  noalloc=True: rt_render_device_noalloc (static scratch, no allocation)."""
        p = self._params(width, height, spp, max_depth, seed, sample_offset, spp_per_launch, stats, force_bvh, wavefront)
        st = F.RtStats()
        cam = camera if camera is not None else self.host.camera
        fn = F.hip.rt_render_f64_device if f64 else (F.hip.rt_render_device_noalloc if noalloc else F.hip.rt_render_device)
        F.check_hip(fn(self._h, C.byref(cam), C.byref(p), C.c_void_p(accum_ptr), 1 if accumulate else 0,
                       C.c_void_p(stream or 0), C.byref(st) if want_stats else None))
        return st.as_dict() if want_stats else None


def _multi_params(width, height, spp, max_depth, seed, sample_offset, spp_per_launch, stats):
    p = F.RtParams()
    p.width, p.height, p.spp, p.max_depth = width, height, spp, max_depth
    p.seed, p.sample_offset, p.spp_per_launch = seed, sample_offset, spp_per_launch
    p.flags = F.RT_FLAG_STATS if stats else 0
    return p


class MultiScene:
    """An rt_multi: the scene resident on every listed GPU of this process, kept across renders."""

    def __init__(self, host_scene, devices):
        self.host = host_scene
        self.devices = list(devices)
        devs = (C.c_int * len(self.devices))(*self.devices)
        h = C.c_void_p()
        F.check_hip(F.hip.rt_multi_create(len(self.devices), devs, C.byref(host_scene.desc), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            F.hip.rt_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, width, height, spp, max_depth, seed=1, sample_offset=0, spp_per_launch=0, stats=False,
               per_device=True, camera=None, f64=False):
        """rt_multi_render (f64=True: rt_multi_render_f64, the reference's arithmetic).
        Returns (accum SUM [h][w][3] float32 or float64, [per-device stats dict] or None)."""
        p = _multi_params(width, height, spp, max_depth, seed, sample_offset, spp_per_launch, stats)
        st = (F.RtStats * len(self.devices))() if per_device else None
        accum = np.zeros((height, width, 3), dtype=np.float64 if f64 else np.float32)
        cam = camera if camera is not None else self.host.camera
        fn = F.hip.rt_multi_render_f64 if f64 else F.hip.rt_multi_render
        F.check_hip(fn(self._h, C.byref(cam), C.byref(p), _ptr(accum), st))
        return accum, ([s.as_dict() for s in st] if per_device else None)


def multi_shard(spp, g, n_dev):
    """rt_multi_shard: (first, count) of shard g of n_dev — host arithmetic, no GPU."""
    first, count = C.c_uint64(), C.c_uint64()
    F.hip.rt_multi_shard(spp, g, n_dev, C.byref(first), C.byref(count))
    return int(first.value), int(count.value)


def render_multi(host_scene, devices, width, height, spp, max_depth, seed=1, sample_offset=0, spp_per_launch=0,
                 stats=False, camera=None):
    """rt_render_multi: one host thread per device, samples sharded, shards summed on devices[0].
    Returns (accum SUM [h][w][3] float32, [per-device stats dict])."""
    p = _multi_params(width, height, spp, max_depth, seed, sample_offset, spp_per_launch, stats)
    devs = (C.c_int * len(devices))(*devices)
    st = (F.RtStats * len(devices))()
    accum = np.zeros((height, width, 3), dtype=np.float32)
    cam = camera if camera is not None else host_scene.camera
    F.check_hip(F.hip.rt_render_multi(len(devices), devs, C.byref(host_scene.desc), C.byref(cam), C.byref(p),
                                      _ptr(accum), st))
    return accum, [s.as_dict() for s in st]


def tonemap(accum, spp):
    """main.rs:167-168 + 51-63: mean, gamma 2, clamp, quantise.  A float64 accumulator (rt_render_f64) is divided and
    quantised in f64 like the reference's own pixel (rt_tonemap_f64); anything else goes through the f32 form."""
    accum = np.asarray(accum)
    out = np.zeros(accum.shape, dtype=np.uint8)
    if accum.dtype == np.float64:
        accum = np.ascontiguousarray(accum)
        F.hip.rt_tonemap_f64(_ptr(accum), accum.size // 3, spp, _ptr(out))
        return out
    accum = np.ascontiguousarray(accum, dtype=np.float32)
    F.hip.rt_tonemap(_ptr(accum), accum.size // 3, spp, _ptr(out))
    return out


def render(scene=3, image_width=1200, aspect_ratio=1.5, samples_per_pixel=500, max_depth=5, device=0, seed=1,
           scene_seed=5, bvh_seed=7, n=0, texels=None, f64=False):
    """main() of the reference (same option names and defaults as `struct Opts`, main.rs:28-49) minus
    the PNG save: returns (rgb8 [h][w][3], accum SUM float32 [h][w][3], stats dict)."""
    scene_id = SCENE_IDS[scene] if isinstance(scene, str) else int(scene)
    if texels is None and scene_id in _NEEDS_EARTHMAP:
        texels = load_earthmap()
    o = F.RthRenderOpts()
    o.scene.scene_id = scene_id
    o.scene.rotate_bbox = -1
    o.scene.aspect_ratio = aspect_ratio
    o.scene.scene_seed = scene_seed
    o.scene.bvh_seed = bvh_seed
    o.scene.n = n
    keep = None
    if texels is not None:
        keep = np.ascontiguousarray(texels, dtype=np.uint8)
        o.scene.tex_height, o.scene.tex_width = keep.shape[:2]
        o.scene.texels = _ptr(keep)
    o.image_width = image_width
    o.samples_per_pixel = samples_per_pixel
    o.max_depth = max_depth
    o.device = device
    o.seed = seed
    o.precision = 1 if f64 else 0  # the f64 audit kernels + f64 tone map: the reference's own arithmetic
    h = image_height(image_width, aspect_ratio)
    rgb8 = np.zeros((h, image_width, 3), dtype=np.uint8)
    accum = np.zeros((h, image_width, 3), dtype=np.float32)
    st = F.RtStats()
    hh = C.c_uint32()
    F.check_host(F.host.rth_render(C.byref(o), _ptr(rgb8), _ptr(accum), C.byref(hh), C.byref(st)))
    return rgb8, accum, st.as_dict()
