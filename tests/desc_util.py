"""Hand-made rt_scene_desc values for the tests (shared by the CPU and the GPU suites)."""
import numpy as np


def chain_desc(n_leaves):
    """A degenerate (chain) BVH of n_leaves small spheres: node i = (leaf i, node i+1); height = n_leaves - 1 nodes
    on the longest path.  Returns (desc, keep-alive list)."""
    from raytracer_amd import _ffi as F
    d = F.RtSceneDesc()
    keep = []

    def put(field, a, ptype):
        a = np.ascontiguousarray(a)
        keep.append(a)
        setattr(d, field, a.ctypes.data_as(ptype))

    n_nodes = n_leaves - 1
    d.abi_version = F.RT_ABI_VERSION
    d.n_nodes, d.root, d.bvh_height = n_nodes, 0, n_nodes
    sph = np.zeros((n_leaves, 4))
    sph[:, 0] = np.linspace(-3.0, 3.0, n_leaves)
    sph[:, 2] = np.linspace(-1.0, 1.0, n_leaves)
    sph[:, 3] = 0.2
    lo, hi = sph[:, :3] - 0.2, sph[:, :3] + 0.2
    box = np.zeros((n_nodes, 6))
    left = np.zeros(n_nodes, dtype=np.uint32)
    right = np.zeros(n_nodes, dtype=np.uint32)
    for i in range(n_nodes):
        box[i, :3] = lo[i:].min(axis=0)
        box[i, 3:] = hi[i:].max(axis=0)
        left[i] = 0x80000000 | i
        right[i] = i + 1 if i + 1 < n_nodes else (0x80000000 | (n_leaves - 1))
    put("node_box", box, F._f64p)
    put("node_left", left, F._u32p)
    put("node_right", right, F._u32p)
    d.n_spheres = n_leaves
    put("sphere", sph, F._f64p)
    put("sphere_material", np.zeros(n_leaves, dtype=np.uint32), F._u32p)
    put("sphere_object", np.arange(n_leaves, dtype=np.uint32), F._u32p)
    d.n_materials = 1
    put("mat_tag", np.array([0], dtype=np.uint8), F._u8p)
    put("mat_texture", np.array([0], dtype=np.uint32), F._u32p)
    put("mat_params", np.zeros(4), F._f64p)
    d.n_textures = 1
    put("tex_tag", np.array([0], dtype=np.uint8), F._u8p)
    put("tex_color", np.array([0.5, 0.6, 0.7]), F._f64p)
    for f, ty, pt in (("tex_odd", np.uint32, F._u32p), ("tex_even", np.uint32, F._u32p),
                      ("tex_texel_offset", np.uint64, F._u64p), ("tex_width", np.uint32, F._u32p),
                      ("tex_height", np.uint32, F._u32p)):
        put(f, np.zeros(1, dtype=ty), pt)
    d.lights_kind = 0
    for k, v in enumerate((0.5, 0.7, 1.0)):
        d.background[k] = v
    return d, keep
