/*
 * rt.h — C ABI of librt_hip.so, the MI355X (gfx950) path-tracing hot path.
 *
 * The reference (dr-BEat/raytracer) has no FFI seam: the hot path is the rayon closure at
 * main.rs:154-170 called in-process over `&Camera`, `&Hittable`, `&Option<Hittable>`.  This
 * header is the seam BASELINE.json's north_star prescribes: the host flattens the
 * Hittable / Material / Texture / PDF enums (hittable.rs:100-111, material.rs:23-30,
 * texture.rs:7-14, pdf.rs:5-10) into tagged structure-of-arrays buffers, and calls these
 * entry points.  Plain pointers and sizes only; every host array is caller-owned and is
 * copied during the call that receives it; no pointer is retained afterwards.
 *
 * All entry points return 0 on success or an rt_status code; rt_last_error() gives the
 * thread-local message.  Nothing aborts or throws across the boundary (the reference's
 * panics — scenes.rs:126, main.rs:176, aabb.rs:50, cam.rs:69 — become RT_ERR_BAD_ARG).
 *
 * An rt_scene is bound to one device and is not thread-safe; distinct handles may be used
 * from distinct threads / processes (one process per GPU is how bench.py scales out).
 */
#ifndef RT_H
#define RT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 3u  /* 2: rt_scene_desc.light_box, RT_LIGHT_CUBE, RT_MAT_AREA_LIGHT; 3: complex_body may be a node
                              (an instance: wrappers over a sub-tree) */
#define RT_MAX_BVH_HEIGHT 63u       /* nodes on a root-to-leaf path (through an instance: + 1 + its sub-tree's): 4 B per level
                                       and lane of LDS, 64 KiB per workgroup */
#define RT_NOALLOC_MAX_LAUNCHES 63u /* rt_render_device_noalloc: kernel launches per call (256 B of static scratch each) */
#define RT_NOALLOC_PARTITIONS 4u    /* ... and how many scenes of one GPU have a partition of that scratch to themselves */

typedef struct rt_scene rt_scene; /* opaque: one per (scene, device) */

typedef enum rt_status {
  RT_OK = 0,
  RT_ERR_BAD_ARG = 1,      /* null pointer, size mismatch, out-of-range index, empty shutter */
  RT_ERR_HIP = 2,          /* a hip* call failed (no device, OOM, launch failure) */
  RT_ERR_UNSUPPORTED = 3,  /* scene uses nesting the flat format cannot express (a medium in a medium, an instance in
                              an instance, a medium whose boundary is a sub-tree) */
  RT_ERR_STACK_DEPTH = 4   /* BVH height exceeds the LDS traversal stack (RT_MAX_BVH_HEIGHT) */
} rt_status;

/* ---- node / leaf references ------------------------------------------------------------
 * A child reference in node_left / node_right / root is either the pre-order index of
 * another node, or a leaf:  RT_REF_LEAF | (kind << 28) | index-into-that-kind's-arrays.
 * RT_REF_NONE as root = Hittable::Empty world (hittable.rs:171, 413).                    */
#define RT_REF_LEAF 0x80000000u
#define RT_REF_NONE 0xFFFFFFFFu
#define RT_REF_KIND(ref) (((ref) >> 28) & 7u)
#define RT_REF_INDEX(ref) ((ref) & 0x0FFFFFFFu)
#define RT_MAKE_LEAF(kind, index) (RT_REF_LEAF | ((uint32_t)(kind) << 28) | (uint32_t)(index))

enum rt_leaf_kind {
  RT_LEAF_SPHERE = 0,   /* Hittable::Sphere        hittable.rs:49-54, 257-287 */
  RT_LEAF_MSPHERE = 1,  /* Hittable::MovingSphere  hittable.rs:56-72, 288-318 */
  RT_LEAF_CUBE = 2,     /* Hittable::Cube          hittable.rs:42-47, 210-256 */
  RT_LEAF_COMPLEX = 3   /* a chain of Translate / Rotate / ConstantMedium wrappers
                           (hittable.rs:74-91, 319-386) around one of the above — or, Translate /
                           Rotate only, around a whole sub-tree (an INSTANCE: complex_body below) */
};

enum rt_op_tag {        /* one wrapper of a complex object's chain, outermost first */
  RT_OP_TRANSLATE = 0,  /* data = offset xyz            hittable.rs:364-373 */
  RT_OP_ROTATE = 1,     /* data = quaternion (w,x,y,z)  hittable.rs:374-386, alias.rs:130-133 */
  RT_OP_MEDIUM = 2      /* data[0] = neg_inv_density    hittable.rs:319-363; material =
                           complex_material (Isotropic, hittable.rs:148) */
};

enum rt_material_tag {  /* material.rs:23-30 */
  RT_MAT_LAMBERTIAN = 0,
  RT_MAT_DIELECTRIC = 1,   /* params[0] = ir */
  RT_MAT_METAL = 2,        /* params[0..2] = albedo, params[3] = fuzz */
  RT_MAT_DIFFUSE_LIGHT = 3,
  RT_MAT_ISOTROPIC = 4,
  RT_MAT_AREA_LIGHT = 5    /* EXTENSION, no counterpart in the reference: emits its texture on EVERY hit.
                              DiffuseLight emits only on front faces (material.rs:120) and a Cube's recorded normal
                              points inward (hittable.rs:234-245), so a DiffuseLight cube is black from outside */
};

enum rt_texture_tag {   /* texture.rs:7-14 */
  RT_TEX_SOLID = 0,     /* color */
  RT_TEX_CHECKER = 1,   /* odd / even = texture indices (may nest) */
  RT_TEX_IMAGE = 2,     /* texel_offset (bytes into texels), width, height; RGB8 row-major */
  RT_TEX_NORMAL = 3,
  RT_TEX_UV = 4,
  RT_TEX_NOISE = 5      /* EXTENSION, no counterpart in the reference: Perlin marble of "Ray Tracing:
                           The Next Week", value = 0.5 (1 + sin(scale z + 10 turb(p, 7))) on all channels;
                           color[0] = scale; texel_offset (multiple of 8) = byte offset into texels of its table:
                           256 unit gradient vectors as f64 xyz (6144 B), then perm_x, perm_y, perm_z (256 B each) */
};
#define RT_NOISE_TABLE_BYTES (256u * 3u * 8u + 3u * 256u)

enum rt_light_tag {     /* hittable.rs:477-515: only Sphere is importance-sampled */
  RT_LIGHT_SPHERE = 0,  /* light_sphere = centre xyz, radius */
  RT_LIGHT_OTHER = 1,   /* pdf_value = 0, random = (1,0,0)   (hittable.rs:498, 513) */
  RT_LIGHT_CUBE = 2     /* EXTENSION, no counterpart in the reference: an axis-aligned box sampled uniformly over its
                           surface.  light_box = p0 xyz, p1 xyz (p0 < p1).  random(): three draws — face by area (x
                           faces, y faces, z faces, low side first), then the two in-face coordinates in axis order;
                           pdf_value(): sum over the points where the probe ray enters (origin outside) and leaves
                           the box of distance^2 / (|cos| * surface area).  A reference `lights` entry that is a
                           Cube is RT_LIGHT_OTHER; a host opts in to this tag explicitly */
};

enum rt_lights_kind {   /* the fold at main.rs:126-131 */
  RT_LIGHTS_NONE = 0,   /* None: pure cosine PDF, no mixture draw */
  RT_LIGHTS_SINGLE = 1, /* Some(x): no List::choose draw */
  RT_LIGHTS_LIST = 2    /* Some(List): uniform choose, mean pdf */
};

/* Flat scene, host side.  Geometry is f64 like the reference (alias.rs:4-7); the library
 * converts to the device precision (f32 for rt_render, f64 for the *_f64 audit entries). */
typedef struct rt_scene_desc {
  uint32_t abi_version;           /* RT_ABI_VERSION */
  uint32_t bvh_height;            /* max number of nodes on a root-to-leaf path (0 if root is a leaf) */

  /* BVH nodes in PRE-ORDER (node, left subtree, right subtree) — hittable.rs:94-98 */
  uint32_t n_nodes;
  uint32_t root;
  const double* node_box;         /* [n_nodes][6] = minimum xyz, maximum xyz (aabb.rs:7-10) */
  const uint32_t* node_left;      /* [n_nodes] */
  const uint32_t* node_right;     /* [n_nodes] */

  uint32_t n_spheres;
  const double* sphere;           /* [n][4] = centre xyz, radius (may be negative, Q15) */
  const uint32_t* sphere_material;
  const uint32_t* sphere_object;  /* world-level object id reported by rt_primary_hits */

  uint32_t n_mspheres;
  const double* msphere;          /* [n][9] = c0 xyz, c1 xyz, radius, time_start, time_end */
  const uint32_t* msphere_material;
  const uint32_t* msphere_object;

  uint32_t n_cubes;
  const double* cube;             /* [n][6] = p0 xyz, p1 xyz */
  const uint32_t* cube_material;
  const uint32_t* cube_object;

  uint32_t n_complex;
  const uint32_t* complex_body;       /* leaf ref of kind SPHERE / MSPHERE / CUBE; or (no RT_REF_LEAF) the index of a
                                         NODE: the root of a sub-tree of its own that the wrappers hold, as
                                         hittable.rs:81-91 hold a Box<Hittable> that is a Bvh (an instance).  Its nodes
                                         live in the node arrays behind the node that references the complex, its
                                         boxes are in the wrappers' inner frame, its leaves are bare primitives; no
                                         RT_OP_MEDIUM in such a chain, no instance inside an instance
                                         (RT_ERR_UNSUPPORTED) */
  const uint32_t* complex_op_first;   /* first wrapper in ops[] */
  const uint32_t* complex_op_count;   /* number of wrappers, outermost first */
  const uint32_t* complex_material;   /* Isotropic material of the medium op, else RT_REF_NONE */
  const uint32_t* complex_object;

  uint32_t n_ops;
  const uint8_t* op_tag;          /* rt_op_tag */
  const double* op_data;          /* [n_ops][4] */

  uint32_t n_materials;
  const uint8_t* mat_tag;         /* rt_material_tag */
  const uint32_t* mat_texture;    /* texture index, or RT_REF_NONE */
  const double* mat_params;       /* [n][4] */

  uint32_t n_textures;
  const uint8_t* tex_tag;         /* rt_texture_tag */
  const double* tex_color;        /* [n][3] */
  const uint32_t* tex_odd;
  const uint32_t* tex_even;
  const uint64_t* tex_texel_offset;
  const uint32_t* tex_width;
  const uint32_t* tex_height;

  uint64_t n_texel_bytes;
  const uint8_t* texels;          /* RGB8 row-major (image::RgbImage, texture.rs:11) */

  uint32_t lights_kind;           /* rt_lights_kind */
  uint32_t n_lights;
  const uint8_t* light_tag;       /* rt_light_tag */
  const double* light_sphere;     /* [n_lights][4] */
  const double* light_box;        /* [n_lights][6], read for RT_LIGHT_CUBE entries only; may be null if there are none */

  double background[3];           /* main.rs:139 */
} rt_scene_desc;

/* = cam.rs:6-18 (w is dead in the reference, cam.rs:13-14) */
typedef struct rt_camera {
  double origin[3], lower_left[3], horizontal[3], vertical[3], u[3], v[3];
  double lens_radius, time_open, time_close;
} rt_camera;

#define RT_FLAG_STATS 1u      /* count nodes / primitive tests / vertices (slower launch) */
#define RT_FLAG_FORCE_BVH 2u  /* f32 only: always walk the BVH, even for scenes small enough for the
                                 leaf-list fast path (same result; used to count the reference
                                 algorithm's node visits for the roofline's algorithmic work) */

#define RT_FLAG_WAVEFRONT 4u  /* f32, BVH-bound scenes only (ignored elsewhere): render with the wavefront form — a pool of
                                 paths in HBM, a TRACE and a SHADE kernel per bounce (device/wavefront.hpp) — instead of the
                                 megakernel.  Same paths, same draws, same image up to f32 summation order.  Measured SLOWER
                                 (DESIGN.md 3.3): kept as the evidence for the megakernel-versus-queue question.  With this
                                 flag (or RT_WAVEFRONT=1 in the environment) rt_render_device is NOT asynchronous and NOT
                                 allocation-free: the call allocates the pool on first use (72 B per slot, 2^21 slots; plus
                                 a pinned word) and synchronises the stream every few bounces.  Ignored (megakernel) when
                                 an eighth-by-eighth-pixel tile count x spp reaches 2^32 (32-bit job cursors).          */

typedef struct rt_params {
  uint32_t width, height;   /* image size; h = trunc(w / aspect) is the caller's job (main.rs:112) */
  uint32_t spp;             /* samples per pixel rendered by THIS call */
  uint32_t max_depth;       /* main.rs:47-48 */
  uint64_t seed;            /* Philox key */
  uint64_t sample_offset;   /* global index of this call's first sample (multi-GPU sharding) */
  uint32_t spp_per_launch;  /* samples per kernel launch; 0 = library default (4096) */
  uint32_t flags;           /* RT_FLAG_* */
} rt_params;

typedef struct rt_stats {
  uint64_t rays;            /* ray_color invocations with depth > 0 (main.rs:77) */
  uint64_t light_probes;    /* Sphere::hit inside pdf_value (hittable.rs:480-482) */
  uint64_t medium_probes;   /* boundary hits inside ConstantMedium (hittable.rs:324-325) */
  uint64_t nodes_visited;   /* RT_FLAG_STATS only, as are the next ten */
  uint64_t sphere_tests, sphere_hits, cube_tests, cube_hits;
  uint64_t rotate_visits, translate_visits, medium_visits;
  uint64_t closest_hits, diffuse_vertices, specular_vertices;
  uint64_t emitter_hits, misses;
  uint64_t nonfinite_samples; /* samples whose radiance was NaN/inf (f32: dropped) */
  uint32_t launches;
  uint32_t reserved;
  double kernel_ms;         /* HIP-event time over this call's render launches */
} rt_stats;

/* Replaces: the immutable borrows `&world`, `&lights` captured by the closure at
 * main.rs:158-169.  Validates every index in the description, copies it to HBM.        */
int rt_scene_create(int device, const rt_scene_desc* desc, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);

/* Replaces: one `world.hit(r, 0.001, inf)` per pixel (main.rs:77) for the deterministic
 * primary ray through the pixel centre (no jitter, no lens offset, ray.time = time), with
 * draw stream (pixel, sample 0, slot 1).  Outputs are host arrays [h][w]...; prim_id is the
 * world-level object id or RT_REF_NONE; uv / front_face may be null.
 * rt_primary_hits (f32) answers with the f32 product walk — the leaf arithmetic rt_render runs — on the pixel-centre
 * ray formed in f64 and rounded once, and AUDITS every decision of that walk with a forward error bound built from
 * the inputs' own f64 -> f32 rounding deltas (device/audit.hpp).  A pixel where some decision (hit or miss, which
 * root, which candidate is nearer, a cube's face pick) sits inside its bound, or whose t may be off by more than
 * 5e-6 relative, is re-decided by the f64 audit kernel; rt_primary_redecided() tells how many pixels of the last
 * call that was (typically 1-3 %: silhouettes, grazing hits, exact ties such as the Cornell box's image diagonal).
 * Result: ids and front_face equal to the f64 reference's, t within 1e-5 relative, on every pixel.           */
int rt_primary_hits(rt_scene* scene, const rt_camera* cam, uint32_t width, uint32_t height,
                    uint64_t seed, double time, uint32_t* prim_id, float* t, float* normal /*[.][3]*/,
                    float* uv /*[.][2]*/, uint8_t* front_face);
int rt_primary_hits_f64(rt_scene* scene, const rt_camera* cam, uint32_t width, uint32_t height,
                        uint64_t seed, double time, uint32_t* prim_id, double* t, double* normal,
                        double* uv, uint8_t* front_face);

/* Pixels of the last rt_primary_hits / rt_primary_hits_fast call on this scene that the audit handed to the f64 kernel. */
uint64_t rt_primary_redecided(const rt_scene* scene);

/* The same query answered by the walk rt_render uses for this scene (the leaf list, or the ordered walk of
 * the library's own SAH tree when every box is proven conservative; else the reference-order walk):
 * lets a caller check, pixel for pixel, that the production walk picks the winner hittable.rs:400-412
 * picks.  No counterpart in the reference beyond rt_primary_hits'.                                   */
int rt_primary_hits_fast(rt_scene* scene, const rt_camera* cam, uint32_t width, uint32_t height,
                         uint64_t seed, double time, uint32_t* prim_id, float* t, float* normal,
                         float* uv, uint8_t* front_face);

/* Replaces: the render closure main.rs:154-167 minus the divide and quantise.  accum_rgb is
 * a HOST array [h][w][3] that receives the SUM (not the mean) of this call's spp samples.  */
int rt_render(rt_scene* scene, const rt_camera* cam, const rt_params* params,
              float* accum_rgb, rt_stats* stats);
/* Audit build of the same kernel in f64 with contraction off: draw-for-draw and
 * branch-for-branch comparable with the CPU oracle.  Not a production path.
 * Every SAMPLE is the oracle's to the last bit (up to libm's sin / cos); a pixel's samples are summed in
 * units that reach the image by f64 atomic adds, so the last bits of a pixel SUM depend on the order in
 * which its units finish (1e-16 relative; the same holds for the f32 entries at 1e-7).             */
int rt_render_f64(rt_scene* scene, const rt_camera* cam, const rt_params* params,
                  double* accum_rgb, rt_stats* stats);
/* Same as rt_render but accum_rgb is DEVICE memory on the scene's device; the launches are
 * enqueued on `stream` (a hipStream_t, may be null).  If accumulate != 0 the samples are
 * added to the buffer's contents, else the buffer is overwritten.  If stats is non-null the
 * call synchronises the stream before returning.  This is what the multi-GPU path uses:
 * each rank renders its sample range, then one RCCL sum all-reduce merges the buffers.
 * STREAMS AND THREADS: a call takes a counter block of its own (work counters, the persistent kernels' unit
 * counter: the next of 16 chunks the scene holds) and, with stats, its own pair of timing events, so calls on
 * ONE scene may be enqueued on different streams, also from different threads, and overlap on the GPU.  A
 * chunk comes round again after 16 further calls on that scene: do not keep more than 16 calls of one scene
 * in flight.  (RT_FLAG_WAVEFRONT calls share the scene's path pool: one at a time.)
 * NON-FINITE SAMPLES (f32 entries: rt_render, rt_render_device, rt_render_multi): a sample whose radiance
 * is NaN or infinite is dropped and counted in rt_stats.nonfinite_samples, the divisor stays spp; the
 * reference adds it to the pixel (main.rs:165), which then quantises to 0 (main.rs:59-61, Q9/Q10 of
 * SURVEY.md 7.3).  rt_render_f64 keeps the reference's behaviour.  Parity of the f32 behaviour on such
 * pixels is unpinned: no reference output holds one.                                              */
int rt_render_device(rt_scene* scene, const rt_camera* cam, const rt_params* params,
                     float* accum_rgb_device, int accumulate, void* stream, rt_stats* stats);
/* The f64 audit kernel into DEVICE memory (f64 [h][w][3] SUM): what bench.py times as the
 * reference-precision leg (the reference computes in f64 throughout, alias.rs:4-7).        */
int rt_render_f64_device(rt_scene* scene, const rt_camera* cam, const rt_params* params,
                         double* accum_rgb_device, int accumulate, void* stream, rt_stats* stats);

/* No counterpart in the reference.  For callers that cannot allocate: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  Same contract and same kernels as
 * rt_render_device, but the kernels' scratch (per launch: the work counters and the persistent
 * kernels' unit counter) lives in a file-scope __device__ array of RT_NOALLOC_MAX_LAUNCHES x 256 B =
 * 15.75 KiB per partition, RT_NOALLOC_PARTITIONS partitions = 63 KiB that every GPU the library runs on holds
 * until the process ends; the call itself makes no
 * hipMalloc / hipFree or other allocating call (memsets, launches, two event records; with stats != NULL
 * also one stream synchronise and a <= 16 KiB copy-back, so pass NULL under capture).  A call that needs
 * more than RT_NOALLOC_MAX_LAUNCHES launches (spp / spp_per_launch, default 4096 per launch) is refused
 * with RT_ERR_BAD_ARG.  A scene uses the partition that its creation order on its GPU selects (ordinal modulo
 * RT_NOALLOC_PARTITIONS): calls through this entry on ONE scene, or on two scenes of one GPU that share a
 * partition (every fifth scene created there), must not overlap — they would race on the unit counters;
 * calls on scenes with different partitions may.  The call zeroes what it uses.                     */
int rt_render_device_noalloc(rt_scene* scene, const rt_camera* cam, const rt_params* params,
                             float* accum_rgb_device, int accumulate, void* stream, rt_stats* stats);

/* Replaces: the same closure (main.rs:154-167) with rayon's worker pool (main.rs:157) lifted from
 * cores to the GPUs of one node, in ONE process.  An rt_multi keeps, per entry of `devices` (a device
 * may be listed more than once), a device copy of the scene, an accumulator and a stream, so that a
 * sequence of renders (progressive passes, camera moves) pays for the G uploads once — the way the
 * reference's workers share one `world` for the whole run (main.rs:138, 157).
 * rt_multi_render: entry g renders samples rt_multi_shard(spp, g, G) of every pixel on its own stream
 * (the union of the shards' Philox streams does not depend on G); each shard but the first is then
 * peer-copied, on that same stream, into its own slot of a staging area on devices[0] — the G-1 copies
 * run concurrently, one per xGMI link — and one kernel on devices[0] adds the slots to the first shard in
 * shard order once all have landed.  The sum is written to the HOST buffer accum_rgb ([h][w][3] f32 SUM);
 * the call returns when it is there.  per_dev: null, or G entries (per-shard counters and kernel time;
 * each shard's thread then waits for its own counters before it starts its copy).
 * One rt_multi serves one rt_multi_render at a time.
 * One process per GPU + RCCL all-reduce (raytracer_amd/distributed.py) is the other way to do it.   */
typedef struct rt_multi rt_multi;
int rt_multi_create(int n_dev, const int* devices, const rt_scene_desc* desc, rt_multi** out);
int rt_multi_render(rt_multi* multi, const rt_camera* cam, const rt_params* params, float* accum_rgb,
                    rt_stats* per_dev);
/* The same in the reference's own arithmetic (f64, alias.rs:4-7): every shard runs the kernels of rt_render_f64_device,
 * the shards are summed in f64 in shard order; accum_rgb is [h][w][3] f64 SUM on the host.  The union of the shards'
 * paths is the single-device render's, path for path; only the order of a pixel's sum differs (1e-16 relative). */
int rt_multi_render_f64(rt_multi* multi, const rt_camera* cam, const rt_params* params, double* accum_rgb,
                        rt_stats* per_dev);
void rt_multi_destroy(rt_multi* multi);
/* The sample range of shard g of n_dev: first = g*spp/n_dev, count = (g+1)*spp/n_dev - first (integer
 * division): contiguous, disjoint, covering [0, spp) for any n_dev, sizes within one of each other.
 * Host arithmetic only (no GPU call): the distributed path's ranks use the same rule.              */
void rt_multi_shard(uint64_t spp, uint32_t g, uint32_t n_dev, uint64_t* first, uint64_t* count);
/* rt_multi_create + one rt_multi_render + rt_multi_destroy.                                         */
int rt_render_multi(int n_dev, const int* devices, const rt_scene_desc* desc, const rt_camera* cam,
                    const rt_params* params, float* accum_rgb, rt_stats* per_dev);

/* No counterpart in the reference (its only tree is new_bvh, hittable.rs:159-206, which the host
 * layer builds): the binned-SAH tree the f32 fast path walks near-child-first on large scenes
 * (DESIGN.md section 3.2), exposed so that it can be checked without a GPU.  Host only.
 * leaf_box: [n_leaves][6] f32 (lo xyz, hi xyz); leaf_refs: [n_leaves] opaque leaf references with
 * RT_REF_LEAF set; nodes_out: [n_leaves - 1] nodes of 16 dwords each: left box lo/hi (6 f32), right
 * box lo/hi (6 f32), left ref, right ref, 2 reserved; node 0 is the root, children follow parents
 * (pre-order).  max_depth caps the height (raised to ceil(log2 n) if smaller).              */
int rt_debug_build_tree(uint32_t n_leaves, const float* leaf_box, const uint32_t* leaf_refs, uint32_t max_depth,
                        uint32_t* nodes_out, uint32_t* height_out);

/* Replaces: `pixel_color /= spp; pixel_from_color(pixel_color)` (main.rs:167-168, 51-63):
 * gamma 2, clamp [0, 0.9999], ×256, truncate; NaN → 0 like Rust's saturating `as u8`.    */
void rt_tonemap(const float* accum_rgb, uint64_t n_pixels, uint64_t spp, uint8_t* rgb8);
int rt_tonemap_device(int device, const float* accum_rgb_device, uint64_t n_pixels, uint64_t spp,
                      uint8_t* rgb8_device, void* stream);

/* The same in the reference's own arithmetic, for the f64 accumulators of rt_render_f64 / rt_render_f64_device:
 * `pixel_color /= spp` and pixel_from_color in f64 (main.rs:167-168, 51-63), quantised exactly as the reference
 * quantises its f64 pixel.                                                                   */
void rt_tonemap_f64(const double* accum_rgb, uint64_t n_pixels, uint64_t spp, uint8_t* rgb8);
int rt_tonemap_f64_device(int device, const double* accum_rgb_device, uint64_t n_pixels, uint64_t spp,
                          uint8_t* rgb8_device, void* stream);

/* Device-side evaluation of the two functions the reference's own tests pin
 * (hittable.rs:547-556 get_sphere_uv; alias.rs:188-215 quaternion rotate), so the KATs can
 * be run against the kernels' code, not only against the oracle.  n items each.          */
int rt_kat_sphere_uv(int device, uint32_t n, const double* p /*[n][3]*/, double* uv /*[n][2]*/, int use_f32);
int rt_kat_rotate(int device, uint32_t n, const double* angle_rad, const double* axis /*[n][3]*/,
                  const double* v /*[n][3]*/, double* out /*[n][3]*/, int use_f32);
/* The f64 kernels' transcendental functions as the device computes them: op 0 sin(a), 1 cos(a), 2 log(a),
 * 3 atan2(a, b), 4 acos(a); n items (b is read for every op).  No counterpart in the reference: the f64 kernel follows
 * the CPU oracle path for path EXCEPT where these differ from the host's libm in the last bit (alias.rs:78-90,
 * hittable.rs:342, 530-540 call them); tests/test_gpu_kats.py measures how often that is.                        */
int rt_kat_libm(int device, uint32_t op, uint32_t n, const double* a, const double* b, double* out);
/* rounds: 7 = the generator of the per-path draw streams (Philox4x32-7), 10 = the generator of the
 * scene and BVH-axis streams (Philox4x32-10).  Both are pinned by Random123's published vectors.     */
int rt_kat_philox(int device, uint32_t n, const uint32_t* ctr /*[n][4]*/, const uint32_t* key /*[n][2]*/,
                  uint32_t rounds, uint32_t* out /*[n][4]*/);

int rt_device_count(void);
uint32_t rt_abi_version(void);
const char* rt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_H */
