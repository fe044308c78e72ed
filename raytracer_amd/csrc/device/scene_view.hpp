// Device-resident scene: the reference's Hittable / Material / Texture enums
// (hittable.rs:100-111, material.rs:23-30, texture.rs:7-14) flattened into tagged arrays in
// HBM.  Hot geometry (what a traversal touches) is split from cold attributes (what only
// the winning hit touches): a sphere test reads one 16-byte record, a node test 32 bytes.
#pragma once
#include "vec.hpp"

namespace rt {

constexpr uint32_t REF_LEAF = 0x80000000u;
constexpr uint32_t REF_NONE = 0xFFFFFFFFu;
constexpr uint32_t LEAF_SPHERE = 0, LEAF_MSPHERE = 1, LEAF_CUBE = 2, LEAF_COMPLEX = 3;
constexpr uint32_t OP_TRANSLATE = 0, OP_ROTATE = 1, OP_MEDIUM = 2;
constexpr uint32_t MAT_LAMBERTIAN = 0, MAT_DIELECTRIC = 1, MAT_METAL = 2, MAT_DIFFUSE_LIGHT = 3,
                   MAT_ISOTROPIC = 4, MAT_AREA_LIGHT = 5 /* extension */;
constexpr uint32_t TEX_SOLID = 0, TEX_CHECKER = 1, TEX_IMAGE = 2, TEX_NORMAL = 3, TEX_UV = 4, TEX_NOISE = 5;
constexpr uint32_t LIGHT_SPHERE = 0, LIGHT_OTHER = 1, LIGHT_CUBE = 2 /* extension */;
constexpr uint32_t LIGHTS_NONE = 0, LIGHTS_SINGLE = 1, LIGHTS_LIST = 2;

__host__ __device__ inline uint32_t ref_kind(uint32_t r) { return (r >> 28) & 7u; }
__host__ __device__ inline uint32_t ref_index(uint32_t r) { return r & 0x0FFFFFFFu; }

template <typename R> struct alignas(4 * sizeof(R)) Real4 {
  R x, y, z, w;
};

// BVH node, pre-order.  float: 32 B (two dwordx4 loads); double: 64 B.
template <typename R> struct Node;
template <> struct alignas(32) Node<float> {
  float lo[3];
  float hi[3];
  uint32_t left, right;
};
template <> struct alignas(64) Node<double> {
  double lo[3];
  double hi[3];
  uint32_t left, right;
  uint32_t pad[2];
};

template <typename R> struct alignas(4 * sizeof(R)) MSphere {  // hittable.rs:56-64
  R c0[3], radius;
  R c1[3], t0;
  R t1, pad[3];
};
template <typename R> struct alignas(4 * sizeof(R)) Cube {  // hittable.rs:42-47
  R p0[3], pad0;
  R p1[3], pad1;
};
struct Attr {  // cold per-primitive attributes
  uint32_t material, object;
  uint32_t rank;  // position of this leaf in the reference's left-first DFS order (tie-break)
  uint32_t pad;
};
// Node of the ORDERED walk (f32): the boxes of BOTH children next to their references, so one
// 64-byte fetch decides which child is nearer.
struct alignas(64) Node2 {
  float lbox[6];
  float rbox[6];
  uint32_t left, right, flags, pad;
};
// flags & CXF_INSTANCE: `body` is not a leaf but the index of a NODE — the root of a sub-tree of its own (an instance:
// Translate / Rotate over a Bvh, hittable.rs:81-91 over hittable.rs:93-98).  The reference-order walk enters it with
// the ray mapped into its frame (traverse()); for every other walk rt_scene_create lists each leaf of the sub-tree
// as a complex of its own (same wrapper chain, body = that leaf: "synthesized" complexes behind the caller's).
constexpr uint32_t CXF_INSTANCE = 1u;
struct Complex {
  uint32_t body, op_first, op_count, medium_pos;  // medium_pos = index of the MEDIUM op in the chain, or REF_NONE
  uint32_t material, object, n_rotate, n_translate;
  uint32_t rank, flags, pad[2];
};
constexpr uint32_t REF_MARKER = REF_LEAF | (7u << 28);  // on a walk's stack: "back in the world's frame" (end of an instance)
// The wrapper chain of a complex object composed on the host (in f64) into rigid maps
// local = M * world + b.  `full` takes a world ray into the primitive's frame, `outer` into the
// constant medium's frame (= full when there is no medium).  Used by the f32 kernels; the f64
// audit kernels walk the chain op by op with Hamilton products exactly like hittable.rs:364-386.
template <typename R> struct alignas(16) ComplexXf {
  R m_full[9], b_full[3];
  R m_outer[9], b_outer[3];
  R neg_inv_density, pad[3];
};
template <typename R> struct alignas(4 * sizeof(R)) Op {
  R d[4];  // offset xyz | quaternion wxyz | neg_inv_density
};
constexpr uint32_t MATF_NEEDS_UV = 1u;  // some texture below this material reads hit.uv (Image / UV)
constexpr uint32_t MATF_SOLID = 2u;     // the texture is Texture::Solid: its colour is copied into `color`
template <typename R> struct Material {
  uint32_t tag, texture, flags, pad;
  R p[4];
  R color[3], pad2;
};

// Leaf-list records (f32, small scenes): one fixed 48-dword record per leaf, copied into LDS at
// kernel start so the traversal loop, the hit record and solid-colour shading read no global
// memory and follow no dependent pointer chain.
#ifndef RT_LIN_MAX_LEAVES
#define RT_LIN_MAX_LEAVES 32  // measured (tools/leaflist_crossover.py): the leaf list beats the ordered BVH walk up to 32 leaves (+21 % at 21, +9 % at 32)
#endif
constexpr uint32_t LIN_DWORDS = 48, LIN_MAX_LEAVES = RT_LIN_MAX_LEAVES, LDS_MAX_LIGHTS = 8;
constexpr uint32_t LIN_FLAGS = 0, LIN_REF = 1, LIN_NID = 2, LIN_OBJECT = 3, LIN_M = 4, LIN_B = 13, LIN_PRIM = 16,
                   LIN_MAT = 24,  // Material<float> image: 12 dwords
                   LIN_INVEXT = 36;  // cube: 1 / (p1 - p0), 3 dwords
constexpr uint32_t LINF_XF = 1u, LINF_CUBE = 2u, LINF_MEDIUM = 4u;
// leaf list over the scene arrays (SceneView::leaf_list): what a workgroup stages in LDS at most
constexpr uint32_t STAGE_MAX_SPHERES = 2 * LIN_MAX_LEAVES, STAGE_MAX_CUBES = LIN_MAX_LEAVES, STAGE_MAX_COMPLEXES = LIN_MAX_LEAVES,
                   STAGE_MAX_OPS = 4 * LIN_MAX_LEAVES;
template <typename R> struct Texture {
  uint32_t tag, odd, even, width, height, pad;
  uint64_t texel_offset;
  R color[3];
};
template <typename R> struct Light {
  uint32_t tag, pad;
  R c[3], radius;  // LIGHT_SPHERE: centre, radius; LIGHT_CUBE: p0, total surface area
  R hi[3], pad2;   // LIGHT_CUBE: p1
};

template <typename R> struct SceneView {
  const Node<R>* nodes;
  const Node2* nodes2;  // ordered walk (f32 side only)
  const Real4<R>* spheres;
  const MSphere<R>* mspheres;
  const Cube<R>* cubes;
  const Attr* sphere_attr;
  const Attr* msphere_attr;
  const Attr* cube_attr;
  const Complex* complexes;
  const ComplexXf<R>* complex_xf;
  const uint32_t* op_tag;
  const Op<R>* ops;
  const Material<R>* materials;
  const Texture<R>* textures;
  const uint8_t* texels;
  const Light<R>* lights;
  const uint32_t* linear_rec;  // n_linear leaf records (LIN_DWORDS each) in BVH left-first DFS order
  uint32_t n_linear;           // 0 = scene not eligible for the leaf-list path
  uint32_t linear_has_media;   // some leaf record carries LINF_MEDIUM
  uint32_t has_noise;          // the scene uses an EXTENSION (a TEX_NOISE texture, a LIGHT_CUBE light, a MAT_AREA_LIGHT
                               // material): only the reference-order kernel's extension instantiation carries that code
  uint32_t linear_xf_mask, linear_cube_mask, linear_medium_mask;  // bit k = LINF_* of record k (kernel arguments:
                                                                  // scalar registers, no per-leaf flag read from LDS)
  // Leaf list read straight from the arrays above (any precision; what the f64 kernel walks on small scenes): the
  // n_leaf_list leaf references in BVH left-first DFS order, and how many records of the hot arrays a workgroup
  // stages in LDS for it (spheres incl. baked, cubes, complexes, ops).  n_leaf_list = 0: not eligible.
  const uint32_t* leaf_list;
  const uint32_t* leaf_by_rank;  // every leaf of the scene by its left-first DFS rank, the leaves of instances as their
                                 // synthesized complexes (f32 side: baked spheres where possible)
  uint32_t has_instances;  // some complex carries CXF_INSTANCE (wave-uniform: the reference-order walk's instance arm)
  uint32_t n_leaf_list;
  uint32_t stage_spheres, stage_cubes, stage_complexes, stage_ops;
  uint32_t root;
  uint32_t root2;  // root of nodes2 (REF_NONE if the ordered walk is not available)
  uint32_t lights_kind, n_lights;
  R background[3];
};

template <typename R> struct CameraView {  // cam.rs:6-18
  Vec3<R> origin, lower_left, horizontal, vertical, u, v;
  Vec3<R> ll_rel;  // lower_left - origin, formed in f64 on the host (used by the f32 kernels only)
  R lens_radius, time_open, time_close;
};

struct DeviceStats {  // one uint64 per rt_stats counter, in rt_stats order
  unsigned long long v[17];
  unsigned long long stamp[8];  // RT_STAMPS diagnostic builds only: wave-cycles per loop phase
  unsigned int unit_counter;    // render_persist_kernel: next (tile, sample range) unit; zeroed per launch
  unsigned int pad;
};
enum StatIdx {
  ST_RAYS = 0, ST_LIGHT_PROBES, ST_MEDIUM_PROBES, ST_NODES, ST_SPHERE_TESTS, ST_SPHERE_HITS,
  ST_CUBE_TESTS, ST_CUBE_HITS, ST_ROTATE, ST_TRANSLATE, ST_MEDIUM, ST_CLOSEST, ST_DIFFUSE,
  ST_SPECULAR, ST_EMITTER, ST_MISSES, ST_NONFINITE, ST_COUNT
};

}  // namespace rt
