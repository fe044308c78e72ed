// C ABI of librt_hip.so (include/rt.h): scene validation and upload, launch orchestration,
// error reporting.  Host code only; the kernels live in device/*.hpp and are instantiated
// by rt_f32.hip / rt_f64.hip.
#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt.h"
#include "accel_build.hpp"
#include "launch.hpp"

using namespace rt;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      (void)hipGetLastError(); /* reported here: do not leave it for a later launch to find */ \
      return fail(RT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));                  \
    }                                                                                          \
  } while (0)

// f64 -> device precision.  Box bounds round OUTWARD in f32 so that a ray that passes the
// reference's f64 slab test also passes the f32 one (conservative traversal).
template <typename R> R conv(double v) { return (R)v; }
float conv_down(double v) {
  float f = (float)v;
  return (double)f > v ? std::nextafterf(f, -INFINITY) : f;
}
float conv_up(double v) {
  float f = (float)v;
  return (double)f < v ? std::nextafterf(f, INFINITY) : f;
}
template <typename R> R conv_lo(double v);
template <typename R> R conv_hi(double v);
template <> double conv_lo<double>(double v) { return v; }
template <> double conv_hi<double>(double v) { return v; }
template <> float conv_lo<float>(double v) { return conv_down(v); }
template <> float conv_hi<float>(double v) { return conv_up(v); }

void host_qrotate(const double v[3], const double q[4], double out[3]) {
  auto ham = [](const double a[4], const double b[4], double r[4]) {
    r[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
    r[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
    r[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
    r[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  };
  const double p[4] = {0.0, v[0], v[1], v[2]}, qi[4] = {q[0], -q[1], -q[2], -q[3]};
  double t[4], r[4];
  ham(q, p, t);
  ham(t, qi, r);
  out[0] = r[1];
  out[1] = r[2];
  out[2] = r[3];
}

struct DeviceBuffers {
  std::vector<void*> ptrs;
  ~DeviceBuffers() {
    for (void* p : ptrs) (void)hipFree(p);
  }
  template <typename T> hipError_t upload(const std::vector<T>& host, const T** out) {
    *out = nullptr;
    size_t bytes = std::max<size_t>(host.size(), 1) * sizeof(T);  // never a null array
    void* d = nullptr;
    hipError_t e = hipMalloc(&d, bytes);
    if (e != hipSuccess) return e;
    ptrs.push_back(d);
    if (!host.empty()) {
      e = hipMemcpy(d, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice);
      if (e != hipSuccess) return e;
    } else {
      e = hipMemset(d, 0, bytes);
      if (e != hipSuccess) return e;
    }
    *out = (const T*)d;
    return hipSuccess;
  }
};

// The reference-order kernels keep a 4-byte stack entry per level and lane in dynamic LDS (launch_impl.hpp:
// stack_depth * 256 * 4 B) beside <= 320 B of static tables (s_lights); the default per-workgroup limit is 64 KiB,
// so 63 levels is what fits (63 * 1024 + 320 = 64832 B).  The ordered walk's own tree (6 B per level and lane) is
// capped at ceil(log2 n) + 6 <= 34 levels by its builder.
constexpr uint32_t MAX_STACK_DEPTH = 63;

// What rt_scene_create learns from one left-first DFS of the tree.
struct TreeFacts {
  std::vector<uint32_t> rank_sphere, rank_msphere, rank_cube, rank_complex;  // DFS rank per leaf
  std::vector<float> leaf_box;      // [n_leaves by rank][6], rounded outward (ordering / culling in the ordered walk)
  std::vector<uint32_t> leaf_refs;  // by rank
  bool conservative = true;         // every node box bounds the node boxes and leaves below it
  double t_lo = -INFINITY, t_hi = INFINITY;  // ... for ray times in [t_lo, t_hi] (moving spheres)
  bool linear_ok = true;            // additionally: few leaves, no moving sphere, media outermost
};

template <typename R> struct SceneSide {
  DeviceBuffers bufs;
  SceneView<R> view{};
  uint32_t ordered_root = RT_REF_NONE, ordered_height = 0;  // the ordered walk's tree, if built
};

template <typename R>
hipError_t build_side(const rt_scene_desc& d, const std::vector<Complex>& complexes, const std::vector<uint32_t>& linear,
                      const TreeFacts& facts, SceneSide<R>& s, std::vector<Node2>& tree2, uint32_t& tree2_root,
                      uint32_t& tree2_height) {
  // BAKED spheres.  A Translate / Rotate chain (no medium) around a static Sphere whose material never reads uv is,
  // for every quantity the path uses, a Sphere at the mapped centre: t is frame-independent, the world normal
  // (p - c') / r is the rotated local normal, front_face follows.  Such complex leaves get a plain sphere record
  // (centre mapped in f64 on the host) appended to the sphere arrays of BOTH precisions; the f32 trees reference that
  // record instead of the complex (one 16-byte fetch and no rigid map per test: final_scene wraps each of its 1000
  // small spheres this way, where the reference has one Translate(Rotate(Bvh)) instance), the f64 audit side keeps
  // walking the literal wrapper chain (hittable.rs:364-386) and only holds the record for rt_primary_hits' audit.
  const uint32_t n_complex = (uint32_t)complexes.size();  // the caller's, then the synthesized ones (instances' leaves)
  std::vector<uint32_t> baked(n_complex, RT_REF_NONE);  // complex index -> index of its baked sphere record
  std::vector<std::array<double, 4>> baked_sphere;
  std::vector<Attr> baked_attr;
  {
    std::vector<uint8_t> tex_uv0(d.n_textures, 0);
    for (uint32_t i = d.n_textures; i-- > 0;) {
      if (d.tex_tag[i] == RT_TEX_IMAGE || d.tex_tag[i] == RT_TEX_UV) tex_uv0[i] = 1;
      if (d.tex_tag[i] == RT_TEX_CHECKER) tex_uv0[i] = tex_uv0[d.tex_odd[i]] | tex_uv0[d.tex_even[i]];
    }
    static const bool no_bake = getenv("RT_NO_BAKE") != nullptr;  // A/B switch
    for (uint32_t i = 0; i < n_complex && !no_bake; ++i) {
      const Complex& cx = complexes[i];
      if ((cx.flags & CXF_INSTANCE) || facts.rank_complex[i] == RT_REF_NONE) continue;  // (a sub-tree; not in the tree)
      if (cx.medium_pos != REF_NONE || RT_REF_KIND(cx.body) != RT_LEAF_SPHERE) continue;
      const uint32_t bi = RT_REF_INDEX(cx.body), m = d.sphere_material[bi];
      if (d.mat_texture[m] < d.n_textures && tex_uv0[d.mat_texture[m]]) continue;
      // c_world: undo the wrappers innermost first (Rotate by q*, Translate by +offset), as leaf_world_bounds does
      double p[3] = {d.sphere[4 * (size_t)bi], d.sphere[4 * (size_t)bi + 1], d.sphere[4 * (size_t)bi + 2]};
      for (uint32_t k = cx.op_count; k-- > 0;) {
        const uint32_t o = cx.op_first + k;
        const double* q = d.op_data + 4 * (size_t)o;
        if (d.op_tag[o] == RT_OP_TRANSLATE) {
          for (int c = 0; c < 3; ++c) p[c] += q[c];
        } else {
          const double qi[4] = {q[0], -q[1], -q[2], -q[3]};
          double r[3];
          host_qrotate(p, qi, r);
          for (int c = 0; c < 3; ++c) p[c] = r[c];
        }
      }
      baked[i] = d.n_spheres + (uint32_t)baked_sphere.size();
      baked_sphere.push_back({p[0], p[1], p[2], d.sphere[4 * (size_t)bi + 3]});
      // pad: the wrapper visits one test of this leaf stands for (STATS builds keep the reference's work counters)
      baked_attr.push_back(Attr{m, cx.object, facts.rank_complex[i], cx.n_rotate | (cx.n_translate << 16)});
    }
  }
  auto f32_ref = [&](uint32_t ref) -> uint32_t {  // the leaf reference the f32 trees use
    if (sizeof(R) == 4 && ref != RT_REF_NONE && (ref & RT_REF_LEAF) && RT_REF_KIND(ref) == RT_LEAF_COMPLEX &&
        baked[RT_REF_INDEX(ref)] != RT_REF_NONE)
      return RT_MAKE_LEAF(RT_LEAF_SPHERE, baked[RT_REF_INDEX(ref)]);
    return ref;
  };
  std::vector<Node<R>> nodes(d.n_nodes);
  for (uint32_t i = 0; i < d.n_nodes; ++i) {
    std::memset(&nodes[i], 0, sizeof(Node<R>));
    for (int k = 0; k < 3; ++k) {
      nodes[i].lo[k] = conv_lo<R>(d.node_box[6 * (size_t)i + k]);
      nodes[i].hi[k] = conv_hi<R>(d.node_box[6 * (size_t)i + 3 + k]);
    }
    nodes[i].left = f32_ref(d.node_left[i]);
    nodes[i].right = f32_ref(d.node_right[i]);
  }
  // ordered-walk tree (f32): its own binned-SAH tree over the same leaves (accel_build.hpp); the
  // reference's tree above stays what the reference-order walk and parity level 1 use.  Each Node2
  // holds both children's boxes beside their references.  RT_ORDERED_REFERENCE_TREE=1 (A/B) mirrors
  // the reference's tree into Node2 form instead.
  // The f64 side walks the SAME tree (traverse_ordered: the f32 boxes are outward-rounded, so they cull conservatively in
  // f64 as well), with the baked sphere records turned back into the complexes they stand for: its leaf tests are the
  // literal ones.  The f32 side is built first and leaves the tree in `tree2`.
  std::vector<Node2> nodes2;
  if constexpr (sizeof(R) == 8) {
    if (tree2_root != RT_REF_NONE) {
      std::vector<uint32_t> unbake(baked_sphere.size(), RT_REF_NONE);
      for (uint32_t i = 0; i < n_complex; ++i)
        if (baked[i] != RT_REF_NONE) unbake[baked[i] - d.n_spheres] = i;
      auto f64_ref = [&](uint32_t ref) -> uint32_t {
        if (ref != RT_REF_NONE && (ref & RT_REF_LEAF) && RT_REF_KIND(ref) == RT_LEAF_SPHERE && RT_REF_INDEX(ref) >= d.n_spheres)
          return RT_MAKE_LEAF(RT_LEAF_COMPLEX, unbake[RT_REF_INDEX(ref) - d.n_spheres]);
        return ref;
      };
      nodes2 = std::move(tree2);
      for (Node2& n : nodes2) {
        n.left = f64_ref(n.left);
        n.right = f64_ref(n.right);
      }
      s.ordered_root = tree2_root;
      s.ordered_height = tree2_height;
    }
  }
  if constexpr (sizeof(R) == 4) {
    bool has_inst = false;
    for (const Complex& c : complexes) has_inst |= (c.flags & CXF_INSTANCE) != 0;
    const bool own_tree = getenv("RT_ORDERED_REFERENCE_TREE") == nullptr || has_inst;  // (the mirror has no instance arm)
    if (facts.conservative && facts.leaf_refs.size() >= 2 && own_tree) {
      const uint32_t n_leaves = (uint32_t)facts.leaf_refs.size();
      std::vector<uint32_t> refs(facts.leaf_refs);
      for (uint32_t& r : refs) r = f32_ref(r);
      static const char* slack_env = getenv("RT_TREE_SLACK");  // tuning override: levels above a balanced tree's height
      const uint32_t slack = slack_env ? (uint32_t)std::max(0, atoi(slack_env)) : 6u;
      nodes2 = accel::build_sah(facts.leaf_box.data(), refs.data(), n_leaves,
                                std::min(MAX_STACK_DEPTH, accel::ceil_log2(n_leaves) + slack), &s.ordered_height);
      s.ordered_root = 0;
    } else if (facts.conservative && d.n_nodes > 0 && !(d.root & RT_REF_LEAF) && d.root != RT_REF_NONE) {
      nodes2.resize(d.n_nodes);
      auto child_box = [&](uint32_t ref, float* out) {
        if (ref & RT_REF_LEAF) {
          const uint32_t k = RT_REF_KIND(ref), i = RT_REF_INDEX(ref);
          const uint32_t rank = k == RT_LEAF_SPHERE ? facts.rank_sphere[i] : k == RT_LEAF_MSPHERE ? facts.rank_msphere[i]
                                : k == RT_LEAF_CUBE ? facts.rank_cube[i] : facts.rank_complex[i];
          for (int c = 0; c < 6; ++c) out[c] = facts.leaf_box[6 * (size_t)rank + c];
        } else {
          for (int c = 0; c < 3; ++c) {
            out[c] = conv_down(d.node_box[6 * (size_t)ref + c]);
            out[3 + c] = conv_up(d.node_box[6 * (size_t)ref + 3 + c]);
          }
        }
      };
      for (uint32_t i = 0; i < d.n_nodes; ++i) {
        std::memset(&nodes2[i], 0, sizeof(Node2));
        child_box(d.node_left[i], nodes2[i].lbox);
        child_box(d.node_right[i], nodes2[i].rbox);
        nodes2[i].left = f32_ref(d.node_left[i]);
        nodes2[i].right = f32_ref(d.node_right[i]);
      }
      s.ordered_root = d.root;
      s.ordered_height = accel::height_of(nodes2, d.root);
    }
    tree2 = nodes2;
    tree2_root = s.ordered_root;
    tree2_height = s.ordered_height;
  }
  std::vector<Real4<R>> spheres(d.n_spheres);
  std::vector<Attr> sphere_attr(d.n_spheres);
  for (uint32_t i = 0; i < d.n_spheres; ++i) {
    const double* p = d.sphere + 4 * (size_t)i;
    spheres[i] = Real4<R>{conv<R>(p[0]), conv<R>(p[1]), conv<R>(p[2]), conv<R>(p[3])};
    sphere_attr[i] = Attr{d.sphere_material[i], d.sphere_object[i], facts.rank_sphere[i], 0};
  }
  for (size_t j = 0; j < baked_sphere.size(); ++j) {
    const auto& p = baked_sphere[j];
    spheres.push_back(Real4<R>{conv<R>(p[0]), conv<R>(p[1]), conv<R>(p[2]), conv<R>(p[3])});
    sphere_attr.push_back(baked_attr[j]);
  }
  std::vector<MSphere<R>> mspheres(d.n_mspheres);
  std::vector<Attr> msphere_attr(d.n_mspheres);
  for (uint32_t i = 0; i < d.n_mspheres; ++i) {
    const double* p = d.msphere + 9 * (size_t)i;
    MSphere<R> m;
    std::memset(&m, 0, sizeof(m));
    for (int k = 0; k < 3; ++k) {
      m.c0[k] = conv<R>(p[k]);
      m.c1[k] = conv<R>(p[3 + k]);
    }
    m.radius = conv<R>(p[6]);
    m.t0 = conv<R>(p[7]);
    m.t1 = conv<R>(p[8]);
    mspheres[i] = m;
    msphere_attr[i] = Attr{d.msphere_material[i], d.msphere_object[i], facts.rank_msphere[i], 0};
  }
  std::vector<Cube<R>> cubes(d.n_cubes);
  std::vector<Attr> cube_attr(d.n_cubes);
  for (uint32_t i = 0; i < d.n_cubes; ++i) {
    const double* p = d.cube + 6 * (size_t)i;
    Cube<R> c;
    std::memset(&c, 0, sizeof(c));
    for (int k = 0; k < 3; ++k) {
      c.p0[k] = conv<R>(p[k]);
      c.p1[k] = conv<R>(p[3 + k]);
    }
    cubes[i] = c;
    cube_attr[i] = Attr{d.cube_material[i], d.cube_object[i], facts.rank_cube[i], 0};
  }
  std::vector<uint32_t> op_tag(d.n_ops);
  std::vector<Op<R>> ops(d.n_ops);
  for (uint32_t i = 0; i < d.n_ops; ++i) {
    op_tag[i] = d.op_tag[i];
    for (int k = 0; k < 4; ++k) ops[i].d[k] = conv<R>(d.op_data[4 * (size_t)i + k]);
  }
  // which textures read hit.uv (validated acyclic: checker children have larger indices)
  std::vector<uint8_t> tex_uv(d.n_textures, 0);
  for (uint32_t i = d.n_textures; i-- > 0;) {
    if (d.tex_tag[i] == RT_TEX_IMAGE || d.tex_tag[i] == RT_TEX_UV) tex_uv[i] = 1;
    if (d.tex_tag[i] == RT_TEX_CHECKER) tex_uv[i] = tex_uv[d.tex_odd[i]] | tex_uv[d.tex_even[i]];
  }
  std::vector<Material<R>> mats(d.n_materials);
  for (uint32_t i = 0; i < d.n_materials; ++i) {
    mats[i].tag = d.mat_tag[i];
    mats[i].texture = d.mat_texture[i];
    mats[i].pad = 0;
    mats[i].flags = (d.mat_texture[i] < d.n_textures && tex_uv[d.mat_texture[i]]) ? MATF_NEEDS_UV : 0u;
    mats[i].pad2 = R(0);
    for (int k = 0; k < 3; ++k) mats[i].color[k] = R(0);
    if (d.mat_texture[i] < d.n_textures && d.tex_tag[d.mat_texture[i]] == RT_TEX_SOLID) {
      mats[i].flags |= MATF_SOLID;
      for (int k = 0; k < 3; ++k) mats[i].color[k] = conv<R>(d.tex_color[3 * (size_t)d.mat_texture[i] + k]);
    }
    for (int k = 0; k < 4; ++k) mats[i].p[k] = conv<R>(d.mat_params[4 * (size_t)i + k]);
  }
  std::vector<Texture<R>> texs(d.n_textures);
  for (uint32_t i = 0; i < d.n_textures; ++i) {
    std::memset(&texs[i], 0, sizeof(Texture<R>));
    texs[i].tag = d.tex_tag[i];
    texs[i].odd = d.tex_odd[i];
    texs[i].even = d.tex_even[i];
    texs[i].width = d.tex_width[i];
    texs[i].height = d.tex_height[i];
    texs[i].texel_offset = d.tex_texel_offset[i];
    for (int k = 0; k < 3; ++k) texs[i].color[k] = conv<R>(d.tex_color[3 * (size_t)i + k]);
  }
  // compose each complex object's wrapper chain into rigid maps (f64 on the host)
  std::vector<ComplexXf<R>> xfs(n_complex);
  for (uint32_t i = 0; i < n_complex; ++i) {
    double M[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, b[3] = {0, 0, 0};
    double Mo[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, bo[3] = {0, 0, 0};
    double nid = 0.0;
    for (uint32_t k = 0; k < complexes[i].op_count; ++k) {
      const uint32_t o = complexes[i].op_first + k;
      const double* q = d.op_data + 4 * (size_t)o;
      if (d.op_tag[o] == RT_OP_TRANSLATE) {
        for (int c = 0; c < 3; ++c) b[c] -= q[c];
      } else if (d.op_tag[o] == RT_OP_ROTATE) {
        // columns of the linear map v -> q v q* (alias.rs:112-117)
        double Rm[9];
        for (int c = 0; c < 3; ++c) {
          double e[3] = {0, 0, 0};
          e[c] = 1.0;
          double r[3];
          host_qrotate(e, q, r);
          Rm[0 + c] = r[0];
          Rm[3 + c] = r[1];
          Rm[6 + c] = r[2];
        }
        double M2[9], b2[3];
        for (int r_ = 0; r_ < 3; ++r_) {
          for (int c = 0; c < 3; ++c)
            M2[3 * r_ + c] = Rm[3 * r_] * M[c] + Rm[3 * r_ + 1] * M[3 + c] + Rm[3 * r_ + 2] * M[6 + c];
          b2[r_] = Rm[3 * r_] * b[0] + Rm[3 * r_ + 1] * b[1] + Rm[3 * r_ + 2] * b[2];
        }
        std::memcpy(M, M2, sizeof(M));
        std::memcpy(b, b2, sizeof(b));
      } else {
        std::memcpy(Mo, M, sizeof(M));
        std::memcpy(bo, b, sizeof(b));
        nid = q[0];
      }
    }
    if (complexes[i].medium_pos == REF_NONE) {
      std::memcpy(Mo, M, sizeof(M));
      std::memcpy(bo, b, sizeof(b));
    }
    std::memset(&xfs[i], 0, sizeof(ComplexXf<R>));
    for (int c = 0; c < 9; ++c) {
      xfs[i].m_full[c] = conv<R>(M[c]);
      xfs[i].m_outer[c] = conv<R>(Mo[c]);
    }
    for (int c = 0; c < 3; ++c) {
      xfs[i].b_full[c] = conv<R>(b[c]);
      xfs[i].b_outer[c] = conv<R>(bo[c]);
    }
    xfs[i].neg_inv_density = conv<R>(nid);
  }
  std::vector<uint8_t> texels(d.texels, d.texels + (d.texels ? d.n_texel_bytes : 0));
  std::vector<Light<R>> lights(d.n_lights);
  for (uint32_t i = 0; i < d.n_lights; ++i) {
    std::memset(&lights[i], 0, sizeof(Light<R>));
    lights[i].tag = d.light_tag[i];
    for (int k = 0; k < 3; ++k) lights[i].c[k] = conv<R>(d.light_sphere[4 * (size_t)i + k]);
    lights[i].radius = conv<R>(d.light_sphere[4 * (size_t)i + 3]);
    if (d.light_tag[i] == RT_LIGHT_CUBE) {  // extension: p0, p1 and the surface area (in f64, rounded once)
      const double* b = d.light_box + 6 * (size_t)i;
      const double ex = b[3] - b[0], ey = b[4] - b[1], ez = b[5] - b[2];
      for (int k = 0; k < 3; ++k) {
        lights[i].c[k] = conv<R>(b[k]);
        lights[i].hi[k] = conv<R>(b[3 + k]);
      }
      lights[i].radius = conv<R>(2.0 * (ey * ez + ex * ez + ex * ey));
    }
  }
  hipError_t e;
#define UP(vec, field) \
  if ((e = s.bufs.upload(vec, &s.view.field)) != hipSuccess) return e;
  UP(nodes, nodes)
  UP(nodes2, nodes2)
  UP(spheres, spheres)
  UP(mspheres, mspheres)
  UP(cubes, cubes)
  UP(sphere_attr, sphere_attr)
  UP(msphere_attr, msphere_attr)
  UP(cube_attr, cube_attr)
  UP(complexes, complexes)
  UP(xfs, complex_xf)
  UP(op_tag, op_tag)
  UP(ops, ops)
  UP(mats, materials)
  UP(texs, textures)
  UP(texels, texels)
  UP(lights, lights)
  // leaf-list records (see device/scene_view.hpp LIN_*): f32 only
  std::vector<uint32_t> linear_rec;
  if constexpr (sizeof(R) == 4) {
    linear_rec.assign(linear.size() * LIN_DWORDS, 0u);
    auto put = [&](size_t k, uint32_t off, float v) { std::memcpy(&linear_rec[k * LIN_DWORDS + off], &v, 4); };
    for (size_t k = 0; k < linear.size(); ++k) {
      uint32_t ref = f32_ref(linear[k]), body = ref, flags = 0, material = 0, object = 0;
      float nid = 0.0f;
      if (RT_REF_KIND(ref) == RT_LEAF_COMPLEX) {
        const uint32_t ci = RT_REF_INDEX(ref);
        const Complex& cx = complexes[ci];
        body = cx.body;
        object = cx.object;
        if (cx.n_rotate + cx.n_translate) flags |= LINF_XF | (cx.n_rotate << 16) | (cx.n_translate << 24);
        for (int c = 0; c < 9; ++c) put(k, LIN_M + c, (float)xfs[ci].m_full[c]);
        for (int c = 0; c < 3; ++c) put(k, LIN_B + c, (float)xfs[ci].b_full[c]);
        if (cx.medium_pos != REF_NONE) {
          flags |= LINF_MEDIUM;
          nid = (float)xfs[ci].neg_inv_density;
          material = cx.material;
        }
      }
      const uint32_t bi = RT_REF_INDEX(body);
      if (RT_REF_KIND(body) == RT_LEAF_CUBE) {
        flags |= LINF_CUBE;
        for (int c = 0; c < 6; ++c) put(k, LIN_PRIM + c, (float)d.cube[6 * (size_t)bi + c]);
        for (int c = 0; c < 3; ++c)
          put(k, LIN_INVEXT + c, (float)(1.0 / ((double)(float)d.cube[6 * (size_t)bi + 3 + c] - (double)(float)d.cube[6 * (size_t)bi + c])));
        if (!(flags & LINF_MEDIUM)) material = d.cube_material[bi];
        if (RT_REF_KIND(ref) != RT_LEAF_COMPLEX) object = d.cube_object[bi];
      } else {
        const Real4<R>& sp = spheres[bi];  // (incl. baked records)
        put(k, LIN_PRIM + 0, (float)sp.x);
        put(k, LIN_PRIM + 1, (float)sp.y);
        put(k, LIN_PRIM + 2, (float)sp.z);
        put(k, LIN_PRIM + 3, (float)sp.w);
        if (!(flags & LINF_MEDIUM)) material = sphere_attr[bi].material;
        if (RT_REF_KIND(ref) != RT_LEAF_COMPLEX) object = sphere_attr[bi].object;
      }
      linear_rec[k * LIN_DWORDS + LIN_FLAGS] = flags;
      linear_rec[k * LIN_DWORDS + LIN_REF] = ref;
      put(k, LIN_NID, nid);
      linear_rec[k * LIN_DWORDS + LIN_OBJECT] = object;
      static_assert(sizeof(Material<float>) == 12 * 4, "LIN_MAT image is 12 dwords");
      if constexpr (sizeof(R) == 4) std::memcpy(&linear_rec[k * LIN_DWORDS + LIN_MAT], &mats[material], sizeof(Material<float>));
    }
  }
  UP(linear_rec, linear_rec)
  // leaf list over the scene arrays (traverse_leaves: the f64 kernel's walk on leaf-list scenes); eligible when the
  // arrays a leaf test reads fit the LDS staging area
  std::vector<uint32_t> leaf_list;
  if (!linear.empty() && spheres.size() <= STAGE_MAX_SPHERES && cubes.size() <= STAGE_MAX_CUBES &&
      complexes.size() <= STAGE_MAX_COMPLEXES && ops.size() <= STAGE_MAX_OPS)
    for (uint32_t ref : linear) leaf_list.push_back(f32_ref(ref));
  UP(leaf_list, leaf_list)
  std::vector<uint32_t> leaf_by_rank(facts.leaf_refs);
  for (uint32_t& r : leaf_by_rank) r = f32_ref(r);
  UP(leaf_by_rank, leaf_by_rank)
#undef UP
  s.view.has_instances = 0;
  for (const Complex& c : complexes) s.view.has_instances |= (c.flags & CXF_INSTANCE) ? 1u : 0u;
  s.view.n_leaf_list = (uint32_t)leaf_list.size();
  s.view.stage_spheres = (uint32_t)spheres.size();
  s.view.stage_cubes = (uint32_t)cubes.size();
  s.view.stage_complexes = (uint32_t)complexes.size();
  s.view.stage_ops = (uint32_t)ops.size();
  s.view.n_linear = sizeof(R) == 4 ? (uint32_t)linear.size() : 0u;
  s.view.has_noise = 0;
  for (uint32_t i = 0; i < d.n_textures; ++i)
    if (d.tex_tag[i] == RT_TEX_NOISE) s.view.has_noise = 1;
  for (uint32_t i = 0; i < d.n_materials; ++i)
    if (d.mat_tag[i] == RT_MAT_AREA_LIGHT) s.view.has_noise = 1;
  for (uint32_t i = 0; i < d.n_lights; ++i)
    if (d.light_tag[i] == RT_LIGHT_CUBE) s.view.has_noise = 1;
  s.view.linear_has_media = 0;
  s.view.linear_xf_mask = s.view.linear_cube_mask = s.view.linear_medium_mask = 0;
  static_assert(LIN_MAX_LEAVES <= 32, "leaf-list flag masks are 32 bits wide");
  for (size_t k = 0; k < linear_rec.size() / LIN_DWORDS; ++k) {
    const uint32_t f = linear_rec[k * LIN_DWORDS + LIN_FLAGS];
    if (f & LINF_MEDIUM) s.view.linear_has_media = 1;
    if (f & LINF_XF) s.view.linear_xf_mask |= 1u << k;
    if (f & LINF_CUBE) s.view.linear_cube_mask |= 1u << k;
    if (f & LINF_MEDIUM) s.view.linear_medium_mask |= 1u << k;
  }
  s.view.root = f32_ref(d.root);
  s.view.root2 = s.ordered_root;
  s.view.lights_kind = d.lights_kind;
  s.view.n_lights = d.n_lights;
  for (int k = 0; k < 3; ++k) s.view.background[k] = conv<R>(d.background[k]);
  return hipSuccess;
}

template <typename R> CameraView<R> to_camera(const rt_camera& c) {
  auto v3 = [](const double* p) { return Vec3<R>((R)p[0], (R)p[1], (R)p[2]); };
  CameraView<R> o;
  o.origin = v3(c.origin);
  o.lower_left = v3(c.lower_left);
  o.horizontal = v3(c.horizontal);
  o.vertical = v3(c.vertical);
  o.u = v3(c.u);
  o.v = v3(c.v);
  o.ll_rel = Vec3<R>((R)(c.lower_left[0] - c.origin[0]), (R)(c.lower_left[1] - c.origin[1]),
                     (R)(c.lower_left[2] - c.origin[2]));
  o.lens_radius = (R)c.lens_radius;
  o.time_open = (R)c.time_open;
  o.time_close = (R)c.time_close;
  return o;
}

bool is_simple_leaf(uint32_t ref, const rt_scene_desc& d) {
  if (ref == RT_REF_NONE || !(ref & RT_REF_LEAF)) return false;
  uint32_t k = RT_REF_KIND(ref), i = RT_REF_INDEX(ref);
  if (k == RT_LEAF_SPHERE) return i < d.n_spheres;
  if (k == RT_LEAF_MSPHERE) return i < d.n_mspheres;
  if (k == RT_LEAF_CUBE) return i < d.n_cubes;
  return false;
}
bool is_valid_leaf(uint32_t ref, const rt_scene_desc& d) {
  if (is_simple_leaf(ref, d)) return true;
  if (ref == RT_REF_NONE || !(ref & RT_REF_LEAF)) return false;
  return RT_REF_KIND(ref) == RT_LEAF_COMPLEX && RT_REF_INDEX(ref) < d.n_complex;
}

// Bounds of a bare primitive in its own frame.  A moving sphere is bounded over its own [time_start, time_end] only;
// *t_lo / *t_hi are narrowed to the ray times for which the bounds hold.
bool prim_local_box(const rt_scene_desc& d, uint32_t r, double plo[3], double phi[3], double* t_lo = nullptr,
                    double* t_hi = nullptr) {
  const uint32_t k = RT_REF_KIND(r), i = RT_REF_INDEX(r);
  if (k == RT_LEAF_SPHERE) {
    const double* s = d.sphere + 4 * (size_t)i;
    for (int c = 0; c < 3; ++c) {
      plo[c] = s[c] - std::fabs(s[3]);
      phi[c] = s[c] + std::fabs(s[3]);
    }
    return true;
  }
  if (k == RT_LEAF_CUBE) {
    const double* q = d.cube + 6 * (size_t)i;
    for (int c = 0; c < 3; ++c) {
      plo[c] = std::fmin(q[c], q[3 + c]);
      phi[c] = std::fmax(q[c], q[3 + c]);
    }
    return true;
  }
  if (k == RT_LEAF_MSPHERE) {  // centre moves linearly between c0 (time_start) and c1 (time_end)
    const double* m = d.msphere + 9 * (size_t)i;
    const double r_ = std::fabs(m[6]);
    if (!(m[7] < m[8])) return false;
    for (int c = 0; c < 3; ++c) {
      plo[c] = std::fmin(m[c], m[3 + c]) - r_;
      phi[c] = std::fmax(m[c], m[3 + c]) + r_;
    }
    if (t_lo) *t_lo = std::fmax(*t_lo, m[7]);
    if (t_hi) *t_hi = std::fmin(*t_hi, m[8]);
    return true;
  }
  return false;
}

// The box [plo, phi] of a complex's body frame, seen from outside its wrapper chain: the eight corners taken back out
// through the wrappers, innermost first — Rotate by q* (hittable.rs:379-383), Translate by +offset — and bounded.
void chain_bounds(const rt_scene_desc& d, const Complex& cx, const double plo[3], const double phi[3], double lo[3],
                  double hi[3]) {
  for (int c = 0; c < 3; ++c) {
    lo[c] = INFINITY;
    hi[c] = -INFINITY;
  }
  for (int corner = 0; corner < 8; ++corner) {
    double p[3] = {(corner & 1) ? phi[0] : plo[0], (corner & 2) ? phi[1] : plo[1], (corner & 4) ? phi[2] : plo[2]};
    for (uint32_t k = cx.op_count; k-- > 0;) {
      const uint32_t o = cx.op_first + k;
      const double* q = d.op_data + 4 * (size_t)o;
      if (d.op_tag[o] == RT_OP_TRANSLATE) {
        for (int c = 0; c < 3; ++c) p[c] += q[c];
      } else if (d.op_tag[o] == RT_OP_ROTATE) {
        const double qi[4] = {q[0], -q[1], -q[2], -q[3]};
        double r[3];
        host_qrotate(p, qi, r);
        for (int c = 0; c < 3; ++c) p[c] = r[c];
      }
    }
    for (int c = 0; c < 3; ++c) {
      lo[c] = std::fmin(lo[c], p[c]);
      hi[c] = std::fmax(hi[c], p[c]);
    }
  }
}

// Conservative world-space bounds of a leaf (for a complex whose body is a primitive).
bool leaf_world_bounds(const rt_scene_desc& d, const std::vector<Complex>& complexes, uint32_t ref, double lo[3],
                       double hi[3], double* t_lo = nullptr, double* t_hi = nullptr) {
  if (RT_REF_KIND(ref) != RT_LEAF_COMPLEX) return prim_local_box(d, ref, lo, hi, t_lo, t_hi);
  const Complex& cx = complexes[RT_REF_INDEX(ref)];
  double plo[3], phi[3];
  if ((cx.flags & CXF_INSTANCE) || !prim_local_box(d, cx.body, plo, phi, t_lo, t_hi)) return false;
  chain_bounds(d, cx, plo, phi, lo, hi);
  return true;
}

// Waves a launch should have: ~16 rounds of the chip's wave slots (256 CUs x 4 SIMDs x 6), so that the last,
// partly filled round costs little; a 1024^2 image has 16384 waves and is split 6 ways along the samples
// (measured: 45.6 Grays/s unsplit, 49.6 at 6 splits, flat to 12).
constexpr uint32_t TARGET_WAVES = 96u * 1024u;

}  // namespace

// ---- scratch of rt_render_device_noalloc (rt.h) ----------------------------------------------------------------
// One 256-byte slot per kernel launch of a call (a DeviceStats block: the work counters and the persistent
// kernels' unit counter), in a file-scope device array instead of memory the library allocates.  Every GPU the
// library runs on holds its own copy of the array at its own address; g_scratch_ptr[device] is that address
// (hipGetSymbolAddress, once per device, with that device current: noalloc_scratch_for below).  Host code must
// never pass the array's NAME to a kernel or a memset: it is a host-side stand-in, not a device address.
constexpr size_t NOALLOC_SLOT_BYTES = 256;
static_assert(sizeof(DeviceStats) <= NOALLOC_SLOT_BYTES, "one DeviceStats block per scratch slot");
// RT_NOALLOC_PARTITIONS partitions of RT_NOALLOC_MAX_LAUNCHES slots: an rt_scene uses the partition its creation
// ordinal on that device selects, so that up to that many scenes (streams, threads) may use the entry side by side.
static __device__ __align__(256) unsigned char scratch[NOALLOC_SLOT_BYTES * RT_NOALLOC_MAX_LAUNCHES * RT_NOALLOC_PARTITIONS];

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[MAX_DEVICES] = {};
uint32_t g_scene_ordinal[MAX_DEVICES] = {};  // scenes created so far per device (under g_scratch_mutex)

// Device address of `scratch` on `device` (which must be the current device).
// *out: the partition of the next scene created on this device.
hipError_t noalloc_scratch_for(int device, unsigned char** out) {
  *out = nullptr;
  if (device < 0 || device >= MAX_DEVICES) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[device]) {
    unsigned char* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!ptr) return hipErrorInvalidSymbol;
    g_scratch_ptr[device] = ptr;
  }
  *out = g_scratch_ptr[device] + (size_t)(g_scene_ordinal[device]++ % RT_NOALLOC_PARTITIONS) * RT_NOALLOC_MAX_LAUNCHES * NOALLOC_SLOT_BYTES;
  return hipSuccess;
}
}  // namespace

struct rt_scene {
  int device = 0;
  unsigned char* static_scratch = nullptr;  // this device's copy of `scratch` (never freed: it is not ours to free)
  uint32_t stack_depth = 1;
  uint32_t n_nodes = 0;
  bool boxes_conservative = false;               // leaf-list / ordered walks allowed ...
  double safe_t_lo = 0, safe_t_hi = 0;           // ... for cameras whose shutter lies in this window
  // Some Cube carries a Dielectric.  A ray refracted INTO a cube re-hits it at t_min (Cube::hit reports the running t_min
  // for a ray that starts inside, hittable.rs:211-228), and the reflection at that vertex leaves through the entry face
  // after EXACTLY t = t_min in exact arithmetic: `t_max <= t_min` (hittable.rs:224 — and aabb.rs:30 for every box above
  // the cube) is then decided by the last bit, box by box.  Only the walk that tests the reference's own boxes in the
  // reference's own order reproduces those decisions, so the f64 kernel — whose contract is the reference's result bit for
  // bit — keeps the reference-order walk on such scenes (no reference scene has one; tests/sweep_mixed.py found 3 in 4004).
  bool glass_cube = false;
  SceneSide<float> f32;
  SceneSide<double> f64;
  uint64_t last_redecided = 0;  // rt_primary_redecided
  // wavefront path (device/wavefront.hpp): the path pool, allocated at the first render that takes it
  void* wf_mem = nullptr;
  uint32_t wf_capacity = 0;       // slots
  WfCounters* wf_ctr = nullptr;   // device
  unsigned int* wf_host = nullptr;  // pinned: the `retired` count as last read
  uint32_t wf_trace_blocks = 0;
  // Counter blocks (DeviceStats: work counters + the persistent kernels' unit counter) of the render calls: a ring of
  // RING_CHUNKS chunks of RING_CHUNK_SLOTS 256-byte slots.  Every call takes the next chunk for itself, so that calls
  // enqueued on different streams (or from different threads) on ONE scene do not share counters; a chunk is reused
  // after RING_CHUNKS further calls, by which time a caller must have let the earlier call finish.
  static constexpr uint32_t RING_CHUNKS = 16, RING_CHUNK_SLOTS = 16;
  unsigned char* dstats = nullptr;
  std::atomic<uint32_t> ring_next{0};
  unsigned char* next_chunk() {
    return dstats + (size_t)(ring_next.fetch_add(1, std::memory_order_relaxed) % RING_CHUNKS) * RING_CHUNK_SLOTS * NOALLOC_SLOT_BYTES;
  }
  hipEvent_t ev0 = nullptr, ev1 = nullptr;  // timing events of the entries that must not allocate (and the wavefront form)
  ~rt_scene() {
    if (dstats) (void)hipFree(dstats);
    if (wf_mem) (void)hipFree(wf_mem);
    if (wf_ctr) (void)hipFree(wf_ctr);
    if (wf_host) (void)hipHostFree(wf_host);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
  }
};

extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }
uint32_t rt_abi_version(void) { return RT_ABI_VERSION; }

int rt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int rt_scene_create(int device, const rt_scene_desc* d, rt_scene** out) {
  if (!d || !out) return fail(RT_ERR_BAD_ARG, "rt_scene_create: null argument");
  *out = nullptr;
  if (d->abi_version != RT_ABI_VERSION)
    return fail(RT_ERR_BAD_ARG, "rt_scene_create: abi_version %u, library is %u", d->abi_version, RT_ABI_VERSION);
#define NEED(cond, what) \
  if (!(cond)) return fail(RT_ERR_BAD_ARG, "rt_scene_create: %s", what);
  NEED(d->n_nodes == 0 || (d->node_box && d->node_left && d->node_right), "null node arrays");
  NEED(d->n_spheres == 0 || (d->sphere && d->sphere_material && d->sphere_object), "null sphere arrays");
  NEED(d->n_mspheres == 0 || (d->msphere && d->msphere_material && d->msphere_object), "null moving-sphere arrays");
  NEED(d->n_cubes == 0 || (d->cube && d->cube_material && d->cube_object), "null cube arrays");
  NEED(d->n_complex == 0 || (d->complex_body && d->complex_op_first && d->complex_op_count && d->complex_material &&
                             d->complex_object),
       "null complex arrays");
  NEED(d->n_ops == 0 || (d->op_tag && d->op_data), "null op arrays");
  NEED(d->n_materials == 0 || (d->mat_tag && d->mat_texture && d->mat_params), "null material arrays");
  NEED(d->n_textures == 0 || (d->tex_tag && d->tex_color && d->tex_odd && d->tex_even && d->tex_texel_offset &&
                              d->tex_width && d->tex_height),
       "null texture arrays");
  NEED(d->n_texel_bytes == 0 || d->texels, "null texels");
  NEED(d->n_lights == 0 || (d->light_tag && d->light_sphere), "null light arrays");
  NEED(d->n_nodes < 0x80000000u && d->n_spheres <= 0x0FFFFFFFu && d->n_mspheres <= 0x0FFFFFFFu &&
           d->n_cubes <= 0x0FFFFFFFu && d->n_complex <= 0x0FFFFFFFu,
       "too many elements for a 28-bit leaf index");
  NEED(d->lights_kind <= RT_LIGHTS_LIST, "bad lights_kind");
  NEED((d->lights_kind == RT_LIGHTS_NONE) == (d->n_lights == 0), "lights_kind / n_lights mismatch");
  NEED(d->lights_kind != RT_LIGHTS_SINGLE || d->n_lights == 1, "RT_LIGHTS_SINGLE needs exactly one light");
  for (uint32_t i = 0; i < d->n_lights; ++i) {
    NEED(d->light_tag[i] <= RT_LIGHT_CUBE, "bad light tag");
    if (d->light_tag[i] == RT_LIGHT_CUBE) {
      NEED(d->light_box, "RT_LIGHT_CUBE without light_box");
      const double* b = d->light_box + 6 * (size_t)i;
      NEED(b[0] < b[3] && b[1] < b[4] && b[2] < b[5] && std::isfinite(b[0] + b[1] + b[2] + b[3] + b[4] + b[5]),
           "RT_LIGHT_CUBE: light_box needs finite p0 < p1");
    }
  }

  // textures: tags, checker children in range and acyclic (children must have a larger index)
  for (uint32_t i = 0; i < d->n_textures; ++i) {
    NEED(d->tex_tag[i] <= RT_TEX_NOISE, "bad texture tag");
    if (d->tex_tag[i] == RT_TEX_NOISE) {  // extension: the gradient table lives in the texel blob
      NEED(d->tex_texel_offset[i] % 8 == 0, "noise table must be 8-byte aligned inside texels");
      NEED(d->tex_texel_offset[i] + RT_NOISE_TABLE_BYTES <= d->n_texel_bytes, "noise table outside texels");
    }
    if (d->tex_tag[i] == RT_TEX_CHECKER)
      NEED(d->tex_odd[i] < d->n_textures && d->tex_even[i] < d->n_textures && d->tex_odd[i] > i && d->tex_even[i] > i,
           "checker children must be textures with a larger index");
    if (d->tex_tag[i] == RT_TEX_IMAGE) {
      NEED(d->tex_width[i] > 0 && d->tex_height[i] > 0, "empty image texture");
      NEED(d->tex_texel_offset[i] + 3ull * d->tex_width[i] * d->tex_height[i] <= d->n_texel_bytes,
           "image texture outside texels");
    }
  }
  for (uint32_t i = 0; i < d->n_materials; ++i) {
    NEED(d->mat_tag[i] <= RT_MAT_AREA_LIGHT, "bad material tag");
    bool textured = d->mat_tag[i] == RT_MAT_LAMBERTIAN || d->mat_tag[i] == RT_MAT_DIFFUSE_LIGHT ||
                    d->mat_tag[i] == RT_MAT_ISOTROPIC || d->mat_tag[i] == RT_MAT_AREA_LIGHT;
    if (textured) NEED(d->mat_texture[i] < d->n_textures, "material texture out of range");
  }
  for (uint32_t i = 0; i < d->n_spheres; ++i) NEED(d->sphere_material[i] < d->n_materials, "sphere material out of range");
  for (uint32_t i = 0; i < d->n_mspheres; ++i)
    NEED(d->msphere_material[i] < d->n_materials, "moving-sphere material out of range");
  for (uint32_t i = 0; i < d->n_cubes; ++i) NEED(d->cube_material[i] < d->n_materials, "cube material out of range");

  std::vector<Complex> complexes(d->n_complex);
  for (uint32_t i = 0; i < d->n_complex; ++i) {
    // a body that is a node index (no RT_REF_LEAF): an INSTANCE, the wrappers over a sub-tree (hittable.rs:81-91 hold
    // a Box<Hittable> of any variant: here a Bvh)
    const bool instance = d->complex_body[i] != RT_REF_NONE && !(d->complex_body[i] & RT_REF_LEAF);
    if (instance) {
      NEED(d->complex_body[i] < d->n_nodes, "complex body: node index out of range");
    } else if (!is_simple_leaf(d->complex_body[i], *d)) {
      return fail(RT_ERR_UNSUPPORTED,
                  "rt_scene_create: complex %u body must be a sphere / moving sphere / cube leaf or a node (a sub-tree)", i);
    }
    NEED((uint64_t)d->complex_op_first[i] + d->complex_op_count[i] <= d->n_ops, "complex op range out of bounds");
    Complex c{};
    c.flags = instance ? CXF_INSTANCE : 0u;
    c.body = d->complex_body[i];
    c.op_first = d->complex_op_first[i];
    c.op_count = d->complex_op_count[i];
    c.medium_pos = REF_NONE;
    c.material = d->complex_material[i];
    c.object = d->complex_object[i];
    for (uint32_t k = 0; k < c.op_count; ++k) {
      uint8_t tag = d->op_tag[c.op_first + k];
      NEED(tag <= RT_OP_MEDIUM, "bad op tag");
      if (tag == RT_OP_ROTATE) {
        const double* q = d->op_data + 4 * (size_t)(c.op_first + k);
        const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
        if (!(std::fabs(n2 - 1.0) <= 1e-9))
          return fail(RT_ERR_UNSUPPORTED, "rt_scene_create: complex %u has a non-unit rotation quaternion", i);
        ++c.n_rotate;
      }
      if (tag == RT_OP_TRANSLATE) ++c.n_translate;
      if (tag == RT_OP_MEDIUM) {
        if (c.medium_pos != REF_NONE)
          return fail(RT_ERR_UNSUPPORTED, "rt_scene_create: complex %u nests two constant media", i);
        if (instance)
          return fail(RT_ERR_UNSUPPORTED, "rt_scene_create: complex %u: a constant medium whose boundary is a sub-tree", i);
        c.medium_pos = k;
      }
    }
    if (c.medium_pos != REF_NONE) NEED(c.material < d->n_materials, "medium material out of range");
    complexes[i] = c;
  }

  // BVH: pre-order (children after parents) makes it acyclic; compute its height bottom-up.
  // (height = stack entries a walk below this node may hold: an instance leaf costs its marker plus its sub-tree)
  std::vector<uint32_t> height(d->n_nodes, 0);
  auto instance_of = [&](uint32_t ref) -> const Complex* {
    if (ref == RT_REF_NONE || !(ref & RT_REF_LEAF) || RT_REF_KIND(ref) != RT_LEAF_COMPLEX) return nullptr;
    const Complex& c = complexes[RT_REF_INDEX(ref)];
    return (c.flags & CXF_INSTANCE) ? &c : nullptr;
  };
  for (uint32_t ii = d->n_nodes; ii-- > 0;) {
    uint32_t h = 0;
    for (uint32_t ref : {d->node_left[ii], d->node_right[ii]}) {
      if (ref & RT_REF_LEAF) {
        NEED(is_valid_leaf(ref, *d), "node child is not a valid leaf");
        if (const Complex* inst = instance_of(ref)) {
          NEED(inst->body > ii, "an instance's sub-tree must come after the node that holds the instance (pre-order)");
          h = std::max(h, 1u + height[inst->body]);
        }
      } else {
        NEED(ref < d->n_nodes && ref > ii, "node child must be a later node (pre-order)");
        h = std::max(h, height[ref]);
      }
    }
    height[ii] = h + 1;
  }
  uint32_t tree_height = 0;
  if (d->root == RT_REF_NONE) {
  } else if (d->root & RT_REF_LEAF) {
    NEED(is_valid_leaf(d->root, *d), "root is not a valid leaf");
    if (const Complex* inst = instance_of(d->root)) tree_height = 1u + height[inst->body];
  } else {
    NEED(d->root < d->n_nodes, "root out of range");
    tree_height = height[d->root];
  }
#undef NEED
  if (tree_height > MAX_STACK_DEPTH)
    return fail(RT_ERR_STACK_DEPTH, "rt_scene_create: BVH height %u exceeds the %u-entry LDS stack", tree_height,
                MAX_STACK_DEPTH);

  // One left-first DFS (the order the reference's recursion reaches the leaves): DFS rank of every
  // leaf, its bounds, and whether every box in the tree really bounds what is below it.  The
  // reference's Rotate::bounding_box does not always (it can clip the object), and then
  // pruning by that box is part of the image: only the reference-order walk reproduces it, so the
  // leaf-list and ordered walks are enabled only for trees whose boxes are verified conservative.
  TreeFacts facts;
  facts.rank_sphere.assign(d->n_spheres, RT_REF_NONE);
  facts.rank_msphere.assign(d->n_mspheres, RT_REF_NONE);
  facts.rank_cube.assign(d->n_cubes, RT_REF_NONE);
  facts.rank_complex.assign(d->n_complex, RT_REF_NONE);
  if (d->root != RT_REF_NONE) {
    struct Todo {
      uint32_t ref, parent;  // parent: the node whose box must bound this entry (RT_REF_NONE: none to check)
      uint32_t inst;         // the instance (complex index) whose sub-tree this entry belongs to, or RT_REF_NONE
    };
    std::vector<Todo> todo{{d->root, RT_REF_NONE, RT_REF_NONE}};
    auto inside = [&](const double lo[3], const double hi[3], uint32_t parent) {
      if (parent == RT_REF_NONE) return true;
      const double* nb = d->node_box + 6 * (size_t)parent;
      for (int k = 0; k < 3; ++k) {
        const double tol = 1e-9 * (1.0 + std::fabs(lo[k]) + std::fabs(hi[k]));
        if (!(nb[k] <= lo[k] + tol && hi[k] <= nb[3 + k] + tol)) return false;
      }
      return true;
    };
    std::vector<uint8_t> seen(d->n_nodes, 0);
    uint32_t nodes_reached = 0;
    while (!todo.empty()) {
      Todo cur = todo.back();
      todo.pop_back();
      if (!(cur.ref & RT_REF_LEAF)) {
        if (seen[cur.ref]) return fail(RT_ERR_BAD_ARG, "rt_scene_create: node %u is referenced more than once", cur.ref);
        seen[cur.ref] = 1;
        ++nodes_reached;
        const double* nb = d->node_box + 6 * (size_t)cur.ref;
        if (!inside(nb, nb + 3, cur.parent)) facts.conservative = false;
        todo.push_back({d->node_right[cur.ref], cur.ref, cur.inst});
        todo.push_back({d->node_left[cur.ref], cur.ref, cur.inst});
        continue;
      }
      const uint32_t k = RT_REF_KIND(cur.ref), i = RT_REF_INDEX(cur.ref);
      if (k == RT_LEAF_COMPLEX && (complexes[i].flags & CXF_INSTANCE)) {
        // An instance: its sub-tree's leaves take their place in the DFS order here.  Its box in the parent's frame =
        // the sub-tree root's box through the wrappers (what hittable.rs:436-456 computes from the child's box).
        if (cur.inst != RT_REF_NONE)
          return fail(RT_ERR_UNSUPPORTED, "rt_scene_create: complex %u: an instance inside an instance", i);
        if (facts.rank_complex[i] != RT_REF_NONE)
          return fail(RT_ERR_BAD_ARG, "rt_scene_create: a leaf is referenced by more than one node");
        facts.rank_complex[i] = (uint32_t)facts.leaf_refs.size();  // (rank of its first leaf: marks it as placed)
        const double* rb = d->node_box + 6 * (size_t)complexes[i].body;
        double lo[3], hi[3];
        chain_bounds(*d, complexes[i], rb, rb + 3, lo, hi);
        if (!inside(lo, hi, cur.parent)) facts.conservative = false;
        todo.push_back({complexes[i].body, RT_REF_NONE, i});
        continue;
      }
      uint32_t leaf = cur.ref;
      if (cur.inst != RT_REF_NONE) {
        // a leaf of an instance's sub-tree: for every walk but the reference-order one it is a complex of its own —
        // the instance's wrapper chain around this primitive — appended behind the caller's complexes
        if (k == RT_LEAF_COMPLEX)
          return fail(RT_ERR_UNSUPPORTED, "rt_scene_create: an instance's sub-tree must hold bare primitives (complex %u inside complex %u)",
                      i, cur.inst);
        double plo[3], phi[3];
        if (!prim_local_box(*d, cur.ref, plo, phi, &facts.t_lo, &facts.t_hi) || !inside(plo, phi, cur.parent)) facts.conservative = false;
        Complex syn = complexes[cur.inst];
        syn.flags = 0;
        syn.body = cur.ref;
        syn.rank = (uint32_t)facts.leaf_refs.size();
        leaf = RT_MAKE_LEAF(RT_LEAF_COMPLEX, (uint32_t)complexes.size());
        complexes.push_back(syn);
        facts.rank_complex.push_back(RT_REF_NONE);
      }
      const uint32_t rank = (uint32_t)facts.leaf_refs.size();
      facts.leaf_refs.push_back(leaf);
      std::vector<uint32_t>& ranks = k == RT_LEAF_SPHERE ? facts.rank_sphere
                                     : k == RT_LEAF_MSPHERE ? facts.rank_msphere
                                     : k == RT_LEAF_CUBE ? facts.rank_cube : facts.rank_complex;
      if (ranks[i] != RT_REF_NONE)
        return fail(RT_ERR_BAD_ARG, "rt_scene_create: a leaf is referenced by more than one node");
      ranks[i] = rank;
      if (leaf != cur.ref) facts.rank_complex[RT_REF_INDEX(leaf)] = rank;
      double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
      if (!leaf_world_bounds(*d, complexes, leaf, lo, hi, &facts.t_lo, &facts.t_hi)) {
        facts.conservative = false;
      } else if (cur.inst == RT_REF_NONE && !inside(lo, hi, cur.parent)) {
        facts.conservative = false;
      }
      // The f32 kernels map the ray into a wrapped object's frame with rounded maps: give such a
      // leaf's culling box a margin well above that rounding (culling only, never a hit test).
      const uint32_t lk = RT_REF_KIND(leaf);
      for (int c = 0; c < 3; ++c) {
        const double pad = lk == RT_LEAF_COMPLEX ? 4e-6 * (std::fabs(lo[c]) + std::fabs(hi[c]) + 1.0) : 0.0;
        lo[c] -= pad;
        hi[c] += pad;
      }
      for (int c = 0; c < 3; ++c) facts.leaf_box.push_back(conv_down(lo[c]));
      for (int c = 0; c < 3; ++c) facts.leaf_box.push_back(conv_up(hi[c]));
      if (k == RT_LEAF_MSPHERE) facts.linear_ok = false;
      if (lk == RT_LEAF_COMPLEX) {
        const Complex& cx = complexes[RT_REF_INDEX(leaf)];
        if (RT_REF_KIND(cx.body) == RT_LEAF_MSPHERE) facts.linear_ok = false;
        if (cx.medium_pos != REF_NONE && cx.medium_pos != 0) facts.linear_ok = false;  // records keep media outermost
      }
    }
    // every node must hang below the root, directly or through an instance: an orphan node's leaves would have no DFS
    // rank, and nothing the reference builds (hittable.rs:159-206) has one
    if (nodes_reached != d->n_nodes)
      return fail(RT_ERR_BAD_ARG, "rt_scene_create: %u of %u nodes are not reachable from the root", d->n_nodes - nodes_reached,
                  d->n_nodes);
  } else if (d->n_nodes != 0) {
    return fail(RT_ERR_BAD_ARG, "rt_scene_create: an empty world (root = RT_REF_NONE) cannot carry nodes");
  }
  if (!facts.conservative || facts.leaf_refs.size() > LIN_MAX_LEAVES) facts.linear_ok = false;
  std::vector<uint32_t> linear;
  if (facts.linear_ok) linear = facts.leaf_refs;
  for (size_t i = 0; i < complexes.size(); ++i) complexes[i].rank = facts.rank_complex[i];

  // (the description is validated in full before the first HIP call: a bad scene is RT_ERR_BAD_ARG with or without a GPU)
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(RT_ERR_BAD_ARG, "rt_scene_create: device %d of %d", device, ndev);
  HIP_TRY(hipSetDevice(device));

  rt_scene* s = new rt_scene;
  s->device = device;
  if (hipError_t e0 = noalloc_scratch_for(device, &s->static_scratch); e0 != hipSuccess) {
    delete s;
    return fail(RT_ERR_HIP, "rt_scene_create: hipGetSymbolAddress(scratch) on device %d: %s", device, hipGetErrorString(e0));
  }
  s->stack_depth = std::max(tree_height, 1u);
  s->n_nodes = d->n_nodes;
  hipError_t e;
  s->boxes_conservative = facts.conservative;
  for (uint32_t i = 0; i < d->n_cubes; ++i)
    if (d->cube_material[i] < d->n_materials && d->mat_tag[d->cube_material[i]] == RT_MAT_DIELECTRIC) s->glass_cube = true;
  s->safe_t_lo = facts.t_lo;
  s->safe_t_hi = facts.t_hi;
  std::vector<Node2> tree2;  // the ordered walk's tree: built by the f32 side, shared with the f64 side
  uint32_t tree2_root = RT_REF_NONE, tree2_height = 0;
  if ((e = build_side<float>(*d, complexes, linear, facts, s->f32, tree2, tree2_root, tree2_height)) != hipSuccess ||
      (e = build_side<double>(*d, complexes, linear, facts, s->f64, tree2, tree2_root, tree2_height)) != hipSuccess ||
      (e = hipMalloc((void**)&s->dstats, (size_t)rt_scene::RING_CHUNKS * rt_scene::RING_CHUNK_SLOTS * NOALLOC_SLOT_BYTES)) != hipSuccess ||
      (e = hipEventCreate(&s->ev0)) != hipSuccess || (e = hipEventCreate(&s->ev1)) != hipSuccess) {
    delete s;
    return fail(RT_ERR_HIP, "rt_scene_create: upload failed: %s", hipGetErrorString(e));
  }
  *out = s;
  return RT_OK;
}

void rt_scene_destroy(rt_scene* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  delete s;
}

}  // extern "C"

namespace {

int check_camera(const rt_camera* cam) {
  if (!cam) return fail(RT_ERR_BAD_ARG, "null camera");
  // gen_range(time_open..time_close) panics on an empty range (cam.rs:69)
  if (!(cam->time_open < cam->time_close)) return fail(RT_ERR_BAD_ARG, "camera shutter interval is empty (cam.rs:69)");
  return RT_OK;
}

// render_persist_kernel: how many units a wave draws AHEAD of need at each visit of the global counter.  Units held by
// a wave when the counter runs dry are the launch's tail, so only launches with hundreds of units per wave draw ahead
// (measured, f64 Cornell 1024^2: 256 spp = 32 units per wave, prefetch 0 / 1 / 3 / 7: 113 / 118 / 128 / 136 ms; 1024 spp
// = 128 per wave, 0 / 3: 402 / 400 ms).  RT_UNIT_PREFETCH overrides.
static uint32_t persist_prefetch(uint64_t units, uint32_t workgroups) {
  static const char* up = getenv("RT_UNIT_PREFETCH");
  if (up) return (uint32_t)std::min(31, std::max(0, atoi(up)));
  const uint64_t per_wave = units / std::max<uint64_t>(1, 4ull * workgroups);
  return per_wave >= 256 ? 3u : 0u;
}

LaunchShape make_shape(const rt_scene* s, uint32_t w, uint32_t h, uint32_t n_samples, bool f32_fast = false,
                       const rt_camera* cam = nullptr, uint32_t max_depth = 0, bool f64 = false) {
  LaunchShape sh;
  sh.trav = 0;  // the reference's left-first walk: always valid
  // (a scene with a Perlin texture, an extension, renders on the reference-order kernel: see launch_impl.hpp)
  const bool fast_ok = f32_fast && s->boxes_conservative && cam && cam->time_open >= s->safe_t_lo && cam->time_close <= s->safe_t_hi;
  if (!f64 && fast_ok && !s->f32.view.has_noise)
    sh.trav = s->f32.view.n_linear > 0 ? 1 : (s->f32.ordered_root != RT_REF_NONE ? 2 : 0);
  // f64: leaf-list scenes walk the leaf list too (traverse_leaves: same winners, t and draws as the reference-order
  // walk, no node tests); RT_FLAG_FORCE_BVH / RT_F64_NO_LEAVES=1 keep the reference-order walk (A/B, work counting)
  static const bool f64_no_leaves = getenv("RT_F64_NO_LEAVES") != nullptr;
  const bool f64_fast_ok = f64 && fast_ok && !s->glass_cube;  // (see rt_scene::glass_cube)
  if (f64_fast_ok && !f64_no_leaves && !s->f64.view.has_noise && s->f64.view.n_leaf_list > 0 && max_depth > 0 && n_samples > 0)
    sh.trav = 3;
  // f64, larger scenes: the ordered walk in f64 (traverse_ordered, persistent form only); RT_F64_NO_ORDERED=1 (A/B)
  static const bool f64_no_ordered = getenv("RT_F64_NO_ORDERED") != nullptr;
  const bool ordered64 = f64_fast_ok && sh.trav == 0 && !f64_no_ordered && !s->f64.view.has_noise &&
                         s->f64.ordered_root != RT_REF_NONE && max_depth > 0 && n_samples > 0;
  if (ordered64) sh.trav = 2;
  uint32_t tx = (w + 15) / 16, ty = (h + 15) / 16;
  sh.tiles = tx * ty;
  sh.stack_depth = sh.trav == 2 ? std::max(f64 ? s->f64.ordered_height : s->f32.ordered_height, 1u) : s->stack_depth;
  // ordered walk: the longer a scene's walks, the smaller the idle fraction worth interrupting them for
  // (measured optimum, re-tuned on the final kernels: 6/8 at ~500-1500 nodes, 4/8 at 20 k, 3/8 at 1 M)
  static const char* re = getenv("RT_REFILL_EIGHTHS");  // tuning override
  const uint32_t n_nodes = s->n_nodes;
  sh.refill_eighths = re ? (uint32_t)std::min(8, std::max(1, atoi(re))) : n_nodes <= 4096 ? 6u : n_nodes <= 65536 ? 4u : 3u;
  // leaf phase once 1/2 (small trees, and again at 10^6 leaves) or 1/4 (in between) of the walking lanes wait at a
  // leaf — measured: shift 1 is +2 % on random_scene / final_scene / 1 M spheres but -8 % at 20 k
  static const char* ls = getenv("RT_LEAF_SHIFT");  // tuning override
  sh.leaf_shift = ls ? (uint32_t)std::min(5, std::max(0, atoi(ls))) : (n_nodes <= 4096 || n_nodes > 65536) ? 1u : 2u;
  if (ordered64) {
    // the f64 ordered walk inside the persistent kernel (refill threshold x leaf-phase threshold, ms on C1 / C3 / C4 f64
    // samples, tools/f64_ordered_tune.sh: 8/8 = run to completion 8.9 / 61.1 / 12.0; 6/8, shift 1: 7.8 / 53.7 / 8.5;
    // 4/8, shift 1: 8.1 / 58.1 / 8.4; shift 0 or 2: 3-70 % slower)
    if (!re) sh.refill_eighths = 6u;
    if (!ls) sh.leaf_shift = 1u;
  }
  static const char* xc = getenv("RT_XCD_CHUNK");
  sh.xcd_chunk = xc ? (uint32_t)std::max(0, atoi(xc)) : 1u;
  static const char* tb = getenv("RT_TILE_BAND");
  sh.tile_band = tb ? (uint32_t)std::max(1, atoi(tb)) : 1u;  // measured: bands of 2..16 rows are no better than rows
  uint32_t waves = sh.tiles * 4;
  static const char* tw = getenv("RT_TARGET_WAVES");  // tuning override
  // The refill kernel (ordered walk) hands a wave's jobs to whichever lane is free, so its waves end together whatever
  // their size, and what is left to balance is the uneven cost of image regions ACROSS waves: many short waves
  // (measured on final_scene 1024^2 x 512 spp: 24 k waves 6.4, 98 k 8.0, 197 k 8.3, 393 k 8.1, 1.5 M 6.3 Grays/s), but
  // no fewer than 6 samples per wave and split (random_scene 400 x 225 x 100 spp: 16 splits 7.0, 64 splits 6.0 Grays/s)
  // The reference-order kernel (lane = pixel; what scenes with non-conservative boxes, RT_FLAG_FORCE_BVH and the f64 audit
  // run on) wants FEWER, longer workgroups: about four rounds of the chip's wave slots, one split once the image alone
  // has that many waves, and otherwise a multiple of 8 (below), so that its workgroups are dealt to the XCDs (measured,
  // ms: simple_light 400 x 267 x 100 spp: 8 splits 1.29, 16: 1.39, 32: 1.61, 64: 1.90; Cornell forced onto it, 256^2 x
  // 256 spp: 8: 2.71, 16: 2.40, 32: 2.42, 64: 2.70; 1920 x 1080 x 64 spp: 1: 8.7, 2: 8.6, 8: 9.5, 16: 11.3; 3840 x 2160 x
  // 64 spp: 1: 30.4, 8: 36.9; the reference's default job 1200 x 800 x 500 spp: 1: 40.4, 2: 37.0, 4: 35.6, 8: 35.8,
  // 16: 37.2; f64 Cornell 1024^2 x 64 spp: 1: 216, 4: 199, 8: 182, 16: 184)
  const uint32_t target = tw ? (uint32_t)std::max(1, atoi(tw)) : (sh.trav == 2 ? 2u * TARGET_WAVES : sh.trav == 0 ? 24u * 1024u : TARGET_WAVES);
  uint32_t splits = waves >= target ? 1 : (target + waves - 1) / waves;
  if (sh.trav == 2) splits = std::min(splits, std::max(1u, n_samples / 6u));
  static const char* fs = getenv("RT_SPLITS");  // tuning override
  static const bool grid_all = getenv("RT_XCD_ALL_OFF") == nullptr;  // A/B: the same for the reference-order kernels
  if (((sh.trav == 2 && splits >= 5) || (grid_all && sh.trav == 0 && splits >= 2)) && sh.xcd_chunk == 1) {
    // a multiple of 8, so that block_slot can give every XCD whole splits (mode 1)
    const uint32_t up = (splits + 7u) / 8u * 8u;
    if (up <= std::max(1u, n_samples / (sh.trav == 0 ? 2u : 4u))) {
      splits = up;
    } else if (sh.trav == 2) {
      splits = splits / 8u * 8u;
      if (splits == 0) splits = std::min(8u, n_samples);
    }  // (reference-order kernel with few samples: as computed, workgroups in grid order)
  }
  if (fs) splits = (uint32_t)std::max(1, atoi(fs));
  sh.splits = std::max(1u, std::min(splits, n_samples));
  // Leaf-list scenes: persistent workers drawing (8x8 tile, sample range) units from a global counter
  // (render_persist_kernel).  Off for max_depth 0 (handled by render_kernel) and with RT_LINEAR_NO_PERSIST (A/B).
  static const bool no_persist = getenv("RT_LINEAR_NO_PERSIST") != nullptr || getenv("RT_LINEAR_SORTED") != nullptr;
  static const char* us = getenv("RT_UNIT_SAMPLES");
  sh.unit_prefetch = 0;  // set below, once the launch's unit count is known
  if (sh.trav == 1 && !no_persist && max_depth > 0) {
    // unit size: large enough to amortise the hand-over (measured optimum 256 at 4096 spp, 64 at 256 spp),
    // small enough that the launch has many units per worker
    const uint32_t auto_unit = std::min(256u, std::max(16u, n_samples / 4));
    sh.unit_samples = std::max(1u, std::min(us ? (uint32_t)std::max(1, atoi(us)) : auto_unit, n_samples));
    sh.wave_tiles_x = (w + 7) / 8;
    sh.wave_tiles = sh.wave_tiles_x * ((h + 7) / 8);
    const uint64_t units = (uint64_t)sh.wave_tiles * ((n_samples + sh.unit_samples - 1) / sh.unit_samples);
    if (units <= 0xFFF00000ull && n_samples > 0) {  // (every wave overshoots the counter once: keep clear of 2^32)
      sh.total_units = (uint32_t)units;
      sh.tiles = (uint32_t)std::min<uint64_t>((units + 3) / 4, 256u * 8u);  // workgroups: at most 8 per CU can be resident
      sh.splits = 1;
      sh.unit_prefetch = persist_prefetch(units, sh.tiles);
    } else {
      sh.unit_samples = 0;
    }
  }
  // The f64 audit kernel (reference-order walk, paths of 1..50 rays walked to their end): the same persistent form
  // (render_persist_kernel<double, ..., TRAV_REFERENCE>) — Cornell 1024^2 x 64 spp 82.7 -> 52.1 ms, x 256 spp 223.7 ->
  // 165 ms (units of 2 / 4 / 8 / 16 / 32 / 64 samples: 169 / 165 / 166 / 167 / 179 / 193 ms).  RT_REF_GRID keeps one
  // workgroup per (tile, split) (A/B).  Not for the f32 reference-order kernel: its jobs are short-path scenes
  // (the reference's default job, depth 5: 35.9 ms as a grid, 38.2 ms at best in this form).
  static const bool ref_grid = getenv("RT_REF_GRID") != nullptr;
  // The f32 reference-order kernel (scenes whose boxes clip — the reference's default scene —, RT_FLAG_FORCE_BVH) takes the
  // queue form too (render_queue_kernel<float, ..., TRAV_REFERENCE>) from 16 samples per launch on: the reference's default
  // job (simple_light 1200 x 800 x 500 spp, depth 5) 35.5 -> 34.1 ms, simple_light 400 x 267 x 100 spp 1.3 -> 1.0 ms, forced
  // walks of Cornell 1024^2 x 256 spp 26.8 -> 23.2 ms, random_scene C1 13.7 -> 10.7 ms, final_scene 512^2 x 16 spp 12.2 -> 11.0 ms;
  // at 4 spp (Cornell 3840 x 2160) the grid form wins, 3.0 vs 3.5 ms.  Not the instantiations that carry the extension
  // code or the instance arm (they stay one workgroup per tile).  RT_QUEUE_REF32=0: the grid form (A/B).
  static const bool queue_ref32 = getenv("RT_QUEUE_REF32") == nullptr || atoi(getenv("RT_QUEUE_REF32")) != 0;
  const bool ref32_units = queue_ref32 && !f64 && sh.trav == 0 && n_samples >= 16 && !s->f32.view.has_noise &&
                           !s->f32.view.has_instances;
  if ((((sh.trav == 0 || sh.trav == 3 || ordered64) && f64 && (!ref_grid || sh.trav != 0)) || ref32_units) && max_depth > 0 && n_samples > 0) {
    // (leaf-list form at the metric's 4096 spp, units of 16 / 32 / 64 / 128 / 256 samples: 1511 / 1518 / 1545 / 1603 / 1751 ms)
    const uint32_t auto_unit = sh.trav == 3 ? std::min(16u, std::max(2u, n_samples / 32)) : std::min(8u, std::max(2u, n_samples / 16));
    sh.unit_samples = std::max(1u, std::min(us ? (uint32_t)std::max(1, atoi(us)) : auto_unit, n_samples));
    sh.wave_tiles_x = (w + 7) / 8;
    sh.wave_tiles = sh.wave_tiles_x * ((h + 7) / 8);
    const uint64_t units = (uint64_t)sh.wave_tiles * ((n_samples + sh.unit_samples - 1) / sh.unit_samples);
    if (units <= 0xFFF00000ull) {
      sh.total_units = (uint32_t)units;
      sh.tiles = (uint32_t)std::min<uint64_t>((units + 3) / 4, 256u * 8u);
      sh.splits = 1;
      sh.unit_prefetch = persist_prefetch(units, sh.tiles);
    } else {
      sh.unit_samples = 0;
      sh.trav = 0;  // (the leaf-list and ordered f64 forms exist as persistent kernels only)
      sh.stack_depth = s->stack_depth;
    }
  }
  // Ordered-walk scenes: the same, with the refill kernel's job sharing (render_units_kernel).  RT_REFILL_GRID keeps
  // the one-workgroup-per-(tile, split) form (A/B).
  static const bool refill_grid = getenv("RT_REFILL_GRID") != nullptr || getenv("RT_ORDERED_NO_REFILL") != nullptr;
  if (sh.trav == 2 && !f64 && !refill_grid && max_depth > 0 && n_samples > 0) {
    // unit size: 8 samples of a tile amortise a unit's hand-over and flush (final_scene 1024^2 x 512 spp: 2 -> 12.3,
    // 4 -> 12.7, 8 -> 12.8, 32 -> 12.2 Grays/s), but a short launch needs >= 16 units per wave slot of the chip to end
    // evenly (random_scene 400 x 225 x 100 spp: 2 -> 8.8, 4 -> 8.3, 8 -> 7.6 Grays/s)
    const uint64_t jobs = (uint64_t)w * h * n_samples;
    const uint32_t auto_unit = (uint32_t)std::min<uint64_t>(8, std::max<uint64_t>(2, jobs / (5120ull * 64ull * 16ull)));
    sh.unit_samples = std::max(1u, std::min(us ? (uint32_t)std::max(1, atoi(us)) : auto_unit, n_samples));
    sh.wave_tiles_x = (w + 15) / 16;  // 16x16 pixel blocks per row; a unit's tile is a quarter of one
    sh.wave_tiles = 4u * sh.wave_tiles_x * ((h + 15) / 16);
    const uint64_t units = (uint64_t)sh.wave_tiles * ((n_samples + sh.unit_samples - 1) / sh.unit_samples);
    if (units <= 0xFFF00000ull) {
      sh.total_units = (uint32_t)units;
      sh.tiles = (uint32_t)std::min<uint64_t>((units + 3) / 4, 256u * 8u);
      sh.splits = 1;
      // LDS per workgroup: the stacks (6 B per entry and lane), the lights, and 6 KiB of accumulators — which are
      // dropped where they would cost a resident workgroup per CU (160 KiB of LDS, at most 5 workgroups by registers)
      static const char* da = getenv("RT_UNITS_DIRECT");  // tuning override
      const size_t stack_bytes = (size_t)sh.stack_depth * 256u * 6u, fixed = 512;
      const size_t with_acc = std::min<size_t>(5, (160u << 10) / (stack_bytes + fixed + 4u * (2u * 192u + 2u) * 4u));
      const size_t without = std::min<size_t>(5, (160u << 10) / (stack_bytes + fixed));
      sh.direct_accum = da ? (atoi(da) != 0) : (without > with_acc);
    } else {
      sh.unit_samples = 0;
    }
  }
  return sh;
}

template <typename R> struct Side;
template <> struct Side<float> {
  static SceneSide<float>& get(rt_scene* s) { return s->f32; }
  static hipError_t render(const SceneView<float>& sc, const CameraView<float>& cam, uint32_t w, uint32_t h, uint32_t sb,
                           uint32_t ns, uint32_t md, uint64_t seed, int acc, bool st, float* a, DeviceStats* ds,
                           LaunchShape sh, hipStream_t str) {
    return launch_render_f32(sc, cam, w, h, sb, ns, md, seed, acc, st, a, ds, sh, str);
  }
  static hipError_t primary(const SceneView<float>& sc, const CameraView<float>& cam, uint32_t w, uint32_t h,
                            uint64_t seed, float time, uint32_t* id, float* t, float* n, float* uv, uint8_t* ff,
                            LaunchShape sh, hipStream_t str) {
    return launch_primary_f32(sc, cam, w, h, seed, time, id, t, n, uv, ff, sh, str);
  }
};
template <> struct Side<double> {
  static SceneSide<double>& get(rt_scene* s) { return s->f64; }
  static hipError_t render(const SceneView<double>& sc, const CameraView<double>& cam, uint32_t w, uint32_t h,
                           uint32_t sb, uint32_t ns, uint32_t md, uint64_t seed, int acc, bool st, double* a,
                           DeviceStats* ds, LaunchShape sh, hipStream_t str) {
    return launch_render_f64(sc, cam, w, h, sb, ns, md, seed, acc, st, a, ds, sh, str);
  }
  static hipError_t primary(const SceneView<double>& sc, const CameraView<double>& cam, uint32_t w, uint32_t h,
                            uint64_t seed, double time, uint32_t* id, double* t, double* n, double* uv, uint8_t* ff,
                            LaunchShape sh, hipStream_t str) {
    return launch_primary_f64(sc, cam, w, h, seed, time, id, t, n, uv, ff, sh, str);
  }
};

void stats_to_abi(const DeviceStats& d, rt_stats* o) {
  o->rays = d.v[ST_RAYS];
  o->light_probes = d.v[ST_LIGHT_PROBES];
  o->medium_probes = d.v[ST_MEDIUM_PROBES];
  o->nodes_visited = d.v[ST_NODES];
  o->sphere_tests = d.v[ST_SPHERE_TESTS];
  o->sphere_hits = d.v[ST_SPHERE_HITS];
  o->cube_tests = d.v[ST_CUBE_TESTS];
  o->cube_hits = d.v[ST_CUBE_HITS];
  o->rotate_visits = d.v[ST_ROTATE];
  o->translate_visits = d.v[ST_TRANSLATE];
  o->medium_visits = d.v[ST_MEDIUM];
  o->closest_hits = d.v[ST_CLOSEST];
  o->diffuse_vertices = d.v[ST_DIFFUSE];
  o->specular_vertices = d.v[ST_SPECULAR];
  o->emitter_hits = d.v[ST_EMITTER];
  o->misses = d.v[ST_MISSES];
  o->nonfinite_samples = d.v[ST_NONFINITE];
}

// rt_params checks shared by every render entry (hoisted so that rt_render_multi can refuse a bad request before it
// builds G scenes): u = (x + xi) / (w - 1) needs w, h >= 2 (main.rs:161-163); pixel and sample indices are 32-bit.
int check_params(const rt_params* p, const char* who) {
  if (!p) return fail(RT_ERR_BAD_ARG, "%s: null params", who);
  if (p->width < 2 || p->height < 2)
    return fail(RT_ERR_BAD_ARG, "%s: image must be at least 2x2 (main.rs:161-163 divides by w-1, h-1)", who);
  if ((uint64_t)p->width * p->height > 0xFFFFFFFFull / 3) return fail(RT_ERR_BAD_ARG, "%s: image too large", who);
  // the persistent kernels pack a unit's tile origin as x | y << 16
  if (p->width > 65535u || p->height > 65535u)
    return fail(RT_ERR_BAD_ARG, "%s: image width and height must not exceed 65535 pixels", who);
  if (p->sample_offset + p->spp > 0xFFFFFFFFull) return fail(RT_ERR_BAD_ARG, "%s: sample index exceeds 32 bits", who);
  return RT_OK;
}

// The wavefront form of one render call (device/wavefront.hpp): BVH-bound scenes, f32.  Synchronises the stream every few
// iterations to read how many pool slots have retired (the host decides when the pool has drained).
int render_wavefront(rt_scene* s, const rt_camera* cam, const rt_params* p, float* d_accum, int accumulate, hipStream_t stream,
                     rt_stats* stats, const LaunchShape& shape) {
  DeviceStats* const dstats = reinterpret_cast<DeviceStats*>(s->next_chunk());
  const bool want_stats = (p->flags & RT_FLAG_STATS) != 0;
  static const char* slots_env = getenv("RT_WF_SLOTS");
  const uint32_t max_slots = slots_env ? (uint32_t)std::max(256, atoi(slots_env)) : (1u << 21);
  WfArgs a{};
  a.width = p->width;
  a.height = p->height;
  a.sample_begin = (uint32_t)p->sample_offset;
  a.n_samples = p->spp;
  a.max_depth = p->max_depth;
  a.tiles_x = (p->width + 7) / 8;
  a.n_tiles = a.tiles_x * ((p->height + 7) / 8);
  a.total_jobs = 64ull * a.n_tiles * p->spp;
  a.stack_depth = shape.stack_depth;
  static const char* re = getenv("RT_WF_REFILL_EIGHTHS");
  a.refill_eighths = re ? (uint32_t)std::min(8, std::max(1, atoi(re))) : 2u;
  a.leaf_shift = shape.leaf_shift;
  a.has_media = 1u;  // (cheap: 20 bytes more per fetched ray; every scene may hold a medium)
  a.seed = p->seed;
  const uint32_t n_slots = (uint32_t)std::min<uint64_t>(max_slots, (a.total_jobs + 255) / 256 * 256);
  a.n_blocks = (uint64_t)a.n_tiles * p->spp;
  if (s->wf_capacity < n_slots) {
    if (s->wf_mem) (void)hipFree(s->wf_mem);
    s->wf_mem = nullptr;
    s->wf_capacity = 0;
    const size_t per_slot = 4 * sizeof(float4) + 5 * sizeof(uint32_t);  // ray_o, ray_d, thr_l, l_ids; hit, slot, list[2], group_next (1/64 used)
    HIP_TRY(hipMalloc(&s->wf_mem, std::max<size_t>((size_t)n_slots * per_slot, 256)));
    s->wf_capacity = n_slots;
  }
  if (!s->wf_ctr) HIP_TRY(hipMalloc((void**)&s->wf_ctr, sizeof(WfCounters)));
  if (!s->wf_host) HIP_TRY(hipHostMalloc((void**)&s->wf_host, 64));
  const size_t lds = (size_t)a.stack_depth * 256 * 6;
  if (!s->wf_trace_blocks) {
    int per_cu = 0, cus = 0;
    HIP_TRY(wf_trace_blocks_per_cu(lds, &per_cu));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s->device));
    s->wf_trace_blocks = (uint32_t)std::max(1, per_cu) * (uint32_t)std::max(1, cus);
  }
  WfPool pool{};
  {
    unsigned char* base = (unsigned char*)s->wf_mem;
    const size_t cap = s->wf_capacity;
    pool.ray_o = (float4*)base;
    pool.ray_d = pool.ray_o + cap;
    pool.thr_l = pool.ray_d + cap;
    pool.l_ids = pool.thr_l + cap;
    pool.hit = (uint32_t*)(pool.l_ids + cap);
    pool.slot = pool.hit + cap;
    pool.list[0] = pool.slot + cap;
    pool.list[1] = pool.list[0] + cap;
    pool.n_slots = n_slots;
  }
  const CameraView<float> cv = to_camera<float>(*cam);
  const size_t accum_bytes = (size_t)p->width * p->height * 3 * sizeof(float);
  HIP_TRY(hipMemsetAsync(dstats, 0, sizeof(DeviceStats), stream));
  HIP_TRY(hipMemsetAsync(s->wf_ctr, 0, sizeof(WfCounters), stream));
  if (!accumulate) HIP_TRY(hipMemsetAsync(d_accum, 0, accum_bytes, stream));
  HIP_TRY(hipEventRecord(s->ev0, stream));
  HIP_TRY(launch_wf_init(cv, a, pool, s->wf_ctr, stream));
  uint32_t iter = 0;
  const uint32_t trace_blocks = (n_slots + 4 * WF_TRACE_ENTRIES - 1) / (4 * WF_TRACE_ENTRIES);
  static const char* ce = getenv("RT_WF_CHECK_EVERY");
  const uint32_t check_every = ce ? (uint32_t)std::max(1, atoi(ce)) : 8u;
  // TAIL mode: once slots retire (their group's jobs are spent), the shade kernel writes the list of the slots still
  // alive and the kernels walk that list instead of all slots
  bool tail = false;
  for (;;) {
    for (uint32_t k = 0; k < check_every; ++k, ++iter) {
      a.in_list = a.out_list;  // (a list exists once an iteration has written it)
      a.out_list = tail ? 1u : 0u;
      HIP_TRY(launch_wf_iteration(s->f32.view, cv, a, pool, s->wf_ctr, iter & 1u, d_accum, dstats, want_stats, trace_blocks,
                                  stream));
    }
    HIP_TRY(hipMemcpyAsync(s->wf_host, &s->wf_ctr->retired, sizeof(unsigned int), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (*s->wf_host >= n_slots) break;
    if (*s->wf_host > 0) tail = true;
    if (iter > (1u << 26)) return fail(RT_ERR_HIP, "rt_render: the wavefront pool did not drain");
  }
  HIP_TRY(hipEventRecord(s->ev1, stream));
  if (stats) {
    DeviceStats hs;
    HIP_TRY(hipMemcpyAsync(&hs, dstats, sizeof(hs), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    std::memset(stats, 0, sizeof(*stats));
    stats_to_abi(hs, stats);
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    stats->kernel_ms = ms;
    stats->launches = 1 + 2 * iter;
  }
  return RT_OK;
}

// Enqueue the launches of one render call on `stream`; accum is device memory.
// Counter blocks (DeviceStats, one 256-byte slot each).  The ordinary entries take the scene's next ring chunk
// (rt_scene::next_chunk: RING_CHUNK_SLOTS slots of this call's own); rt_render_device_noalloc passes its partition of
// the file-scope scratch (`noalloc`, n_slots = RT_NOALLOC_MAX_LAUNCHES) and is refused if it needs more launches than
// that.  Launch i uses slot i % n_slots; all slots are zeroed by one kernel up front, a slot that is used again has its
// unit counter re-zeroed first (launches of one call are ordered on the stream), and the work counters — which simply
// keep adding up in a reused slot — are summed over the slots on the host.
template <typename R>
int render_on_device(rt_scene* s, const rt_camera* cam, const rt_params* p, R* d_accum, int accumulate,
                     hipStream_t stream, rt_stats* stats, unsigned char* noalloc = nullptr, uint32_t n_slots = 0) {
  if (!s || !p || !d_accum) return fail(RT_ERR_BAD_ARG, "rt_render: null argument");
  if (int rc = check_camera(cam)) return rc;
  if (int rc = check_params(p, "rt_render")) return rc;
  // One launch renders up to 4096 samples of every pixel: a lane keeps regenerating samples in its loop, so
  // long launches amortise the wave start-up and the tail in which the lanes of a wave run out of
  // samples one by one (measured on Cornell 1024^2 x 4096 spp: 64 per launch 40.3, 256: 43.1, 4096: 45.2 Grays/s).
  const uint32_t per_launch = p->spp_per_launch ? p->spp_per_launch : 4096u;
  const uint32_t n_launches = (p->spp + per_launch - 1) / per_launch;
  if (noalloc && n_launches > n_slots)
    return fail(RT_ERR_BAD_ARG,
                "rt_render_device_noalloc: %u spp at %u per launch is %u launches, the static scratch holds %u: raise "
                "spp_per_launch or split the call",
                p->spp, per_launch, n_launches, n_slots);
  HIP_TRY(hipSetDevice(s->device));
  const bool want_stats = (p->flags & RT_FLAG_STATS) != 0;
  if constexpr (sizeof(R) == 4) {
    // BVH-bound scenes: the wavefront form, opt-in (RT_FLAG_WAVEFRONT, or RT_WAVEFRONT=1 in the environment); never for
    // the entry that must not allocate
    static const bool wavefront_env = getenv("RT_WAVEFRONT") != nullptr;
    const bool wavefront = wavefront_env || (p->flags & RT_FLAG_WAVEFRONT) != 0;
    if (wavefront && !noalloc && p->spp > 0 && !(p->flags & RT_FLAG_FORCE_BVH)) {
      const LaunchShape sh0 = make_shape(s, p->width, p->height, p->spp, true, cam, p->max_depth);
      // (its job cursors are 32-bit, one per 1/64 of the jobs: a shard hands out about n_tiles x spp of them — beyond
      // that the call stays on the megakernel)
      const uint64_t shard_jobs = (uint64_t)((p->width + 7) / 8) * ((p->height + 7) / 8) * p->spp;
      if (sh0.trav == 2 && shard_jobs < 0xFFF00000ull)
        return render_wavefront(s, cam, p, (float*)d_accum, accumulate, stream, stats, sh0);
    }
  }
  unsigned char* const blocks = noalloc ? noalloc : s->next_chunk();
  if (!noalloc) n_slots = rt_scene::RING_CHUNK_SLOTS;
  const uint32_t used = std::max(1u, std::min(n_launches, n_slots));
  auto block = [&](uint32_t i) { return reinterpret_cast<DeviceStats*>(blocks + (size_t)(i % used) * NOALLOC_SLOT_BYTES); };
  CameraView<R> cv = to_camera<R>(*cam);
  const SceneView<R>& sv = Side<R>::get(s).view;
  static_assert(sizeof(DeviceStats) % 4 == 0 && NOALLOC_SLOT_BYTES % 4 == 0, "cleared in dwords");
  HIP_TRY(launch_zero(blocks, (size_t)used * NOALLOC_SLOT_BYTES, stream));
  // Timing events only when somebody will read them: without stats the call is kernels only — what a caller that
  // captures it into a HIP graph wants to see on its stream.  The ordinary entries create this call's own pair (two
  // calls on one scene may be timed at once); the entry that must not allocate uses the scene's.
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  struct EventPair {
    hipEvent_t *a, *b;
    bool owned;
    ~EventPair() {
      if (owned && *a) (void)hipEventDestroy(*a);
      if (owned && *b) (void)hipEventDestroy(*b);
    }
  } pair{&ev0, &ev1, false};
  if (stats) {
    if (noalloc) {
      ev0 = s->ev0;
      ev1 = s->ev1;
    } else {
      pair.owned = true;
      HIP_TRY(hipEventCreate(&ev0));
      HIP_TRY(hipEventCreate(&ev1));
    }
    HIP_TRY(hipEventRecord(ev0, stream));
  }
  uint32_t done = 0, launches = 0;
  bool first = true;
  const size_t accum_bytes = (size_t)p->width * p->height * 3 * sizeof(R);
  if (p->spp == 0 && !accumulate) HIP_TRY(launch_zero(d_accum, accum_bytes, stream));
  while (done < p->spp) {
    uint32_t n = std::min(per_launch, p->spp - done);
    LaunchShape sh = make_shape(s, p->width, p->height, n, !(p->flags & RT_FLAG_FORCE_BVH), cam, p->max_depth, sizeof(R) == 8);
    int acc = (accumulate || !first) ? 1 : 0;
    if ((sh.splits > 1 || sh.unit_samples > 0) && !acc) HIP_TRY(launch_zero(d_accum, accum_bytes, stream));
    if (sh.unit_samples > 0 && launches >= used) HIP_TRY(launch_zero(&block(launches)->unit_counter, sizeof(unsigned int), stream));
    HIP_TRY(Side<R>::render(sv, cv, p->width, p->height, (uint32_t)(p->sample_offset + done), n, p->max_depth, p->seed,
                            acc, want_stats, d_accum, block(launches), sh, stream));
    done += n;
    first = false;
    ++launches;
  }
  if (stats) {
    HIP_TRY(hipEventRecord(ev1, stream));
    DeviceStats hs;
    std::memset(&hs, 0, sizeof(hs));
    static_assert(NOALLOC_SLOT_BYTES * RT_NOALLOC_MAX_LAUNCHES <= 64 * 1024, "the slot image is read back through the stack");
    alignas(16) unsigned char image[NOALLOC_SLOT_BYTES * RT_NOALLOC_MAX_LAUNCHES];
    static_assert(rt_scene::RING_CHUNK_SLOTS <= RT_NOALLOC_MAX_LAUNCHES, "a ring chunk fits the read-back image");
    HIP_TRY(hipMemcpyAsync(image, blocks, (size_t)used * NOALLOC_SLOT_BYTES, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    for (uint32_t i = 0; i < used; ++i) {
      DeviceStats one;
      std::memcpy(&one, image + (size_t)i * NOALLOC_SLOT_BYTES, sizeof(one));
      for (int k = 0; k < 17; ++k) hs.v[k] += one.v[k];
      for (int k = 0; k < 8; ++k) hs.stamp[k] += one.stamp[k];
    }
    std::memset(stats, 0, sizeof(*stats));
    stats_to_abi(hs, stats);
    if (getenv("RT_PRINT_STAMPS")) {
      fprintf(stderr, "stamps:");
      for (int i = 0; i < 8; ++i) fprintf(stderr, " %llu", hs.stamp[i]);
      fprintf(stderr, "\n");
    }
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    stats->kernel_ms = ms;
    stats->launches = launches;
  }
  return RT_OK;
}

template <typename R>
int render_to_host(rt_scene* s, const rt_camera* cam, const rt_params* p, R* h_accum, rt_stats* stats) {
  if (!s || !p || !h_accum) return fail(RT_ERR_BAD_ARG, "rt_render: null argument");
  if (int rc = check_params(p, "rt_render")) return rc;  // before the accumulator is sized from them
  HIP_TRY(hipSetDevice(s->device));
  size_t bytes = (size_t)p->width * p->height * 3 * sizeof(R);
  R* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, std::max<size_t>(bytes, 16)));
  rt_stats local;
  int rc = render_on_device<R>(s, cam, p, d, 0, nullptr, stats ? stats : &local);
  if (rc == RT_OK) {
    hipError_t e = hipMemcpy(h_accum, d, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(RT_ERR_HIP, "rt_render: copy back failed: %s", hipGetErrorString(e));
  }
  (void)hipFree(d);
  return rc;
}

template <typename R>
int primary_to_host(rt_scene* s, const rt_camera* cam, uint32_t w, uint32_t h, uint64_t seed, double time,
                    uint32_t* prim_id, R* t, R* normal, R* uv, uint8_t* front_face, bool fast = false) {
  if (!s || !cam || !prim_id || !t || !normal) return fail(RT_ERR_BAD_ARG, "rt_primary_hits: null argument");
  if (w < 2 || h < 2) return fail(RT_ERR_BAD_ARG, "rt_primary_hits: image must be at least 2x2");
  if ((uint64_t)w * h > 0xFFFFFFFFull / 3) return fail(RT_ERR_BAD_ARG, "rt_primary_hits: image too large");
  HIP_TRY(hipSetDevice(s->device));
  size_t n = (size_t)w * h;
  DeviceBuffers tmp;
  auto alloc = [&](size_t bytes, void** out) -> hipError_t {
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipSuccess) tmp.ptrs.push_back(*out);
    return e;
  };
  uint32_t* d_id = nullptr;
  R *d_t = nullptr, *d_n = nullptr, *d_uv = nullptr;
  uint8_t *d_ff = nullptr, *d_unsure = nullptr;
  HIP_TRY(alloc(n * sizeof(uint32_t), (void**)&d_id));
  HIP_TRY(alloc(n * sizeof(R), (void**)&d_t));
  HIP_TRY(alloc(n * 3 * sizeof(R), (void**)&d_n));
  if (uv) HIP_TRY(alloc(n * 2 * sizeof(R), (void**)&d_uv));
  if (front_face) HIP_TRY(alloc(n, (void**)&d_ff));
  LaunchShape sh = make_shape(s, w, h, 1, fast && sizeof(R) == 4, cam, 0);
  sh.tiles = ((w + 15) / 16) * ((h + 15) / 16);  // the primary kernel is tile-per-workgroup whatever the render shape
  sh.ref_stack_depth = s->stack_depth;
  if (fast && (time < s->safe_t_lo || time > s->safe_t_hi)) {
    sh.trav = 0;
    sh.stack_depth = s->stack_depth;
  }
  if constexpr (sizeof(R) == 4) {
    // f32: the product walk, audited (device/audit.hpp); the pixels it marks are re-decided by the f64 audit kernel
    HIP_TRY(alloc(n, (void**)&d_unsure));
    const CameraView<double> cam64 = to_camera<double>(*cam);
    HIP_TRY(launch_primary_audited_f32(s->f32.view, s->f64.view, cam64, w, h, seed, time, d_id, (float*)d_t, (float*)d_n,
                                       (float*)d_uv, d_ff, d_unsure, sh, nullptr));
    HIP_TRY(launch_primary_fix_f64(s->f64.view, cam64, w, h, seed, time, d_unsure, d_id, (float*)d_t, (float*)d_n, (float*)d_uv,
                                   d_ff, s->stack_depth, nullptr));
  } else {
    HIP_TRY(Side<R>::primary(Side<R>::get(s).view, to_camera<R>(*cam), w, h, seed, (R)time, d_id, d_t, d_n, d_uv, d_ff, sh,
                             nullptr));
  }
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(prim_id, d_id, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(t, d_t, n * sizeof(R), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(normal, d_n, n * 3 * sizeof(R), hipMemcpyDeviceToHost));
  if (uv) HIP_TRY(hipMemcpy(uv, d_uv, n * 2 * sizeof(R), hipMemcpyDeviceToHost));
  if (front_face) HIP_TRY(hipMemcpy(front_face, d_ff, n, hipMemcpyDeviceToHost));
  if (d_unsure) {
    std::vector<uint8_t> flags(n);
    HIP_TRY(hipMemcpy(flags.data(), d_unsure, n, hipMemcpyDeviceToHost));
    uint64_t c = 0, why[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint8_t f : flags) {
      c += f != 0;
      for (int b = 0; b < 8; ++b) why[b] += (f >> b) & 1;
    }
    s->last_redecided = c;
    if (getenv("RT_AUDIT_REASONS"))  // diagnostic: which kind of decision sent pixels to the f64 kernel (device/audit.hpp AUD_*)
      fprintf(stderr, "audit: %llu of %zu re-decided; node %llu cube %llu sphere-hit %llu sphere-window %llu medium %llu tie %llu accuracy %llu direction %llu\n",
              (unsigned long long)c, n, (unsigned long long)why[0], (unsigned long long)why[1], (unsigned long long)why[2],
              (unsigned long long)why[3], (unsigned long long)why[4], (unsigned long long)why[5], (unsigned long long)why[6],
              (unsigned long long)why[7]);
  }
  return RT_OK;
}

}  // namespace

extern "C" {

int rt_primary_hits(rt_scene* s, const rt_camera* cam, uint32_t w, uint32_t h, uint64_t seed, double time,
                    uint32_t* prim_id, float* t, float* normal, float* uv, uint8_t* front_face) {
  return primary_to_host<float>(s, cam, w, h, seed, time, prim_id, t, normal, uv, front_face);
}
int rt_primary_hits_f64(rt_scene* s, const rt_camera* cam, uint32_t w, uint32_t h, uint64_t seed, double time,
                        uint32_t* prim_id, double* t, double* normal, double* uv, uint8_t* front_face) {
  return primary_to_host<double>(s, cam, w, h, seed, time, prim_id, t, normal, uv, front_face);
}

int rt_primary_hits_fast(rt_scene* s, const rt_camera* cam, uint32_t w, uint32_t h, uint64_t seed, double time,
                         uint32_t* prim_id, float* t, float* normal, float* uv, uint8_t* front_face) {
  return primary_to_host<float>(s, cam, w, h, seed, time, prim_id, t, normal, uv, front_face, true);
}

uint64_t rt_primary_redecided(const rt_scene* s) { return s ? s->last_redecided : 0; }

int rt_render(rt_scene* s, const rt_camera* cam, const rt_params* p, float* accum, rt_stats* stats) {
  return render_to_host<float>(s, cam, p, accum, stats);
}
int rt_render_f64(rt_scene* s, const rt_camera* cam, const rt_params* p, double* accum, rt_stats* stats) {
  return render_to_host<double>(s, cam, p, accum, stats);
}
int rt_render_device(rt_scene* s, const rt_camera* cam, const rt_params* p, float* d_accum, int accumulate,
                     void* stream, rt_stats* stats) {
  return render_on_device<float>(s, cam, p, d_accum, accumulate, (hipStream_t)stream, stats);
}
int rt_render_f64_device(rt_scene* s, const rt_camera* cam, const rt_params* p, double* d_accum, int accumulate,
                         void* stream, rt_stats* stats) {
  return render_on_device<double>(s, cam, p, d_accum, accumulate, (hipStream_t)stream, stats);
}
// rt_render_device without any allocation behind it: the launches' DeviceStats blocks (work counters + the
// persistent kernels' unit counter) live in the file-scope `scratch` array, one 256-byte slot per launch, addressed
// through the per-device pointer hipGetSymbolAddress returned (never through the array's name).  No hipMalloc /
// hipFree here or below: memsets, kernel launches, event records, and (only if stats != NULL) one small copy-back.
int rt_render_device_noalloc(rt_scene* s, const rt_camera* cam, const rt_params* p, float* d_accum, int accumulate,
                             void* stream, rt_stats* stats) {
  if (!s) return fail(RT_ERR_BAD_ARG, "rt_render_device_noalloc: null scene");
  if (!s->static_scratch) return fail(RT_ERR_HIP, "rt_render_device_noalloc: the scratch array has no address on device %d", s->device);
  return render_on_device<float>(s, cam, p, d_accum, accumulate, (hipStream_t)stream, stats, s->static_scratch,
                                 RT_NOALLOC_MAX_LAUNCHES);
}

// ---- one process, G devices -----------------------------------------------------------------------------------------
// rayon's "shared immutable scene, independent workers" model (main.rs:157-169) lifted from cores to GPUs.  An rt_multi
// keeps what does not change between renders: one device copy of the scene per entry of `devices`, one accumulator and
// one stream per entry, and on devices[0] a staging area of G-1 images.  A render: one host thread per entry enqueues
// its sample shard on its own stream, then — on that same stream, so it starts the moment the shard is done, whatever
// the other devices are doing — a peer copy of the shard into ITS slot of the staging area (every peer has its own
// xGMI link to devices[0]; the G-1 copies overlap), and records an event.  devices[0]'s stream waits for the G-1 events,
// ONE kernel adds the G-1 slots to devices[0]'s own shard in shard order, and the image goes to the host; the only host
// wait is the final one.  The result is needed on one device, so this is a reduce onto devices[0], not an all-reduce;
// the process-per-GPU path (raytracer_amd/distributed.py) is the one that uses RCCL.
struct rt_multi {
  std::vector<int> devices;
  std::vector<rt_scene*> scenes;
  std::vector<void*> accum;  // per entry, on its device: `capacity` bytes (f32 or f64 sums, whichever the render asks for)
  std::vector<hipStream_t> streams;
  std::vector<hipEvent_t> done;
  void* staging = nullptr;  // on devices[0]: (G-1) x capacity bytes
  size_t capacity = 0;
};

void rt_multi_shard(uint64_t spp, uint32_t g, uint32_t n_dev, uint64_t* first, uint64_t* count) {
  // [g*spp/G, (g+1)*spp/G): contiguous, disjoint, covers [0, spp) for every G; sizes differ by at most one
  const uint64_t lo = n_dev ? spp * (uint64_t)g / (uint64_t)n_dev : 0;
  const uint64_t hi = n_dev ? spp * ((uint64_t)g + 1) / (uint64_t)n_dev : 0;
  if (first) *first = lo;
  if (count) *count = hi - lo;
}

void rt_multi_destroy(rt_multi* m) {
  if (!m) return;
  for (size_t g = 0; g < m->devices.size(); ++g) {
    if (hipSetDevice(m->devices[g]) == hipSuccess) {
      if (g < m->streams.size() && m->streams[g]) (void)hipStreamSynchronize(m->streams[g]);
      if (g < m->accum.size() && m->accum[g]) (void)hipFree(m->accum[g]);
      if (g < m->done.size() && m->done[g]) (void)hipEventDestroy(m->done[g]);
      if (g < m->streams.size() && m->streams[g]) (void)hipStreamDestroy(m->streams[g]);
      if (g == 0 && m->staging) (void)hipFree(m->staging);
    }
    if (g < m->scenes.size()) rt_scene_destroy(m->scenes[g]);
  }
  delete m;
}

int rt_multi_create(int n_dev, const int* devices, const rt_scene_desc* desc, rt_multi** out) {
  if (!out) return fail(RT_ERR_BAD_ARG, "rt_multi_create: null out");
  *out = nullptr;
  if (n_dev < 1 || !devices || !desc) return fail(RT_ERR_BAD_ARG, "rt_multi_create: bad argument");
  if (n_dev > 64) return fail(RT_ERR_BAD_ARG, "rt_multi_create: at most 64 devices");
  rt_multi* m = new rt_multi;
  const size_t G = (size_t)n_dev;
  m->devices.assign(devices, devices + n_dev);
  m->scenes.assign(G, nullptr);
  m->accum.assign(G, nullptr);
  m->streams.assign(G, nullptr);
  m->done.assign(G, nullptr);
  std::vector<int> rcs(G, RT_OK);
  std::vector<std::string> errs(G);
  auto work = [&](size_t g) {  // the G scene uploads (tree builds included) run side by side
    auto bail = [&](int rc) {
      rcs[g] = rc;
      errs[g] = g_err;
    };
    if (int rc = rt_scene_create(m->devices[g], desc, &m->scenes[g])) return bail(rc);
    auto hip_try = [&](hipError_t e, const char* what) {
      if (e != hipSuccess) bail(fail(RT_ERR_HIP, "rt_multi_create: %s: %s", what, hipGetErrorString(e)));
      return e == hipSuccess;
    };
    if (!hip_try(hipSetDevice(m->devices[g]), "hipSetDevice")) return;
    if (!hip_try(hipStreamCreateWithFlags(&m->streams[g], hipStreamNonBlocking), "hipStreamCreate")) return;
    if (!hip_try(hipEventCreateWithFlags(&m->done[g], hipEventDisableTiming), "hipEventCreate")) return;
    if (g > 0 && m->devices[g] != m->devices[0]) {
      // direct xGMI copies into devices[0]'s staging area; without peer access the runtime stages through the host
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, m->devices[g], m->devices[0]) == hipSuccess && can) {
        hipError_t e = hipDeviceEnablePeerAccess(m->devices[0], 0);
        if (e != hipSuccess) (void)hipGetLastError();  // already enabled (an earlier rt_multi, or the caller): fine
      }
    }
  };
  std::vector<std::thread> pool;
  for (size_t g = 1; g < G; ++g) pool.emplace_back(work, g);
  work(0);
  for (auto& t : pool) t.join();
  for (size_t g = 0; g < G; ++g)
    if (rcs[g] != RT_OK) {
      const int rc = fail(rcs[g], "device %d: %s", m->devices[g], errs[g].c_str());
      rt_multi_destroy(m);
      return rc;
    }
  *out = m;
  return RT_OK;
}

static int multi_reserve(rt_multi* m, size_t want_bytes) {
  if (want_bytes <= m->capacity) return RT_OK;
  const size_t G = m->devices.size();
  const size_t bytes = std::max<size_t>(want_bytes, 16);
  for (size_t g = 0; g < G; ++g) {
    HIP_TRY(hipSetDevice(m->devices[g]));
    HIP_TRY(hipStreamSynchronize(m->streams[g]));
    if (m->accum[g]) (void)hipFree(m->accum[g]);
    m->accum[g] = nullptr;
    m->capacity = 0;
    HIP_TRY(hipMalloc(&m->accum[g], bytes));
    if (g == 0) {
      if (m->staging) (void)hipFree(m->staging);
      m->staging = nullptr;
      if (G > 1) HIP_TRY(hipMalloc(&m->staging, bytes * (G - 1)));
    }
  }
  m->capacity = bytes;
  return RT_OK;
}

}  // extern "C"
template <typename R>
static int multi_render(rt_multi* m, const rt_camera* cam, const rt_params* p, R* accum, rt_stats* per_dev, const char* who) {
  if (!m || !p || !accum) return fail(RT_ERR_BAD_ARG, "%s: null argument", who);
  if (int rc = check_camera(cam)) return rc;
  if (int rc = check_params(p, who)) return rc;  // before any accumulator is sized from them
  const size_t G = m->devices.size();
  const size_t n_floats = (size_t)p->width * p->height * 3;
  if (int rc = multi_reserve(m, n_floats * sizeof(R))) return rc;
  std::vector<int> rcs(G, RT_OK);
  std::vector<std::string> errs(G);
  auto work = [&](size_t g) {
    auto bail = [&](int rc) {
      rcs[g] = rc;
      errs[g] = g_err;
    };
    rt_params q = *p;  // samples [first, first + count) of every pixel: the union of the shards' Philox streams is
    uint64_t first = 0, count = 0;  // independent of G
    rt_multi_shard(p->spp, (uint32_t)g, (uint32_t)G, &first, &count);
    q.spp = (uint32_t)count;
    q.sample_offset = p->sample_offset + first;
    // stats == NULL keeps the whole call asynchronous; with per_dev the shard's own thread waits for its counters
    if (int rc = render_on_device<R>(m->scenes[g], cam, &q, static_cast<R*>(m->accum[g]), 0, m->streams[g], per_dev ? &per_dev[g] : nullptr))
      return bail(rc);
    hipError_t e = hipSuccess;
    if (g > 0)
      e = hipMemcpyPeerAsync(static_cast<R*>(m->staging) + (g - 1) * n_floats, m->devices[0], m->accum[g], m->devices[g],
                             n_floats * sizeof(R), m->streams[g]);
    if (e == hipSuccess && g > 0) e = hipEventRecord(m->done[g], m->streams[g]);
    if (e != hipSuccess) bail(fail(RT_ERR_HIP, "%s: shard %zu copy: %s", who, g, hipGetErrorString(e)));
  };
  std::vector<std::thread> pool;
  for (size_t g = 1; g < G; ++g) pool.emplace_back(work, g);
  work(0);
  for (auto& t : pool) t.join();
  int rc = RT_OK;
  for (size_t g = 0; g < G && rc == RT_OK; ++g)
    if (rcs[g] != RT_OK) rc = fail(rcs[g], "device %d: %s", m->devices[g], errs[g].c_str());
  auto hip_ok = [&](hipError_t e, const char* what) {
    if (e != hipSuccess && rc == RT_OK) rc = fail(RT_ERR_HIP, "%s: %s: %s", who, what, hipGetErrorString(e));
    return e == hipSuccess;
  };
  if (rc == RT_OK && hip_ok(hipSetDevice(m->devices[0]), "hipSetDevice")) {
    for (size_t g = 1; g < G && rc == RT_OK; ++g) hip_ok(hipStreamWaitEvent(m->streams[0], m->done[g], 0), "hipStreamWaitEvent");
    if (rc == RT_OK && G > 1) {
      if constexpr (sizeof(R) == 8)
        hip_ok(launch_accum_add_f64(static_cast<double*>(m->accum[0]), static_cast<const double*>(m->staging), n_floats,
                                    (uint32_t)(G - 1), m->streams[0]), "accum_add");
      else
        hip_ok(launch_accum_add(static_cast<float*>(m->accum[0]), static_cast<const float*>(m->staging), n_floats,
                                (uint32_t)(G - 1), m->streams[0]), "accum_add");
    }
    if (rc == RT_OK)
      hip_ok(hipMemcpyAsync(accum, m->accum[0], n_floats * sizeof(R), hipMemcpyDeviceToHost, m->streams[0]), "copy back");
    if (rc == RT_OK) hip_ok(hipStreamSynchronize(m->streams[0]), "hipStreamSynchronize");
  }
  if (rc != RT_OK)  // leave nothing of this call in flight behind an error
    for (size_t g = 0; g < G; ++g)
      if (hipSetDevice(m->devices[g]) == hipSuccess) (void)hipStreamSynchronize(m->streams[g]);
  return rc;
}

extern "C" {
int rt_multi_render(rt_multi* m, const rt_camera* cam, const rt_params* p, float* accum, rt_stats* per_dev) {
  return multi_render<float>(m, cam, p, accum, per_dev, "rt_multi_render");
}
int rt_multi_render_f64(rt_multi* m, const rt_camera* cam, const rt_params* p, double* accum, rt_stats* per_dev) {
  return multi_render<double>(m, cam, p, accum, per_dev, "rt_multi_render_f64");
}

int rt_render_multi(int n_dev, const int* devices, const rt_scene_desc* desc, const rt_camera* cam, const rt_params* p,
                    float* accum, rt_stats* per_dev) {
  if (n_dev < 1 || !devices || !desc || !p || !accum) return fail(RT_ERR_BAD_ARG, "rt_render_multi: bad argument");
  if (int rc = check_camera(cam)) return rc;
  if (int rc = check_params(p, "rt_render_multi")) return rc;  // before G scenes are built
  rt_multi* m = nullptr;
  if (int rc = rt_multi_create(n_dev, devices, desc, &m)) return rc;
  const int rc = rt_multi_render(m, cam, p, accum, per_dev);
  rt_multi_destroy(m);
  return rc;
}

int rt_debug_build_tree(uint32_t n_leaves, const float* leaf_box, const uint32_t* leaf_refs, uint32_t max_depth,
                        uint32_t* nodes_out, uint32_t* height_out) {
  if (n_leaves < 2 || !leaf_box || !leaf_refs || !nodes_out || !height_out)
    return fail(RT_ERR_BAD_ARG, "rt_debug_build_tree: need >= 2 leaves and non-null arrays");
  for (uint32_t i = 0; i < n_leaves; ++i)
    if (!(leaf_refs[i] & RT_REF_LEAF)) return fail(RT_ERR_BAD_ARG, "rt_debug_build_tree: leaf_refs[%u] lacks RT_REF_LEAF", i);
  static_assert(sizeof(Node2) == 64, "Node2 is one 64-byte line");
  std::vector<Node2> nodes = accel::build_sah(leaf_box, leaf_refs, n_leaves, max_depth, height_out);
  std::memcpy(nodes_out, nodes.data(), nodes.size() * sizeof(Node2));
  return RT_OK;
}

void rt_tonemap(const float* accum, uint64_t n_pixels, uint64_t spp, uint8_t* rgb8) {
  if (!accum || !rgb8) return;
  const float inv = 1.0f / (float)spp;
  for (uint64_t i = 0; i < 3 * n_pixels; ++i) {
    float g = std::sqrt(accum[i] * inv);
    uint8_t q = 0;
    if (g == g) {
      g = g < 0.0f ? 0.0f : (g > 0.9999f ? 0.9999f : g);
      q = (uint8_t)(g * 256.0f);
    }
    rgb8[i] = q;
  }
}

int rt_tonemap_device(int device, const float* d_accum, uint64_t n_pixels, uint64_t spp, uint8_t* d_rgb8, void* stream) {
  // spp == 0 is not refused: like rt_tonemap (and like main.rs:167, `pixel_color / 0`) it divides by zero, and
  // 0/0 = NaN quantises to 0 (main.rs:59-61), anything else to 255
  if (!d_accum || !d_rgb8) return fail(RT_ERR_BAD_ARG, "rt_tonemap_device: null argument");
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(launch_tonemap(d_accum, 3 * n_pixels, 1.0f / (float)spp, d_rgb8, (hipStream_t)stream));
  return RT_OK;
}

void rt_tonemap_f64(const double* accum, uint64_t n_pixels, uint64_t spp, uint8_t* rgb8) {
  if (!accum || !rgb8) return;
  for (uint64_t i = 0; i < 3 * n_pixels; ++i) {
    double g = std::sqrt(accum[i] / (double)spp);
    uint8_t q = 0;
    if (g == g) {
      g = g < 0.0 ? 0.0 : (g > 0.9999 ? 0.9999 : g);
      q = (uint8_t)(g * 256.0);
    }
    rgb8[i] = q;
  }
}

int rt_tonemap_f64_device(int device, const double* d_accum, uint64_t n_pixels, uint64_t spp, uint8_t* d_rgb8, void* stream) {
  if (!d_accum || !d_rgb8) return fail(RT_ERR_BAD_ARG, "rt_tonemap_f64_device: null argument");
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(launch_tonemap_f64(d_accum, 3 * n_pixels, (double)spp, d_rgb8, (hipStream_t)stream));
  return RT_OK;
}

static int kat_common(int device, size_t in_bytes[], const void* in[], int n_in, size_t out_bytes, void* out,
                      hipError_t (*run)(void** d_in, void* d_out, void* ctx), void* ctx) {
  HIP_TRY(hipSetDevice(device));
  DeviceBuffers tmp;
  void* d_in[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int i = 0; i < n_in; ++i) {
    HIP_TRY(hipMalloc(&d_in[i], std::max<size_t>(in_bytes[i], 16)));
    tmp.ptrs.push_back(d_in[i]);
    HIP_TRY(hipMemcpy(d_in[i], in[i], in_bytes[i], hipMemcpyHostToDevice));
  }
  void* d_out = nullptr;
  HIP_TRY(hipMalloc(&d_out, std::max<size_t>(out_bytes, 16)));
  tmp.ptrs.push_back(d_out);
  HIP_TRY(run(d_in, d_out, ctx));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, d_out, out_bytes, hipMemcpyDeviceToHost));
  return RT_OK;
}

struct KatCtx {
  uint32_t n;
  int variant;  // sphere_uv / rotate: use the f32 kernels; philox: 1 = the path streams' generator (7 rounds)
};

int rt_kat_sphere_uv(int device, uint32_t n, const double* p, double* uv, int use_f32) {
  if (!p || !uv || n == 0) return fail(RT_ERR_BAD_ARG, "rt_kat_sphere_uv: bad argument");
  size_t ib[] = {n * 3 * sizeof(double)};
  const void* in[] = {p};
  KatCtx c{n, use_f32};
  return kat_common(device, ib, in, 1, n * 2 * sizeof(double), uv,
                    [](void** di, void* dout, void* ctx) -> hipError_t {
                      auto* k = (KatCtx*)ctx;
                      return k->variant ? launch_kat_sphere_uv_f32(k->n, (const double*)di[0], (double*)dout, nullptr)
                                        : launch_kat_sphere_uv_f64(k->n, (const double*)di[0], (double*)dout, nullptr);
                    },
                    &c);
}

int rt_kat_rotate(int device, uint32_t n, const double* angle, const double* axis, const double* v, double* out,
                  int use_f32) {
  if (!angle || !axis || !v || !out || n == 0) return fail(RT_ERR_BAD_ARG, "rt_kat_rotate: bad argument");
  size_t ib[] = {n * sizeof(double), n * 3 * sizeof(double), n * 3 * sizeof(double)};
  const void* in[] = {angle, axis, v};
  KatCtx c{n, use_f32};
  return kat_common(device, ib, in, 3, n * 3 * sizeof(double), out,
                    [](void** di, void* dout, void* ctx) -> hipError_t {
                      auto* k = (KatCtx*)ctx;
                      return k->variant ? launch_kat_rotate_f32(k->n, (const double*)di[0], (const double*)di[1],
                                                                (const double*)di[2], (double*)dout, nullptr)
                                        : launch_kat_rotate_f64(k->n, (const double*)di[0], (const double*)di[1],
                                                                (const double*)di[2], (double*)dout, nullptr);
                    },
                    &c);
}

int rt_kat_libm(int device, uint32_t op, uint32_t n, const double* a, const double* b, double* out) {
  if (!a || !b || !out || n == 0 || op > 4) return fail(RT_ERR_BAD_ARG, "rt_kat_libm: bad argument");
  size_t ib[] = {n * sizeof(double), n * sizeof(double)};
  const void* in[] = {a, b};
  KatCtx c{n, (int)op};
  return kat_common(device, ib, in, 2, n * sizeof(double), out,
                    [](void** di, void* dout, void* ctx) -> hipError_t {
                      auto* k = (KatCtx*)ctx;
                      return launch_kat_libm(k->n, (uint32_t)k->variant, (const double*)di[0], (const double*)di[1],
                                             (double*)dout, nullptr);
                    },
                    &c);
}

int rt_kat_philox(int device, uint32_t n, const uint32_t* ctr, const uint32_t* key, uint32_t rounds, uint32_t* out) {
  if (!ctr || !key || !out || n == 0) return fail(RT_ERR_BAD_ARG, "rt_kat_philox: bad argument");
  if (rounds != 7 && rounds != 10)
    return fail(RT_ERR_BAD_ARG, "rt_kat_philox: rounds must be 7 (path streams) or 10 (scene streams)");
  size_t ib[] = {n * 4 * sizeof(uint32_t), n * 2 * sizeof(uint32_t)};
  const void* in[] = {ctr, key};
  KatCtx c{n, rounds == 7 ? 1 : 0};
  return kat_common(device, ib, in, 2, n * 4 * sizeof(uint32_t), out,
                    [](void** di, void* dout, void* ctx) -> hipError_t {
                      auto* k = (KatCtx*)ctx;
                      return launch_kat_philox(k->n, (const uint32_t*)di[0], (const uint32_t*)di[1], k->variant,
                                               (uint32_t*)dout, nullptr);
                    },
                    &c);
}

}  // extern "C"
