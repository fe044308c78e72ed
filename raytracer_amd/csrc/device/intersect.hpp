// Flattened-BVH traversal and primitive intersection: the device form of Hittable::hit
// (hittable.rs:208-415) and AxisAlignedBoundingBox::hit (aabb.rs:17-34).
//
// The reference recurses (Bvh arm, hittable.rs:400-412): box test, left child, then the right
// child clipped to the left hit's t, right wins only if strictly closer.  Every arm returns
// t <= the t_max it was given, so that recursion is equivalent to an iterative left-first
// DFS with a LIFO stack, ONE running `closest`, node boxes tested against (t_min, closest),
// and replacement only on strictly smaller t: same nodes visited, same primitives tested
// with the same bounds, same medium draws in the same order, same winner on ties.
//
// Traversal keeps only (t, leaf ref) per candidate; the full HitRecord (p, normal, uv,
// front_face — hittable.rs:17-40) is rebuilt once, for the winner, by make_hit_record().
#pragma once
#include "philox.hpp"
#include "scene_view.hpp"

namespace rt {

template <bool STATS> struct Cnt {
  uint32_t v[STATS ? ST_COUNT : 1];
  RT_DEV void clear() {
#pragma unroll
    for (int i = 0; i < (STATS ? (int)ST_COUNT : 1); ++i) v[i] = 0;
  }
  template <int I> RT_DEV void inc(uint32_t n = 1) {
    if constexpr (I == ST_RAYS)
      v[0] += n;
    else if constexpr (STATS)
      v[I] += n;
  }
};

template <typename R> struct RayT {
  Vec3<R> o, d;
  R time;
};

// ---- aabb.rs:17-34, with inv_d hoisted out of the per-node loop (the reference multiplies
// by inv_d = 1/d, so hoisting is bit-identical) -------------------------------------------
// The reference leaves the loop at the first axis whose interval is empty; the interval only
// shrinks from axis to axis (max / min are monotone and ignore NaN), so "some axis failed" is
// exactly "the final t_max <= t_min": all three axes are evaluated without branches and tested
// once.  Same booleans, same t_min out.
template <typename R>
RT_DEV bool slab_hit(const R* lo, const R* hi, Vec3<R> o, Vec3<R> inv_d, R& t_min, R& t_max) {
  const R lx = lo[0], ly = lo[1], lz = lo[2], hx = hi[0], hy = hi[1], hz = hi[2];
  R t0 = (lx - o.x) * inv_d.x, t1 = (hx - o.x) * inv_d.x;
  R lo_t = rmax(inv_d.x < R(0) ? t1 : t0, t_min);
  R hi_t = rmin(inv_d.x < R(0) ? t0 : t1, t_max);
  t0 = (ly - o.y) * inv_d.y;
  t1 = (hy - o.y) * inv_d.y;
  lo_t = rmax(inv_d.y < R(0) ? t1 : t0, lo_t);
  hi_t = rmin(inv_d.y < R(0) ? t0 : t1, hi_t);
  t0 = (lz - o.z) * inv_d.z;
  t1 = (hz - o.z) * inv_d.z;
  lo_t = rmax(inv_d.z < R(0) ? t1 : t0, lo_t);
  hi_t = rmin(inv_d.z < R(0) ? t0 : t1, hi_t);
  t_min = lo_t;
  t_max = hi_t;
  return !(hi_t <= lo_t);
}

// ---- hittable.rs:257-276 (Sphere) -------------------------------------------------------
template <typename R>
RT_DEV bool sphere_t(Vec3<R> o, Vec3<R> d, R a, Vec3<R> c, R radius, R t_min, R t_max, R& t_out) {
  Vec3<R> oc = o - c;
  R half_b = dot(oc, d);
  R cc = sqrlen(oc) - radius * radius;
  R disc = half_b * half_b - a * cc;
  if (disc < R(0)) return false;
  R sqrtd = rroot(disc);
  R root = (-half_b - sqrtd) / a;
  if (root < t_min || root > t_max) {
    root = (-half_b + sqrtd) / a;
    if (root < t_min || root > t_max) return false;
  }
  t_out = root;
  return true;
}
// f32 production form.  The reference computes in f64; in f32 the two subtractions
// |oc|^2 - r^2 and half_b^2 - a*c cancel catastrophically for the r = 1000 ground spheres
// (scenes.rs:59,148) and for small far spheres, which would break the 1e-5 bound on t and
// normals.  The inputs stay f32; only the cancelling dot products are carried in f64
// (vector f64 runs at half the f32 rate on gfx950), then everything returns to f32.
struct SphereRoots {
  float near_t, far_t;  // (-half_b -/+ sqrt(disc)) / a
  bool real;            // disc >= 0 (a NaN discriminant counts as real, as in the reference's `< 0.0` test)
};
// Both roots at once: a constant medium probes its boundary twice with different windows
// (hittable.rs:324-331); the quadratic is solved once and each probe only picks its root.
RT_DEV SphereRoots sphere_roots(Vec3<float> o, Vec3<float> d, float a, Vec3<float> c, float radius) {
  // Keep the f32 -> f64 conversions of the ray HERE (an empty asm the optimiser cannot see through): hoisted out
  // of the leaf-list loop they occupy 12 vector registers for the whole walk, which costs the persistent kernel
  // its sixth wave per SIMD (measured: 60.4 -> 65.3 Grays/s on Cornell with the barrier and 6 waves).
  asm volatile("" : "+v"(o.x), "+v"(o.y), "+v"(o.z), "+v"(d.x), "+v"(d.y), "+v"(d.z));
  const double ocx = (double)o.x - (double)c.x, ocy = (double)o.y - (double)c.y, ocz = (double)o.z - (double)c.z;
  const double dx = d.x, dy = d.y, dz = d.z;
  // explicit FMAs: one fixed form at every inlined copy (see mul_add3)
  const double half_b = __builtin_fma(ocz, dz, __builtin_fma(ocy, dy, ocx * dx));
  const double cc = __builtin_fma(ocz, ocz, __builtin_fma(ocy, ocy, ocx * ocx)) - (double)radius * (double)radius;
  const double a64 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
  const double disc = __builtin_fma(half_b, half_b, -(a64 * cc));
  SphereRoots s;
  s.real = !(disc < 0.0);
  const double sqrtd = (double)rroot((float)disc);
  const float inv_a = rrcp(a);
  s.near_t = (float)(-half_b - sqrtd) * inv_a;
  s.far_t = (float)(-half_b + sqrtd) * inv_a;
  return s;
}
// hittable.rs:266-273: the nearest root inside [t_min, t_max]
RT_DEV bool sphere_pick(const SphereRoots& s, float t_min, float t_max, float& t_out) {
  if (!s.real) return false;
  float root = s.near_t;
  if (root < t_min || root > t_max) {
    root = s.far_t;
    if (root < t_min || root > t_max) return false;
  }
  t_out = root;
  return true;
}
template <>
RT_DEV bool sphere_t<float>(Vec3<float> o, Vec3<float> d, float a, Vec3<float> c, float radius, float t_min,
                            float t_max, float& t_out) {
  return sphere_pick(sphere_roots(o, d, a, c, radius), t_min, t_max, t_out);
}

// outward normal (p - c) / radius, hittable.rs:278; the f32 form builds p - c as
// (o - c) + t d with o - c rounded once, instead of rounding p first.
template <typename R>
RT_DEV Vec3<R> sphere_outward(Vec3<R> o, Vec3<R> d, R t, Vec3<R> p, Vec3<R> c, R radius) {
  return (p - c) / radius;
}
template <>
RT_DEV Vec3<float> sphere_outward<float>(Vec3<float> o, Vec3<float> d, float t, Vec3<float> p, Vec3<float> c,
                                         float radius) {
  Vec3<float> oc((float)((double)o.x - (double)c.x), (float)((double)o.y - (double)c.y),
                 (float)((double)o.z - (double)c.z));
  return madd(oc, t, d) / radius;
}

// hittable.rs:66-72
template <typename R> RT_DEV Vec3<R> moving_center(const MSphere<R>& m, R time) {
  Vec3<R> c0(m.c0[0], m.c0[1], m.c0[2]), c1(m.c1[0], m.c1[1], m.c1[2]);
  const R s = rdiv(time - m.t0, m.t1 - m.t0);
  if constexpr (sizeof(R) == 4)  // one fixed fused form (see mul_add3)
    return {__builtin_fmaf(s, c1.x - c0.x, c0.x), __builtin_fmaf(s, c1.y - c0.y, c0.y), __builtin_fmaf(s, c1.z - c0.z, c0.z)};
  else
    return c0 + s * (c1 - c0);
}

// t1 + hit_distance / ray_length (hittable.rs:346-348).  The f32 kernels evaluate it in three places (leaf list,
// reference-order walk, ordered walk); under HIP's default -ffp-contract=fast the compiler may fuse the multiply-add in one
// and not in another, and the walks would then disagree in the last bit of t.  One explicit FMA for all of them.
RT_DEV float medium_exit_t(float t1, float hit_distance, float ray_length) {
  return __builtin_fmaf(hit_distance, __builtin_amdgcn_rcpf(ray_length), t1);
}
RT_DEV double medium_exit_t(double t1, double hit_distance, double ray_length) { return t1 + hit_distance / ray_length; }

// One simple primitive (SPHERE / MSPHERE / CUBE) against a ray in that primitive's frame.
template <typename R, bool STATS>
RT_DEV bool prim_t(const SceneView<R>& sc, uint32_t ref, const RayT<R>& r, Vec3<R> inv_d, R a, R t_min,
                   R t_max, R& t_out, Cnt<STATS>& cnt) {
  uint32_t kind = ref_kind(ref), idx = ref_index(ref);
  if (kind == LEAF_SPHERE) {
    cnt.template inc<ST_SPHERE_TESTS>();
    Real4<R> s = sc.spheres[idx];
    bool h = sphere_t(r.o, r.d, a, Vec3<R>(s.x, s.y, s.z), s.w, t_min, t_max, t_out);
    if (h) cnt.template inc<ST_SPHERE_HITS>();
    return h;
  } else if (kind == LEAF_MSPHERE) {
    cnt.template inc<ST_SPHERE_TESTS>();
    MSphere<R> m = sc.mspheres[idx];
    bool h = sphere_t(r.o, r.d, a, moving_center(m, r.time), m.radius, t_min, t_max, t_out);
    if (h) cnt.template inc<ST_SPHERE_HITS>();
    return h;
  } else {  // LEAF_CUBE: hittable.rs:210-229 — returns the ENTRY t only (running t_min, Q2)
    cnt.template inc<ST_CUBE_TESTS>();
    Cube<R> c = sc.cubes[idx];
    R lo = t_min, hi = t_max;
    if (!slab_hit(c.p0, c.p1, r.o, inv_d, lo, hi)) return false;
    cnt.template inc<ST_CUBE_HITS>();
    t_out = lo;
    return true;
  }
}

template <typename R> RT_DEV Vec3<R> inverse3(Vec3<R> d) { return {rrcp(d.x), rrcp(d.y), rrcp(d.z)}; }

template <typename R> RT_DEV Vec3<R> xf_dir(const R* m, Vec3<R> v) {
  return {mul_add3(m[0], v.x, m[1], v.y, m[2], v.z), mul_add3(m[3], v.x, m[4], v.y, m[5], v.z),
          mul_add3(m[6], v.x, m[7], v.y, m[8], v.z)};
}
template <typename R> RT_DEV Vec3<R> xf_point(const R* m, const R* b, Vec3<R> v) {
  return {mul_add3(m[0], v.x, m[1], v.y, m[2], v.z) + b[0], mul_add3(m[3], v.x, m[4], v.y, m[5], v.z) + b[1],
          mul_add3(m[6], v.x, m[7], v.y, m[8], v.z) + b[2]};
}
template <typename R> RT_DEV Vec3<R> xf_dir_transposed(const R* m, Vec3<R> v) {
  return {mul_add3(m[0], v.x, m[3], v.y, m[6], v.z), mul_add3(m[1], v.x, m[4], v.y, m[7], v.z),
          mul_add3(m[2], v.x, m[5], v.y, m[8], v.z)};
}

// hittable.rs:331: the medium's second boundary probe starts at hit1.t + 0.0001.  In f32 that sum
// rounds back to hit1.t once |hit1.t| >= 2048 (a ray deep inside a large medium: the entry lies far
// behind the origin), the probe would find the entry again and the medium would never scatter; the
// f32 kernels therefore start the probe at least one float above hit1.t.
template <typename R> RT_DEV R second_probe_start(R t1) {
  const R t = t1 + R(0.0001);
  if constexpr (sizeof(R) == 4) {
    const uint32_t b = __float_as_uint(t1);
    const float up = t1 > 0.0f ? __uint_as_float(b + 1u) : (t1 < 0.0f ? __uint_as_float(b - 1u) : 1e-45f);
    return t > up ? t : up;
  }
  return t;
}

// A chain of Translate / Rotate / ConstantMedium wrappers around one primitive
// (hittable.rs:319-386).  All wrappers are rigid, so t is the same in every frame.
template <typename R, bool STATS>
RT_DEV bool complex_t(const SceneView<R>& sc, uint32_t idx, const RayT<R>& r, R t_min, R t_max, Rng& rng,
                      R& t_out, Cnt<STATS>& cnt) {
  Complex cx = sc.complexes[idx];
  RayT<R> lr = r;
  Vec3<R> dm = r.d;  // direction in the medium's frame
  R neg_inv_density = R(0);
  if constexpr (sizeof(R) == 4) {
    const ComplexXf<R>& xf = sc.complex_xf[idx];
    cnt.template inc<ST_TRANSLATE>(cx.n_translate);
    cnt.template inc<ST_ROTATE>(cx.n_rotate);
    if (cx.medium_pos != REF_NONE) cnt.template inc<ST_MEDIUM>();
    lr.o = xf_point(xf.m_full, xf.b_full, r.o);
    lr.d = xf_dir(xf.m_full, r.d);
    neg_inv_density = xf.neg_inv_density;  // |dm| = |r.d|: the maps are rigid
  } else
  for (uint32_t k = 0; k < cx.op_count; ++k) {
    uint32_t tag = sc.op_tag[cx.op_first + k];
    Op<R> op = sc.ops[cx.op_first + k];
    if (tag == OP_TRANSLATE) {  // hittable.rs:365
      cnt.template inc<ST_TRANSLATE>();
      lr.o = lr.o - Vec3<R>(op.d[0], op.d[1], op.d[2]);
    } else if (tag == OP_ROTATE) {  // hittable.rs:375-377
      cnt.template inc<ST_ROTATE>();
      Quat<R> q{op.d[0], op.d[1], op.d[2], op.d[3]};
      lr.o = qrotate(lr.o, q);
      lr.d = qrotate(lr.d, q);
    } else {
      cnt.template inc<ST_MEDIUM>();
      dm = lr.d;
      neg_inv_density = op.d[0];
    }
  }
  Vec3<R> inv_d = inverse3(lr.d);
  R a = sqrlen(lr.d);
  if (cx.medium_pos == REF_NONE) return prim_t(sc, cx.body, lr, inv_d, a, t_min, t_max, t_out, cnt);

  // hittable.rs:319-349
  R t1, t2;
  cnt.template inc<ST_MEDIUM_PROBES>();
  if (!prim_t(sc, cx.body, lr, inv_d, a, -Consts<R>::inf, Consts<R>::inf, t1, cnt)) return false;
  cnt.template inc<ST_MEDIUM_PROBES>();
  if (!prim_t(sc, cx.body, lr, inv_d, a, second_probe_start(t1), Consts<R>::inf, t2, cnt)) return false;
  t1 = rmax(t1, t_min);
  t2 = rmin(t2, t_max);
  if (t1 >= t2) return false;
  t1 = rmax(t1, R(0));
  R ray_length = length(dm);
  R distance_inside_boundary = (t2 - t1) * ray_length;
  R hit_distance = neg_inv_density * rng.template keyed_log<R>(cx.object);
  if (hit_distance > distance_inside_boundary) return false;
  t_out = medium_exit_t(t1, hit_distance, ray_length);
  return true;
}

// f32, any leaf, ONE copy of each primitive's code: a wave whose lanes sit at leaves of different
// kinds (bare sphere, moving sphere, wrapped sphere, medium around a sphere ...) runs the sphere test
// once for all of them instead of once per kind (prim_t inlined under complex_t and under the medium's
// probes).  Same arithmetic as prim_t / complex_t<float>.
template <bool STATS>
RT_DEV bool leaf_t_unified(const SceneView<float>& sc, uint32_t ref, const RayT<float>& r, Vec3<float> inv_d, float a,
                           float t_min, float t_max, Rng& rng, float& t_out, Cnt<STATS>& cnt) {
  uint32_t kind = ref_kind(ref), idx = ref_index(ref);
  Vec3<float> o = r.o, d = r.d;
  bool medium = false;
  float neg_inv_density = 0.0f;
  uint32_t object = 0;
  if (kind == LEAF_COMPLEX) {
    const Complex cx = sc.complexes[idx];
    const ComplexXf<float>& xf = sc.complex_xf[idx];
    cnt.template inc<ST_TRANSLATE>(cx.n_translate);
    cnt.template inc<ST_ROTATE>(cx.n_rotate);
    o = xf_point(xf.m_full, xf.b_full, r.o);
    d = xf_dir(xf.m_full, r.d);
    inv_d = inverse3(d);
    a = sqrlen(d);
    medium = cx.medium_pos != REF_NONE;
    neg_inv_density = xf.neg_inv_density;
    object = cx.object;
    kind = ref_kind(cx.body);
    idx = ref_index(cx.body);
    if (medium) cnt.template inc<ST_MEDIUM>();
  }
  // the primitive in its own frame: a box (p0, p1) or a sphere (c, radius)
  const bool is_cube = kind == LEAF_CUBE;
  Vec3<float> p0, p1;  // cube corners, or (centre, radius in p1.x)
  if (is_cube) {
    const Cube<float> c = sc.cubes[idx];
    p0 = Vec3<float>(c.p0[0], c.p0[1], c.p0[2]);
    p1 = Vec3<float>(c.p1[0], c.p1[1], c.p1[2]);
  } else if (kind == LEAF_SPHERE) {
    const Real4<float> sp = sc.spheres[idx];
    p0 = Vec3<float>(sp.x, sp.y, sp.z);
    p1 = Vec3<float>(sp.w, 0.0f, 0.0f);
    if constexpr (STATS) {  // a baked sphere (rt_api.hip) stands for its Rotate / Translate wrappers in the work counters
      const uint32_t w = sc.sphere_attr[idx].pad;
      cnt.template inc<ST_ROTATE>(w & 0xFFFFu);
      cnt.template inc<ST_TRANSLATE>(w >> 16);
    }
  } else {
    const MSphere<float> m = sc.mspheres[idx];
    p0 = moving_center(m, r.time);
    p1 = Vec3<float>(m.radius, 0.0f, 0.0f);
  }
  SphereRoots roots;
  if (!is_cube) roots = sphere_roots(o, d, a, p0, p1.x);  // once, for both probes of a medium
  // probe 0: the hit test proper, or a medium's first boundary probe; probe 1: a medium's second
  float lo = medium ? -Consts<float>::inf : t_min, hi = medium ? Consts<float>::inf : t_max;
  float t1 = 0.0f, t2 = 0.0f;
  bool pending = true, ok = false;
  for (int probe = 0; probe < 2; ++probe) {
    if (pending) {
      float t;
      bool h;
      if (medium) cnt.template inc<ST_MEDIUM_PROBES>();
      if (is_cube) {  // hittable.rs:210-229
        cnt.template inc<ST_CUBE_TESTS>();
        const float blo[3] = {p0.x, p0.y, p0.z}, bhi[3] = {p1.x, p1.y, p1.z};
        float l = lo, u = hi;
        h = slab_hit(blo, bhi, o, inv_d, l, u);
        t = l;
        if (h) cnt.template inc<ST_CUBE_HITS>();
      } else {  // hittable.rs:257-276 / 288-307
        cnt.template inc<ST_SPHERE_TESTS>();
        h = sphere_pick(roots, lo, hi, t);
        if (h) cnt.template inc<ST_SPHERE_HITS>();
      }
      if (probe == 0) {
        t1 = t;
        ok = h;
        pending = medium && h;
        lo = second_probe_start(t1);  // hittable.rs:331
      } else {
        t2 = t;
        ok = h;
        pending = false;
      }
    }
    if (!__any(pending)) break;
  }
  if (!ok) return false;
  if (!medium) {
    t_out = t1;
    return true;
  }
  // hittable.rs:332-349
  t1 = rmax(t1, t_min);
  t2 = rmin(t2, t_max);
  if (t1 >= t2) return false;
  t1 = rmax(t1, 0.0f);
  const float ray_length = length(r.d);  // rigid maps keep |d|
  const float hit_distance = neg_inv_density * rng.template keyed_log<float>(object);
  if (hit_distance > (t2 - t1) * ray_length) return false;
  t_out = medium_exit_t(t1, hit_distance, ray_length);
  return true;
}

template <typename R> RT_DEV uint32_t leaf_rank(const SceneView<R>& sc, uint32_t ref);
// world.hit(r, t_min, +inf) — hittable.rs:400-412 as an iterative DFS.  `stack` is this
// lane's column of the LDS stack: entry i lives at stack[i * stride].
// Same per-lane visiting order as the recursion, run in two wave phases: every lane first descends through inner nodes
// until it stands at a leaf (or has nothing left), then the lanes at a leaf run their leaf tests TOGETHER — instead of
// one lane's box test and another's (much longer) leaf test taking turns in every iteration.
// INSTANCES (a complex whose body is a node, CXF_INSTANCE: Translate / Rotate over a Bvh, hittable.rs:364-386 calling
// hittable.rs:400-412): the walk maps the ray into the instance's frame exactly as the wrappers' hit arms do (f64: op by
// op; f32: the composed rigid map), pushes REF_MARKER and descends into the sub-tree with the SAME running `closest`
// (the arms pass t_min / t_max through; t is the same in every frame); popping the marker restores the world ray.  A
// hit inside is reported as the leaf's synthesized complex (SceneView::leaf_by_rank), whose record make_hit_record
// builds through the same wrapper chain.
// INST: the instantiation that carries the instance arm (scenes without instances run the one without: the mapped
// ray's registers cost the f32 reference-order kernel a third of its speed on the reference's default job otherwise).
template <bool INST, typename R, bool STATS>
RT_DEV void traverse_impl(const SceneView<R>& sc, const RayT<R>& r, R t_min, Rng& rng, uint32_t* stack,
                          uint32_t stride, R& closest, uint32_t& best, Cnt<STATS>& cnt) {
  closest = Consts<R>::inf;
  best = REF_NONE;
  uint32_t cur = sc.root;
  if (cur == REF_NONE) return;
  RayT<R> lr = r;  // the ray in the frame being walked (INST = false: always the world ray)
  Vec3<R> inv_d = inverse3(r.d);
  R a = sqrlen(r.d);
  bool in_instance = false;
  uint32_t sp = 0;
  bool walking = true;
  for (;;) {
    while (walking && !(cur & REF_LEAF)) {
      cnt.template inc<ST_NODES>();
      Node<R> n = sc.nodes[cur];
      R lo = t_min, hi = closest;
      if (slab_hit(n.lo, n.hi, lr.o, inv_d, lo, hi)) {
        stack[sp * stride] = n.right;
        ++sp;
        cur = n.left;
      } else if (sp == 0) {
        walking = false;
      } else {
        --sp;
        cur = stack[sp * stride];
      }
    }
    if (!walking) break;
    if (INST && cur == REF_MARKER) {  // the instance's sub-tree is done: back to the world's frame
      lr = r;
      inv_d = inverse3(r.d);
      a = sqrlen(r.d);
      in_instance = false;
    } else if (INST && ref_kind(cur) == LEAF_COMPLEX && (sc.complexes[ref_index(cur)].flags & CXF_INSTANCE)) {
      const Complex cx = sc.complexes[ref_index(cur)];
      cnt.template inc<ST_TRANSLATE>(cx.n_translate);
      cnt.template inc<ST_ROTATE>(cx.n_rotate);
      if constexpr (sizeof(R) == 4) {
        const ComplexXf<R>& xf = sc.complex_xf[ref_index(cur)];
        lr.o = xf_point(xf.m_full, xf.b_full, r.o);
        lr.d = xf_dir(xf.m_full, r.d);
      } else {
        for (uint32_t k = 0; k < cx.op_count; ++k) {  // hittable.rs:365, 375-377, outermost first
          const Op<R> op = sc.ops[cx.op_first + k];
          if (sc.op_tag[cx.op_first + k] == OP_TRANSLATE) {
            lr.o = lr.o - Vec3<R>(op.d[0], op.d[1], op.d[2]);
          } else {
            const Quat<R> q{op.d[0], op.d[1], op.d[2], op.d[3]};
            lr.o = qrotate(lr.o, q);
            lr.d = qrotate(lr.d, q);
          }
        }
      }
      inv_d = inverse3(lr.d);
      a = sqrlen(lr.d);
      in_instance = true;
      stack[sp * stride] = REF_MARKER;
      ++sp;
      cur = cx.body;  // a node: the sub-tree's root, its box tested against the mapped ray like any other
      continue;
    } else {
      R t;
      bool h;
      uint32_t leaf = cur;
      if (INST && in_instance) leaf = sc.leaf_by_rank[leaf_rank(sc, cur)];  // this leaf as a complex of its own (f32: or baked)
      if constexpr (sizeof(R) == 4) {
        // f32: a leaf of an instance is tested as every other f32 walk tests it — its synthesized complex (or baked
        // sphere) against the WORLD ray — so that all f32 walks compute the same bits for it; the mapped ray only
        // culls (the sub-tree's boxes).  The f64 walk below is literal: the primitive against the mapped ray.
        if (INST && in_instance)
          h = leaf_t_unified(sc, leaf, r, inverse3(r.d), sqrlen(r.d), t_min, closest, rng, t, cnt);
        else
          h = leaf_t_unified(sc, cur, lr, inv_d, a, t_min, closest, rng, t, cnt);
      } else if (ref_kind(cur) == LEAF_COMPLEX)
        h = complex_t(sc, ref_index(cur), lr, t_min, closest, rng, t, cnt);
      else
        h = prim_t(sc, cur, lr, inv_d, a, t_min, closest, t, cnt);
      // hittable.rs:405-411: the first record found is kept whatever its t (a NaN t from a NaN ray included — the
      // reference returns it, and every later `right.t < left.t` is then false); a later one must be strictly closer.
      // KNOWN GAP (NaN t only, never seen in any sweep): in the recursion a NaN-t record that is the first one found
      // inside a RIGHT subtree becomes that subtree's t_max, so the rest of that subtree is discarded, and the
      // subtree's answer (NaN) then loses against the left record; here `closest` stays finite across that subtree, so
      // a later leaf of it with t < closest is still accepted.  The two agree whenever the NaN record is the walk's
      // first record (handled above) or no NaN t arises at all (every finite ray against finite geometry).
      if (h && (best == REF_NONE || t < closest)) {
        closest = t;
        best = leaf;
      }
    }
    if (sp == 0) break;
    --sp;
    cur = stack[sp * stride];
  }
}

template <typename R, bool STATS>
RT_DEV void traverse(const SceneView<R>& sc, const RayT<R>& r, R t_min, Rng& rng, uint32_t* stack,
                     uint32_t stride, R& closest, uint32_t& best, Cnt<STATS>& cnt) {
  if (sc.has_instances)  // wave-uniform (a kernel argument)
    traverse_impl<true>(sc, r, t_min, rng, stack, stride, closest, best, cnt);
  else
    traverse_impl<false>(sc, r, t_min, rng, stack, stride, closest, best, cnt);
}

// ORDERED walk (f32 production path for scenes too large for the leaf list).
//
// The reference always descends left first, so on a large scene it wades through hundreds of
// nodes behind the nearest hit.  The result it computes is nevertheless order-independent: every
// leaf yields a candidate t that does not depend on the t_max it is given (sphere: nearest root
// >= t_min; cube: entry t; medium: boundary entry + a draw KEYED by the object), the winner is the
// smallest candidate, and among equal candidates the leaf reached first by the left-first DFS,
// i.e. the one with the smallest DFS rank.  So: visit the nearer child first, cull popped entries
// whose box entry is beyond the current closest, keep the bounds inclusive (next float above
// `closest`) and break exact ties by the stored DFS rank — same winner as hittable.rs:400-412,
// a fraction of the node visits.
template <typename R> RT_DEV uint32_t leaf_rank(const SceneView<R>& sc, uint32_t ref) {
  const uint32_t k = ref_kind(ref), i = ref_index(ref);
  if (k == LEAF_SPHERE) return sc.sphere_attr[i].rank;
  if (k == LEAF_MSPHERE) return sc.msphere_attr[i].rank;
  if (k == LEAF_CUBE) return sc.cube_attr[i].rank;
  return sc.complexes[i].rank;
}
RT_DEV float next_up(float x) {  // smallest float above a positive finite x; +inf stays +inf
  return x < Consts<float>::inf ? __uint_as_float(__float_as_uint(x) + 1u) : x;
}

// The walk as a resumable state machine: the render kernel advances every lane of a wave one step at
// a time and hands lanes whose walk has finished a new ray while the others keep walking (their
// state stays in registers, their stack in LDS).
struct OrderedWalk {
  // A pending entry's box-entry distance is kept in 16 bits (the float's upper half, i.e. rounded
  // toward zero: never above the true distance, so culling by it stays conservative); with the 32-bit
  // reference that is 6 B of LDS per stack level and lane instead of 8.
  static RT_DEV uint16_t entry_bits(float t) { return (uint16_t)(__float_as_uint(t) >> 16); }
  static RT_DEV float entry_float(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
  uint32_t cur, sp, best;
  float closest;
  Vec3<float> inv_d;
  float a;

  // returns false if there is nothing to walk (empty world)
  RT_DEV bool start(const SceneView<float>& sc, const RayT<float>& r) {
    closest = Consts<float>::inf;
    best = REF_NONE;
    cur = sc.root2;
    sp = 0;
    inv_d = inverse3(r.d);
    a = sqrlen(r.d);
    return cur != REF_NONE;
  }

  // one node (both child boxes) or one leaf test, then the next entry; false when the walk is over
  template <bool STATS>
  RT_DEV bool step(const SceneView<float>& sc, const RayT<float>& r, float t_min, Rng& rng, uint32_t* stack,
                   uint16_t* stack_t, uint32_t stride, Cnt<STATS>& cnt) {
    const float t_incl = next_up(closest);
    if (!(cur & REF_LEAF)) {
      cnt.template inc<ST_NODES>(2);
      const Node2 n = sc.nodes2[cur];
      float l0 = t_min, l1 = t_incl, r0 = t_min, r1 = t_incl;
      const bool hl = slab_hit(n.lbox, n.lbox + 3, r.o, inv_d, l0, l1);
      const bool hr = slab_hit(n.rbox, n.rbox + 3, r.o, inv_d, r0, r1);
      if (hl && hr) {
        const bool left_first = l0 <= r0;
        stack[sp * stride] = left_first ? n.right : n.left;
        stack_t[sp * stride] = entry_bits(left_first ? r0 : l0);
        ++sp;
        cur = left_first ? n.left : n.right;
        return true;
      }
      if (hl || hr) {
        cur = hl ? n.left : n.right;
        return true;
      }
    } else {
      float t;
      const bool h = leaf_t_unified(sc, cur, r, inv_d, a, t_min, t_incl, rng, t, cnt);
      if (h) {
        bool take = t < closest;
        if (t == closest && best != REF_NONE) take = leaf_rank(sc, cur) < leaf_rank(sc, best);
        if (take) {
          closest = t;
          best = cur;
        }
      }
    }
    return pop(stack, stack_t, stride);
  }

  // next stack entry whose box entry is not beyond the closest hit; false when the stack is empty
  RT_DEV bool pop(const uint32_t* stack, const uint16_t* stack_t, uint32_t stride) {
    for (;;) {
      if (sp == 0) return false;
      --sp;
#ifndef RT_POP_SPLIT_READS
      // both halves of the entry are requested together (one LDS round trip per pop; left to itself the compiler
      // reads the reference only after the distance has passed the test: two dependent round trips)
      uint32_t c = stack[sp * stride];
      uint32_t tb = stack_t[sp * stride];
      asm volatile("" : "+v"(c), "+v"(tb));
      cur = c;
      if (!(entry_float((uint16_t)tb) > closest)) return true;
#else
      cur = stack[sp * stride];
      if (!(entry_float(stack_t[sp * stride]) > closest)) return true;
#endif
    }
  }

  // The same walk in two halves, so that a wave can run its lanes' node steps together and their
  // (rarer, costlier) leaf tests together: node_step needs `cur` to be an inner node, leaf_step a leaf.
  template <bool STATS>
  RT_DEV bool node_step(const SceneView<float>& sc, const RayT<float>& r, float t_min, uint32_t* stack, uint16_t* stack_t,
                        uint32_t stride, Cnt<STATS>& cnt) {
    const float t_incl = next_up(closest);
    cnt.template inc<ST_NODES>(2);
    const Node2 n = sc.nodes2[cur];
    float l0 = t_min, l1 = t_incl, r0 = t_min, r1 = t_incl;
    const bool hl = slab_hit(n.lbox, n.lbox + 3, r.o, inv_d, l0, l1);
    const bool hr = slab_hit(n.rbox, n.rbox + 3, r.o, inv_d, r0, r1);
    // select-only bookkeeping (one divergent region, the pop): the far child is written to the free
    // slot `sp` whether or not it is kept (a node at level L has sp <= L - 1 < height: in bounds)
    const bool left_first = hl && (!hr || l0 <= r0);
    stack[sp * stride] = left_first ? n.right : n.left;
    stack_t[sp * stride] = entry_bits(left_first ? r0 : l0);
    sp += (hl && hr) ? 1u : 0u;
    cur = left_first ? n.left : n.right;
    if (hl || hr) return true;
    return pop(stack, stack_t, stride);
  }
  template <bool STATS>
  RT_DEV bool leaf_step(const SceneView<float>& sc, const RayT<float>& r, float t_min, Rng& rng, const uint32_t* stack,
                        const uint16_t* stack_t, uint32_t stride, Cnt<STATS>& cnt) {
    const float t_incl = next_up(closest);
    float t;
    const bool h = leaf_t_unified(sc, cur, r, inv_d, a, t_min, t_incl, rng, t, cnt);
    if (h) {
      bool take = t < closest;
      if (t == closest && best != REF_NONE) take = leaf_rank(sc, cur) < leaf_rank(sc, best);
      if (take) {
        closest = t;
        best = cur;
      }
    }
    return pop(stack, stack_t, stride);
  }
};

// The ordered walk in any precision, run to completion (what the f64 kernel walks on scenes too large for the leaf
// list): the same tree — Node2's f32 boxes are rounded OUTWARD from the f64 bounds, so testing them in R arithmetic
// (a float converts to a double exactly) only culls what no leaf below can be hit in; the leaves are tested by the
// literal R-precision code (prim_t / complex_t, an instance's leaves as their synthesized complexes).  By the argument
// above the winner, its t and the keyed draws are those of the reference-order walk bit for bit; only the work
// counters (nodes visited, leaves tested) differ.  Phased like traverse(): node steps together, leaf tests together.
RT_DEV double next_up(double x) {
  return x < Consts<double>::inf ? __longlong_as_double(__double_as_longlong(x) + 1ll) : x;
}
template <typename R> RT_DEV uint16_t entry_bits_below(R t) {  // a positive distance in 16 bits, never above t
  float f = (float)t;
  if constexpr (sizeof(R) == 8) {
    if ((double)f > t) f = __uint_as_float(__float_as_uint(f) - 1u);  // (the conversion rounded up)
  }
  return OrderedWalk::entry_bits(f);
}
// ... as a resumable state machine (like OrderedWalk, in the caller's precision): the f64 kernel advances every lane of
// a wave by node steps and leaf steps and lets the lanes whose walk is over go on to their vertex while the others keep
// walking (state in registers, stack in LDS).
template <typename R> struct OrderedWalkR {
  uint32_t cur, sp, best;
  R closest;
  Vec3<R> inv_d;
  R a;
  RT_DEV bool start(const SceneView<R>& sc, const RayT<R>& r) {  // false: nothing to walk (empty world)
    closest = Consts<R>::inf;
    best = REF_NONE;
    cur = sc.root2;
    sp = 0;
    inv_d = inverse3(r.d);
    a = sqrlen(r.d);
    return cur != REF_NONE;
  }
  RT_DEV bool pop(const uint32_t* stack, const uint16_t* stack_t, uint32_t stride) {  // next entry not entered beyond the closest hit
    for (;;) {
      if (sp == 0) return false;
      --sp;
      uint32_t c = stack[sp * stride];  // (both halves requested together: see OrderedWalk::pop)
      uint32_t tb = stack_t[sp * stride];
      asm volatile("" : "+v"(c), "+v"(tb));
      cur = c;
      if (!((R)OrderedWalk::entry_float((uint16_t)tb) > closest)) return true;
    }
  }
  template <bool STATS>  // `cur` is an inner node
  RT_DEV bool node_step(const SceneView<R>& sc, const RayT<R>& r, R t_min, uint32_t* stack, uint16_t* stack_t, uint32_t stride,
                        Cnt<STATS>& cnt) {
    cnt.template inc<ST_NODES>(2);
    const Node2 n = sc.nodes2[cur];
    const R t_incl = next_up(closest);
    R lb[6], rb[6];
    for (int c = 0; c < 6; ++c) {
      lb[c] = (R)n.lbox[c];
      rb[c] = (R)n.rbox[c];
    }
    R l0 = t_min, l1 = t_incl, r0 = t_min, r1 = t_incl;
    const bool hl = slab_hit(lb, lb + 3, r.o, inv_d, l0, l1);
    const bool hr = slab_hit(rb, rb + 3, r.o, inv_d, r0, r1);
    const bool left_first = hl && (!hr || l0 <= r0);
    if (hl && hr) {
      stack[sp * stride] = left_first ? n.right : n.left;
      stack_t[sp * stride] = entry_bits_below(left_first ? r0 : l0);
      ++sp;
    }
    cur = left_first ? n.left : n.right;
    if (hl || hr) return true;
    return pop(stack, stack_t, stride);
  }
  template <bool STATS>  // `cur` is a leaf
  RT_DEV bool leaf_step(const SceneView<R>& sc, const RayT<R>& r, R t_min, Rng& rng, const uint32_t* stack, const uint16_t* stack_t,
                        uint32_t stride, Cnt<STATS>& cnt) {
    const R t_incl = next_up(closest);
    R t;
    bool h;
    if constexpr (sizeof(R) == 4)
      h = leaf_t_unified(sc, cur, r, inv_d, a, t_min, t_incl, rng, t, cnt);
    else if (ref_kind(cur) == LEAF_COMPLEX)
      h = complex_t(sc, ref_index(cur), r, t_min, t_incl, rng, t, cnt);
    else
      h = prim_t(sc, cur, r, inv_d, a, t_min, t_incl, t, cnt);
    if (h) {
      bool take = t < closest;
      if (t == closest && best != REF_NONE) take = leaf_rank(sc, cur) < leaf_rank(sc, best);
      if (take) {
        closest = t;
        best = cur;
      }
    }
    return pop(stack, stack_t, stride);
  }
};
// run to completion (A/B form: RT_F64_ORDERED_NO_REFILL)
template <typename R, bool STATS>
RT_DEV void traverse_ordered(const SceneView<R>& sc, const RayT<R>& r, R t_min, Rng& rng, uint32_t* stack, uint16_t* stack_t,
                             uint32_t stride, R& closest, uint32_t& best, Cnt<STATS>& cnt) {
  OrderedWalkR<R> w;
  bool walking = w.start(sc, r);
  while (walking) {
    while (walking && !(w.cur & REF_LEAF)) walking = w.node_step(sc, r, t_min, stack, stack_t, stride, cnt);
    if (!walking) break;
    walking = w.leaf_step(sc, r, t_min, rng, stack, stack_t, stride, cnt);
  }
  closest = w.closest;
  best = w.best;
}

// Small scenes (a handful of leaves): test every leaf in BVH left-first DFS order with the
// same shrinking t_max and the same strict `<` — exactly the reference's List arm
// (hittable.rs:387-399).  A leaf that the BVH would have pruned can only be hit outside
// (t_min, closest), which its own test rejects (and a medium returns before drawing), so the
// winner and the draw sequence equal the BVH walk's (rt_scene_create verifies that every node
// box really bounds its leaves before enabling this path).  The loop counter is wave-uniform:
// every lane tests the same leaf at the same time (no traversal divergence), and the leaf
// records sit in LDS in a fixed layout (no dependent loads).
template <bool STATS>
RT_DEV bool lin_prim_t(const float* rec, uint32_t flags, Vec3<float> o, Vec3<float> d, Vec3<float> inv_d, float a,
                       float t_min, float t_max, float& t_out, Cnt<STATS>& cnt) {
  if (flags & LINF_CUBE) {
    cnt.template inc<ST_CUBE_TESTS>();
    float lo = t_min, hi = t_max;
    if (!slab_hit(rec + LIN_PRIM, rec + LIN_PRIM + 3, o, inv_d, lo, hi)) return false;
    cnt.template inc<ST_CUBE_HITS>();
    t_out = lo;
    return true;
  }
  cnt.template inc<ST_SPHERE_TESTS>();
  const bool h = sphere_t(o, d, a, Vec3<float>(rec[LIN_PRIM], rec[LIN_PRIM + 1], rec[LIN_PRIM + 2]), rec[LIN_PRIM + 3],
                          t_min, t_max, t_out);
  if (h) cnt.template inc<ST_SPHERE_HITS>();
  return h;
}

// MEDIA = false compiles the loop without the constant-medium arm, so the RNG state is not
// loop-carried (no per-iteration register copies) in scenes that have no medium.
// the three LINF_* bits of every record, bit k = record k: kernel arguments (SceneView), i.e. scalar registers
struct LinMasks {
  uint32_t xf, cube, medium;
};
template <typename SV> RT_DEV LinMasks lin_masks(const SV& sc) {
  return LinMasks{sc.linear_xf_mask, sc.linear_cube_mask, sc.linear_medium_mask};
}
template <bool STATS, bool MEDIA>
RT_DEV void traverse_linear(const float* lin, uint32_t n_linear, LinMasks masks, const RayT<float>& r, float t_min, Rng& rng,
                            float& closest, uint32_t& best_k, Cnt<STATS>& cnt) {
  closest = Consts<float>::inf;
  best_k = REF_NONE;
  const Vec3<float> inv_d0 = inverse3(r.d);
  const float a0 = sqrlen(r.d);
  for (uint32_t k = 0; k < n_linear; ++k) {
    const float* rec = lin + k * LIN_DWORDS;
    // wave-uniform, from scalar registers (reading rec[LIN_FLAGS] costs an LDS round trip and a
    // v_readfirstlane per leaf before the first branch can be taken)
    const uint32_t flags = ((masks.xf >> k) & 1u) * LINF_XF | ((masks.cube >> k) & 1u) * LINF_CUBE |
                           ((masks.medium >> k) & 1u) * LINF_MEDIUM;
    float t;
    bool h;
    // one leaf in a given frame; instantiated twice (world frame / the record's rigid map) so the
    // common world-frame case carries no register copies of the ray
    auto test = [&](const Vec3<float> o, const Vec3<float> d, const Vec3<float> inv_d, const float a) {
      if (!MEDIA || !(flags & LINF_MEDIUM)) return lin_prim_t(rec, flags, o, d, inv_d, a, t_min, closest, t, cnt);
      // hittable.rs:319-349
      cnt.template inc<ST_MEDIUM>();
      float t1, t2;
      cnt.template inc<ST_MEDIUM_PROBES>();
      if (!lin_prim_t(rec, flags, o, d, inv_d, a, -Consts<float>::inf, Consts<float>::inf, t1, cnt)) return false;
      cnt.template inc<ST_MEDIUM_PROBES>();
      if (!lin_prim_t(rec, flags, o, d, inv_d, a, second_probe_start(t1), Consts<float>::inf, t2, cnt)) return false;
      t1 = rmax(t1, t_min);
      t2 = rmin(t2, closest);
      if (!(t1 < t2)) return false;
      t1 = rmax(t1, 0.0f);
      const float ray_length = length(r.d);  // rigid maps keep |d|
      const float hit_distance = rec[LIN_NID] * rng.keyed_log<float>(__float_as_uint(rec[LIN_OBJECT]));
      if (hit_distance > (t2 - t1) * ray_length) return false;
      t = medium_exit_t(t1, hit_distance, ray_length);
      return true;
    };
    if (flags & LINF_XF) {  // wave-uniform branch
      const Vec3<float> o = xf_point(rec + LIN_M, rec + LIN_B, r.o);
      const Vec3<float> d = xf_dir(rec + LIN_M, r.d);
      if constexpr (STATS) {
        cnt.template inc<ST_ROTATE>(__float_as_uint(rec[LIN_FLAGS]) >> 16 & 0xFF);
        cnt.template inc<ST_TRANSLATE>(__float_as_uint(rec[LIN_FLAGS]) >> 24);
      }
      h = test(o, d, inverse3(d), sqrlen(d));
    } else {
      h = test(r.o, r.d, inv_d0, a0);
    }
    if (h && t < closest) {
      closest = t;
      best_k = k;
    }
  }
}

// The same leaf list for ANY precision, over the scene arrays themselves (`sc` may be a copy of the scene view whose
// hot arrays point into LDS): every leaf of the tree in left-first DFS order, tested by the very code the
// reference-order walk runs at a leaf (prim_t / complex_t), with the same shrinking t_max and the same acceptance
// rule as traverse() — so the winner, its t (bit for bit) and the medium draws are the BVH walk's; what changes is the
// work: no node tests, and the leaves a node box would have pruned are tested (and rejected).  The loop counter and
// the leaf reference are wave-uniform: no traversal divergence, no stack.  This is what the f64 kernel walks on scenes
// the f32 kernels take the leaf list for (same eligibility: verified-conservative boxes, <= LIN_MAX_LEAVES leaves).
template <typename R, bool STATS>
RT_DEV void traverse_leaves(const SceneView<R>& sc, const RayT<R>& r, R t_min, Rng& rng, R& closest, uint32_t& best,
                            Cnt<STATS>& cnt) {
  closest = Consts<R>::inf;
  best = REF_NONE;
  const Vec3<R> inv_d = inverse3(r.d);
  const R a = sqrlen(r.d);
  for (uint32_t k = 0; k < sc.n_leaf_list; ++k) {
    const uint32_t ref = __builtin_amdgcn_readfirstlane(sc.leaf_list[k]);  // wave-uniform
    R t;
    bool h;
    if (ref_kind(ref) == LEAF_COMPLEX)
      h = complex_t(sc, ref_index(ref), r, t_min, closest, rng, t, cnt);
    else
      h = prim_t(sc, ref, r, inv_d, a, t_min, closest, t, cnt);
    if (h && (best == REF_NONE || t < closest)) {  // as traverse(): hittable.rs:405-411
      closest = t;
      best = ref;
    }
  }
}

// ---- the winner's HitRecord (hittable.rs:17-40) -------------------------------------------
template <typename R> struct HitRec {
  Vec3<R> p, normal;
  Vec2<R> uv;
  bool front_face;
  uint32_t material, object;
};

template <typename R>
RT_DEV void face_normal(Vec3<R> d, Vec3<R> outward, HitRec<R>& h) {  // hittable.rs:26-36
  h.front_face = dot(d, outward) < R(0);
  h.normal = h.front_face ? outward : -outward;
}

// hittable.rs:233-249: the "outward" normal by which face rel_p lies within epsilon of (Q1: it
// points inward), else the (0.5,0.5,0.5) fallback (Q2)
template <typename R> RT_DEV Vec3<R> cube_normal(Vec3<R> rel) {
  const R eps = R(0.002);
  if (rel.x < eps) return Vec3<R>(1, 0, 0);
  if (rel.x > R(1) - eps) return Vec3<R>(-1, 0, 0);
  if (rel.y < eps) return Vec3<R>(0, 1, 0);
  if (rel.y > R(1) - eps) return Vec3<R>(0, -1, 0);
  if (rel.z < eps) return Vec3<R>(0, 0, 1);
  if (rel.z > R(1) - eps) return Vec3<R>(0, 0, -1);
  if constexpr (sizeof(R) == 4)  // a literal: the compiler does not fold the hardware rsq and keeps (then spills) a register for it
    return Vec3<R>(0.57735026f, 0.57735026f, 0.57735026f);
  else
    return normalize(Vec3<R>(R(0.5), R(0.5), R(0.5)));
}

template <typename R>
RT_DEV void prim_record(const SceneView<R>& sc, uint32_t ref, const RayT<R>& r, R t, bool force_uv, HitRec<R>& h) {
  uint32_t kind = ref_kind(ref), idx = ref_index(ref);
  h.p = madd(r.o, t, r.d);  // Ray::at, ray.rs:19-21
  if (kind == LEAF_CUBE) {  // hittable.rs:229-255
    Cube<R> c = sc.cubes[idx];
    Attr at = sc.cube_attr[idx];
    h.material = at.material;
    h.object = at.object;
    Vec3<R> p0(c.p0[0], c.p0[1], c.p0[2]), p1(c.p1[0], c.p1[1], c.p1[2]);
    Vec3<R> rel = (h.p - p0) / (p1 - p0);
    Vec3<R> n = cube_normal(rel);
    R az = rabs(n.z), ax = rabs(n.x), ay = rabs(n.y);
    h.uv.x = rel.x * az + rel.y * ax + rel.x * ay;
    h.uv.y = rel.y * az + rel.z * ax + rel.z * ay;
    (void)force_uv;
    face_normal(r.d, n, h);
  } else {  // hittable.rs:277-286 / 308-317
    Vec3<R> c;
    R radius;
    Attr at;
    if (kind == LEAF_SPHERE) {
      Real4<R> s = sc.spheres[idx];
      c = Vec3<R>(s.x, s.y, s.z);
      radius = s.w;
      at = sc.sphere_attr[idx];
    } else {
      MSphere<R> m = sc.mspheres[idx];
      c = moving_center(m, r.time);
      radius = m.radius;
      at = sc.msphere_attr[idx];
    }
    h.material = at.material;
    h.object = at.object;
    Vec3<R> outward = sphere_outward(r.o, r.d, t, h.p, c, radius);
    // get_sphere_uv costs an acos and an atan2: evaluate it only when some texture reads uv
    if (force_uv || (sc.materials[at.material].flags & MATF_NEEDS_UV)) {
      h.uv = get_sphere_uv(outward);
    } else {
      h.uv.x = R(0);
      h.uv.y = R(0);
    }
    face_normal(r.d, outward, h);
  }
}

template <typename R>
RT_DEV void make_hit_record(const SceneView<R>& sc, uint32_t ref, const RayT<R>& r, R t, bool force_uv,
                            HitRec<R>& h) {
  if constexpr (sizeof(R) == 4) {
    // f32: one shared copy of the primitive code for bare and wrapped primitives (lanes of a
    // wave that hit a wall and lanes that hit a rotated box run the same cube code together).
    const bool is_cx = ref_kind(ref) == LEAF_COMPLEX;
    bool is_medium = false;
    uint32_t body = ref, cx_material = 0, cx_object = 0;
    const R* m = nullptr;
    RayT<R> lr = r;
    if (is_cx) {
      const Complex cx = sc.complexes[ref_index(ref)];
      const ComplexXf<R>& xf = sc.complex_xf[ref_index(ref)];
      is_medium = cx.medium_pos != REF_NONE;
      body = cx.body;
      cx_material = cx.material;
      cx_object = cx.object;
      m = is_medium ? xf.m_outer : xf.m_full;
      lr.o = xf_point(xf.m_full, xf.b_full, r.o);
      lr.d = xf_dir(m, r.d);
    }
    if (!is_medium) {
      prim_record(sc, body, lr, t, force_uv, h);
    } else {  // hittable.rs:349-362: arbitrary normal (1,0,0), uv (0,0), in the medium's frame
      h.uv.x = R(0);
      h.uv.y = R(0);
      face_normal(lr.d, Vec3<R>(1, 0, 0), h);
      h.material = cx_material;
    }
    if (is_cx) {
      h.normal = xf_dir_transposed(m, h.normal);
      h.object = cx_object;
    }
    h.p = madd(r.o, t, r.d);  // rigid maps: the world-space point is o + t d in any frame
    return;
  }
  // f64: the literal wrapper chain (hittable.rs:364-386), in three steps so that the wave runs ONE copy of the
  // primitive's record code for bare and wrapped primitives alike (lanes at a wall and lanes at a rotated box run the
  // cube code together): into the body's frame, the body's record, back out.
  const bool is_cx = ref_kind(ref) == LEAF_COMPLEX;
  Complex cx{};
  RayT<R> lr = r;
  uint32_t n_outer = 0, body = ref;
  if (is_cx) {
    cx = sc.complexes[ref_index(ref)];
    n_outer = cx.medium_pos == REF_NONE ? cx.op_count : cx.medium_pos;
    body = cx.body;
    for (uint32_t k = 0; k < n_outer; ++k) {
      uint32_t tag = sc.op_tag[cx.op_first + k];
      Op<R> op = sc.ops[cx.op_first + k];
      if (tag == OP_TRANSLATE) {
        lr.o = lr.o - Vec3<R>(op.d[0], op.d[1], op.d[2]);
      } else {
        Quat<R> q{op.d[0], op.d[1], op.d[2], op.d[3]};
        lr.o = qrotate(lr.o, q);
        lr.d = qrotate(lr.d, q);
      }
    }
  }
  if (!is_cx || cx.medium_pos == REF_NONE) {
    prim_record(sc, body, lr, t, force_uv, h);
  } else {  // hittable.rs:349-362: arbitrary normal (1,0,0), uv (0,0), in the medium's frame
    h.p = madd(lr.o, t, lr.d);
    h.uv.x = R(0);
    h.uv.y = R(0);
    face_normal(lr.d, Vec3<R>(1, 0, 0), h);
    h.material = cx.material;
  }
  if (!is_cx) return;
  h.object = cx.object;
  // unwind the outer wrappers, innermost first (hittable.rs:369-372, 378-385)
  for (uint32_t k = n_outer; k-- > 0;) {
    uint32_t tag = sc.op_tag[cx.op_first + k];
    Op<R> op = sc.ops[cx.op_first + k];
    if (tag == OP_TRANSLATE) {
      h.p = h.p + Vec3<R>(op.d[0], op.d[1], op.d[2]);
    } else {
      Quat<R> qi = qinvert(Quat<R>{op.d[0], op.d[1], op.d[2], op.d[3]});
      h.p = qrotate(h.p, qi);
      h.normal = qrotate(h.normal, qi);
    }
  }
}

// HitRecord + material of the winning leaf from its LDS record (f32 leaf-list path).
RT_DEV void lin_hit_record(const SceneView<float>& sc, const float* rec, const RayT<float>& r, float t, HitRec<float>& h,
                           Material<float>& mat) {
  const uint32_t flags = __float_as_uint(rec[LIN_FLAGS]);
  Vec3<float> o = r.o, d = r.d;
  if (flags & LINF_XF) {
    o = xf_point(rec + LIN_M, rec + LIN_B, r.o);
    d = xf_dir(rec + LIN_M, r.d);
  }
  mat.tag = __float_as_uint(rec[LIN_MAT + 0]);
  mat.texture = __float_as_uint(rec[LIN_MAT + 1]);
  mat.flags = __float_as_uint(rec[LIN_MAT + 2]);
  mat.pad = 0;
  for (int i = 0; i < 4; ++i) mat.p[i] = rec[LIN_MAT + 4 + i];
  for (int i = 0; i < 3; ++i) mat.color[i] = rec[LIN_MAT + 8 + i];
  h.material = REF_NONE;
  h.object = __float_as_uint(rec[LIN_OBJECT]);
  h.uv.x = 0.0f;
  h.uv.y = 0.0f;
  if (flags & LINF_MEDIUM) {  // hittable.rs:349-362 (the medium op is outermost in this path's records)
    face_normal(r.d, Vec3<float>(1, 0, 0), h);
  } else if (flags & LINF_CUBE) {
    const Vec3<float> pl = madd(o, t, d);
    const Vec3<float> p0(rec[LIN_PRIM], rec[LIN_PRIM + 1], rec[LIN_PRIM + 2]);
    const Vec3<float> inv_ext(rec[LIN_INVEXT], rec[LIN_INVEXT + 1], rec[LIN_INVEXT + 2]);  // 1 / (p1 - p0)
    const Vec3<float> rel = (pl - p0) * inv_ext;
    const Vec3<float> n = cube_normal(rel);
    if (mat.flags & MATF_NEEDS_UV) {
      const float az = rabs(n.z), ax = rabs(n.x), ay = rabs(n.y);
      h.uv.x = rel.x * az + rel.y * ax + rel.x * ay;
      h.uv.y = rel.y * az + rel.z * ax + rel.z * ay;
    }
    face_normal(d, n, h);
  } else {
    const Vec3<float> pl = madd(o, t, d);
    const Vec3<float> c(rec[LIN_PRIM], rec[LIN_PRIM + 1], rec[LIN_PRIM + 2]);
    const Vec3<float> outward = sphere_outward(o, d, t, pl, c, rec[LIN_PRIM + 3]);
    if (mat.flags & MATF_NEEDS_UV) h.uv = get_sphere_uv(outward);
    face_normal(d, outward, h);
  }
  if ((flags & LINF_XF) && !(flags & LINF_MEDIUM)) h.normal = xf_dir_transposed(rec + LIN_M, h.normal);
  h.p = madd(r.o, t, r.d);
  (void)sc;
}

}  // namespace rt
