// Error-tracked restatement of the f32 walk, for rt_primary_hits (parity levels 1-2 of the f32 product).
//
// The f32 kernels decide "hit or miss", "which root", "which candidate is closer" on f32-rounded inputs (ray
// direction, centres, boxes, rigid maps).  Almost every such decision has a margin far above that rounding and then
// equals the f64 reference's (hittable.rs:208-415 evaluated in f64, alias.rs:4-7).  This header re-walks the
// reference-order tree with the SAME primitives (slab_hit, sphere_roots, xf_point ...: the product walks' bits) and
// carries a first-order forward error bound beside every t.  The bound's inputs are the rounding deltas of the
// inputs themselves — |x_f64 - x_f32| for the ray, the centres, the radii, the box corners, the mapped ray of a
// wrapped object, read from the f64 copy of the scene that rt_scene_create keeps on the device anyway — plus a few
// units of 2^-23 for the f32 arithmetic.  A decision whose operands are closer than their bounds, or a winner whose t
// may be off by more than 5e-6 relative, marks the pixel UNSURE; rt_primary_hits re-decides exactly those pixels
// with the f64 audit kernel (primary_fix_kernel, rt_f64.hip: contraction off, bit-comparable with the CPU oracle).
// The bounds drive flags only: no f64 quantity computed here ever reaches a decision or an output.
// The render kernels do not do this: their camera rays carry 24-bit jitter and have no f64 counterpart to agree
// with ray for ray.
#pragma once
#include "intersect.hpp"

namespace rt {

constexpr float AUD_EPS = 1.1920929e-7f;  // 2^-23
constexpr float AUD_SAFETY = 2.0f;        // a decision needs twice the sum of its operands' bounds as margin
constexpr float AUD_T_REL = 5e-6f;        // the winner's t: bound must stay below this (the bar is 1e-5)

RT_DEV float linf(Vec3<float> v) { return rmax(rabs(v.x), rmax(rabs(v.y), rabs(v.z))); }
RT_DEV Vec3<float> delta3(Vec3<double> a, Vec3<float> b) {  // |a - b| per component
  return Vec3<float>((float)fabs(a.x - (double)b.x), (float)fabs(a.y - (double)b.y), (float)fabs(a.z - (double)b.z));
}

struct AuditFrame {  // a ray in some frame, as the f32 kernels see it, with the rounding deltas of its origin / direction
  Vec3<float> o, d, inv_d;
  float a;             // |d|^2
  Vec3<float> eo, ed;  // |o64_k - o_k|, |d64_k - d_k| per component
};

// why a pixel is unsure (a mask, for the diagnostic histogram RT_AUDIT_REASONS=1 prints; any bit = re-decide)
constexpr uint32_t AUD_NODE = 1, AUD_CUBE = 2, AUD_SPHERE_HIT = 4, AUD_SPHERE_WINDOW = 8, AUD_MEDIUM = 16, AUD_TIE = 32,
                   AUD_ACCURACY = 64, AUD_DIRECTION = 128;

struct Cand {  // outcome of one leaf test
  bool hit;
  uint32_t unsure;  // AUD_* mask
  float t, e;       // candidate t and its absolute error bound
};

// Per-axis slab crossings of (p0, p1) in frame f, each with its bound; dp0 / dp1 = rounding deltas of the corners.
// Only crossings that can BE the interval's ends count: the entry is the largest near crossing, so a near crossing far
// below it (by more than both bounds) plays no part, however ill-conditioned it is (a big box's far face met at a
// shallow angle has a huge |t| and a huge bound, and nothing hangs on it); same for the exit and the far crossings.
struct SlabBounds {
  float e_entry, e_exit;      // bounds of the raw entry / exit
  float entry_raw, exit_raw;  // before clamping to a window
  bool bad;                   // some direction component is too small to trust its sign
};
RT_DEV SlabBounds slab_bounds(const float* p0, const float* p1, const float* dp0, const float* dp1, const AuditFrame& f) {
  SlabBounds b;
  b.bad = false;
  float near_t[3], far_t[3], near_e[3], far_e[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float dk = comp(f.d, k), ik = comp(f.inv_d, k), ok = comp(f.o, k);
    const float edk = comp(f.ed, k), eok = comp(f.eo, k);
    if (!(rabs(dk) > 4.0f * edk + 1e-30f)) b.bad = true;  // a component that rounding could flip or zero (or NaN)
    const float rel = edk * rabs(ik) + 3.0f * AUD_EPS;    // of 1/d_k, plus the subtraction, the rcp and the product
    const float t0 = (p0[k] - ok) * ik, t1 = (p1[k] - ok) * ik;
    const float e0 = rabs(ik) * (dp0[k] + eok) + rabs(t0) * rel;
    const float e1 = rabs(ik) * (dp1[k] + eok) + rabs(t1) * rel;
    const bool neg = ik < 0.0f;
    near_t[k] = neg ? t1 : t0;
    far_t[k] = neg ? t0 : t1;
    near_e[k] = neg ? e1 : e0;
    far_e[k] = neg ? e0 : e1;
  }
  b.entry_raw = rmax(near_t[0], rmax(near_t[1], near_t[2]));
  b.exit_raw = rmin(far_t[0], rmin(far_t[1], far_t[2]));
  b.e_entry = 0.0f;
  b.e_exit = 0.0f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {  // the binding crossing's bound ...
    if (near_t[k] == b.entry_raw) b.e_entry = rmax(b.e_entry, near_e[k]);
    if (far_t[k] == b.exit_raw) b.e_exit = rmax(b.e_exit, far_e[k]);
  }
  const float e_entry0 = b.e_entry, e_exit0 = b.e_exit;
#pragma unroll
  for (int k = 0; k < 3; ++k) {  // ... and every other crossing that could overtake it
    if (rfinite(near_t[k]) && near_t[k] + near_e[k] >= b.entry_raw - e_entry0) b.e_entry = rmax(b.e_entry, near_e[k]);
    if (rfinite(far_t[k]) && far_t[k] - far_e[k] <= b.exit_raw + e_exit0) b.e_exit = rmax(b.e_exit, far_e[k]);
  }
  return b;
}

// Is "the interval [max(entry, t_min), min(exit, t_max)] is empty" (slab_hit: hi <= lo) decided with margin?
RT_DEV bool slab_marginal(const SlabBounds& b, float t_min, float t_max, float e_tmax) {
  const bool lo_is_entry = b.entry_raw > t_min, hi_is_exit = b.exit_raw < t_max;
  const float lo = lo_is_entry ? b.entry_raw : t_min, hi = hi_is_exit ? b.exit_raw : t_max;
  // an end counts with its bound if it is, or could be, the binding one
  const float e_lo = (lo_is_entry || b.entry_raw + b.e_entry >= t_min) ? b.e_entry : 0.0f;
  const float e_hi = rmax((hi_is_exit || b.exit_raw - b.e_exit <= t_max) ? b.e_exit : 0.0f,
                          (!hi_is_exit || t_max - e_tmax <= b.exit_raw + b.e_exit) && rfinite(t_max) ? e_tmax : 0.0f);
  return !(rabs(hi - lo) > AUD_SAFETY * (e_lo + e_hi));
}

// A node box (aabb.rs:17-34): a cull decision against (t_min, closest).  The decision is slab_hit's.
// e_tmax = bound of t_max (the running closest).  Node boxes are the reference's f64 boxes rounded OUTWARD (rt_api.hip
// conv_lo / conv_hi); n64 is the f64 original.
RT_DEV bool audit_node(const Node<float>& n, const Node<double>& n64, const AuditFrame& f, float t_min, float t_max,
                       float e_tmax, uint32_t& unsure) {
  float lo = t_min, hi = t_max;
  const bool hit = slab_hit(n.lo, n.hi, f.o, f.inv_d, lo, hi);
  float dp0[3], dp1[3];
  for (int k = 0; k < 3; ++k) {
    dp0[k] = (float)fabs(n64.lo[k] - (double)n.lo[k]);
    dp1[k] = (float)fabs(n64.hi[k] - (double)n.hi[k]);
  }
  const SlabBounds b = slab_bounds(n.lo, n.hi, dp0, dp1, f);
  if (b.bad) unsure |= AUD_DIRECTION;
  if (slab_marginal(b, t_min, t_max, e_tmax)) {
    unsure |= AUD_NODE;
#ifdef RT_AUDIT_DEBUG
    if ((blockIdx.x & 7) == 0 && (threadIdx.x & 31) == 0)
      printf("node lo %g %g %g hi %g %g %g | entry %g +- %g exit %g +- %g | t_min %g t_max %g +- %g | d %g %g %g\n", n.lo[0], n.lo[1],
             n.lo[2], n.hi[0], n.hi[1], n.hi[2], b.entry_raw, b.e_entry, b.exit_raw, b.e_exit, t_min, t_max, e_tmax, f.d.x, f.d.y, f.d.z);
#endif
  }
  return hit;
}

// A cube leaf (hittable.rs:210-229): returns the ENTRY t only (Q2).
RT_DEV Cand audit_slab(const float* p0, const float* p1, const float* dp0, const float* dp1, const AuditFrame& f, float t_min,
                       float t_max, float e_tmax) {
  Cand c;
  float lo = t_min, hi = t_max;
  c.hit = slab_hit(p0, p1, f.o, f.inv_d, lo, hi);
  c.t = lo;
  const SlabBounds b = slab_bounds(p0, p1, dp0, dp1, f);
  c.unsure = (b.bad ? AUD_DIRECTION : 0u) | (slab_marginal(b, t_min, t_max, e_tmax) ? AUD_CUBE : 0u);
  // Q2: "the ray starts inside the slab" (entry t = t_min, fallback normal) against "enters at entry_raw"
  if (c.hit && !(rabs(b.entry_raw - t_min) > AUD_SAFETY * b.e_entry)) c.unsure |= AUD_CUBE;
  c.e = lo == t_min ? 0.0f : b.e_entry;  // t_min itself is exact
  return c;
}

// Sphere (hittable.rs:257-276): decision and roots from sphere_roots / sphere_pick (the product's code).
// Bounds, first order in the input deltas (ec: centre, er: radius, f.eo / f.ed: ray):
//   hit or miss:  q = r^2 - b^2 with b the distance of the centre from the ray's line, closest approach at
//                 t_c = -hb / a:  |delta b| <= |delta o| + |delta c| + |t_c| |delta d|,  |delta q| <= 2 |r| er + 2 b |delta b| + |delta b|^2
//   a root t:     F(t) = |o + t d - c|^2 - r^2 = 0  =>  delta t = -[n . (delta o + t delta d - delta c) - delta r] / (n . d)
//                 with n the unit normal at the root and |n . d| |r| = sqrt(disc)
// (the quadratic itself is solved in f64 from the f32 inputs, sphere_roots: its arithmetic adds ~2 ulp of f32 at the end)
// near_is_below: the window's lower end was derived from this sphere's own near root (a medium's second probe starts at
// hit1.t + 0.0001, hittable.rs:325, 331; in f32 at least one float above hit1.t, second_probe_start): "near root < lower
// end" then holds by construction and is no decision.
RT_DEV Cand audit_sphere(Vec3<float> c, float radius, float ec, float er, const AuditFrame& f, float t_min, float t_max,
                         float e_tmax, bool near_is_below = false) {
  Cand out;
  const SphereRoots roots = sphere_roots(f.o, f.d, f.a, c, radius);
  out.hit = sphere_pick(roots, t_min, t_max, out.t);
  out.e = 0.0f;
  out.unsure = 0;
  const double ocx = (double)f.o.x - (double)c.x, ocy = (double)f.o.y - (double)c.y, ocz = (double)f.o.z - (double)c.z;
  const double hb = ocx * f.d.x + ocy * f.d.y + ocz * f.d.z;
  const double oc2 = ocx * ocx + ocy * ocy + ocz * ocz;
  const double a = (double)f.d.x * f.d.x + (double)f.d.y * f.d.y + (double)f.d.z * f.d.z;
  const double r2 = (double)radius * (double)radius;
  const double disc = hb * hb - a * (oc2 - r2);
  const float q = (float)(disc / a);  // r^2 - b^2
  const float b = (float)sqrt(fmax(r2 - disc / a, 0.0));
  const float tc = (float)(-hb / a);
  const float eo = length(f.eo), ed = length(f.ed);  // |delta o|, |delta d| (Euclidean)
  const float eb = eo + ec + rabs(tc) * ed;
  const float eq = 2.0f * rabs(radius) * er + 2.0f * b * eb + eb * eb + 1e-12f * (float)r2;
  if (!(rabs(q) > AUD_SAFETY * eq)) {  // hit or miss hangs on rounding (also NaN)
    out.unsure = AUD_SPHERE_HIT;
    return out;
  }
  if (disc < 0.0) return out;  // a clear miss
  const float sq = (float)sqrt(disc);  // = |n . d| |r|
  auto root_err = [&](float root) {
    return rabs(radius) * (eo + ec + rabs(root) * ed + er) / sq + 4.0f * AUD_EPS * rabs(root);
  };
  // window tests of hittable.rs:266-273
  const float e_hi = rfinite(t_max) ? e_tmax : 0.0f;
  for (int k = 0; k < 2; ++k) {
    const float root = k == 0 ? roots.near_t : roots.far_t;
    const float er_ = root_err(root);
    if (!(k == 0 && near_is_below) && !(rabs(root - t_min) > AUD_SAFETY * er_)) out.unsure |= AUD_SPHERE_WINDOW;
    if (rfinite(t_max) && !(rabs(root - t_max) > AUD_SAFETY * (er_ + e_hi))) out.unsure |= AUD_SPHERE_WINDOW;
    if (k == 0 && !(root < t_min || root > t_max)) break;  // the near root is taken: the far one is never looked at
  }
  if (out.hit) out.e = root_err(out.t);
  return out;
}

// One leaf of any kind: mirrors leaf_t_unified (same frames, same probes, same medium arithmetic).  sc64 / r64: the
// f64 scene and ray, read for rounding deltas only.
// best_sphere: index of the current closest hit's sphere if that hit is a bare Sphere leaf, else REF_NONE — for the one
// tie that is STRUCTURAL: Shirley's fog ball is a glass Sphere plus a ConstantMedium around a second Sphere with the same
// centre and radius.  Once the glass is the closest hit, the medium's entry hit1.t and the clip bound t_max are the same
// function of the same inputs, equal bit for bit in f32 and in f64 alike (hittable.rs:332-335: t2 = t_max = t1, no hit).
RT_DEV Cand audit_leaf(const SceneView<float>& sc, const SceneView<double>& sc64, uint32_t ref, const RayT<float>& r,
                       const RayT<double>& r64, const AuditFrame& world, float t_min, float t_max, float e_tmax, Rng& rng,
                       uint32_t best_sphere) {
  uint32_t kind = ref_kind(ref), idx = ref_index(ref);
  AuditFrame f = world;
  bool medium = false, bare_body = false;
  float neg_inv_density = 0.0f;
  uint32_t object = 0;
  if (kind == LEAF_COMPLEX) {
    const Complex cx = sc.complexes[idx];
    bare_body = cx.n_rotate + cx.n_translate == 0;
    const ComplexXf<float>& xf = sc.complex_xf[idx];
    const ComplexXf<double>& xf64 = sc64.complex_xf[idx];
    f.o = xf_point(xf.m_full, xf.b_full, r.o);
    f.d = xf_dir(xf.m_full, r.d);
    f.inv_d = inverse3(f.d);
    f.a = sqrlen(f.d);
    // the mapped ray's deltas: against the f64 map of the f64 ray (the composed map equals the reference's chain of
    // Translate / Rotate wrappers, hittable.rs:364-386, to f64 rounding)
    f.eo = delta3(xf_point(xf64.m_full, xf64.b_full, r64.o), f.o);
    f.ed = delta3(xf_dir(xf64.m_full, r64.d), f.d);
    medium = cx.medium_pos != REF_NONE;
    neg_inv_density = xf.neg_inv_density;
    object = cx.object;
    kind = ref_kind(cx.body);
    idx = ref_index(cx.body);
  }
  const bool is_cube = kind == LEAF_CUBE;
  float p0[3] = {0, 0, 0}, p1[3] = {0, 0, 0}, dp0[3] = {0, 0, 0}, dp1[3] = {0, 0, 0};
  Vec3<float> c;
  float radius = 0.0f, ec = 0.0f, er = 0.0f;
  if (is_cube) {
    const Cube<float> q = sc.cubes[idx];
    const Cube<double> q64 = sc64.cubes[idx];
    for (int k = 0; k < 3; ++k) {
      p0[k] = q.p0[k];
      p1[k] = q.p1[k];
      dp0[k] = (float)fabs(q64.p0[k] - (double)q.p0[k]);
      dp1[k] = (float)fabs(q64.p1[k] - (double)q.p1[k]);
    }
  } else if (kind == LEAF_SPHERE) {
    const Real4<float> sp = sc.spheres[idx];
    const Real4<double> sp64 = sc64.spheres[idx];
    c = Vec3<float>(sp.x, sp.y, sp.z);
    radius = sp.w;
    ec = length(delta3(Vec3<double>(sp64.x, sp64.y, sp64.z), c));
    er = (float)fabs(sp64.w - (double)sp.w);
  } else {
    const MSphere<float> m = sc.mspheres[idx];
    const MSphere<double> m64 = sc64.mspheres[idx];
    c = moving_center(m, r.time);
    radius = m.radius;
    ec = length(delta3(moving_center(m64, r64.time), c));
    er = (float)fabs(m64.radius - (double)m.radius);
  }
  auto probe = [&](float lo, float hi, float e_hi, bool second = false) {
    return is_cube ? audit_slab(p0, p1, dp0, dp1, f, lo, hi, e_hi) : audit_sphere(c, radius, ec, er, f, lo, hi, e_hi, second);
  };
  if (!medium) return probe(t_min, t_max, e_tmax);
  // hittable.rs:319-349
  Cand out;
  out.hit = false;
  out.t = 0.0f;
  out.e = 0.0f;
  const Cand h1 = probe(-Consts<float>::inf, Consts<float>::inf, 0.0f);
  out.unsure = h1.unsure;
  if (!h1.hit) return out;
  const Cand h2 = probe(second_probe_start(h1.t), Consts<float>::inf, 0.0f, true);
  out.unsure |= h2.unsure;
  if (!h2.hit) return out;
  if (bare_body && kind == LEAF_SPHERE && best_sphere != REF_NONE && h1.t == t_max) {
    const Real4<double> mine = sc64.spheres[idx], other = sc64.spheres[best_sphere];
    if (mine.x == other.x && mine.y == other.y && mine.z == other.z && mine.w == other.w) return out;  // structural tie: no hit
  }
  float t1 = rmax(h1.t, t_min), t2 = rmin(h2.t, t_max);
  const float e1 = h1.t > t_min ? h1.e : 0.0f;
  const float e2 = h2.t < t_max ? h2.e : (rfinite(t_max) ? e_tmax : 0.0f);
  if (!(rabs(h1.t - t_min) > AUD_SAFETY * h1.e)) out.unsure |= AUD_MEDIUM;
  if (rfinite(t_max) && !(rabs(h2.t - t_max) > AUD_SAFETY * (h2.e + e_tmax))) out.unsure |= AUD_MEDIUM;
  if (!(rabs(t2 - t1) > AUD_SAFETY * (e1 + e2))) out.unsure |= AUD_MEDIUM;
  if (t1 >= t2) return out;
  if (t1 != t_min && !(rabs(t1) > AUD_SAFETY * e1)) out.unsure |= AUD_MEDIUM;  // the clamp at 0
  t1 = rmax(t1, 0.0f);
  const float ray_length = length(r.d);
  const float inside = (t2 - t1) * ray_length;
  const float hit_distance = neg_inv_density * rng.template keyed_log<float>(object);
  const float e_inside = (e1 + e2) * ray_length + 8.0f * AUD_EPS * inside;
  const float e_hd = 8.0f * AUD_EPS * rabs(hit_distance);  // ln from all 32 bits of the draw (Rng::keyed_log): f32 rounding only
  if (!(rabs(hit_distance - inside) > AUD_SAFETY * (e_inside + e_hd))) out.unsure |= AUD_MEDIUM;
  if (hit_distance > inside) return out;
  out.hit = true;
  out.t = medium_exit_t(t1, hit_distance, ray_length);
  out.e = e1 + e_hd * __builtin_amdgcn_rcpf(ray_length) + 4.0f * AUD_EPS * rabs(out.t);
  return out;
}

// hittable.rs:400-412 as in traverse(), with bounds.  Returns non-zero (AUD_* mask) if the pixel must be re-decided in f64.
RT_DEV uint32_t traverse_audit(const SceneView<float>& sc, const SceneView<double>& sc64, const RayT<float>& r,
                           const RayT<double>& r64, float t_min, Rng& rng, uint32_t* stack, uint32_t stride, float& closest,
                           uint32_t& best, float& e_best) {
  closest = Consts<float>::inf;
  best = REF_NONE;
  e_best = 0.0f;
  uint32_t cur = sc.root;
  if (cur == REF_NONE) return false;
  AuditFrame world;
  world.o = r.o;
  world.d = r.d;
  world.inv_d = inverse3(r.d);
  world.a = sqrlen(r.d);
  world.eo = delta3(r64.o, r.o);
  world.ed = delta3(r64.d, r.d);
  uint32_t unsure = 0, best_sphere = REF_NONE;
  uint32_t sp = 0;
  for (;;) {
    if (!(cur & REF_LEAF)) {
      const Node<float> n = sc.nodes[cur];
      if (audit_node(n, sc64.nodes[cur], world, t_min, closest, e_best, unsure)) {
        stack[sp * stride] = n.right;
        ++sp;
        cur = n.left;
        continue;
      }
    } else if (sc.has_instances && ref_kind(cur) == LEAF_COMPLEX && (sc.complexes[ref_index(cur)].flags & CXF_INSTANCE)) {
      // an instance (a sub-tree behind a rigid map): not audited — a ray whose walk reaches one is re-decided in f64
      unsure |= AUD_ACCURACY;
    } else {
      const Cand c = audit_leaf(sc, sc64, cur, r, r64, world, t_min, closest, e_best, rng, best_sphere);
      unsure |= c.unsure;
      if (c.hit) {
        if (rfinite(closest) && !(rabs(c.t - closest) > AUD_SAFETY * (c.e + e_best))) unsure |= AUD_TIE;  // a near tie
        if (c.t < closest) {
          closest = c.t;
          e_best = c.e;
          best = cur;
          best_sphere = ref_kind(cur) == LEAF_SPHERE ? ref_index(cur) : REF_NONE;
        }
      }
    }
    if (sp == 0) break;
    --sp;
    cur = stack[sp * stride];
  }
  if (best != REF_NONE && !(e_best <= AUD_T_REL * rabs(closest))) unsure |= AUD_ACCURACY;
  return unsure;
}

// The winner's record can also hang on rounding: a cube's face pick compares rel_p with 0.002 (hittable.rs:233-249).
RT_DEV bool audit_record_unsure(const SceneView<float>& sc, const SceneView<double>& sc64, uint32_t ref, const RayT<float>& r,
                                const RayT<double>& r64, float t, float e_t) {
  uint32_t kind = ref_kind(ref), idx = ref_index(ref);
  Vec3<float> o = r.o, d = r.d;
  Vec3<float> e_o = delta3(r64.o, r.o), e_d = delta3(r64.d, r.d);
  if (kind == LEAF_COMPLEX) {
    const Complex cx = sc.complexes[idx];
    if (cx.medium_pos != REF_NONE) return false;  // fixed normal (1,0,0)
    const ComplexXf<float>& xf = sc.complex_xf[idx];
    const ComplexXf<double>& xf64 = sc64.complex_xf[idx];
    o = xf_point(xf.m_full, xf.b_full, r.o);
    d = xf_dir(xf.m_full, r.d);
    e_o = delta3(xf_point(xf64.m_full, xf64.b_full, r64.o), o);
    e_d = delta3(xf_dir(xf64.m_full, r64.d), d);
    kind = ref_kind(cx.body);
    idx = ref_index(cx.body);
  }
  if (kind != LEAF_CUBE) return false;
  const Cube<float> q = sc.cubes[idx];
  const Cube<double> q64 = sc64.cubes[idx];
  const Vec3<float> p = madd(o, t, d);
  const float eps = 0.002f;
  for (int k = 0; k < 3; ++k) {
    const float ep = comp(e_o, k) + rabs(t) * comp(e_d, k) + e_t * rabs(comp(d, k)) +
                     2.0f * AUD_EPS * (rabs(comp(o, k)) + rabs(t * comp(d, k)));
    const float ext = q.p1[k] - q.p0[k];
    const float rel = (comp(p, k) - q.p0[k]) / ext;
    const float dp = (float)fmax(fabs(q64.p0[k] - (double)q.p0[k]), fabs(q64.p1[k] - (double)q.p1[k]));
    const float er = (ep + 3.0f * dp) / rabs(ext) + 4.0f * AUD_EPS * (rabs(rel) + 1.0f);
    if (!(rabs(rel - eps) > AUD_SAFETY * er) || !(rabs(rel - (1.0f - eps)) > AUD_SAFETY * er)) return true;
  }
  return false;
}

}  // namespace rt
