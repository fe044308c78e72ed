// The path-tracing megakernel and its helpers: the device form of the render closure
// (main.rs:154-167), Camera::get_ray (cam.rs:63-71), ray_color (main.rs:65-105),
// Material::{scatter, scattering_pdf, emit} (material.rs:57-133), PDF::{value, generate}
// (pdf.rs:13-39), Hittable::{pdf_value, random} (hittable.rs:477-515) and Texture::value
// (texture.rs:33-69).
//
// Execution shape (CDNA4): one 64-lane wavefront owns an 8x8 pixel tile, four waves make a
// 16x16 workgroup tile.  Each lane runs ONE flat loop whose iteration is one path vertex;
// a lane whose path ends regenerates its next sample inside the same loop, so lanes drift
// apart in sample index but stay busy (the recursion of ray_color becomes
// throughput/radiance registers).  Per-lane BVH stack lives in LDS, column-major by lane.
#pragma once
#include <type_traits>

#include "audit.hpp"

namespace rt {

constexpr int BLOCK = 256;
// waves per SIMD the register allocator must leave room for (measured on MI355X):
// reference-order f32 kernel: 4: 25.0, 5: 28.0, 6: 29.4, 7: 26.6, 8: 20.9 Grays/s on simple_light (the reference's
// default scene), 6 is also +25..37 % over 4 on the forced reference walk of the BVH scenes; the f64 audit
// kernels stay at 4; the leaf-list kernel gains at 6; the refill kernel is best at 5 (spills at 6)
#ifndef RT_RENDER_MIN_WAVES
#define RT_RENDER_MIN_WAVES 6
#endif
#ifndef RT_RENDER_F64_MIN_WAVES
#define RT_RENDER_F64_MIN_WAVES 4
#endif
#ifndef RT_LINEAR_MIN_WAVES
#define RT_LINEAR_MIN_WAVES 6
#endif
#ifndef RT_REFILL_MIN_WAVES
#define RT_REFILL_MIN_WAVES 5
#endif

// ---- EXTENSION (rt.h RT_TEX_NOISE, no counterpart in the reference): Perlin noise of "The Next Week" ---------
// tab: 256 unit vectors (f64 xyz) then perm_x, perm_y, perm_z.  Same expression order as the oracle's
// NoiseTable (the f64 audit kernel reproduces it bit for bit).
template <typename R> RT_DEV R perlin_noise(const uint8_t* tab, Vec3<R> p) {
  const double* vec = reinterpret_cast<const double*>(tab);
  const uint8_t* perm = tab + 256 * 3 * sizeof(double);
  const R fx = __builtin_floor(p.x), fy = __builtin_floor(p.y), fz = __builtin_floor(p.z);
  const R u = p.x - fx, v = p.y - fy, w = p.z - fz;
  const int i = (int)fx, j = (int)fy, k = (int)fz;
  const R uu = u * u * (R(3) - R(2) * u), vv = v * v * (R(3) - R(2) * v), ww = w * w * (R(3) - R(2) * w);
  R accum = R(0);
#pragma unroll
  for (int di = 0; di < 2; ++di)
#pragma unroll
    for (int dj = 0; dj < 2; ++dj)
#pragma unroll
      for (int dk = 0; dk < 2; ++dk) {
        const uint32_t idx = perm[(i + di) & 255] ^ perm[256 + ((j + dj) & 255)] ^ perm[512 + ((k + dk) & 255)];
        const R wx = u - R(di), wy = v - R(dj), wz = w - R(dk);
        const R d = (R)vec[3 * idx] * wx + (R)vec[3 * idx + 1] * wy + (R)vec[3 * idx + 2] * wz;
        accum += (di ? uu : R(1) - uu) * (dj ? vv : R(1) - vv) * (dk ? ww : R(1) - ww) * d;
      }
  return accum;
}
template <typename R> RT_DEV R perlin_turb(const uint8_t* tab, Vec3<R> p) {
  R accum = R(0), weight = R(1);
  for (int o = 0; o < 7; ++o) {
    accum += weight * perlin_noise(tab, p);
    weight *= R(0.5);
    p = R(2) * p;
  }
  return rabs(accum);
}

// ---- texture.rs:33-69 ---------------------------------------------------------------------
// NOISE: whether this instantiation carries the Perlin code (extension).  Only the reference-order kernel has a
// NOISE = true variant (and the f64 audit kernel): inlined into the fast kernels' bounce loops the noise code
// costs the common, noise-free path registers (measured: 116 B of scratch, -6 % on the BVH scenes), and as a
// real call it costs 25-30 % (AMDGPU call ABI).  Scenes with a noise texture are routed to that variant.
template <typename R, bool NOISE = false>
RT_DEV Vec3<R> texture_value(const SceneView<R>& sc, uint32_t idx, Vec2<R> uv, Vec3<R> p, Vec3<R> normal) {
  for (;;) {
    Texture<R> t = sc.textures[idx];
    if (t.tag == TEX_CHECKER) {
      R sines = sin_scaled10(p.x) * sin_scaled10(p.y) * sin_scaled10(p.z);
      idx = sines < R(0) ? t.odd : t.even;
      continue;
    }
    if (t.tag == TEX_SOLID) return Vec3<R>(t.color[0], t.color[1], t.color[2]);
    if (t.tag == TEX_IMAGE) {
      R u = rmin(rmax(uv.x, R(0)), R(1));
      R v = rmin(rmax(uv.y, R(0)), R(1));
      v = R(1) - v;
      uint32_t x = (uint32_t)(u * (R)t.width);
      uint32_t y = (uint32_t)(v * (R)t.height);
      x = x < t.width - 1 ? x : t.width - 1;
      y = y < t.height - 1 ? y : t.height - 1;
      const uint8_t* px = sc.texels + t.texel_offset + 3ull * ((uint64_t)y * t.width + x);
      const R color_scale = R(1) / R(255);
      return Vec3<R>((R)px[0] * color_scale, (R)px[1] * color_scale, (R)px[2] * color_scale);
    }
    if (NOISE && t.tag == TEX_NOISE) {
      const R m = R(0.5) * (R(1) + rsin(t.color[0] * p.z + R(10) * perlin_turb(sc.texels + t.texel_offset, p)));
      return Vec3<R>(m, m, m);
    }
    if (t.tag == TEX_NORMAL) return Vec3<R>(rabs(normal.x), rabs(normal.y), rabs(normal.z));
    return Vec3<R>(uv.x, uv.y, R(0));  // TEX_UV
  }
}

// ---- alias.rs:44-94 -------------------------------------------------------------------------
template <typename R> RT_DEV Vec3<R> random_in_unit_sphere(Rng& rng) {
  for (;;) {
    R a = rng.range<R>(R(-1), R(1));
    R b = rng.range<R>(R(-1), R(1));
    R c = rng.range<R>(R(-1), R(1));
    Vec3<R> p(a, b, c);
    if (sqrlen(p) < R(1)) return p;
  }
}
template <typename R> RT_DEV Vec3<R> random_in_unit_disk(Rng& rng) {
  for (;;) {
    R a = rng.range<R>(R(-1), R(1));
    R b = rng.range<R>(R(-1), R(1));
    Vec3<R> p(a, b, R(0));
    if (sqrlen(p) < R(1)) return p;
  }
}
// random_cosine_direction (alias.rs:72-82) and random_to_sphere (alias.rs:84-94) live inside shade(): the two
// share one "draw r1 r2, map (z, radial) into an ONB" code path there.

// ---- hittable.rs:477-500 on the flattened light list (random(): see shade()) -------------------
template <typename R, bool STATS>
RT_DEV R light_pdf_one(const Light<R>& l, Vec3<R> origin, Vec3<R> direction, Cnt<STATS>& cnt) {
  if (l.tag != LIGHT_SPHERE) return R(0);
  cnt.template inc<ST_LIGHT_PROBES>();
  cnt.template inc<ST_SPHERE_TESTS>();
  Vec3<R> c(l.c[0], l.c[1], l.c[2]);
  R t;
  if (!sphere_t(origin, direction, sqrlen(direction), c, l.radius, R(0), Consts<R>::inf, t)) return R(0);
  cnt.template inc<ST_SPHERE_HITS>();
  R cos_theta_max = rroot(R(1) - rdiv(l.radius * l.radius, sqrlen(c - origin)));
  R solid_angle = R(2) * Consts<R>::pi * (R(1) - cos_theta_max);
  return rrcp(solid_angle);
}
// f32 production form of the same test.  For an origin outside the sphere "the probe ray hits
// it at t >= 0" is "the direction lies inside the tangent cone": cos(angle to the centre) >=
// cos_theta_max — every quantity of which the pdf needs anyway.  An origin inside (or on) the
// sphere always hits, and the pdf is the same NaN / inf the reference produces (Q9).
struct LightCone {  // what both Hittable::random and pdf_value need of a sphere light seen from `origin`
  Vec3<float> to_c;   // centre - origin
  float d2, r2;       // |to_c|^2, radius^2
  float cos_theta_max, inv_len;  // sqrt(1 - r^2/d^2) (hittable.rs:487, alias.rs:87), 1/|to_c|
};
RT_DEV LightCone light_cone(const Light<float>& l, Vec3<float> origin) {
  LightCone c;
  c.to_c = Vec3<float>(l.c[0], l.c[1], l.c[2]) - origin;
  c.d2 = sqrlen(c.to_c);
  c.r2 = l.radius * l.radius;
  c.cos_theta_max = rroot(1.0f - rdiv(c.r2, c.d2));
  c.inv_len = __builtin_amdgcn_rsqf(c.d2);
  return c;
}
template <bool STATS> RT_DEV float light_pdf_cone(const LightCone& c, Vec3<float> unit_direction, Cnt<STATS>& cnt) {
  cnt.template inc<ST_LIGHT_PROBES>();
  cnt.template inc<ST_SPHERE_TESTS>();
  const float cos_to_centre = dot(unit_direction, c.to_c) * c.inv_len;
  if (!(c.d2 <= c.r2 || cos_to_centre >= c.cos_theta_max)) return 0.0f;
  cnt.template inc<ST_SPHERE_HITS>();
  return rrcp(2.0f * Consts<float>::pi * (1.0f - c.cos_theta_max));
}
template <bool STATS>
RT_DEV float light_pdf_cone(const Light<float>& l, Vec3<float> origin, Vec3<float> unit_direction, Cnt<STATS>& cnt) {
  if (l.tag != LIGHT_SPHERE) return 0.0f;
  return light_pdf_cone(light_cone(l, origin), unit_direction, cnt);
}

// EXTENSION (RT_LIGHT_CUBE, no counterpart in the reference): the density, per solid angle, of "a point uniform on the
// box's surface" in direction v from `origin`: the sum over the points where the ray enters (origin outside) and leaves
// the box of distance^2 / (|cos| area) = t^2 |v|^3 / (|v_k| area), k the axis of the face.  The slab loop is
// aabb.rs:17-34 over (0, inf) with the axes remembered; l.c = p0, l.hi = p1, l.radius = surface area.
template <typename R, bool STATS>
RT_DEV R light_pdf_cube(const Light<R>& l, Vec3<R> origin, Vec3<R> v, Cnt<STATS>& cnt) {
  cnt.template inc<ST_LIGHT_PROBES>();
  cnt.template inc<ST_CUBE_TESTS>();
  const R o[3] = {origin.x, origin.y, origin.z}, d[3] = {v.x, v.y, v.z};
  R t_in = R(0), t_out = Consts<R>::inf, d_in = R(0), d_out = R(0);  // |v_k| of the entry / exit face (0: none)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const R inv_d = rdiv(R(1), d[i]);
    R t0 = (l.c[i] - o[i]) * inv_d, t1 = (l.hi[i] - o[i]) * inv_d;
    if (inv_d < R(0)) {
      const R tmp = t0;
      t0 = t1;
      t1 = tmp;
    }
    if (t0 > t_in) {
      t_in = t0;
      d_in = rabs(d[i]);
    }
    if (t1 < t_out) {
      t_out = t1;
      d_out = rabs(d[i]);
    }
    if (t_out <= t_in) return R(0);
  }
  cnt.template inc<ST_CUBE_HITS>();
  const R len2 = sqrlen(v), len3 = len2 * rroot(len2);
  R sum = R(0);
  if (d_in != R(0)) sum += rdiv(t_in * t_in * len3, d_in * l.radius);
  if (d_out != R(0)) sum += rdiv(t_out * t_out * len3, d_out * l.radius);
  return sum;
}

template <typename R, bool STATS, bool EXT = false>
RT_DEV R lights_pdf(const SceneView<R>& sc, const Light<R>* lights, Vec3<R> origin, Vec3<R> direction,
                    Vec3<R> unit_direction, Cnt<STATS>& cnt) {
  auto one = [&](const Light<R>& l) -> R {
    if constexpr (EXT)
      if (l.tag == LIGHT_CUBE) return light_pdf_cube(l, origin, direction, cnt);
    if constexpr (sizeof(R) == 4)
      return light_pdf_cone(l, origin, unit_direction, cnt);
    else
      return light_pdf_one(l, origin, direction, cnt);
  };
  if (sc.lights_kind == LIGHTS_SINGLE) return one(lights[0]);
  R weight = rrcp((R)sc.n_lights);
  R sum = R(0);
  for (uint32_t i = 0; i < sc.n_lights; ++i) sum += weight * one(lights[i]);
  return sum;
}

// texture.value() of a material's texture; Solid textures were folded into the material record
template <typename R, bool NOISE = false>
RT_DEV Vec3<R> material_color(const SceneView<R>& sc, const Material<R>& m, const HitRec<R>& h) {
  if (m.flags & MATF_SOLID) return Vec3<R>(m.color[0], m.color[1], m.color[2]);
  return texture_value<R, NOISE>(sc, m.texture, h.uv, h.p, h.normal);
}

// material.rs:128-133
template <typename R> RT_DEV R reflectance(R cosine, R ref_idx) {
  R r0 = rdiv(R(1) - ref_idx, R(1) + ref_idx);
  r0 = r0 * r0;
  R m = R(1) - cosine;
  R m2 = m * m;
  R m4 = m2 * m2;
  return r0 + (R(1) - r0) * (m4 * m);
}

// One vertex of ray_color (main.rs:77-102) after the closest hit is known.  Updates the
// running throughput / radiance, overwrites `ray` with the scattered ray and returns true if
// the path continues.
template <typename R, bool STATS, bool NOISE = false>
RT_DEV bool shade(const SceneView<R>& sc, const Light<R>* lights, const Material<R>& m, const HitRec<R>& h,
                  RayT<R>& ray, Vec3<R>& thr, Vec3<R>& L, Rng& rng, Cnt<STATS>& cnt) {
  // (NOISE = the extension instantiation: Perlin textures, RT_MAT_AREA_LIGHT, RT_LIGHT_CUBE)
  if (m.tag == MAT_DIFFUSE_LIGHT || (NOISE && m.tag == MAT_AREA_LIGHT)) {  // scatter -> None: return emitted (main.rs:101)
    cnt.template inc<ST_EMITTER>();
    if (h.front_face || (NOISE && m.tag == MAT_AREA_LIGHT))
      L = madd(L, thr, material_color<R, NOISE>(sc, m, h));
    else  // material.rs:118-126: the back face emits Color::new(), and the callers still multiply that zero into their
          // attenuation * scattering_pdf / pdf_value (main.rs:93-97): NaN where a vertex up the path had pdf_value 0 (Q9)
      L = madd(L, thr, Vec3<R>(0, 0, 0));
    return false;
  }
  if (m.tag == MAT_DIELECTRIC) {  // material.rs:66-87; emitted is dropped (main.rs:81-83)
    cnt.template inc<ST_SPECULAR>();
    R ir = m.p[0];
    R refraction_ratio = h.front_face ? rrcp(ir) : ir;
    Vec3<R> unit_direction = normalize(ray.d);
    R cos_theta = rmin(dot(-unit_direction, h.normal), R(1));
    R sin_theta = rroot(R(1) - cos_theta * cos_theta);
    bool cannot_refract = refraction_ratio * sin_theta > R(1);
    Vec3<R> direction;
    if (cannot_refract || reflectance(cos_theta, refraction_ratio) > rng.next<R>())
      direction = reflect(unit_direction, h.normal);
    else
      direction = refract(unit_direction, h.normal, refraction_ratio);
    ray.o = h.p;
    ray.d = direction;
    return true;
  }
  if (m.tag == MAT_METAL) {  // material.rs:88-95
    cnt.template inc<ST_SPECULAR>();
    Vec3<R> reflected = reflect(normalize(ray.d), h.normal);
    Vec3<R> direction = reflected + m.p[3] * random_in_unit_sphere<R>(rng);
    ray.o = h.p;
    ray.d = direction;
    thr = thr * Vec3<R>(m.p[0], m.p[1], m.p[2]);
    return true;
  }
  // Lambertian / Isotropic: ScatterRecord::PDF with a cosine PDF (material.rs:59-65, 97-100)
  cnt.template inc<ST_DIFFUSE>();
  Vec3<R> attenuation = material_color<R, NOISE>(sc, m, h);
  const bool have_lights = sc.lights_kind != LIGHTS_NONE;
  // PDF::generate (pdf.rs:33-39): Mixture[light, cosine].choose(), then either
  // Hittable::random (hittable.rs:502-515) or the cosine direction (alias.rs:72-82).  Both
  // are "build an ONB on an axis, draw r1 r2, map (z, radial) into it": only the axis and the
  // z mapping differ, so the two branches share one copy of that code (same arithmetic).
  bool from_light = false;
  if (have_lights) from_light = rng.below(2) == 0;
  Vec3<R> axis = h.normal;
  R light_radius = R(0), distance_squared = R(1);
  bool no_sample = false;  // a light that is not a Sphere: random() = (1,0,0), no draws
  // f32, one sphere light (Cornell): Hittable::random and pdf_value need the same geometry of the light as
  // seen from the hit point — computed once here, for every lane, instead of once in the (divergent)
  // light-sampling branch and again in the pdf
  [[maybe_unused]] bool one_cone = false;
  [[maybe_unused]] LightCone cone{};
  if constexpr (sizeof(R) == 4) {
    if (sc.lights_kind == LIGHTS_SINGLE && lights[0].tag == LIGHT_SPHERE) {  // wave-uniform
      one_cone = true;
      cone = light_cone(lights[0], h.p);
    }
  }
  R cos_theta_max_sample = R(0);
  Vec3<R> direction(1, 0, 0);
  if (from_light) {
    uint32_t i = 0;
    if (sc.lights_kind == LIGHTS_LIST) i = rng.below(sc.n_lights);  // list.choose()
    const Light<R> l = lights[i];
    if (NOISE && l.tag == LIGHT_CUBE) {
      // EXTENSION: a point uniform on the box's surface — the face by area (x faces, y faces, z faces, low side
      // first), then the two in-face coordinates in axis order; three draws
      no_sample = true;
      const R ex = l.hi[0] - l.c[0], ey = l.hi[1] - l.c[1], ez = l.hi[2] - l.c[2];
      const R ax = ey * ez, ay = ex * ez, az = ex * ey;
      const R sel = rng.next<R>() * (R(2) * (ax + ay + az));
      const R r1 = rng.next<R>();
      const R r2 = rng.next<R>();
      Vec3<R> q;
      if (sel < R(2) * ax)
        q = Vec3<R>(sel < ax ? l.c[0] : l.hi[0], l.c[1] + r1 * ey, l.c[2] + r2 * ez);
      else if (sel < R(2) * ax + R(2) * ay)
        q = Vec3<R>(l.c[0] + r1 * ex, sel < R(2) * ax + ay ? l.c[1] : l.hi[1], l.c[2] + r2 * ez);
      else
        q = Vec3<R>(l.c[0] + r1 * ex, l.c[1] + r2 * ey, sel < R(2) * ax + R(2) * ay + az ? l.c[2] : l.hi[2]);
      direction = q - h.p;
    } else if (l.tag != LIGHT_SPHERE) {
      no_sample = true;
    } else {
      bool shared = false;
      if constexpr (sizeof(R) == 4) {
        if (one_cone) {
          axis = cone.to_c * cone.inv_len;  // unit
          cos_theta_max_sample = cone.cos_theta_max;
          shared = true;
        }
      }
      if (!shared) {
        axis = Vec3<R>(l.c[0], l.c[1], l.c[2]) - h.p;
        distance_squared = sqrlen(axis);
        light_radius = l.radius;
        cos_theta_max_sample = rroot(R(1) - rdiv(light_radius * light_radius, distance_squared));
        // f32: the hit normal is unit length already (cube faces, (p-c)/r, rigid maps of those); only the
        // axis towards a light needs normalising
        if constexpr (sizeof(R) == 4) axis = normalize(axis);
      }
    }
  }
  if (!no_sample) {
    const ONB<R> uvw = sizeof(R) == 4 ? ONB<R>::from_unit_w(axis) : ONB<R>::from_w(axis);
    const R r1 = rng.next<R>();
    const R r2 = rng.next<R>();
    R z, radial;
    if (from_light) {  // alias.rs:84-94
      z = R(1) + r2 * (cos_theta_max_sample - R(1));
      radial = rroot(R(1) - z * z);
    } else {  // alias.rs:72-82
      z = rroot(R(1) - r2);
      radial = rroot(r2);
    }
    R sn, cs;
    sincos_2pi(r1, sn, cs);
    direction = uvw.local(Vec3<R>(cs * radial, sn * radial, z));
  }
  // cos_w = ONB::from_w(hit.normal).w of PDF::Cosine; unit = normalize(direction) (pdf.rs:13-31).  In f32
  // both are unit already to rounding: the normal as above, the direction as an orthonormal basis
  // applied to (cos phi * radial, sin phi * radial, z) with radial^2 + z^2 = 1.
  const Vec3<R> cos_w = sizeof(R) == 4 ? h.normal : normalize(h.normal);
  // (f32: a direction towards a point of a box light, or the constant (1,0,0), is not unit by construction)
  Vec3<R> unit = (sizeof(R) == 4 && !(NOISE && no_sample)) ? direction : normalize(direction);
  R cosine = dot(unit, cos_w);
  R cos_pdf = cosine <= R(0) ? R(0) : rdiv(cosine, Consts<R>::pi);
  R pdf_value;
  if (have_lights) {
    pdf_value = R(0);
    R light_pdf = R(0);
    bool shared = false;
    if constexpr (sizeof(R) == 4) {
      if (one_cone) {
        light_pdf = light_pdf_cone(cone, unit, cnt);
        shared = true;
      }
    }
    if (!shared) light_pdf = lights_pdf<R, STATS, NOISE>(sc, lights, h.p, direction, unit, cnt);
    pdf_value += R(0.5) * light_pdf;
    pdf_value += R(0.5) * cos_pdf;
  } else {
    pdf_value = cos_pdf;
  }
  // material.rs:104-116
  R spdf = R(1);
  if (m.tag == MAT_LAMBERTIAN) {
    R c2 = dot(h.normal, unit);
    spdf = c2 <= R(0) ? R(0) : rdiv(c2, Consts<R>::pi);
  }
  thr = thr * ((attenuation * spdf) / pdf_value);  // main.rs:93-97
  ray.o = h.p;
  ray.d = direction;
  return true;
}

// lower_left + s*horizontal + t*vertical - origin (cam.rs:68).  In f32 that sum cancels
// catastrophically when the camera sits far from the world origin (Cornell: |origin| ~ 900,
// viewport ~ 7), so the f32 kernels use the host's f64 (lower_left - origin) instead.
template <typename R> RT_DEV Vec3<R> primary_direction(const CameraView<R>& cam, R s, R t) {
  if constexpr (sizeof(R) == 4)
    return madd(madd(cam.ll_rel, s, cam.horizontal), t, cam.vertical);
  else
    return cam.lower_left + s * cam.horizontal + t * cam.vertical - cam.origin;
}

// cam.rs:63-71.  Draw order in slot 0: jitter u, jitter v, shutter time, then the lens-disk
// rejection pairs (the reference draws the time last; the streams are ours, the quantities are
// independent).  A pinhole camera (lens_radius == 0) multiplies the disk sample by zero, so its
// draws are skipped: same ray, one Philox block instead of two or more.
template <typename R> RT_DEV RayT<R> camera_ray(const CameraView<R>& cam, R s, R t, Rng& rng) {
  RayT<R> r;
  r.time = rng.range<R>(cam.time_open, cam.time_close);
  if (cam.lens_radius == R(0)) {
    r.o = cam.origin;
    r.d = primary_direction(cam, s, t);
    return r;
  }
  Vec3<R> rd = cam.lens_radius * random_in_unit_disk<R>(rng);
  Vec3<R> offset = cam.u * rd.x + cam.v * rd.y;
  r.o = cam.origin + offset;
  r.d = primary_direction(cam, s, t) - offset;
  return r;
}

// The path wants to go on but the depth budget is spent: the reference still multiplies the zero that
// ray_color(depth 0) returns (main.rs:73-75) into `attenuation * scattering_pdf * ... / pdf_value`
// (main.rs:93-97), so a vertex whose pdf_value is 0 or NaN (Q9: a scatter point on or inside a light sphere)
// turns the sample into NaN even here.  thr already holds that product's factor: add thr * 0.
template <typename R> RT_DEV void depth_exhausted(Vec3<R>& L, Vec3<R> thr) { L = L + thr * Vec3<R>(0, 0, 0); }

RT_DEV uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <bool STATS> RT_DEV void flush_stats(const Cnt<STATS>& cnt, DeviceStats* stats) {
  constexpr int N = STATS ? (int)ST_COUNT : 1;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    uint32_t s = wave_sum(cnt.v[i]);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(&stats->v[i], (unsigned long long)s);
  }
}

// Which (tile, sample split) a workgroup renders.  Workgroups go to the 8 XCDs round-robin by linear id, and each
// XCD has its own L2.  mode 0: tile = blockIdx.x, split = blockIdx.y — the workgroups resident on one XCD at a time are
// every 8th tile of a stretch 8x as long as it need be.  mode 1 (gridDim.y a multiple of 8): XCD x renders the splits
// x, x + 8, ... of EVERY tile, tiles in order — the workgroups it runs side by side are neighbours in the image (they
// walk the same part of the tree and find it in their L2), and all 8 XCDs have the same work whatever the image
// looks like (each renders the whole image, 1/8 of the samples).  mode >= 2: slots dealt in chunks of `mode` (kept for
// experiments).  All are bijections on the grid's slots: which pixels and samples are rendered does not change.
RT_DEV void block_slot(uint32_t mode, uint32_t& tile, uint32_t& split) {
  tile = blockIdx.x;
  split = blockIdx.y;
  if (mode == 0) return;
  const uint32_t total = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
  const uint32_t x = id & 7u, i = id >> 3;  // XCD x runs ids x, x + 8, ...: this is its i-th workgroup
  if (mode == 1) {
    if (gridDim.y & 7u) return;
    const uint32_t k = i / gridDim.x;
    split = x + 8u * k;
    tile = i - k * gridDim.x;
    return;
  }
  const uint32_t round = mode * 8u;
  if (id >= total - total % round) return;
  const uint32_t f = ((i / mode) * 8u + x) * mode + i % mode;
  split = f / gridDim.x;
  tile = f - split * gridDim.x;
}

// Tile order within the image: bands of `band` tile rows, walked column by column, so that a run of consecutive
// tiles (what one XCD has resident at a time) is a compact block of the image rather than a few long rows.
// band <= 1: plain row-major.  A bijection on [0, tiles_x * tiles_y).
RT_DEV uint32_t band_order(uint32_t i, uint32_t tiles_x, uint32_t tiles_y, uint32_t band) {
  if (band <= 1u) return i;
  const uint32_t per_band = band * tiles_x, b = i / per_band, within = i - b * per_band;
  const uint32_t rows = min(band, tiles_y - b * band);  // the last band may be lower
  const uint32_t col = within / rows, row = b * band + (within - col * rows);
  return row * tiles_x + col;
}

RT_DEV void tile_pixel(uint32_t width, uint32_t& px, uint32_t& py, uint32_t tile = blockIdx.x) {
  const uint32_t tiles_x = (width + 15u) / 16u;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  px = (tile % tiles_x) * 16u + (wave & 1u) * 8u + (lane & 7u);
  py = (tile / tiles_x) * 16u + (wave >> 1) * 8u + (lane >> 3);
}

// Diagnostic build (-DRT_STAMPS): s_memtime stamps at wave-convergent points of the bounce loop;
// the per-phase wave-cycle sums leave the kernel only through DeviceStats::stamp.
#ifdef RT_STAMPS
#define RT_STAMP(i)                                                                    \
  do {                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    unsigned long long now_;                                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");       \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    stamp_acc[i] += now_ - stamp_last;                                                 \
    stamp_last = now_;                                                                 \
  } while (0)
#elif defined(RT_NO_PHASE_MARKS)  // A/B: the loops without the markers below
#define RT_STAMP(i) \
  do {              \
  } while (0)
#else
// Product build: a phase boundary is an (empty) inline-asm statement.  It emits nothing but a comment, yet it is a
// CONVERGENT operation the optimiser may neither duplicate nor make control-dependent on new conditions, so the
// bounce loop keeps the shape it is written in: one copy of trace / hit record / shade that all lanes of the wave run
// together.  Without the markers the f64 audit kernel came out with the loop body duplicated per branch history
// (lane utilisation 0.12) and ran 2.16x slower (Cornell 1024^2 x 64 spp: 179.4 -> 82.9 ms); the f32 kernels are
// unchanged by them (C1 / C3 / C4 / Cornell within 0.3 %).
#define RT_STAMP(i) asm volatile("; phase %0" ::"n"(i))
#endif

struct RenderArgs {
  uint32_t width, height;
  uint32_t sample_begin;  // global index of this launch's first sample
  uint32_t n_samples;
  uint32_t max_depth;
  uint32_t accumulate;  // add to accum instead of overwriting it (ignored when splits > 1: always atomic add)
  uint32_t splits;      // gridDim.y: each block row renders a slice of the launch's samples
  uint32_t stack_depth; // LDS traversal-stack entries per lane
  uint32_t refill_eighths;  // render_refill_kernel: refill when this many eighths of the unfinished lanes idle (1..8)
  uint32_t unit_samples;    // render_persist_kernel: samples per (8x8 tile, sample range) unit
  uint32_t wave_tiles_x, wave_tiles, total_units;
  uint32_t leaf_shift;  // refill kernel: run the leaf tests once (walking lanes >> leaf_shift) lanes wait at a leaf
  uint32_t direct_accum;  // units kernel: no LDS accumulators, finished samples go to the image by float atomics
  uint32_t unit_prefetch; // render_persist_kernel: units a wave draws ahead of need per visit of the global counter (< UNIT_RING)
  uint32_t xcd_chunk;   // render_kernel, refill kernel: block_slot's mode
  uint32_t tile_band;   // refill kernel: band_order's band height in tiles
  uint64_t seed;
};

// TRAV: 0 = the reference's left-first BVH walk, 1 = leaf list (small scenes), 2 = ordered BVH walk
// run to completion per ray (kept as the A/B baseline of render_refill_kernel, which is what
// LaunchShape::trav == 2 launches unless RT_ORDERED_NO_REFILL is set in the environment)
// 3 = the leaf list over the scene arrays staged in LDS (traverse_leaves: any precision; what the f64 kernel runs on
// leaf-list scenes)
constexpr int TRAV_REFERENCE = 0, TRAV_LINEAR = 1, TRAV_ORDERED = 2, TRAV_LEAVES = 3;

// INST (TRAV_REFERENCE only): -1 = the walk picks its instance / no-instance form at run time (SceneView::has_instances);
// 0 / 1 = compiled for scenes without / with instances — the f32 product instantiation of the reference-order kernel is
// launched that way, so that the kernel scenes without instances run (the reference's default scene among them) is the
// one it was before instances existed
template <typename R, bool STATS, bool EARLY_EXIT, int TRAV, bool NOISE = false, int INST = -1>
__global__ void __launch_bounds__(BLOCK, TRAV == 1 ? RT_LINEAR_MIN_WAVES : (sizeof(R) == 8 ? RT_RENDER_F64_MIN_WAVES : RT_RENDER_MIN_WAVES)) render_kernel(SceneView<R> sc, CameraView<R> cam, RenderArgs a,
                                                       R* __restrict__ accum, DeviceStats* stats) {
  constexpr bool LINEAR = TRAV == TRAV_LINEAR;
  extern __shared__ uint32_t lds_stack[];
  uint32_t* stack = lds_stack + threadIdx.x;
  // small read-mostly tables staged in LDS once per workgroup: the lights, and (leaf-list path)
  // the fixed-layout leaf records
  __shared__ Light<R> s_lights[LDS_MAX_LIGHTS];
  __shared__ float s_lin[LINEAR ? LIN_MAX_LEAVES * LIN_DWORDS : 1];
  {
    const uint32_t n_l = sc.n_lights < LDS_MAX_LIGHTS ? sc.n_lights : LDS_MAX_LIGHTS;
    constexpr uint32_t LW = sizeof(Light<R>) / 4;
    for (uint32_t i = threadIdx.x; i < n_l * LW; i += BLOCK)
      reinterpret_cast<uint32_t*>(s_lights)[i] = reinterpret_cast<const uint32_t*>(sc.lights)[i];
    if constexpr (LINEAR)
      for (uint32_t i = threadIdx.x; i < sc.n_linear * LIN_DWORDS; i += BLOCK)
        s_lin[i] = __uint_as_float(sc.linear_rec[i]);
    __syncthreads();
  }
  const Light<R>* lights = sc.n_lights <= LDS_MAX_LIGHTS ? s_lights : sc.lights;
  uint32_t px, py, my_tile, my_split;
  block_slot(a.xcd_chunk, my_tile, my_split);  // (tile, split) slots dealt to the XCDs split-wise: see block_slot
  tile_pixel(a.width, px, py, my_tile);
  const bool active = px < a.width && py < a.height;
  const uint32_t pixel = py * a.width + px;

  Cnt<STATS> cnt;
  cnt.clear();
  uint32_t nonfinite = 0;
  Vec3<R> acc;

  if (active) {
    const Vec3<R> background(sc.background[0], sc.background[1], sc.background[2]);
    // this workgroup's slice of the launch's samples
    uint32_t s = (uint32_t)(((uint64_t)a.n_samples * my_split) / a.splits);
    const uint32_t s_end = (uint32_t)(((uint64_t)a.n_samples * (my_split + 1)) / a.splits);
    bool alive = false;
    RayT<R> ray;
    Vec3<R> thr;
    uint32_t depth_left = 0, slot = 0;
    Rng rng;
    rng.init(a.seed, pixel, 0);
#ifdef RT_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif
    for (;;) {
      RT_STAMP(0);
      if (!alive) {
        if (s == s_end) break;
        rng.sample = a.sample_begin + s;
        ++s;
        rng.begin(0);
        R ju = rng.next<R>();
        R jv = rng.next<R>();
        R u = rdiv((R)px + ju, (R)(a.width - 1));                  // main.rs:161
        R v = rdiv((R)(a.height - py - 1) + jv, (R)(a.height - 1));  // main.rs:162-163
        ray = camera_ray(cam, u, v, rng);
        thr = Vec3<R>(1, 1, 1);
        depth_left = a.max_depth;
        slot = 1;
        alive = true;
        if (depth_left == 0) {  // main.rs:73-75
          alive = false;
          continue;
        }
      }
      RT_STAMP(1);
      rng.begin(slot);
      cnt.template inc<ST_RAYS>();
      RT_STAMP(2);
      R t;
      uint32_t best;
      if constexpr (LINEAR) {
        if (sc.linear_has_media)  // wave-uniform
          traverse_linear<STATS, true>(s_lin, sc.n_linear, lin_masks(sc), ray, R(0.001), rng, t, best, cnt);
        else
          traverse_linear<STATS, false>(s_lin, sc.n_linear, lin_masks(sc), ray, R(0.001), rng, t, best, cnt);
      } else if constexpr (TRAV == TRAV_ORDERED) {  // run-to-completion form of the ordered walk (A/B baseline)
        OrderedWalk walk;
        bool more = walk.start(sc, ray);
        while (more)
          more = walk.step(sc, ray, R(0.001), rng, stack, reinterpret_cast<uint16_t*>(lds_stack + a.stack_depth * BLOCK) + threadIdx.x,
                           (uint32_t)BLOCK, cnt);
        t = walk.closest;
        best = walk.best;
      } else if constexpr (INST == 0) {
        traverse_impl<false>(sc, ray, R(0.001), rng, stack, (uint32_t)BLOCK, t, best, cnt);
      } else if constexpr (INST == 1) {
        traverse_impl<true>(sc, ray, R(0.001), rng, stack, (uint32_t)BLOCK, t, best, cnt);
      } else {
        traverse(sc, ray, R(0.001), rng, stack, (uint32_t)BLOCK, t, best, cnt);
      }
      RT_STAMP(3);
      Vec3<R> L(0, 0, 0);  // set once, at the path's last vertex (see render_persist_kernel)
      bool cont;
      if (best == REF_NONE) {  // main.rs:104
        cnt.template inc<ST_MISSES>();
        L = madd(L, thr, background);
        cont = false;
      } else {
        cnt.template inc<ST_CLOSEST>();
        HitRec<R> h;
        Material<R> mat;
        if constexpr (LINEAR) {
          lin_hit_record(sc, s_lin + best * LIN_DWORDS, ray, t, h, mat);
        } else {
          make_hit_record(sc, best, ray, t, false, h);
          mat = sc.materials[h.material];
        }
        RT_STAMP(4);
        cont = shade<R, STATS, NOISE>(sc, lights, mat, h, ray, thr, L, rng, cnt);
      }
      RT_STAMP(5);
      ++slot;
      --depth_left;
      if (depth_left == 0 && cont) {  // the next ray_color call would return 0
        depth_exhausted(L, thr);
        cont = false;
      }
      if (EARLY_EXIT && cont && thr.x == R(0) && thr.y == R(0) && thr.z == R(0)) cont = false;
      if (!cont) {
        const bool fin = rfinite(L.x) && rfinite(L.y) && rfinite(L.z);
        if (!fin) ++nonfinite;
        // f32 production kernel drops a non-finite sample (DESIGN.md "f32 conditioning");
        // the f64 audit kernel lets it poison the pixel like the reference does.
        if (fin || sizeof(R) == 8) acc = acc + L;
        alive = false;
      }
    }
#ifdef RT_STAMPS
    if ((threadIdx.x & 63) == __builtin_ctzll(__ballot(1)))
      for (int i = 0; i < 8; ++i) atomicAdd(&stats->stamp[i], stamp_acc[i]);
#endif
    R* out = accum + 3ull * pixel;
    if (a.splits > 1) {
      atomicAdd(out + 0, acc.x);
      atomicAdd(out + 1, acc.y);
      atomicAdd(out + 2, acc.z);
    } else if (a.accumulate) {
      out[0] += acc.x;
      out[1] += acc.y;
      out[2] += acc.z;
    } else {
      out[0] = acc.x;
      out[1] = acc.y;
      out[2] = acc.z;
    }
  }
  flush_stats(cnt, stats);
  uint32_t nf = wave_sum(nonfinite);
  if ((threadIdx.x & 63) == 0 && nf) atomicAdd(&stats->v[ST_NONFINITE], (unsigned long long)nf);
}

// Large scenes, f32: the ordered walk with in-wave ray refill.
//
// In a BVH-bound scene the number of walk steps per ray is heavy-tailed (1M spheres: mean ~170,
// tail in the thousands), so a wave that walks "until its slowest lane is done" idles most lanes
// most of the time (measured: 9 % VALU lane utilisation).  Here the walk is a per-lane state
// machine (OrderedWalk): all lanes step together, and as soon as a set fraction (RenderArgs::refill_eighths) of the wave's
// unfinished lanes have no ray left to walk, the wave leaves the step loop, shades exactly those lanes (hit
// record, Material::scatter, next ray or next camera sample) and re-enters the loop with them
// refilled — the lanes still walking keep their state in registers and their stack in LDS.  Same
// paths, same draws as render_kernel: a lane's vertices are still processed in order.
#ifdef RT_BLOCK_TRACE
// Diagnostic build (tools/build_variant.sh trace -DRT_BLOCK_TRACE): per wave {start, end (100 MHz ticks), XCC id, HW id},
// read back by rt_debug_block_trace.  Not in the product library.
constexpr uint32_t BLOCK_TRACE_MAX = 1u << 20;
__device__ uint4 g_block_trace[BLOCK_TRACE_MAX];
#endif

template <bool STATS>
__global__ void __launch_bounds__(BLOCK, RT_REFILL_MIN_WAVES) render_refill_kernel(SceneView<float> sc, CameraView<float> cam,
                                                                                 RenderArgs a, float* __restrict__ accum,
                                                                                 DeviceStats* stats) {
  using R = float;
  extern __shared__ uint32_t lds_stack[];
  uint32_t* stack = lds_stack + threadIdx.x;
  uint16_t* stack_t = reinterpret_cast<uint16_t*>(lds_stack + a.stack_depth * BLOCK) + threadIdx.x;
  __shared__ Light<R> s_lights[LDS_MAX_LIGHTS];
  {
    const uint32_t n_l = sc.n_lights < LDS_MAX_LIGHTS ? sc.n_lights : LDS_MAX_LIGHTS;
    constexpr uint32_t LW = sizeof(Light<R>) / 4;
    for (uint32_t i = threadIdx.x; i < n_l * LW; i += BLOCK)
      reinterpret_cast<uint32_t*>(s_lights)[i] = reinterpret_cast<const uint32_t*>(sc.lights)[i];
    __syncthreads();
  }
  const Light<R>* lights = sc.n_lights <= LDS_MAX_LIGHTS ? s_lights : sc.lights;
  // A wave owns an 8x8 pixel tile and this block row's slice of the launch's samples: 64 x n_s (pixel, sample) JOBS.
  // Jobs are not tied to lanes: whichever lane needs work takes the next job (j -> pixel j % 64, sample j / 64), so the
  // lanes of a wave run out of work together, within one path of each other, whatever their paths' lengths (with a
  // fixed quota per lane — its own pixel's samples — a wave lasts as long as its unluckiest lane, which costs most
  // where quotas are small: BASELINE config C1, 400 x 225 at 100 spp, is 27 samples per lane of the chip).  A finished
  // sample is added to its pixel's sum in LDS (ds_add_f32).  Paths and draws are unchanged: a sample's Philox stream is
  // keyed by (pixel, sample), not by the lane that renders it; only the f32 order of a pixel's sample sum changes.
  __shared__ float s_acc[BLOCK / 64][64 * 3];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#ifdef RT_BLOCK_TRACE
  const uint32_t trace_t0 = (uint32_t)wall_clock64();
#endif
  uint32_t px, py, my_tile, my_split;
  block_slot(a.xcd_chunk, my_tile, my_split);
  my_tile = band_order(my_tile, (a.width + 15u) / 16u, (a.height + 15u) / 16u, a.tile_band);
  tile_pixel(a.width, px, py, my_tile);
  const uint32_t tx0 = px - (lane & 7u), ty0 = py - (lane >> 3);  // the wave's tile origin
  s_acc[wave][lane] = 0.0f;
  s_acc[wave][64 + lane] = 0.0f;
  s_acc[wave][128 + lane] = 0.0f;
  __builtin_amdgcn_wave_barrier();

  Cnt<STATS> cnt;
  cnt.clear();
  uint32_t nonfinite = 0;
  const Vec3<R> background(sc.background[0], sc.background[1], sc.background[2]);
  const uint32_t s_begin = (uint32_t)(((uint64_t)a.n_samples * my_split) / a.splits);
  const uint32_t s_end = (uint32_t)(((uint64_t)a.n_samples * (my_split + 1)) / a.splits);
  const uint32_t total_jobs = 64u * (s_end - s_begin);  // wave-uniform
  uint32_t next_job = 0;                                 // wave-uniform: jobs handed out so far
  bool done = total_jobs == 0 || a.max_depth == 0;       // main.rs:73-75: depth 0 -> black
  bool alive = false, walking = false;
  uint32_t lp = lane;  // the pixel of the job in flight (0..63 within the tile)
  RayT<R> ray;
  Vec3<R> thr;
  uint32_t depth_left = 0, slot = 0;
  Rng rng;
  rng.init(a.seed, 0, 0);
  OrderedWalk walk;
  walk.cur = REF_NONE;
  walk.sp = 0;
  walk.best = REF_NONE;
  walk.closest = 0.0f;
  walk.a = 0.0f;

#ifdef RT_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif
  for (;;) {
    RT_STAMP(0);
    // (1) lanes without a walk in progress: next job if the path is over, then start the walk
    {
      const bool needy = !walking && !done && !alive;
      const uint64_t m = __ballot(needy);
      if (needy) {
        const uint32_t j = next_job + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (j >= total_jobs) {
          done = true;
        } else {
          lp = j & 63u;
          const uint32_t jx = tx0 + (lp & 7u), jy = ty0 + (lp >> 3);
          if (jx < a.width && jy < a.height) {  // (a tile that straddles the image edge: its outside jobs are dropped)
            rng.pixel = jy * a.width + jx;
            rng.sample = a.sample_begin + s_begin + (j >> 6);
            rng.begin(0);
            R ju = rng.next<R>();
            R jv = rng.next<R>();
            R u = rdiv((R)jx + ju, (R)(a.width - 1));                  // main.rs:161
            R v = rdiv((R)(a.height - jy - 1) + jv, (R)(a.height - 1));  // main.rs:162-163
            ray = camera_ray(cam, u, v, rng);
            thr = Vec3<R>(1, 1, 1);
            depth_left = a.max_depth;
            slot = 1;
            alive = true;
          }
        }
      }
      next_job += (uint32_t)__popcll(m);
    }
    if (!walking && alive) {
      rng.begin(slot);
      cnt.template inc<ST_RAYS>();
      walking = walk.start(sc, ray);
    }
    RT_STAMP(1);
    const uint32_t n_unfinished = (uint32_t)__popcll(__ballot(!done));
    if (n_unfinished == 0) break;
    // refill once this many of the unfinished lanes are idle (a.refill_eighths / 8 of them, at least one)
    const uint32_t refill_q = (n_unfinished * a.refill_eighths) >> 3;
    const uint32_t refill_at = refill_q > 0 ? refill_q : 1u;
    // (2) step all walking lanes until enough of them have run out of work
    //     node steps run together until a quarter of the walking lanes sit at a leaf, then those
    //     leaf tests run together (a wave that interleaves them runs the costly leaf code for one
    //     or two lanes in most iterations)
    for (;;) {
      const uint32_t n_walking = (uint32_t)__popcll(__ballot(walking));
      if (n_walking + refill_at <= n_unfinished) break;
      const uint32_t leaves_at = (n_walking >> a.leaf_shift) > 0 ? (n_walking >> a.leaf_shift) : 1u;
      RT_STAMP(2);
      for (;;) {
        const bool at_node = walking && !(walk.cur & REF_LEAF);
        const uint32_t n_node = (uint32_t)__popcll(__ballot(at_node));
        const uint32_t n_walk_now = (uint32_t)__popcll(__ballot(walking));
        if (n_node == 0 || n_walk_now - n_node >= leaves_at || n_walk_now + refill_at <= n_unfinished) break;
        if (at_node) walking = walk.node_step(sc, ray, R(0.001), stack, stack_t, (uint32_t)BLOCK, cnt);
      }
      RT_STAMP(3);
      if (walking && (walk.cur & REF_LEAF))
        walking = walk.leaf_step(sc, ray, R(0.001), rng, stack, stack_t, (uint32_t)BLOCK, cnt);
      RT_STAMP(4);
    }
    RT_STAMP(5);
    // (3) lanes whose walk is over: one vertex of ray_color (main.rs:77-104)
    if (!walking && alive) {
      Vec3<R> L(0, 0, 0);  // set once, at the path's last vertex (see render_persist_kernel)
      bool cont;
      if (walk.best == REF_NONE) {  // main.rs:104
        cnt.template inc<ST_MISSES>();
        L = madd(L, thr, background);
        cont = false;
      } else {
        cnt.template inc<ST_CLOSEST>();
        HitRec<R> h;
        make_hit_record(sc, walk.best, ray, walk.closest, false, h);
        const Material<R> mat = sc.materials[h.material];
        cont = shade(sc, lights, mat, h, ray, thr, L, rng, cnt);
      }
      ++slot;
      --depth_left;
      if (depth_left == 0 && cont) {  // the next ray_color call would return 0
        depth_exhausted(L, thr);
        cont = false;
      }
      if (cont && thr.x == R(0) && thr.y == R(0) && thr.z == R(0)) cont = false;
      if (!cont) {
        const bool fin = rfinite(L.x) && rfinite(L.y) && rfinite(L.z);
        if (!fin) ++nonfinite;
        if (fin) {
          atomicAdd(&s_acc[wave][3u * lp + 0u], L.x);
          atomicAdd(&s_acc[wave][3u * lp + 1u], L.y);
          atomicAdd(&s_acc[wave][3u * lp + 2u], L.z);
        }
        alive = false;
      }
    }
    RT_STAMP(6);
  }
#ifdef RT_STAMPS
  if ((threadIdx.x & 63) == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&stats->stamp[i], stamp_acc[i]);
#endif
  __builtin_amdgcn_wave_barrier();
  if (px < a.width && py < a.height) {
    R* out = accum + 3ull * (py * a.width + px);
    const R ax = s_acc[wave][3u * lane + 0u], ay = s_acc[wave][3u * lane + 1u], az = s_acc[wave][3u * lane + 2u];
    if (a.splits > 1) {
      atomicAdd(out + 0, ax);
      atomicAdd(out + 1, ay);
      atomicAdd(out + 2, az);
    } else if (a.accumulate) {
      out[0] += ax;
      out[1] += ay;
      out[2] += az;
    } else {
      out[0] = ax;
      out[1] = ay;
      out[2] = az;
    }
  }
#ifdef RT_BLOCK_TRACE
  {
    const uint32_t w = ((blockIdx.y * gridDim.x + blockIdx.x) * (BLOCK / 64)) + wave;
    if (lane == 0 && w < BLOCK_TRACE_MAX)
      g_block_trace[w] = make_uint4(trace_t0, (uint32_t)wall_clock64(), __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20),
                                    __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4));
  }
#endif
  flush_stats(cnt, stats);
  uint32_t nf = wave_sum(nonfinite);
  if ((threadIdx.x & 63) == 0 && nf) atomicAdd(&stats->v[ST_NONFINITE], (unsigned long long)nf);
}

// render_refill_kernel as PERSISTENT waves drawing work UNITS (one 8x8 tile x unit_samples samples = 64 x unit_samples
// jobs) from a global counter — what trav == 2 launches by default.  Why (measured with the block-trace build, final_scene
// 1024^2 x 512 spp): with one workgroup per (tile, split) the hardware dispatcher kept only ~3000 of the chip's 5120
// wave slots filled (4500 once the grid was dealt XCD-wise), a workgroup holds its LDS until its slowest wave is done
// (13-16 % of the resident wave-time idle), and the launch ends with a tail as long as its longest wave (a third of the
// run on random_scene 400 x 225 x 100 spp).  Here every wave slot of the chip is taken once, at the start, and a wave
// whose unit has no jobs left to hand out draws the next unit AT ONCE, while its slower lanes are still on paths of
// the previous one: two units are open per wave (LDS accumulators s_acc[wave][0/1], a job carries its unit's parity),
// a unit's sums go to the image (float atomics) when its last job finishes.  The only tail left is the launch's last
// path.  Same jobs, same Philox streams, same walk and shade code as render_refill_kernel.
#ifndef RT_NODE_BURST
#define RT_NODE_BURST 3
#endif
constexpr uint32_t UNITS_LDS_FLOATS = 2 * 192 + 2;  // per wave: two units' pixel sums and pixel origins

template <bool STATS>
__global__ void __launch_bounds__(BLOCK, RT_REFILL_MIN_WAVES) render_units_kernel(SceneView<float> sc, CameraView<float> cam,
                                                                                RenderArgs a, float* __restrict__ accum,
                                                                                DeviceStats* stats) {
  using R = float;
  extern __shared__ uint32_t lds_stack[];
  uint32_t* stack = lds_stack + threadIdx.x;
  uint16_t* stack_t = reinterpret_cast<uint16_t*>(lds_stack + a.stack_depth * BLOCK) + threadIdx.x;
  __shared__ Light<R> s_lights[LDS_MAX_LIGHTS];
  {
    const uint32_t n_l = sc.n_lights < LDS_MAX_LIGHTS ? sc.n_lights : LDS_MAX_LIGHTS;
    constexpr uint32_t LW = sizeof(Light<R>) / 4;
    for (uint32_t i = threadIdx.x; i < n_l * LW; i += BLOCK)
      reinterpret_cast<uint32_t*>(s_lights)[i] = reinterpret_cast<const uint32_t*>(sc.lights)[i];
    __syncthreads();
  }
  const Light<R>* lights = sc.n_lights <= LDS_MAX_LIGHTS ? s_lights : sc.lights;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  // the wave's LDS accumulators, behind the stacks in dynamic LDS: [2 parities][64 pixels x 3] floats + 2 origins.
  // a.direct_accum != 0: there are none (the launch would lose a resident workgroup per CU to them — deep trees): every
  // job is an orphan from the start and adds its sample to the image itself
  const uint32_t direct = a.direct_accum;  // wave-uniform
  float* s_acc = reinterpret_cast<float*>(lds_stack + a.stack_depth * BLOCK + a.stack_depth * (BLOCK / 2)) + wave * UNITS_LDS_FLOATS;
  uint32_t* s_origin = reinterpret_cast<uint32_t*>(s_acc + 2 * 192);  // pixel origin (x | y << 16) of the two open units
#ifdef RT_BLOCK_TRACE
  const uint32_t trace_t0 = (uint32_t)wall_clock64();
#endif
  if (!direct) {
#pragma unroll
    for (uint32_t p = 0; p < 2; ++p) {
      s_acc[p * 192u + lane] = 0.0f;
      s_acc[p * 192u + 64 + lane] = 0.0f;
      s_acc[p * 192u + 128 + lane] = 0.0f;
    }
  }
  __builtin_amdgcn_wave_barrier();

  Cnt<STATS> cnt;
  cnt.clear();
  uint32_t nonfinite = 0;
  const Vec3<R> background(sc.background[0], sc.background[1], sc.background[2]);
  // wave-uniform: the unit being handed out (parity cp) and the unfinished-job counts of both open units
  uint32_t cp = 0, n_jobs = 0, next_job = 0, s0 = 0, tx0 = 0, ty0 = 0;
  uint32_t remaining[2] = {0, 0};
  bool exhausted = a.max_depth == 0;  // main.rs:73-75: depth 0 -> black (the host has zeroed the image)
  bool alive = false, walking = false;
  uint32_t lp = lane;  // pixel (0..63 within its unit's tile) | unit parity << 6 of the job in flight
  RayT<R> ray;
  Vec3<R> thr;
  uint32_t depth_left = 0, slot = 0;
  Rng rng;
  rng.init(a.seed, 0, 0);
  OrderedWalk walk;
  walk.cur = REF_NONE;
  walk.sp = 0;
  walk.best = REF_NONE;
  walk.closest = 0.0f;
  walk.a = 0.0f;

  // unit parity p is complete: its sums leave LDS
  auto flush = [&](uint32_t p) {
    __builtin_amdgcn_wave_barrier();
    const uint32_t xy = s_origin[p];
    const uint32_t fx = (xy & 0xFFFFu) + (lane & 7u), fy = (xy >> 16) + (lane >> 3);
    if (fx < a.width && fy < a.height) {
      R* out = accum + 3ull * (fy * a.width + fx);
      atomicAdd(out + 0, s_acc[p * 192u + 3u * lane + 0u]);
      atomicAdd(out + 1, s_acc[p * 192u + 3u * lane + 1u]);
      atomicAdd(out + 2, s_acc[p * 192u + 3u * lane + 2u]);
    }
    __builtin_amdgcn_wave_barrier();
    s_acc[p * 192u + lane] = 0.0f;
    s_acc[p * 192u + 64 + lane] = 0.0f;
    s_acc[p * 192u + 128 + lane] = 0.0f;
    __builtin_amdgcn_wave_barrier();
  };

#ifdef RT_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif
  for (;;) {
    RT_STAMP(0);
    // (1) lanes whose path is over take the next jobs; a unit that is handed out makes room for the next one
    {
      bool needy = !alive;
      for (;;) {
        const uint64_t m = __ballot(needy);
        if (m == 0) break;
        if (next_job >= n_jobs) {
          if (exhausted) break;
          if (!direct && remaining[cp ^ 1u] != 0) {
            // the other slot still has paths in flight (paths that outlast a whole unit): rather than wait for them,
            // its sums so far leave LDS now and those lanes become ORPHANS (bit 7 of lp) that add their sample to
            // the image themselves when they finish
            if (alive && ((lp >> 6) & 1u) == (cp ^ 1u)) lp |= 128u;
            flush(cp ^ 1u);
            remaining[cp ^ 1u] = 0;
          }
          uint32_t u = 0;
          if (lane == 0) u = atomicAdd(&stats->unit_counter, 1u);
          u = __builtin_amdgcn_readfirstlane(u);
          if (u >= a.total_units) {
            exhausted = true;
            break;
          }
          cp ^= 1u;
          // unit u: sample chunk u / wave_tiles of 8x8 tile u % wave_tiles; four consecutive tiles are the quarters of
          // a 16x16 pixel block (wave_tiles_x of those per row), so that the waves of a workgroup start out as neighbours
          const uint32_t chunk = u / a.wave_tiles, tile = u - chunk * a.wave_tiles;
          const uint32_t t16 = tile >> 2, trow = t16 / a.wave_tiles_x;
          tx0 = (t16 - trow * a.wave_tiles_x) * 16u + (tile & 1u) * 8u;
          ty0 = trow * 16u + ((tile >> 1) & 1u) * 8u;
          const uint32_t sb = chunk * a.unit_samples;
          const uint32_t ns = a.n_samples - sb < a.unit_samples ? a.n_samples - sb : a.unit_samples;
          s0 = a.sample_begin + sb;
          n_jobs = 64u * ns;
          next_job = 0;
          if (!direct) {
            remaining[cp] = n_jobs;
            if (lane == 0) s_origin[cp] = tx0 | (ty0 << 16);
          }
        }
        const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        const uint32_t avail = n_jobs - next_job, asked = (uint32_t)__popcll(m);
        bool dropped = false;
        if (needy && rank < avail) {
          const uint32_t j = next_job + rank;
          lp = (j & 63u) | (cp << 6) | (direct << 7);
          const uint32_t jx = tx0 + (j & 7u), jy = ty0 + ((j >> 3) & 7u);
          if (jx < a.width && jy < a.height) {
            rng.pixel = jy * a.width + jx;
            rng.sample = s0 + (j >> 6);
            rng.begin(0);
            R ju = rng.next<R>();
            R jv = rng.next<R>();
            R u = rdiv((R)jx + ju, (R)(a.width - 1));                  // main.rs:161
            R v = rdiv((R)(a.height - jy - 1) + jv, (R)(a.height - 1));  // main.rs:162-163
            ray = camera_ray(cam, u, v, rng);
            thr = Vec3<R>(1, 1, 1);
            depth_left = a.max_depth;
            slot = 1;
            alive = true;
            needy = false;
          } else {
            dropped = true;  // a tile that straddles the image edge: its outside jobs are over at once (the lane asks again)
          }
        }
        next_job += asked < avail ? asked : avail;
        const uint32_t n_dropped = (uint32_t)__popcll(__ballot(dropped));
        if (n_dropped && !direct) {
          remaining[cp] -= n_dropped;
          if (remaining[cp] == 0) flush(cp);
        }
      }
    }
    if (!walking && alive) {
      // (no walk reads the slot's block — a medium's draw is keyed by the slot's counter alone — so it is generated
      // after the walk, before the vertex: its words are not live while the lane walks)
#ifndef RT_NO_LATE_BLOCK
      rng.set_slot(slot);
#else
      rng.begin(slot);
#endif
      cnt.template inc<ST_RAYS>();
      walking = walk.start(sc, ray);
    }
    RT_STAMP(1);
    const uint32_t n_alive = (uint32_t)__popcll(__ballot(alive));
    if (n_alive == 0) break;  // nothing in flight and nothing handed out: the counter has run dry
    // refill once this many of the lanes with a path are idle (a.refill_eighths / 8 of them, at least one)
    const uint32_t refill_q = (n_alive * a.refill_eighths) >> 3;
    const uint32_t refill_at = refill_q > 0 ? refill_q : 1u;
    // (2) as render_refill_kernel
    for (;;) {
      const uint32_t n_walking = (uint32_t)__popcll(__ballot(walking));
      if (n_walking + refill_at <= n_alive) break;
      const uint32_t leaves_at = (n_walking >> a.leaf_shift) > 0 ? (n_walking >> a.leaf_shift) : 1u;
      RT_STAMP(2);
      for (;;) {
        const bool at_node = walking && !(walk.cur & REF_LEAF);
        const uint32_t n_node = (uint32_t)__popcll(__ballot(at_node));
        const uint32_t n_walk_now = (uint32_t)__popcll(__ballot(walking));
        if (n_node == 0 || n_walk_now - n_node >= leaves_at || n_walk_now + refill_at <= n_alive) break;
        if (at_node) walking = walk.node_step(sc, ray, R(0.001), stack, stack_t, (uint32_t)BLOCK, cnt);
#if RT_NODE_BURST > 1
        // a burst of node steps per round of ballots: the wave-level bookkeeping (three ballots + counts, scalar) is
        // paid once per RT_NODE_BURST steps of a lane instead of once per step (measured, Grays/s on random_scene C1 /
        // final_scene / 1 M spheres / random_scene 1920 wide: burst 1: 8.72 12.8 2.65 13.5; 2: 9.13 13.0 2.73 14.1;
        // 3: 9.16 13.4 2.75 14.1; 4: 9.14 13.1 2.74 13.9; as a rolled loop of 8: 8.50 11.6 2.69 13.0)
#pragma unroll
        for (int b = 1; b < RT_NODE_BURST; ++b)
          if (walking && !(walk.cur & REF_LEAF)) walking = walk.node_step(sc, ray, R(0.001), stack, stack_t, (uint32_t)BLOCK, cnt);
#endif
      }
      RT_STAMP(3);
      if (walking && (walk.cur & REF_LEAF))
        walking = walk.leaf_step(sc, ray, R(0.001), rng, stack, stack_t, (uint32_t)BLOCK, cnt);
      RT_STAMP(4);
    }
    RT_STAMP(5);
    // (3) lanes whose walk is over: one vertex of ray_color (main.rs:77-104)
    bool finished = false;
    if (!walking && alive) {
      Vec3<R> L(0, 0, 0);  // set once, at the path's last vertex (see render_persist_kernel)
      bool cont;
      if (walk.best == REF_NONE) {  // main.rs:104
        cnt.template inc<ST_MISSES>();
        L = madd(L, thr, background);
        cont = false;
      } else {
        cnt.template inc<ST_CLOSEST>();
        HitRec<R> h;
        make_hit_record(sc, walk.best, ray, walk.closest, false, h);
        const Material<R> mat = sc.materials[h.material];
        RT_STAMP(7);
#ifndef RT_NO_LATE_BLOCK
        rng.refill();
#endif
        cont = shade(sc, lights, mat, h, ray, thr, L, rng, cnt);
      }
      ++slot;
      --depth_left;
      if (depth_left == 0 && cont) {  // the next ray_color call would return 0
        depth_exhausted(L, thr);
        cont = false;
      }
      if (cont && thr.x == R(0) && thr.y == R(0) && thr.z == R(0)) cont = false;
      if (!cont) {
        const bool fin = rfinite(L.x) && rfinite(L.y) && rfinite(L.z);
        if (!fin) ++nonfinite;
        if (fin && !(lp & 128u)) {
          float* acc = &s_acc[((lp >> 6) & 1u) * 192u + 3u * (lp & 63u)];
          atomicAdd(acc + 0, L.x);
          atomicAdd(acc + 1, L.y);
          atomicAdd(acc + 2, L.z);
        } else if (fin) {
          R* out = accum + 3ull * rng.pixel;
          atomicAdd(out + 0, L.x);
          atomicAdd(out + 1, L.y);
          atomicAdd(out + 2, L.z);
        }
        alive = false;
        finished = !(lp & 128u);  // (an orphan's unit has been closed already)
      }
    }
    {
      const uint64_t f = __ballot(finished);
      if (f) {
        const uint32_t f1 = (uint32_t)__popcll(__ballot(finished && (lp & 64u))), f0 = (uint32_t)__popcll(f) - f1;
        remaining[0] -= f0;
        remaining[1] -= f1;
        if (f0 && remaining[0] == 0) flush(0);
        if (f1 && remaining[1] == 0) flush(1);
      }
    }
    RT_STAMP(6);
  }
#ifdef RT_STAMPS
  if ((threadIdx.x & 63) == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&stats->stamp[i], stamp_acc[i]);
#endif
#ifdef RT_BLOCK_TRACE
  {
    const uint32_t w = blockIdx.x * (BLOCK / 64) + wave;
    if (lane == 0 && w < BLOCK_TRACE_MAX)
      g_block_trace[w] = make_uint4(trace_t0, (uint32_t)wall_clock64(), __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20),
                                    __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4));
  }
#endif
  flush_stats(cnt, stats);
  uint32_t nf = wave_sum(nonfinite);
  if ((threadIdx.x & 63) == 0 && nf) atomicAdd(&stats->v[ST_NONFINITE], (unsigned long long)nf);
}

// Small scenes, f32, PERSISTENT form of the leaf-list megakernel (the default for TRAV_LINEAR).
//
// render_kernel gives a wave one 8x8 tile and every lane a fixed quota of its pixel's samples: the wave
// lasts as long as its unluckiest lane (path lengths vary: ~6 % idle at a 683-sample quota), and the
// launch as long as its last round of waves.  Here the work is cut into UNITS (one 8x8 tile x
// unit_samples samples) handed out from a global counter; a wave keeps a ring of the units it has
// drawn in LDS and every lane walks that list AT ITS OWN PACE — a lane that finishes its pixel's
// samples of unit i goes straight on to unit i+1 while its neighbours are still in unit i.  Nobody
// waits for anybody until the global counter runs dry, when the stragglers are at most one unit apart.
// A finished (pixel, unit) sum is added to the accumulator with a float atomic (units of one pixel run
// on different waves).  Same paths and draws as render_kernel: a sample's Philox stream is keyed by
// (pixel, sample), not by who renders it.
constexpr uint32_t UNIT_RING = 32;
// LDS words that the lanes of ONE wave hand to each other (the unit ring): plain ds_read / ds_write — a wave's LDS
// operations execute in program order, so all that is needed is that the compiler keeps them in that order and does
// not cache them in registers (relaxed atomics at wavefront scope + a wavefront-scope fence: no instruction is emitted
// for either).  Declared `volatile`, the same words were accessed through FLAT instructions with system-scope
// coherence bits and a full vmcnt(0) wait each — in the hand-over block that some lane of a wave enters in every
// second iteration of the f64 kernel.
RT_DEV uint32_t lds_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
RT_DEV void lds_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
RT_DEV void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#ifndef RT_PERSIST_MIN_WAVES
#define RT_PERSIST_MIN_WAVES 6  // measured (after sphere_roots stopped the f64 hoist): 4: 54.2, 5: 61.1, 6 (80 VGPRs, 32 B scratch): 65.3, 7: 63.1 Grays/s
#endif
// TRAV_REFERENCE (any precision): the same hand-out around the reference-order BVH walk — what scenes with
// non-conservative boxes, RT_FLAG_FORCE_BVH and the f64 audit run on; EARLY_EXIT / NOISE as in render_kernel.
// TRAV_LEAVES (f64): around traverse_leaves, the hot scene arrays staged in LDS once per workgroup.
// TRAV_ORDERED (f64): around traverse_ordered — the SAH tree walked nearer child first in the caller's precision, for
// scenes too large for the leaf list (same images as the reference-order walk, a fraction of its node visits).
#ifndef RT_LEAVES_MIN_WAVES
#define RT_LEAVES_MIN_WAVES 4
#endif
template <typename R, bool STATS, int TRAV = TRAV_LINEAR, bool EARLY_EXIT = true, bool NOISE = false>
__global__ void __launch_bounds__(BLOCK, TRAV == TRAV_LINEAR ? RT_PERSIST_MIN_WAVES : TRAV == TRAV_LEAVES ? RT_LEAVES_MIN_WAVES : (sizeof(R) == 8 ? RT_RENDER_F64_MIN_WAVES : RT_RENDER_MIN_WAVES))
    render_persist_kernel(SceneView<R> sc_in, CameraView<R> cam, RenderArgs a, R* __restrict__ accum, DeviceStats* stats) {
  constexpr bool LINEAR = TRAV == TRAV_LINEAR, LEAVES = TRAV == TRAV_LEAVES, ORDERED = TRAV == TRAV_ORDERED;
  static_assert(TRAV == TRAV_LINEAR || TRAV == TRAV_REFERENCE || TRAV == TRAV_LEAVES || TRAV == TRAV_ORDERED,
                "leaf list, reference-order or (run to completion) ordered walk");
  extern __shared__ uint32_t lds_stack[];  // TRAV_REFERENCE / TRAV_ORDERED: the walk's stack
  [[maybe_unused]] uint32_t* stack = lds_stack + threadIdx.x;
  // TRAV_ORDERED: 16-bit box-entry distances behind the 32-bit references (launch_impl.hpp sizes both)
  [[maybe_unused]] uint16_t* stack_t = reinterpret_cast<uint16_t*>(lds_stack + (size_t)a.stack_depth * BLOCK) + threadIdx.x;
  __shared__ Light<R> s_lights[LDS_MAX_LIGHTS];
  __shared__ float s_lin[LINEAR ? LIN_MAX_LEAVES * LIN_DWORDS : 1];
  // TRAV_LEAVES: the arrays a leaf test reads (hittable.rs:42-91 flattened), staged so that the wave-uniform leaf loop
  // reads LDS; `sc` is the scene view with those arrays re-pointed
  __shared__ Real4<R> s_spheres[LEAVES ? STAGE_MAX_SPHERES : 1];
  __shared__ Cube<R> s_cubes[LEAVES ? STAGE_MAX_CUBES : 1];
  __shared__ Complex s_complexes[LEAVES ? STAGE_MAX_COMPLEXES : 1];
  __shared__ Op<R> s_ops[LEAVES ? STAGE_MAX_OPS : 1];
  __shared__ uint32_t s_op_tag[LEAVES ? STAGE_MAX_OPS : 1];
  __shared__ uint32_t s_leaf_list[LEAVES ? LIN_MAX_LEAVES : 1];
  // ... and the camera (24 doubles that are read once per sample): in LDS they cost no scalar registers between samples
  __shared__ alignas(16) unsigned char s_cam_raw[LEAVES ? sizeof(CameraView<R>) : 16];
  // f64 leaf-list form: the lane's running pixel sum lives in LDS (3 reals, column-major by lane), not in 6 vector
  // registers the walk would spill; it is touched once per finished sample.  (Not the stack-walking forms: their LDS
  // decides how many workgroups a CU holds — 10^6 spheres in f64: 116.5 -> 130.9 ms with these 6 KB.)
#ifndef RT_NO_F64_ACC_LDS
  constexpr bool ACC_LDS = sizeof(R) == 8 && LEAVES;
#else
  constexpr bool ACC_LDS = false;
#endif
  __shared__ R s_acc_all[ACC_LDS ? 3 * BLOCK : 1];
  [[maybe_unused]] R* s_acc = s_acc_all + (ACC_LDS ? threadIdx.x : 0u);
  if constexpr (ACC_LDS) {
    s_acc[0] = R(0);
    s_acc[BLOCK] = R(0);
    s_acc[2 * BLOCK] = R(0);
  }
  SceneView<R> sc = sc_in;
  if constexpr (LEAVES) {
    if (threadIdx.x == 0) *reinterpret_cast<CameraView<R>*>(s_cam_raw) = cam;
    auto stage = [&](void* dst, const void* src, uint32_t bytes) {
      for (uint32_t i = threadIdx.x; i < bytes / 4u; i += BLOCK)
        reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(src)[i];
    };
    stage(s_spheres, sc_in.spheres, sc_in.stage_spheres * (uint32_t)sizeof(Real4<R>));
    stage(s_cubes, sc_in.cubes, sc_in.stage_cubes * (uint32_t)sizeof(Cube<R>));
    stage(s_complexes, sc_in.complexes, sc_in.stage_complexes * (uint32_t)sizeof(Complex));
    stage(s_ops, sc_in.ops, sc_in.stage_ops * (uint32_t)sizeof(Op<R>));
    stage(s_op_tag, sc_in.op_tag, sc_in.stage_ops * 4u);
    stage(s_leaf_list, sc_in.leaf_list, sc_in.n_leaf_list * 4u);
    sc.spheres = s_spheres;
    sc.cubes = s_cubes;
    sc.complexes = s_complexes;
    sc.ops = s_ops;
    sc.op_tag = s_op_tag;
    sc.leaf_list = s_leaf_list;
  }
  __shared__ uint32_t s_unit_xy[BLOCK / 64][UNIT_RING];  // tile origin x | y << 16
  __shared__ uint32_t s_unit_k[BLOCK / 64][UNIT_RING];   // which sample range of the launch
  __shared__ uint32_t s_red[BLOCK / 64][2];              // wave-wide min scratch
  {
    const uint32_t n_l = sc.n_lights < LDS_MAX_LIGHTS ? sc.n_lights : LDS_MAX_LIGHTS;
    constexpr uint32_t LW = sizeof(Light<R>) / 4;
    for (uint32_t i = threadIdx.x; i < n_l * LW; i += BLOCK)
      reinterpret_cast<uint32_t*>(s_lights)[i] = reinterpret_cast<const uint32_t*>(sc.lights)[i];
    if constexpr (LINEAR)
      for (uint32_t i = threadIdx.x; i < sc.n_linear * LIN_DWORDS; i += BLOCK) s_lin[i] = __uint_as_float(sc.linear_rec[i]);
    __syncthreads();
  }
  const Light<R>* lights = sc.n_lights <= LDS_MAX_LIGHTS ? s_lights : sc.lights;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const Vec3<R> background(sc.background[0], sc.background[1], sc.background[2]);
  // wave-uniform reciprocals of main.rs:161-163, pinned to scalar registers (the compiler keeps the result
  // of a VALU rcp in a vector register otherwise, and at 96 VGPRs that register is spilled to scratch)
  [[maybe_unused]] float inv_wm1 = 0.0f, inv_hm1 = 0.0f;
  if constexpr (LINEAR) {
    inv_wm1 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(rrcp((float)(a.width - 1)))));
    inv_hm1 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(rrcp((float)(a.height - 1)))));
  }

  Cnt<STATS> cnt;
  cnt.clear();
  uint32_t nonfinite = 0;
  [[maybe_unused]] Vec3<R> acc;
  uint32_t list_len = 0;    // wave-uniform: units this wave has drawn so far
  bool exhausted = false;   // wave-uniform: the global counter has run dry
  uint32_t ui = 0xFFFFFFFFu;  // this lane's position in the wave's list (none yet)
  uint32_t s = 0, s_end = 0, px = 0, py = 0;
  bool have_pixel = false, lane_done = false, alive = false;
  RayT<R> ray;
  Vec3<R> thr;
  Rng rng;
  rng.init(a.seed, 0, 0);
  // TRAV_ORDERED: the walk is a resumable state machine — a lane whose walk is over goes on to its vertex (and its next
  // ray) while its neighbours keep walking; the wave leaves the step loop once a.refill_eighths / 8 of its lanes wait
  [[maybe_unused]] bool walking = false;
  [[maybe_unused]] OrderedWalkR<R> walk;
  walk.cur = REF_NONE;
  walk.sp = 0;
  walk.best = REF_NONE;
  walk.closest = R(0);
  walk.a = R(0);
  walk.inv_d = Vec3<R>(0, 0, 0);
#ifdef RT_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif

  for (;;) {
    RT_STAMP(0);  // loop tail + re-convergence
    // (0) lanes that have finished their pixel's samples of a unit move on to the wave's next unit.  The common case is
    // lane-local: the list already holds the next entry (two LDS reads).  Only when a lane has run to the END of the
    // list does the wave act together: it draws the missing unit plus UNIT_PREFETCH more with ONE atomic on the global
    // counter (consecutive units; lane i fills entry i), so that this wave-wide part runs once per UNIT_PREFETCH + 1
    // hand-overs instead of at every one (with 16-sample units some lane of the wave is at a hand-over in every
    // second iteration).  Units drawn ahead are at most UNIT_PREFETCH per wave: a fraction of a millisecond of work
    // at the launch's end.
    const bool needy = !alive && s == s_end && !lane_done;
    if (__ballot(needy)) {  // wave-uniform
      if (needy && have_pixel) {
        R* out = accum + 3ull * (py * a.width + px);
        if constexpr (ACC_LDS) {
          atomicAdd(out + 0, s_acc[0]);
          atomicAdd(out + 1, s_acc[BLOCK]);
          atomicAdd(out + 2, s_acc[2 * BLOCK]);
          s_acc[0] = R(0);
          s_acc[BLOCK] = R(0);
          s_acc[2 * BLOCK] = R(0);
        } else {
          atomicAdd(out + 0, acc.x);
          atomicAdd(out + 1, acc.y);
          atomicAdd(out + 2, acc.z);
          acc = Vec3<R>(0, 0, 0);
        }
        have_pixel = false;
      }
      // a needy lane stands at ui and wants entry ui + 1 <= list_len: the list is short for it iff ui + 1 == list_len
      if (__ballot(needy && ui + 1u >= list_len) && !exhausted) {  // wave-uniform
        // the oldest entry anybody may still read: entries before min(ui + 1) over the unfinished lanes are dead
        const uint32_t leader = (uint32_t)__builtin_ctzll(__ballot(1));
        if (lane == leader) lds_store(&s_red[wave][1], 0xFFFFFFFFu);
        wave_lds_fence();
        if (!lane_done) atomicMin(&s_red[wave][1], ui + 1u);
        wave_lds_fence();
        const uint32_t oldest = __builtin_amdgcn_readfirstlane(lds_load(&s_red[wave][1]));
        const uint32_t room = UNIT_RING - (list_len - oldest);  // (list_len - oldest <= UNIT_RING always)
        uint32_t n_new = 1u + a.unit_prefetch < room ? 1u + a.unit_prefetch : room;
        if (n_new) {
          uint32_t u0 = 0;
          if (lane == leader) u0 = atomicAdd(&stats->unit_counter, n_new);
          u0 = __shfl(u0, (int)leader, 64);
          if (u0 + n_new >= a.total_units) {
            exhausted = true;
            n_new = u0 < a.total_units ? a.total_units - u0 : 0u;
          }
          if (lane < n_new) {
            const uint32_t u = u0 + lane, tile = u % a.wave_tiles, e = (list_len + lane) % UNIT_RING;
            lds_store(&s_unit_xy[wave][e], ((tile % a.wave_tiles_x) * 8u) | (((tile / a.wave_tiles_x) * 8u) << 16));
            lds_store(&s_unit_k[wave][e], u / a.wave_tiles);
          }
          list_len += n_new;
          wave_lds_fence();
        }
      }
      if (needy) {
        if (ui + 1u < list_len) {
          ++ui;
          const uint32_t xy = lds_load(&s_unit_xy[wave][ui % UNIT_RING]), k = lds_load(&s_unit_k[wave][ui % UNIT_RING]);
          px = (xy & 0xFFFFu) + (lane & 7u);
          py = (xy >> 16) + (lane >> 3);
          // s, s_end: GLOBAL sample indices (what the Philox counter takes), so that the sample in flight and the
          // loop counter are one register
          s = k * a.unit_samples;
          s_end = a.sample_begin + (s + a.unit_samples < a.n_samples ? s + a.unit_samples : a.n_samples);
          s += a.sample_begin;
          have_pixel = px < a.width && py < a.height;
          if (!have_pixel) s = s_end;  // a tile that straddles the image edge: nothing to do for this lane
          rng.pixel = py * a.width + px;
        } else if (exhausted) {
          lane_done = true;
        }  // else: the ring is full; try again once the slowest lane has moved on
      }
    }
    RT_STAMP(1);  // unit hand-over
    if (!alive) {
      if (s == s_end) {
        if (lane_done) break;
        continue;
      }
      rng.sample = s;  // ++s when the path ends
      rng.begin(0);
      R ju = rng.next<R>();
      R jv = rng.next<R>();
      R u, v;
      if constexpr (LINEAR) {
        u = ((R)px + ju) * inv_wm1;                  // main.rs:161
        v = ((R)(a.height - py - 1) + jv) * inv_hm1;  // main.rs:162-163
      } else {  // (as render_kernel: the quotient itself, which is what the oracle computes)
        u = rdiv((R)px + ju, (R)(a.width - 1));
        v = rdiv((R)(a.height - py - 1) + jv, (R)(a.height - 1));
      }
      if constexpr (LEAVES) {
        ray = camera_ray(*reinterpret_cast<const CameraView<R>*>(s_cam_raw), u, v, rng);
#ifndef RT_KEEP_LEAVES_TIME
        ray.time = R(0);  // leaf-list scenes hold no moving sphere (rt_scene_create): the shutter time is drawn, never read
#endif
      } else
        ray = camera_ray(cam, u, v, rng);
      thr = Vec3<R>(1, 1, 1);
      alive = true;
    }
    RT_STAMP(2);  // camera ray
    R t;
    uint32_t best;
    if constexpr (ORDERED) {
      if (!walking) {  // a fresh ray (camera or scattered)
        rng.begin(rng.slot + 1u);
        cnt.template inc<ST_RAYS>();
        walking = walk.start(sc, ray);
      }
      RT_STAMP(3);
      const uint32_t n_active = (uint32_t)__popcll(__ballot(true));  // lanes with a path
      const uint32_t refill_q = (n_active * a.refill_eighths) >> 3;
      const uint32_t refill_at = refill_q > 0 ? refill_q : 1u;
      for (;;) {
        const uint32_t n_walking = (uint32_t)__popcll(__ballot(walking));
        if (n_walking == 0 || n_walking + refill_at <= n_active) break;
        const uint32_t leaves_at = (n_walking >> a.leaf_shift) > 0 ? (n_walking >> a.leaf_shift) : 1u;
        for (;;) {  // node steps together ...
          const bool at_node = walking && !(walk.cur & REF_LEAF);
          const uint32_t n_node = (uint32_t)__popcll(__ballot(at_node));
          const uint32_t n_walk_now = (uint32_t)__popcll(__ballot(walking));
          if (n_node == 0 || n_walk_now - n_node >= leaves_at || n_walk_now + refill_at <= n_active) break;
          if (at_node) walking = walk.node_step(sc, ray, R(0.001), stack, stack_t, (uint32_t)BLOCK, cnt);
        }
        // ... then the leaf tests together
        if (walking && (walk.cur & REF_LEAF))
          walking = walk.leaf_step(sc, ray, R(0.001), rng, stack, stack_t, (uint32_t)BLOCK, cnt);
      }
      RT_STAMP(4);
      if (walking) continue;  // still on its way: no vertex for this lane in this round
      t = walk.closest;
      best = walk.best;
    } else {
    // slot 0 = camera, slot b = the b-th ray of the path (also its depth counter).  No walk reads the slot's block (a
    // medium's draw is keyed by the slot's counter alone): with RT_LATE_BLOCK it is generated after the walk, for the
    // lanes that hit something, and its four words are not live across the walk
#ifndef RT_NO_LATE_BLOCK
    rng.set_slot(rng.slot + 1u);
#else
    rng.begin(rng.slot + 1u);
#endif
    cnt.template inc<ST_RAYS>();
    RT_STAMP(3);  // bounce Philox block
    if constexpr (LINEAR) {
      if (sc.linear_has_media)  // wave-uniform
        traverse_linear<STATS, true>(s_lin, sc.n_linear, lin_masks(sc), ray, R(0.001), rng, t, best, cnt);
      else
        traverse_linear<STATS, false>(s_lin, sc.n_linear, lin_masks(sc), ray, R(0.001), rng, t, best, cnt);
    } else if constexpr (LEAVES) {
      traverse_leaves(sc, ray, R(0.001), rng, t, best, cnt);
    } else {
      traverse(sc, ray, R(0.001), rng, stack, (uint32_t)BLOCK, t, best, cnt);
    }
    RT_STAMP(4);  // leaf-list walk
    }
    // A path's radiance is set ONCE, at its last vertex (an emitter, a miss, the depth limit: main.rs:73-75, 101, 104) —
    // up to there ray_color's sum is the zero it starts from, so it need not be carried from vertex to vertex.
    Vec3<R> L(0, 0, 0);
    bool cont;
    if (best == REF_NONE) {  // main.rs:104
      cnt.template inc<ST_MISSES>();
      L = madd(L, thr, background);
      cont = false;
    } else {
      cnt.template inc<ST_CLOSEST>();
      HitRec<R> h;
      Material<R> mat;
      if constexpr (LINEAR) {
        lin_hit_record(sc, s_lin + best * LIN_DWORDS, ray, t, h, mat);
      } else {
        make_hit_record(sc, best, ray, t, false, h);
        mat = sc.materials[h.material];
      }
      RT_STAMP(5);  // hit record
#ifndef RT_NO_LATE_BLOCK
      if constexpr (!ORDERED) rng.refill();
#endif
      cont = shade<R, STATS, NOISE>(sc, lights, mat, h, ray, thr, L, rng, cnt);
    }
    RT_STAMP(6);  // shade (or miss)
    if (rng.slot == a.max_depth && cont) {  // the next ray_color call would return 0
      depth_exhausted(L, thr);
      cont = false;
    }
    if (EARLY_EXIT && cont && thr.x == R(0) && thr.y == R(0) && thr.z == R(0)) cont = false;
    if (!cont) {
      const bool fin = rfinite(L.x) && rfinite(L.y) && rfinite(L.z);
      if (!fin) ++nonfinite;
      if constexpr (ACC_LDS) {  // (the f64 kernel lets a non-finite sample poison the pixel like the reference does)
        s_acc[0] += L.x;
        s_acc[BLOCK] += L.y;
        s_acc[2 * BLOCK] += L.z;
      } else if (fin || sizeof(R) == 8)
        acc = acc + L;
      alive = false;
      ++s;
    }
  }
#ifdef RT_STAMPS
  if ((threadIdx.x & 63) == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&stats->stamp[i], stamp_acc[i]);
#endif
  flush_stats(cnt, stats);
  uint32_t nf = wave_sum(nonfinite);
  if ((threadIdx.x & 63) == 0 && nf) atomicAdd(&stats->v[ST_NONFINITE], (unsigned long long)nf);
}

// Leaf-list scenes, jobs NOT tied to lanes and camera rays generated A WAVE AT A TIME (render_queue_kernel; the default
// for TRAV_LINEAR in f32 and TRAV_LEAVES in f64 unless RT_NO_QUEUE is set).
//
// In render_persist_kernel a lane owns a pixel, so in every iteration of the bounce loop the wave runs Camera::get_ray
// (a Philox block, the jitter, the ray) for the lanes whose path has just ended — 41 % of them in the f32 Cornell
// kernel (2.43 rays per sample), 19 % in f64 (5.13): the camera code, a tenth of the kernel's instructions, runs at
// that lane utilisation.  Here a wave cuts its unit (8x8 tile x unit_samples samples) into BATCHES of one sample of
// all 64 pixels; a batch's 64 camera rays are generated by all 64 lanes together (lane i = pixel i: full lanes) into a
// wave-private LDS queue, and a lane whose path has ended pops the next ray from the queue (a few LDS reads), whatever
// pixel it belongs to.  A finished sample is added to its pixel's sum in LDS (two units may be open: a job carries its
// unit's parity; a unit's sums go to the image by atomics when its last job finishes; if a third unit is needed while
// the oldest still has paths in flight, that unit is closed early and its stragglers — ORPHANS — add their sample to
// the image themselves: render_units_kernel's scheme).  Same paths and draws as the other forms: a sample's Philox
// streams are keyed by (pixel, sample), not by the lane that renders it; only the order of a pixel's sum changes.
#ifndef RT_QUEUE_F32_MIN_WAVES
#define RT_QUEUE_F32_MIN_WAVES 6
#endif
#ifndef RT_QUEUE_F64_MIN_WAVES
#define RT_QUEUE_F64_MIN_WAVES 4
#endif
template <typename R, bool STATS, int TRAV, bool EARLY_EXIT, bool NOISE = false, int INST = -1>
__global__ void __launch_bounds__(BLOCK, sizeof(R) == 4 ? (TRAV == TRAV_ORDERED ? RT_REFILL_MIN_WAVES : RT_QUEUE_F32_MIN_WAVES) : RT_QUEUE_F64_MIN_WAVES)
    render_queue_kernel(SceneView<R> sc_in, CameraView<R> cam_in, RenderArgs a, R* __restrict__ accum, DeviceStats* stats) {
  constexpr bool LINEAR = TRAV == TRAV_LINEAR, LEAVES = TRAV == TRAV_LEAVES, ORDERED = TRAV == TRAV_ORDERED;
  static_assert(LINEAR || LEAVES || ORDERED || TRAV == TRAV_REFERENCE,
                "leaf-list forms, the ordered walk as a state machine, or the reference-order walk run to completion");
  // TRAV_REFERENCE (f64: scenes whose boxes clip, extensions — NOISE —, RT_FLAG_FORCE_BVH): the reference's left-first walk,
  // its 32-bit stack in dynamic LDS
  static_assert(!LINEAR || sizeof(R) == 4, "the LDS leaf records are f32");
  // TRAV_ORDERED (f64 on scenes too large for the leaf list): the walk's stack in dynamic LDS, 32-bit references and,
  // behind them, 16-bit box-entry distances (launch_impl.hpp sizes both); the walk is OrderedWalkR's state machine —
  // lanes whose walk is over go on to their vertex and take their next ray while the others keep walking
  extern __shared__ uint32_t lds_stack[];
  [[maybe_unused]] uint32_t* stack = lds_stack + threadIdx.x;
  [[maybe_unused]] uint16_t* stack_t = reinterpret_cast<uint16_t*>(lds_stack + (size_t)a.stack_depth * BLOCK) + threadIdx.x;
  __shared__ Light<R> s_lights[LDS_MAX_LIGHTS];
  __shared__ float s_lin[LINEAR ? LIN_MAX_LEAVES * LIN_DWORDS : 1];
  __shared__ Real4<R> s_spheres[LEAVES ? STAGE_MAX_SPHERES : 1];
  __shared__ Cube<R> s_cubes[LEAVES ? STAGE_MAX_CUBES : 1];
  __shared__ Complex s_complexes[LEAVES ? STAGE_MAX_COMPLEXES : 1];
  __shared__ Op<R> s_ops[LEAVES ? STAGE_MAX_OPS : 1];
  __shared__ uint32_t s_op_tag[LEAVES ? STAGE_MAX_OPS : 1];
  __shared__ uint32_t s_leaf_list[LEAVES ? LIN_MAX_LEAVES : 1];
  __shared__ alignas(16) unsigned char s_cam_raw[sizeof(CameraView<R>)];
  // per wave: the ray queue (component-major: d.xyz, time, o.xyz of entry e at [c][e]), the entries' pixels, and the
  // two open units' pixel sums ([parity][channel][pixel])
  __shared__ R s_q[BLOCK / 64][7][64];
  __shared__ uint32_t s_qpix[BLOCK / 64][64];
  __shared__ R s_sum[BLOCK / 64][2][3][64];
  SceneView<R> sc = sc_in;
  {
    if (threadIdx.x == 0) *reinterpret_cast<CameraView<R>*>(s_cam_raw) = cam_in;
    const uint32_t n_l = sc.n_lights < LDS_MAX_LIGHTS ? sc.n_lights : LDS_MAX_LIGHTS;
    constexpr uint32_t LW = sizeof(Light<R>) / 4;
    for (uint32_t i = threadIdx.x; i < n_l * LW; i += BLOCK)
      reinterpret_cast<uint32_t*>(s_lights)[i] = reinterpret_cast<const uint32_t*>(sc.lights)[i];
    if constexpr (LINEAR)
      for (uint32_t i = threadIdx.x; i < sc.n_linear * LIN_DWORDS; i += BLOCK) s_lin[i] = __uint_as_float(sc.linear_rec[i]);
    if constexpr (LEAVES) {
      auto stage = [&](void* dst, const void* src, uint32_t bytes) {
        for (uint32_t i = threadIdx.x; i < bytes / 4u; i += BLOCK)
          reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(src)[i];
      };
      stage(s_spheres, sc_in.spheres, sc_in.stage_spheres * (uint32_t)sizeof(Real4<R>));
      stage(s_cubes, sc_in.cubes, sc_in.stage_cubes * (uint32_t)sizeof(Cube<R>));
      stage(s_complexes, sc_in.complexes, sc_in.stage_complexes * (uint32_t)sizeof(Complex));
      stage(s_ops, sc_in.ops, sc_in.stage_ops * (uint32_t)sizeof(Op<R>));
      stage(s_op_tag, sc_in.op_tag, sc_in.stage_ops * 4u);
      stage(s_leaf_list, sc_in.leaf_list, sc_in.n_leaf_list * 4u);
      sc.spheres = s_spheres;
      sc.cubes = s_cubes;
      sc.complexes = s_complexes;
      sc.ops = s_ops;
      sc.op_tag = s_op_tag;
      sc.leaf_list = s_leaf_list;
    }
    __syncthreads();
  }
  const CameraView<R>& cam = *reinterpret_cast<const CameraView<R>*>(s_cam_raw);
  const Light<R>* lights = sc.n_lights <= LDS_MAX_LIGHTS ? s_lights : sc.lights;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint64_t lanes_below = (1ull << lane) - 1ull;
  R* const q = &s_q[wave][0][0];
  uint32_t* const qpix = s_qpix[wave];
  R* const sum = &s_sum[wave][0][0][0];
#pragma unroll
  for (uint32_t i = 0; i < 6; ++i) sum[i * 64u + lane] = R(0);
  __builtin_amdgcn_wave_barrier();
  const Vec3<R> background(sc.background[0], sc.background[1], sc.background[2]);
  const bool lens = cam_in.lens_radius != R(0);  // wave-uniform (a kernel argument)
  [[maybe_unused]] float inv_wm1 = 0.0f, inv_hm1 = 0.0f;
  if constexpr (LINEAR) {  // (as render_persist_kernel: wave-uniform reciprocals of main.rs:161-163 in scalar registers)
    inv_wm1 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(rrcp((float)(a.width - 1)))));
    inv_hm1 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(rrcp((float)(a.height - 1)))));
  }

  Cnt<STATS> cnt;
  cnt.clear();
  uint32_t nonfinite = 0;
  // wave-uniform: the unit being handed out (parity cp), its batches, the queue, the unfinished jobs of both open units
  uint32_t cp = 0, tx0 = 0, ty0 = 0, s0 = 0, ns = 0, batch = 0, q_head = 0, q_count = 0, q_sample = 0;
  uint32_t remaining[2] = {0, 0}, ox[2] = {0, 0}, oy[2] = {0, 0};
  bool exhausted = false;
  // per lane: the path in flight
  bool alive = false;
  uint32_t lp = 0;  // pixel (0..63 of its unit's tile) | unit parity << 6 | orphan << 7
  RayT<R> ray;
  Vec3<R> thr;
  Rng rng;
  rng.init(a.seed, 0, 0);
  [[maybe_unused]] bool walking = false;
  [[maybe_unused]] std::conditional_t<sizeof(R) == 4, OrderedWalk, OrderedWalkR<R>> walk;  // (f32: the select-only node step)
  walk.cur = REF_NONE;
  walk.sp = 0;
  walk.best = REF_NONE;
  walk.closest = R(0);
  walk.a = R(0);
  walk.inv_d = Vec3<R>(0, 0, 0);

  // unit parity p is complete (or is closed early): its sums leave LDS
  auto flush = [&](uint32_t p) {
    wave_lds_fence();
    const uint32_t fx = ox[p] + (lane & 7u), fy = oy[p] + (lane >> 3);
    if (fx < a.width && fy < a.height) {
      R* out = accum + 3ull * (fy * a.width + fx);
      atomicAdd(out + 0, sum[(p * 3u + 0u) * 64u + lane]);
      atomicAdd(out + 1, sum[(p * 3u + 1u) * 64u + lane]);
      atomicAdd(out + 2, sum[(p * 3u + 2u) * 64u + lane]);
    }
    sum[(p * 3u + 0u) * 64u + lane] = R(0);
    sum[(p * 3u + 1u) * 64u + lane] = R(0);
    sum[(p * 3u + 2u) * 64u + lane] = R(0);
    wave_lds_fence();
  };

#ifdef RT_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif
  for (;;) {
    RT_STAMP(0);
    // (1) lanes without a path take the next rays of the queue; an empty queue is refilled by the whole wave
    {
      bool needy = !alive;
      for (;;) {
        const uint64_t m = __ballot(needy);
        if (m == 0) break;
        if (q_count == 0) {
          if (batch == ns) {  // every batch of the unit has been generated: the next unit
            if (exhausted) break;
            if (remaining[cp ^ 1u] != 0) {  // (paths that outlast a whole unit: close it, they finish as orphans)
              if (alive && ((lp >> 6) & 1u) == (cp ^ 1u)) lp |= 128u;
              flush(cp ^ 1u);
              remaining[cp ^ 1u] = 0;
            }
            uint32_t u = 0;
            if (lane == 0) u = atomicAdd(&stats->unit_counter, 1u);
            u = __builtin_amdgcn_readfirstlane(u);
            if (u >= a.total_units) {
              exhausted = true;
              break;
            }
            cp ^= 1u;
            const uint32_t chunk = u / a.wave_tiles, tile = u - chunk * a.wave_tiles, trow = tile / a.wave_tiles_x;
            tx0 = (tile - trow * a.wave_tiles_x) * 8u;
            ty0 = trow * 8u;
            const uint32_t sb = chunk * a.unit_samples;
            ns = a.n_samples - sb < a.unit_samples ? a.n_samples - sb : a.unit_samples;
            s0 = a.sample_begin + sb;
            batch = 0;
            ox[cp] = tx0;
            oy[cp] = ty0;
            const uint32_t in_image =
                (uint32_t)__popcll(__ballot(tx0 + (lane & 7u) < a.width && ty0 + (lane >> 3) < a.height));
            remaining[cp] = in_image * ns;  // (a tile that straddles the image edge has fewer than 64 pixels)
          }
          // one batch: sample s0 + batch of every pixel of the tile, generated by all lanes together (main.rs:160-164)
          const uint32_t gx = tx0 + (lane & 7u), gy = ty0 + (lane >> 3);
          const bool inside = gx < a.width && gy < a.height;
          const uint64_t vm = __ballot(inside);
          wave_lds_fence();  // (the entries popped so far have been read)
          if (inside) {
            Rng cr;
            cr.init(a.seed, gy * a.width + gx, s0 + batch);
            cr.begin(0);
            const R ju = cr.next<R>();
            const R jv = cr.next<R>();
            R u, v;
            if constexpr (LINEAR) {
              u = ((R)gx + ju) * inv_wm1;
              v = ((R)(a.height - gy - 1) + jv) * inv_hm1;
            } else {  // (the quotient itself, which is what the oracle computes)
              u = rdiv((R)gx + ju, (R)(a.width - 1));
              v = rdiv((R)(a.height - gy - 1) + jv, (R)(a.height - 1));
            }
            const RayT<R> cray = camera_ray(cam, u, v, cr);
            const uint32_t e = (uint32_t)__popcll(vm & lanes_below);
            q[0 * 64u + e] = cray.d.x;
            q[1 * 64u + e] = cray.d.y;
            q[2 * 64u + e] = cray.d.z;
            q[3 * 64u + e] = cray.time;
            if (lens) {
              q[4 * 64u + e] = cray.o.x;
              q[5 * 64u + e] = cray.o.y;
              q[6 * 64u + e] = cray.o.z;
            }
            qpix[e] = lane;
          }
          wave_lds_fence();
          q_head = 0;
          q_count = (uint32_t)__popcll(vm);
          q_sample = s0 + batch;
          ++batch;
          if (q_count == 0) continue;
        }
        const uint32_t rank = (uint32_t)__popcll(m & lanes_below);
        if (needy && rank < q_count) {
          const uint32_t e = q_head + rank;
          ray.d = Vec3<R>(q[0 * 64u + e], q[1 * 64u + e], q[2 * 64u + e]);
          ray.time = q[3 * 64u + e];
          if (lens)
            ray.o = Vec3<R>(q[4 * 64u + e], q[5 * 64u + e], q[6 * 64u + e]);
          else
            ray.o = cam.origin;
          if constexpr (LEAVES) ray.time = R(0);  // leaf-list scenes hold no moving sphere: drawn, never read
          const uint32_t pl = qpix[e];
          lp = pl | (cp << 6);
          rng.pixel = (ty0 + (pl >> 3)) * a.width + tx0 + (pl & 7u);
          rng.sample = q_sample;
          rng.set_slot(0);
          thr = Vec3<R>(1, 1, 1);
          alive = true;
          needy = false;
        }
        const uint32_t asked = (uint32_t)__popcll(m), taken = asked < q_count ? asked : q_count;
        q_head += taken;
        q_count -= taken;
      }
    }
    RT_STAMP(1);  // job hand-over (and camera batches)
    if (__ballot(alive) == 0) break;  // nothing in flight and nothing left to hand out
    bool finished = false;
    [[maybe_unused]] bool at_vertex = alive;
    if constexpr (ORDERED) {
      if (alive && !walking) {  // a fresh ray (camera or scattered)
        rng.set_slot(rng.slot + 1u);
        cnt.template inc<ST_RAYS>();
        walking = walk.start(sc, ray);
      }
      RT_STAMP(3);
      const uint32_t n_alive = (uint32_t)__popcll(__ballot(alive));
      const uint32_t refill_q = (n_alive * a.refill_eighths) >> 3;
      const uint32_t refill_at = refill_q > 0 ? refill_q : 1u;
      for (;;) {
        const uint32_t n_walking = (uint32_t)__popcll(__ballot(walking));
        if (n_walking == 0 || n_walking + refill_at <= n_alive) break;
        const uint32_t leaves_at = (n_walking >> a.leaf_shift) > 0 ? (n_walking >> a.leaf_shift) : 1u;
        for (;;) {  // node steps together ...
          const bool at_node = walking && !(walk.cur & REF_LEAF);
          const uint32_t n_node = (uint32_t)__popcll(__ballot(at_node));
          const uint32_t n_walk_now = (uint32_t)__popcll(__ballot(walking));
          if (n_node == 0 || n_walk_now - n_node >= leaves_at || n_walk_now + refill_at <= n_alive) break;
          if (at_node) walking = walk.node_step(sc, ray, R(0.001), stack, stack_t, (uint32_t)BLOCK, cnt);
          if constexpr (sizeof(R) == 4) {  // (a burst of node steps per round of ballots: see render_units_kernel)
#pragma unroll
            for (int nb = 1; nb < RT_NODE_BURST; ++nb)
              if (walking && !(walk.cur & REF_LEAF)) walking = walk.node_step(sc, ray, R(0.001), stack, stack_t, (uint32_t)BLOCK, cnt);
          }
        }
        // ... then the leaf tests together
        if (walking && (walk.cur & REF_LEAF))
          walking = walk.leaf_step(sc, ray, R(0.001), rng, stack, stack_t, (uint32_t)BLOCK, cnt);
      }
      at_vertex = alive && !walking;
    }
    if (at_vertex) {  // (leaf-list forms: every lane, except in the launch's last iterations)
      R t;
      uint32_t best;
      if constexpr (ORDERED) {
        t = walk.closest;
        best = walk.best;
      } else {
      // slot b = the b-th ray of the path (also its depth counter); its block is generated after the walk
      rng.set_slot(rng.slot + 1u);
      cnt.template inc<ST_RAYS>();
      RT_STAMP(3);
      }
      if constexpr (LINEAR) {
        if (sc.linear_has_media)  // wave-uniform
          traverse_linear<STATS, true>(s_lin, sc.n_linear, lin_masks(sc), ray, R(0.001), rng, t, best, cnt);
        else
          traverse_linear<STATS, false>(s_lin, sc.n_linear, lin_masks(sc), ray, R(0.001), rng, t, best, cnt);
      } else if constexpr (LEAVES) {
        traverse_leaves(sc, ray, R(0.001), rng, t, best, cnt);
      } else if constexpr (!ORDERED) {
        if constexpr (INST == 0)  // (as render_kernel: compiled for scenes without / with instances)
          traverse_impl<false>(sc, ray, R(0.001), rng, stack, (uint32_t)BLOCK, t, best, cnt);
        else if constexpr (INST == 1)
          traverse_impl<true>(sc, ray, R(0.001), rng, stack, (uint32_t)BLOCK, t, best, cnt);
        else
          traverse(sc, ray, R(0.001), rng, stack, (uint32_t)BLOCK, t, best, cnt);
      }
      RT_STAMP(4);  // walk
      Vec3<R> L(0, 0, 0);  // set once, at the path's last vertex (see render_persist_kernel)
      bool cont;
      if (best == REF_NONE) {  // main.rs:104
        cnt.template inc<ST_MISSES>();
        L = madd(L, thr, background);
        cont = false;
      } else {
        cnt.template inc<ST_CLOSEST>();
        HitRec<R> h;
        Material<R> mat;
        if constexpr (LINEAR) {
          lin_hit_record(sc, s_lin + best * LIN_DWORDS, ray, t, h, mat);
        } else {
          make_hit_record(sc, best, ray, t, false, h);
          mat = sc.materials[h.material];
        }
        RT_STAMP(5);  // hit record
        rng.refill();
        cont = shade<R, STATS, NOISE>(sc, lights, mat, h, ray, thr, L, rng, cnt);
      }
      RT_STAMP(6);  // shade (or miss)
      if (rng.slot == a.max_depth && cont) {  // the next ray_color call would return 0
        depth_exhausted(L, thr);
        cont = false;
      }
      if (EARLY_EXIT && cont && thr.x == R(0) && thr.y == R(0) && thr.z == R(0)) cont = false;
      if (!cont) {
        const bool fin = rfinite(L.x) && rfinite(L.y) && rfinite(L.z);
        if (!fin) ++nonfinite;
        // (f32 drops a non-finite sample; the f64 kernel lets it poison the pixel like the reference does)
        if (fin || sizeof(R) == 8) {
          if (!(lp & 128u)) {
            R* acc = sum + ((lp >> 6) & 1u) * 192u + (lp & 63u);
            atomicAdd(acc + 0, L.x);
            atomicAdd(acc + 64, L.y);
            atomicAdd(acc + 128, L.z);
          } else {
            R* out = accum + 3ull * rng.pixel;
            atomicAdd(out + 0, L.x);
            atomicAdd(out + 1, L.y);
            atomicAdd(out + 2, L.z);
          }
        }
        alive = false;
        finished = !(lp & 128u);  // (an orphan's unit has been closed already)
      }
    }
    {
      const uint64_t f = __ballot(finished);
      if (f) {
        const uint32_t f1 = (uint32_t)__popcll(__ballot(finished && (lp & 64u))), f0 = (uint32_t)__popcll(f) - f1;
        remaining[0] -= f0;
        remaining[1] -= f1;
        if (f0 && remaining[0] == 0) flush(0);
        if (f1 && remaining[1] == 0) flush(1);
      }
    }
    RT_STAMP(7);
  }
#ifdef RT_STAMPS
  if ((threadIdx.x & 63) == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&stats->stamp[i], stamp_acc[i]);
#endif
  flush_stats(cnt, stats);
  uint32_t nf = wave_sum(nonfinite);
  if ((threadIdx.x & 63) == 0 && nf) atomicAdd(&stats->v[ST_NONFINITE], (unsigned long long)nf);
}

// Small scenes, f32: the leaf-list megakernel with the workgroup's paths SORTED through LDS once per
// vertex ("a wavefront queue that never leaves the CU").
//
// In render_kernel a lane owns a pixel, so every wave runs the camera code for the ~40 % of its
// lanes whose path just ended and the scatter code for the ~57 % that hit a diffuse surface: VALU lane
// utilisation 0.63.  Here a lane is a SLOT: after the closest hit and its record are known, the 256
// slots of the workgroup are permuted through LDS so that the paths that need Material::scatter sit
// in the low waves and the slots that need a new camera sample in the high waves; both code blocks
// then run in (almost) full waves.  (pixel, sample) jobs are handed out from an LDS counter, the
// radiance of a finished sample is added to the workgroup's LDS tile accumulator with ds_add_f32.
// Same paths and draws as render_kernel (a job's Philox stream is keyed by pixel and sample, not by
// lane); only the order in which a pixel's samples are summed differs.
// MEASURED (Cornell 1024^2, profiles/r01_sorted_kernel_experiment.md): lane utilisation 0.63 -> 0.76 and
// 8 % fewer VALU wave-instructions per ray, but three workgroup barriers per vertex and the state
// traffic through LDS cost more than that buys: 22.0 ms against 18.6 ms per 256 spp.  Kept as an
// opt-in experiment (RT_LINEAR_SORTED=1), not the default.
constexpr uint32_t SORT_DWORDS = 20;
#ifndef RT_SORTED_MIN_WAVES
#define RT_SORTED_MIN_WAVES 5
#endif
template <bool STATS>
__global__ void __launch_bounds__(BLOCK, RT_SORTED_MIN_WAVES) render_sorted_kernel(SceneView<float> sc, CameraView<float> cam,
                                                                                 RenderArgs a, float* __restrict__ accum,
                                                                                 DeviceStats* stats) {
  using R = float;
  __shared__ Light<R> s_lights[LDS_MAX_LIGHTS];
  __shared__ float s_lin[LIN_MAX_LEAVES * LIN_DWORDS];
  __shared__ uint32_t s_state[SORT_DWORDS * BLOCK];
  __shared__ float s_acc[3 * BLOCK];
  __shared__ uint32_t s_wave_flagged[BLOCK / 64];
  __shared__ uint32_t s_job;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  {
    const uint32_t n_l = sc.n_lights < LDS_MAX_LIGHTS ? sc.n_lights : LDS_MAX_LIGHTS;
    constexpr uint32_t LW = sizeof(Light<R>) / 4;
    for (uint32_t i = tid; i < n_l * LW; i += BLOCK)
      reinterpret_cast<uint32_t*>(s_lights)[i] = reinterpret_cast<const uint32_t*>(sc.lights)[i];
    for (uint32_t i = tid; i < sc.n_linear * LIN_DWORDS; i += BLOCK) s_lin[i] = __uint_as_float(sc.linear_rec[i]);
    for (uint32_t i = tid; i < 3 * BLOCK; i += BLOCK) s_acc[i] = 0.0f;
    if (tid == 0) s_job = 0;
    __syncthreads();
  }
  const Light<R>* lights = sc.n_lights <= LDS_MAX_LIGHTS ? s_lights : sc.lights;
  const uint32_t tiles_x = (a.width + 15u) / 16u;
  const uint32_t tx0 = (blockIdx.x % tiles_x) * 16u, ty0 = (blockIdx.x / tiles_x) * 16u;
  // local pixel lp of the 16x16 tile -> image coordinates (8x8 sub-tiles, like tile_pixel)
  auto pixel_of = [&](uint32_t lp, uint32_t& px, uint32_t& py) {
    px = tx0 + ((lp >> 6) & 1u) * 8u + (lp & 7u);
    py = ty0 + (lp >> 7) * 8u + ((lp >> 3) & 7u);
  };
  const uint32_t s_begin = (uint32_t)(((uint64_t)a.n_samples * blockIdx.y) / a.splits);
  const uint32_t s_end = (uint32_t)(((uint64_t)a.n_samples * (blockIdx.y + 1)) / a.splits);
  const uint32_t total_jobs = BLOCK * (s_end - s_begin);  // job j: local pixel j % 256, sample s_begin + j / 256
  const Vec3<R> background(sc.background[0], sc.background[1], sc.background[2]);

  Cnt<STATS> cnt;
  cnt.clear();
  uint32_t nonfinite = 0;
  bool alive = false;
  Vec3<R> ro, rd, thr, L;
  uint32_t lp = 0, smp = 0, depth_left = 0, slot = 0;
  uint32_t boundary = 0;  // slots >= boundary may take a new job this iteration
  Rng rng;
  rng.init(a.seed, 0, 0);

  // a finished sample: drop it if non-finite (DESIGN.md "f32 conditioning"), else add it to the tile
  auto finish = [&]() {
    const bool fin = rfinite(L.x) && rfinite(L.y) && rfinite(L.z);
    if (!fin) {
      ++nonfinite;
    } else {
      atomicAdd(&s_acc[3 * lp + 0], L.x);
      atomicAdd(&s_acc[3 * lp + 1], L.y);
      atomicAdd(&s_acc[3 * lp + 2], L.z);
    }
    alive = false;
  };

  for (;;) {
    // (A) free slots in the high waves take the next (pixel, sample) job: Camera::get_ray
    if (!alive && tid >= boundary) {
      const uint64_t m = __ballot(1);
      const uint32_t leader = (uint32_t)__builtin_ctzll(m);
      uint32_t base = 0;
      if (lane == leader) base = atomicAdd(&s_job, (uint32_t)__popcll(m));
      base = __shfl(base, (int)leader, 64);
      const uint32_t j = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      if (j < total_jobs) {
        lp = j & (BLOCK - 1u);
        smp = a.sample_begin + s_begin + (j >> 8);
        uint32_t px, py;
        pixel_of(lp, px, py);
        if (px < a.width && py < a.height) {
          rng.pixel = py * a.width + px;
          rng.sample = smp;
          rng.begin(0);
          R ju = rng.next<R>();
          R jv = rng.next<R>();
          R u = rdiv((R)px + ju, (R)(a.width - 1));                  // main.rs:161
          R v = rdiv((R)(a.height - py - 1) + jv, (R)(a.height - 1));  // main.rs:162-163
          const RayT<R> cr = camera_ray(cam, u, v, rng);
          ro = cr.o;
          rd = cr.d;
          thr = Vec3<R>(1, 1, 1);
          L = Vec3<R>(0, 0, 0);
          depth_left = a.max_depth;
          slot = 1;
          alive = true;
        }
      }
    }
    // all paths done and all jobs handed out?
    if (!__syncthreads_or(alive ? 1 : 0) && s_job >= total_jobs) break;

    // (B) closest hit: the leaf list (hittable.rs:387-399 over the BVH's leaves)
    R t = R(0);
    uint32_t best = REF_NONE;
    if (alive) {
      uint32_t px, py;
      pixel_of(lp, px, py);
      rng.pixel = py * a.width + px;
      rng.sample = smp;
      rng.set_slot(slot);
      cnt.template inc<ST_RAYS>();
      RayT<R> ray;
      ray.o = ro;
      ray.d = rd;
      ray.time = R(0);  // leaf-list scenes hold no moving sphere
      if (sc.linear_has_media)  // wave-uniform
        traverse_linear<STATS, true>(s_lin, sc.n_linear, lin_masks(sc), ray, R(0.001), rng, t, best, cnt);
      else
        traverse_linear<STATS, false>(s_lin, sc.n_linear, lin_masks(sc), ray, R(0.001), rng, t, best, cnt);
    }
    // (C) the winner's HitRecord; misses and emitters end here (main.rs:101-104), the rest needs scatter
    bool needs_scatter = false;
    HitRec<R> h;
    h.p = Vec3<R>(0, 0, 0);
    h.normal = Vec3<R>(0, 0, 0);
    h.uv.x = h.uv.y = R(0);
    h.front_face = false;
    if (alive) {
      if (best == REF_NONE) {
        cnt.template inc<ST_MISSES>();
        L = madd(L, thr, background);
        finish();
      } else {
        cnt.template inc<ST_CLOSEST>();
        RayT<R> ray;
        ray.o = ro;
        ray.d = rd;
        ray.time = R(0);
        Material<R> mat;
        lin_hit_record(sc, s_lin + best * LIN_DWORDS, ray, t, h, mat);
        if (mat.tag == MAT_DIFFUSE_LIGHT) {
          cnt.template inc<ST_EMITTER>();
          if (h.front_face) L = madd(L, thr, material_color(sc, mat, h));
          finish();
        } else {
          needs_scatter = true;
        }
      }
    }
    // (D) permute the workgroup's slots: scatter work to the low end, free slots to the high end
    const uint64_t fm = __ballot(needs_scatter);
    if (lane == 0) s_wave_flagged[wave] = (uint32_t)__popcll(fm);
    __syncthreads();
    uint32_t flagged_before = 0, flagged_total = 0;
#pragma unroll
    for (uint32_t w = 0; w < BLOCK / 64; ++w) {
      const uint32_t c = s_wave_flagged[w];
      flagged_before += w < wave ? c : 0u;
      flagged_total += c;
    }
    const uint32_t in_wave = (uint32_t)__popcll(fm & ((1ull << lane) - 1ull));
    const uint32_t pos = needs_scatter ? flagged_before + in_wave
                                       : BLOCK - 1u - ((wave * 64u - flagged_before) + (lane - in_wave));
    {
      uint32_t* st = s_state + pos;
      st[0 * BLOCK] = __float_as_uint(h.p.x);
      st[1 * BLOCK] = __float_as_uint(h.p.y);
      st[2 * BLOCK] = __float_as_uint(h.p.z);
      st[3 * BLOCK] = __float_as_uint(rd.x);
      st[4 * BLOCK] = __float_as_uint(rd.y);
      st[5 * BLOCK] = __float_as_uint(rd.z);
      st[6 * BLOCK] = __float_as_uint(h.normal.x);
      st[7 * BLOCK] = __float_as_uint(h.normal.y);
      st[8 * BLOCK] = __float_as_uint(h.normal.z);
      st[9 * BLOCK] = __float_as_uint(thr.x);
      st[10 * BLOCK] = __float_as_uint(thr.y);
      st[11 * BLOCK] = __float_as_uint(thr.z);
      st[12 * BLOCK] = __float_as_uint(L.x);
      st[13 * BLOCK] = __float_as_uint(L.y);
      st[14 * BLOCK] = __float_as_uint(L.z);
      st[15 * BLOCK] = __float_as_uint(h.uv.x);
      st[16 * BLOCK] = __float_as_uint(h.uv.y);
      st[17 * BLOCK] = (best & 0xFFu) | (h.front_face ? 0x100u : 0u) | (lp << 16);
      st[18 * BLOCK] = smp;
      st[19 * BLOCK] = (depth_left & 0xFFFFu) | (slot << 16);
    }
    __syncthreads();
    {
      const uint32_t* st = s_state + tid;
      h.p = Vec3<R>(__uint_as_float(st[0 * BLOCK]), __uint_as_float(st[1 * BLOCK]), __uint_as_float(st[2 * BLOCK]));
      rd = Vec3<R>(__uint_as_float(st[3 * BLOCK]), __uint_as_float(st[4 * BLOCK]), __uint_as_float(st[5 * BLOCK]));
      h.normal = Vec3<R>(__uint_as_float(st[6 * BLOCK]), __uint_as_float(st[7 * BLOCK]), __uint_as_float(st[8 * BLOCK]));
      thr = Vec3<R>(__uint_as_float(st[9 * BLOCK]), __uint_as_float(st[10 * BLOCK]), __uint_as_float(st[11 * BLOCK]));
      L = Vec3<R>(__uint_as_float(st[12 * BLOCK]), __uint_as_float(st[13 * BLOCK]), __uint_as_float(st[14 * BLOCK]));
      h.uv.x = __uint_as_float(st[15 * BLOCK]);
      h.uv.y = __uint_as_float(st[16 * BLOCK]);
      const uint32_t m0 = st[17 * BLOCK], m2 = st[19 * BLOCK];
      best = m0 & 0xFFu;
      h.front_face = (m0 & 0x100u) != 0;
      lp = m0 >> 16;
      smp = st[18 * BLOCK];
      depth_left = m2 & 0xFFFFu;
      slot = m2 >> 16;
    }
    boundary = flagged_total;
    alive = tid < boundary;
    // (E) Material::scatter + the mixture pdf for the sorted paths (main.rs:77-100)
    if (alive) {
      uint32_t px, py;
      pixel_of(lp, px, py);
      rng.pixel = py * a.width + px;
      rng.sample = smp;
      rng.begin(slot);
      const float* rec = s_lin + best * LIN_DWORDS;
      Material<R> mat;
      mat.tag = __float_as_uint(rec[LIN_MAT + 0]);
      mat.texture = __float_as_uint(rec[LIN_MAT + 1]);
      mat.flags = __float_as_uint(rec[LIN_MAT + 2]);
      mat.pad = 0;
      for (int i = 0; i < 4; ++i) mat.p[i] = rec[LIN_MAT + 4 + i];
      for (int i = 0; i < 3; ++i) mat.color[i] = rec[LIN_MAT + 8 + i];
      h.material = REF_NONE;
      h.object = __float_as_uint(rec[LIN_OBJECT]);
      RayT<R> ray;
      ray.o = h.p;
      ray.d = rd;
      ray.time = R(0);
      bool cont = shade(sc, lights, mat, h, ray, thr, L, rng, cnt);
      ro = ray.o;
      rd = ray.d;
      ++slot;
      --depth_left;
      if (depth_left == 0 && cont) {  // the next ray_color call would return 0
        depth_exhausted(L, thr);
        cont = false;
      }
      if (cont && thr.x == R(0) && thr.y == R(0) && thr.z == R(0)) cont = false;
      if (!cont) finish();
    }
  }
  __syncthreads();
  {
    uint32_t px, py;
    pixel_of(tid, px, py);
    if (px < a.width && py < a.height) {
      R* out = accum + 3ull * (py * a.width + px);
      const R x = s_acc[3 * tid + 0], y = s_acc[3 * tid + 1], z = s_acc[3 * tid + 2];
      if (a.splits > 1) {
        atomicAdd(out + 0, x);
        atomicAdd(out + 1, y);
        atomicAdd(out + 2, z);
      } else if (a.accumulate) {
        out[0] += x;
        out[1] += y;
        out[2] += z;
      } else {
        out[0] = x;
        out[1] = y;
        out[2] = z;
      }
    }
  }
  flush_stats(cnt, stats);
  uint32_t nf = wave_sum(nonfinite);
  if ((threadIdx.x & 63) == 0 && nf) atomicAdd(&stats->v[ST_NONFINITE], (unsigned long long)nf);
}

// world.hit() for the deterministic primary ray of every pixel: parity levels 1-2.
// TRAV = 0: the reference's left-first walk; 1 / 2 (f32): the production walks (leaf list / ordered walk of the
// SAH tree), so that their winner can be compared with the reference walk's pixel for pixel.
template <typename R, int TRAV = 0>
__global__ void __launch_bounds__(BLOCK) primary_kernel(SceneView<R> sc, CameraView<R> cam, uint32_t width,
                                                        uint32_t height, uint64_t seed, R time,
                                                        uint32_t* __restrict__ prim_id, R* __restrict__ t_out,
                                                        R* __restrict__ normal, R* __restrict__ uv,
                                                        uint8_t* __restrict__ front_face, uint32_t stack_depth) {
  extern __shared__ uint32_t lds_stack[];
  uint32_t* stack = lds_stack + threadIdx.x;
  uint32_t px, py;
  tile_pixel(width, px, py);
  if (px >= width || py >= height) return;
  const uint32_t pixel = py * width + px;
  R u = ((R)px + R(0.5)) / (R)(width - 1);
  R v = ((R)(height - py - 1) + R(0.5)) / (R)(height - 1);
  RayT<R> ray;
  ray.o = cam.origin;
  ray.d = primary_direction(cam, u, v);
  ray.time = time;
  Rng rng;
  rng.init(seed, pixel, 0);
  rng.begin(1);
  Cnt<false> cnt;
  cnt.clear();
  R t;
  uint32_t best;
  if constexpr (TRAV == TRAV_ORDERED) {
    OrderedWalk walk;
    bool more = walk.start(sc, ray);
    while (more)
      more = walk.step(sc, ray, R(0.001), rng, stack, reinterpret_cast<uint16_t*>(lds_stack + stack_depth * BLOCK) + threadIdx.x,
                       (uint32_t)BLOCK, cnt);
    t = walk.closest;
    best = walk.best;
  } else if constexpr (TRAV == TRAV_LINEAR) {
    uint32_t k;
    traverse_linear<false, true>(reinterpret_cast<const float*>(sc.linear_rec), sc.n_linear, lin_masks(sc), ray, R(0.001), rng, t, k, cnt);
    best = k == REF_NONE ? REF_NONE : sc.linear_rec[k * LIN_DWORDS + LIN_REF];
  } else {
    traverse(sc, ray, R(0.001), rng, stack, (uint32_t)BLOCK, t, best, cnt);
  }
  HitRec<R> h;
  h.object = REF_NONE;
  h.normal = Vec3<R>(0, 0, 0);
  h.uv.x = h.uv.y = R(0);
  h.front_face = false;
  if (best != REF_NONE) make_hit_record(sc, best, ray, t, true, h);
  prim_id[pixel] = h.object;
  t_out[pixel] = t;
  normal[3ull * pixel + 0] = h.normal.x;
  normal[3ull * pixel + 1] = h.normal.y;
  normal[3ull * pixel + 2] = h.normal.z;
  if (uv) {
    uv[2ull * pixel + 0] = h.uv.x;
    uv[2ull * pixel + 1] = h.uv.y;
  }
  if (front_face) front_face[pixel] = h.front_face ? 1 : 0;
}

// rt_primary_hits / rt_primary_hits_fast in f32: the product walk's answer, plus the error-tracked audit of audit.hpp.
// The ray is the deterministic pixel-centre ray formed in f64 (the reference's expression, cam.rs:68) and rounded
// ONCE to f32, so that its rounding delta is half an ulp per component (the render kernels form their jittered rays
// in f32: their direction error is set by the 24-bit jitter anyway).  unsure_out[pixel] = 1 asks the host to have the
// pixel re-decided by primary_fix_kernel (f64 TU).
template <int TRAV>
__global__ void __launch_bounds__(BLOCK) primary_f32_kernel(SceneView<float> sc, SceneView<double> sc64, CameraView<double> cam64,
                                                            uint32_t width, uint32_t height, uint64_t seed, double time,
                                                            uint32_t* __restrict__ prim_id, float* __restrict__ t_out,
                                                            float* __restrict__ normal, float* __restrict__ uv,
                                                            uint8_t* __restrict__ front_face, uint8_t* __restrict__ unsure_out,
                                                            uint32_t stack_depth) {
  using R = float;
  extern __shared__ uint32_t lds_stack[];
  uint32_t* stack = lds_stack + threadIdx.x;
  uint32_t px, py;
  tile_pixel(width, px, py);
  if (px >= width || py >= height) return;
  const uint32_t pixel = py * width + px;
  RayT<double> r64;
  {
    const double u = ((double)px + 0.5) / (double)(width - 1);
    const double v = ((double)(height - py - 1) + 0.5) / (double)(height - 1);
    r64.o = cam64.origin;
    r64.d = cam64.lower_left + u * cam64.horizontal + v * cam64.vertical - cam64.origin;
    r64.time = time;
  }
  RayT<R> ray;
  ray.o = Vec3<R>((R)r64.o.x, (R)r64.o.y, (R)r64.o.z);
  ray.d = Vec3<R>((R)r64.d.x, (R)r64.d.y, (R)r64.d.z);
  ray.time = (R)time;
  Rng rng;
  rng.init(seed, pixel, 0);
  rng.begin(1);
  Cnt<false> cnt;
  cnt.clear();
  R t;
  uint32_t best;
  if constexpr (TRAV == TRAV_ORDERED) {
    OrderedWalk walk;
    bool more = walk.start(sc, ray);
    while (more)
      more = walk.step(sc, ray, R(0.001), rng, stack, reinterpret_cast<uint16_t*>(lds_stack + stack_depth * BLOCK) + threadIdx.x,
                       (uint32_t)BLOCK, cnt);
    t = walk.closest;
    best = walk.best;
  } else if constexpr (TRAV == TRAV_LINEAR) {
    uint32_t k;
    traverse_linear<false, true>(reinterpret_cast<const float*>(sc.linear_rec), sc.n_linear, lin_masks(sc), ray, R(0.001), rng, t, k, cnt);
    best = k == REF_NONE ? REF_NONE : sc.linear_rec[k * LIN_DWORDS + LIN_REF];
  } else {
    traverse(sc, ray, R(0.001), rng, stack, (uint32_t)BLOCK, t, best, cnt);
  }
  // the audit: same walk order as the reference, same leaf arithmetic, with error bounds
  R ta, e_t;
  uint32_t besta;
  uint32_t unsure = traverse_audit(sc, sc64, ray, r64, R(0.001), rng, stack, (uint32_t)BLOCK, ta, besta, e_t);
  if (besta != best || !(ta == t)) unsure |= AUD_ACCURACY;  // (never observed: the walks share their leaf code)
  HitRec<R> h;
  h.object = REF_NONE;
  h.normal = Vec3<R>(0, 0, 0);
  h.uv.x = h.uv.y = R(0);
  h.front_face = false;
  if (best != REF_NONE) {
    make_hit_record(sc, best, ray, t, true, h);
    if (audit_record_unsure(sc, sc64, best, ray, r64, t, e_t)) unsure |= AUD_CUBE;
  }
  prim_id[pixel] = h.object;
  t_out[pixel] = t;
  normal[3ull * pixel + 0] = h.normal.x;
  normal[3ull * pixel + 1] = h.normal.y;
  normal[3ull * pixel + 2] = h.normal.z;
  if (uv) {
    uv[2ull * pixel + 0] = h.uv.x;
    uv[2ull * pixel + 1] = h.uv.y;
  }
  if (front_face) front_face[pixel] = h.front_face ? 1 : 0;
  unsure_out[pixel] = (uint8_t)unsure;  // AUD_* mask: non-zero = re-decide
}

#ifdef RT_DEFINE_F64_FIX
// The pixels primary_f32_kernel marked: world.hit() in f64 exactly as primary_kernel<double> (and the oracle) evaluates
// it, written into the f32 outputs.  Lives in the f64 translation unit (contraction off): an exact tie — the Cornell
// box's image diagonal runs along the seam of two walls — must break the way the reference's arithmetic breaks it.
__global__ void __launch_bounds__(BLOCK) primary_fix_kernel(SceneView<double> sc, CameraView<double> cam, uint32_t width,
                                                            uint32_t height, uint64_t seed, double time,
                                                            const uint8_t* __restrict__ unsure, uint32_t* __restrict__ prim_id,
                                                            float* __restrict__ t_out, float* __restrict__ normal,
                                                            float* __restrict__ uv, uint8_t* __restrict__ front_face) {
  using R = double;
  extern __shared__ uint32_t lds_stack[];
  uint32_t* stack = lds_stack + threadIdx.x;
  uint32_t px, py;
  tile_pixel(width, px, py);
  if (px >= width || py >= height) return;
  const uint32_t pixel = py * width + px;
  if (!unsure[pixel]) return;
  R u = ((R)px + R(0.5)) / (R)(width - 1);
  R v = ((R)(height - py - 1) + R(0.5)) / (R)(height - 1);
  RayT<R> ray;
  ray.o = cam.origin;
  ray.d = primary_direction(cam, u, v);
  ray.time = time;
  Rng rng;
  rng.init(seed, pixel, 0);
  rng.begin(1);
  Cnt<false> cnt;
  cnt.clear();
  R t;
  uint32_t best;
  traverse(sc, ray, R(0.001), rng, stack, (uint32_t)BLOCK, t, best, cnt);
  HitRec<R> h;
  h.object = REF_NONE;
  h.normal = Vec3<R>(0, 0, 0);
  h.uv.x = h.uv.y = R(0);
  h.front_face = false;
  if (best != REF_NONE) make_hit_record(sc, best, ray, t, true, h);
  prim_id[pixel] = h.object;
  t_out[pixel] = (float)t;
  normal[3ull * pixel + 0] = (float)h.normal.x;
  normal[3ull * pixel + 1] = (float)h.normal.y;
  normal[3ull * pixel + 2] = (float)h.normal.z;
  if (uv) {
    uv[2ull * pixel + 0] = (float)h.uv.x;
    uv[2ull * pixel + 1] = (float)h.uv.y;
  }
  if (front_face) front_face[pixel] = h.front_face ? 1 : 0;
}

// main.rs:167-168, 51-63 in the reference's own arithmetic: the f64 SUM divided by spp, gamma 2, clamp, x256, truncate
// (NaN -> 0 like Rust's saturating `as u8`)
__global__ void tonemap_f64_kernel(const double* __restrict__ accum, uint64_t n, double spp, uint8_t* __restrict__ out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    double g = sqrt(accum[i] / spp);
    uint8_t q = 0;
    if (g == g) {
      g = g < 0.0 ? 0.0 : (g > 0.9999 ? 0.9999 : g);
      q = (uint8_t)(g * 256.0);
    }
    out[i] = q;
  }
}
#endif  // RT_DEFINE_F64_FIX

#ifdef RT_DEFINE_COMMON
// main.rs:167-168, 51-63
__global__ void tonemap_kernel(const float* __restrict__ accum, uint64_t n, float inv_spp, uint8_t* __restrict__ out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float g = sqrtf(accum[i] * inv_spp);
    uint8_t q = 0;
    if (g == g) {
      g = g < 0.0f ? 0.0f : (g > 0.9999f ? 0.9999f : g);
      q = (uint8_t)(g * 256.0f);
    }
    out[i] = q;
  }
}

// n dwords of zeros.  The render entries clear their accumulator and counter blocks with this, not with hipMemsetAsync:
// captured into a HIP graph (rt_render_device_noalloc's callers), a memset node followed by a kernel node was replayed
// without the kernel waiting for it (ROCm 7.2: units that finished early lost their sums from the second replay on);
// kernel-after-kernel on one stream is ordered in a graph as it is on the stream.
__global__ void zero_kernel(uint32_t* __restrict__ dst, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = 0u;
}

// dst[i] = ((dst[i] + src[i]) + src[n + i]) + ... over n_src images of n values each, in that order (the accumulators
// of the other devices' sample shards, rt_multi_render / rt_multi_render_f64)
template <typename T>
__global__ void accum_add_kernel(T* __restrict__ dst, const T* __restrict__ src, uint64_t n, uint32_t n_src) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    T a = dst[i];
    for (uint32_t k = 0; k < n_src; ++k) a += src[(uint64_t)k * n + i];
    dst[i] = a;
  }
}

#endif  // RT_DEFINE_COMMON

template <typename R>
__global__ void kat_sphere_uv_kernel(uint32_t n, const double* p, double* uv) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Vec2<R> r = get_sphere_uv(Vec3<R>((R)p[3 * i], (R)p[3 * i + 1], (R)p[3 * i + 2]));
  uv[2 * i] = (double)r.x;
  uv[2 * i + 1] = (double)r.y;
}
// The f64 kernels' transcendental functions (device libm), for tests that measure where they differ from the host's:
// op 0 sin(a), 1 cos(a), 2 log(a), 3 atan2(a, b), 4 acos(a).
#ifdef RT_DEFINE_LIBM_KAT  // (one definition: the f64 translation unit's)
__global__ void kat_libm_kernel(uint32_t n, uint32_t op, const double* a, const double* b, double* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = a[i], y = b[i];
  out[i] = op == 0 ? sin(x) : op == 1 ? cos(x) : op == 2 ? log(x) : op == 3 ? atan2(x, y) : acos(x);
}
#endif

template <typename R>
__global__ void kat_rotate_kernel(uint32_t n, const double* angle, const double* axis, const double* v, double* out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Quat<R> q = new_quaternion((R)angle[i], Vec3<R>((R)axis[3 * i], (R)axis[3 * i + 1], (R)axis[3 * i + 2]));
  Vec3<R> r = qrotate(Vec3<R>((R)v[3 * i], (R)v[3 * i + 1], (R)v[3 * i + 2]), q);
  out[3 * i] = (double)r.x;
  out[3 * i + 1] = (double)r.y;
  out[3 * i + 2] = (double)r.z;
}
#ifdef RT_DEFINE_COMMON
// path_stream != 0: the function the render kernels draw from (philox_path, RT_PATH_PHILOX_ROUNDS);
// else Philox4x32-10, the host's scene/BVH-axis generator, for the published 10-round vectors.
__global__ void kat_philox_kernel(uint32_t n, const uint32_t* ctr, const uint32_t* key, int path_stream,
                                  uint32_t* out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t c0 = ctr[4 * i], c1 = ctr[4 * i + 1], c2 = ctr[4 * i + 2], c3 = ctr[4 * i + 3];
  const uint32_t k0 = key[2 * i], k1 = key[2 * i + 1];
  U4 r = path_stream ? philox_path(c0, c1, c2, c3, k0, k1) : philox4x32<10>(c0, c1, c2, c3, k0, k1);
  out[4 * i] = r.x;
  out[4 * i + 1] = r.y;
  out[4 * i + 2] = r.z;
  out[4 * i + 3] = r.w;
}
#endif  // RT_DEFINE_COMMON

}  // namespace rt

#if RT_HAS_LINEAR
#include "wavefront.hpp"  // f32 only: the queue form of the ordered-walk path
#endif
