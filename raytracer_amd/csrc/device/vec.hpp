// Device-side vector math, templated on the scalar type (float for the production
// kernels, double for the audit kernels).  Semantics follow the reference's use of the
// un-vendored `vecmat` crate and alias.rs:101-181: element-wise * and /, normalize = v / |v|.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

#define RT_DEV __device__ __forceinline__

template <typename R> struct Consts;
template <> struct Consts<float> {
  static constexpr float pi = 3.14159265358979323846f;
  static constexpr float inf = __builtin_huge_valf();
};
template <> struct Consts<double> {
  static constexpr double pi = 3.14159265358979323846;
  static constexpr double inf = __builtin_huge_val();
};

RT_DEV float rroot(float x) { return __builtin_amdgcn_sqrtf(x); }  // v_sqrt_f32, 1 ulp (sqrtf's IEEE fix-up costs ~15 more)
RT_DEV double rroot(double x) { return sqrt(x); }
RT_DEV float rsin(float x) { return sinf(x); }
RT_DEV double rsin(double x) { return sin(x); }
RT_DEV float rcos(float x) { return cosf(x); }
RT_DEV double rcos(double x) { return cos(x); }
RT_DEV float rlog(float x) { return logf(x); }
RT_DEV double rlog(double x) { return log(x); }
RT_DEV float racos(float x) { return acosf(x); }
RT_DEV double racos(double x) { return acos(x); }
RT_DEV float ratan2(float y, float x) { return atan2f(y, x); }
RT_DEV double ratan2(double y, double x) { return atan2(y, x); }
RT_DEV float rabs(float x) { return fabsf(x); }
RT_DEV double rabs(double x) { return fabs(x); }
RT_DEV float rmin(float a, float b) { return fminf(a, b); }  // ignores a NaN operand, like f64::min
RT_DEV double rmin(double a, double b) { return fmin(a, b); }
RT_DEV float rmax(float a, float b) { return fmaxf(a, b); }
RT_DEV double rmax(double a, double b) { return fmax(a, b); }
// Division / normalisation: the f64 audit kernels divide exactly like the reference; the f32
// production kernels use the hardware reciprocal and reciprocal square root (1 ulp), which is
// far inside the 1e-5 parity bound and several times cheaper than IEEE division on gfx950.
RT_DEV float rrcp(float x) { return __builtin_amdgcn_rcpf(x); }
RT_DEV double rrcp(double x) { return 1.0 / x; }
RT_DEV float rdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
RT_DEV double rdiv(double a, double b) { return a / b; }
// sin and cos of 2*pi*x for x in [0,1): v_sin_f32 / v_cos_f32 take revolutions directly.
RT_DEV void sincos_2pi(float x, float& s, float& c) {
  s = __builtin_amdgcn_sinf(x);
  c = __builtin_amdgcn_cosf(x);
}
RT_DEV void sincos_2pi(double x, double& s, double& c) {
  const double phi = 2.0 * 3.14159265358979323846 * x;  // alias.rs:78, 90
  c = cos(phi);  // (sincos(phi, &s, &c) measured: 2 % fewer instructions, no faster)
  s = sin(phi);
}
// sin(10 * x) for the checker texture (texture.rs:38): f32 reduces the argument in revolutions.
RT_DEV float sin_scaled10(float x) { return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(x * 1.5915494309189535f)); }
RT_DEV double sin_scaled10(double x) { return sin(10.0 * x); }
RT_DEV bool rfinite(float x) { return __builtin_isfinite(x); }
RT_DEV bool rfinite(double x) { return __builtin_isfinite(x); }

template <typename R> struct Vec3 {
  R x, y, z;
  __host__ __device__ __forceinline__ Vec3() : x(0), y(0), z(0) {}
  __host__ __device__ __forceinline__ Vec3(R x_, R y_, R z_) : x(x_), y(y_), z(z_) {}
};
template <typename R> struct Vec2 {
  R x, y;
};

template <typename R> RT_DEV Vec3<R> operator+(Vec3<R> a, Vec3<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename R> RT_DEV Vec3<R> operator-(Vec3<R> a, Vec3<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename R> RT_DEV Vec3<R> operator-(Vec3<R> a) { return {-a.x, -a.y, -a.z}; }
template <typename R> RT_DEV Vec3<R> operator*(Vec3<R> a, Vec3<R> b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
template <typename R> RT_DEV Vec3<R> operator/(Vec3<R> a, Vec3<R> b) {
  return {rdiv(a.x, b.x), rdiv(a.y, b.y), rdiv(a.z, b.z)};
}
template <typename R> RT_DEV Vec3<R> operator*(R s, Vec3<R> a) { return {s * a.x, s * a.y, s * a.z}; }
template <typename R> RT_DEV Vec3<R> operator*(Vec3<R> a, R s) { return {a.x * s, a.y * s, a.z * s}; }
template <typename R> RT_DEV Vec3<R> operator/(Vec3<R> a, R s) {
  if constexpr (sizeof(R) == 4) {
    const R inv = rrcp(s);
    return {a.x * inv, a.y * inv, a.z * inv};
  } else {
    return {a.x / s, a.y / s, a.z / s};
  }
}
// a0 b0 + a1 b1 + a2 b2.  f32: ONE fixed fused form.  Under HIP's default -ffp-contract=fast the compiler decides per inlined
// copy whether and how to fuse `x*y + z*w + ...`; two walks that reach the same leaf through different code
// (leaf list, reference-order walk, ordered walk) would then disagree in the last bit of a dot product and,
// now and then, of t.  f64 (the audit kernels, -ffp-contract=off) keeps the oracle's plain expression.
template <typename R> RT_DEV R mul_add3(R a0, R b0, R a1, R b1, R a2, R b2) {
  if constexpr (sizeof(R) == 4)
    return __builtin_fmaf(a2, b2, __builtin_fmaf(a1, b1, a0 * b0));
  else
    return a0 * b0 + a1 * b1 + a2 * b2;
}
template <typename R> RT_DEV R dot(Vec3<R> a, Vec3<R> b) { return mul_add3(a.x, b.x, a.y, b.y, a.z, b.z); }
// a + s * b and a + b * c per component.  The f32 translation unit is compiled with -ffp-contract=on, which
// fuses only inside one source expression: the same source gives the same bits in every kernel, but a sum
// built from Vec3 operators (two function calls) would stay unfused — these helpers fuse it explicitly.
// f64 (audit kernels, contraction off): the oracle's plain operations.
template <typename R> RT_DEV Vec3<R> madd(Vec3<R> a, R s, Vec3<R> b) {
  if constexpr (sizeof(R) == 4)
    return {__builtin_fmaf(s, b.x, a.x), __builtin_fmaf(s, b.y, a.y), __builtin_fmaf(s, b.z, a.z)};
  else
    return a + s * b;
}
template <typename R> RT_DEV Vec3<R> madd(Vec3<R> a, Vec3<R> b, Vec3<R> c) {
  if constexpr (sizeof(R) == 4)
    return {__builtin_fmaf(b.x, c.x, a.x), __builtin_fmaf(b.y, c.y, a.y), __builtin_fmaf(b.z, c.z, a.z)};
  else
    return a + b * c;
}
template <typename R> RT_DEV Vec3<R> cross(Vec3<R> a, Vec3<R> b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <typename R> RT_DEV R sqrlen(Vec3<R> a) { return dot(a, a); }
template <typename R> RT_DEV R length(Vec3<R> a) { return rroot(sqrlen(a)); }
template <typename R> RT_DEV Vec3<R> normalize(Vec3<R> a) {
  if constexpr (sizeof(R) == 4) {
    const R inv = __builtin_amdgcn_rsqf(sqrlen(a));
    return {a.x * inv, a.y * inv, a.z * inv};
  } else {
    return a / length(a);
  }
}
template <typename R> RT_DEV R comp(Vec3<R> a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// alias.rs:101-103
template <typename R> RT_DEV Vec3<R> reflect(Vec3<R> v, Vec3<R> n) {
  if constexpr (sizeof(R) == 4)
    return madd(v, -(R(2) * dot(v, n)), n);
  else
    return v - (R(2) * dot(v, n)) * n;
}
// alias.rs:105-110
template <typename R> RT_DEV Vec3<R> refract(Vec3<R> v, Vec3<R> n, R etai_over_etat) {
  R cos_theta = rmin(dot(-v, n), R(1));
  Vec3<R> perp = etai_over_etat * (v + cos_theta * n);
  Vec3<R> par = (-rroot(rabs(R(1) - sqrlen(perp)))) * n;
  return perp + par;
}

// alias.rs:129-147: quaternion (w, x, y, z)
template <typename R> struct Quat {
  R w, x, y, z;
};
template <typename R> RT_DEV Quat<R> qinvert(Quat<R> q) { return {q.w, -q.x, -q.y, -q.z}; }
template <typename R> RT_DEV Quat<R> hamilton(Quat<R> a, Quat<R> b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
// alias.rs:112-117: v ↦ q v q⁻¹
template <typename R> RT_DEV Vec3<R> qrotate(Vec3<R> v, Quat<R> q) {
  Quat<R> p{R(0), v.x, v.y, v.z};
  Quat<R> r = hamilton(hamilton(q, p), qinvert(q));
  return {r.x, r.y, r.z};
}
// alias.rs:130-133
template <typename R> RT_DEV Quat<R> new_quaternion(R angle, Vec3<R> axis) {
  Vec3<R> v = normalize(axis) * rsin(angle / R(2));
  return {rcos(angle / R(2)), v.x, v.y, v.z};
}

// alias.rs:149-181
template <typename R> struct ONB {
  Vec3<R> u, v, w;
  RT_DEV static ONB from_w(Vec3<R> w_in) {
    ONB o;
    o.w = normalize(w_in);
    Vec3<R> a = rabs(o.w.x) > R(0.9) ? Vec3<R>(0, 1, 0) : Vec3<R>(1, 0, 0);
    o.v = normalize(cross(o.w, a));
    o.u = cross(o.w, o.v);
    return o;
  }
  // Same basis for a w that is already unit length (f32 kernels: hit normals, which the reference
  // normalises again); cross(w, a) written out for the two axis choices.
  RT_DEV static ONB from_unit_w(Vec3<R> w) {
    ONB o;
    o.w = w;
    const bool big = rabs(w.x) > R(0.9);
    o.v = normalize(big ? Vec3<R>(-w.z, R(0), w.x) : Vec3<R>(R(0), w.z, -w.y));
    o.u = cross(o.w, o.v);
    return o;
  }
  RT_DEV Vec3<R> local(Vec3<R> a) const {
    if constexpr (sizeof(R) == 4)
      return {mul_add3(u.x, a.x, v.x, a.y, w.x, a.z), mul_add3(u.y, a.x, v.y, a.y, w.y, a.z),
              mul_add3(u.z, a.x, v.z, a.y, w.z, a.z)};
    else
      return u * a.x + v * a.y + w * a.z;
  }
};

// hittable.rs:533-540
template <typename R> RT_DEV Vec2<R> get_sphere_uv(Vec3<R> p) {
  R theta = racos(-p.y);
  R phi = ratan2(-p.z, p.x) + Consts<R>::pi;
  return {rdiv(phi, R(2) * Consts<R>::pi), rdiv(theta, Consts<R>::pi)};
}

}  // namespace rt
