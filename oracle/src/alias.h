// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Not part of the product path.
//
// f64 vector math + sampling helpers, restating alias.rs:1-181 and the subset of the
// un-vendored crate `vecmat` 0.4.2 (Cargo.lock:743-744) that the reference calls:
// element-wise + - * / for vec∘vec and scalar∘vec, dot, cross, sqrlen, length,
// normalize (= v / length), map, new() = zeros.
#pragma once
#include <cmath>

#include "philox.h"

namespace orc {

constexpr double PI = 3.14159265358979323846;  // std::f64::consts::PI

struct V2 {
  double x = 0, y = 0;
};

struct V3 {
  double x = 0, y = 0, z = 0;
  V3() = default;
  V3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
  double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  double& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};

inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator/(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double sqrlen(V3 a) { return dot(a, a); }
inline double length(V3 a) { return std::sqrt(sqrlen(a)); }
inline V3 normalize(V3 a) { return a / length(a); }

// alias.rs:16-21, 96-99
inline bool near_zero(V3 a) {
  return std::fabs(a.x) < 1e-8 && std::fabs(a.y) < 1e-8 && std::fabs(a.z) < 1e-8;
}
inline bool near_zero(V2 a) { return std::fabs(a.x) < 1e-8 && std::fabs(a.y) < 1e-8; }

// alias.rs:40-42
inline V3 random_vector(Rng& rng) {
  double a = rng.next_f64();
  double b = rng.next_f64();
  double c = rng.next_f64();
  return {a, b, c};
}

// alias.rs:44-56 — rejection, 3 draws per iteration in [-1,1)
inline V3 random_in_unit_sphere(Rng& rng) {
  for (;;) {
    double a = rng.range(-1.0, 1.0);
    double b = rng.range(-1.0, 1.0);
    double c = rng.range(-1.0, 1.0);
    V3 p(a, b, c);
    if (sqrlen(p) < 1.0) return p;
  }
}

// alias.rs:62-70 — rejection, 2 draws per iteration
inline V3 random_in_unit_disk(Rng& rng) {
  for (;;) {
    double a = rng.range(-1.0, 1.0);
    double b = rng.range(-1.0, 1.0);
    V3 p(a, b, 0.0);
    if (sqrlen(p) < 1.0) return p;
  }
}

// alias.rs:72-82 — r1 first, then r2
inline V3 random_cosine_direction(Rng& rng) {
  double r1 = rng.next_f64();
  double r2 = rng.next_f64();
  double z = std::sqrt(1.0 - r2);
  double phi = 2.0 * PI * r1;
  double x = std::cos(phi) * std::sqrt(r2);
  double y = std::sin(phi) * std::sqrt(r2);
  return {x, y, z};
}

// alias.rs:84-94
inline V3 random_to_sphere(Rng& rng, double radius, double distance_squared) {
  double r1 = rng.next_f64();
  double r2 = rng.next_f64();
  double z = 1.0 + r2 * (std::sqrt(1.0 - radius * radius / distance_squared) - 1.0);
  double phi = 2.0 * PI * r1;
  double x = std::cos(phi) * std::sqrt(1.0 - z * z);
  double y = std::sin(phi) * std::sqrt(1.0 - z * z);
  return {x, y, z};
}

// alias.rs:101-103
inline V3 reflect(V3 v, V3 n) { return v - (2.0 * dot(v, n)) * n; }

// alias.rs:105-110
inline V3 refract(V3 v, V3 n, double etai_over_etat) {
  double cos_theta = std::fmin(dot(-v, n), 1.0);
  V3 r_out_perp = etai_over_etat * (v + cos_theta * n);
  V3 r_out_parallel = (-std::sqrt(std::fabs(1.0 - sqrlen(r_out_perp)))) * n;
  return r_out_perp + r_out_parallel;
}

// Quaternion = Vec4<f64> (w, x, y, z) — alias.rs:7, 129-147
struct Quat {
  double w = 1, x = 0, y = 0, z = 0;
};

inline Quat new_quaternion(double angle, V3 axis) {
  V3 v = normalize(axis) * std::sin(angle / 2.0);
  return {std::cos(angle / 2.0), v.x, v.y, v.z};
}
inline Quat invert(Quat q) { return {q.w, -q.x, -q.y, -q.z}; }
inline Quat hamiltonian_prod(Quat a, Quat b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
          a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
          a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
// alias.rs:112-117 — v ↦ q v q⁻¹
inline V3 rotate(V3 v, Quat q) {
  Quat p{0.0, v.x, v.y, v.z};
  Quat r = hamiltonian_prod(q, p);
  r = hamiltonian_prod(r, invert(q));
  return {r.x, r.y, r.z};
}

// alias.rs:149-181
struct ONB {
  V3 u, v, w;
  static ONB from_w(V3 w_in) {
    V3 w = normalize(w_in);
    V3 a = std::fabs(w.x) > 0.9 ? V3(0.0, 1.0, 0.0) : V3(1.0, 0.0, 0.0);
    V3 v = normalize(cross(w, a));
    V3 u = cross(w, v);
    return {u, v, w};
  }
  V3 local(V3 a) const { return u * a.x + v * a.y + w * a.z; }
};

// ray.rs:4-22
struct Ray {
  V3 origin, direction;
  double time = 0;
  V3 at(double t) const { return origin + t * direction; }
};

}  // namespace orc
