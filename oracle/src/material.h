// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Not part of the product path.
//
// Texture (texture.rs:7-70), Material (material.rs:7-133) and the cosine PDF pieces of
// pdf.rs:13-39, restated in C++17 / f64.
#pragma once
#include <cstdint>
#include <memory>
#include <utility>
#include <vector>

#include "alias.h"
#include "philox.h"

namespace orc {

// ---- EXTENSION (no counterpart in the reference): Perlin noise of "Ray Tracing: The
// Next Week" (gradient noise with trilinear Hermite interpolation, turbulence of depth 7), so that the
// final-scene composition can carry its marble sphere.  Table = 256 random unit vectors + three permutations.
struct NoiseTable {
  double vec[256][3];
  uint8_t perm[3][256];

  // draw order: 256 x (x, y, z in [-1,1)), then for each of the three permutations the 255 swap targets
  // of a Fisher-Yates shuffle running from the top: target = floor(next * (i + 1)), i = 255 .. 1
  static std::shared_ptr<const NoiseTable> make(Rng& rng) {
    auto t = std::make_shared<NoiseTable>();
    for (int i = 0; i < 256; ++i) {
      const double x = rng.range(-1.0, 1.0);
      const double y = rng.range(-1.0, 1.0);
      const double z = rng.range(-1.0, 1.0);
      const double len = std::sqrt(x * x + y * y + z * z);
      t->vec[i][0] = x / len;
      t->vec[i][1] = y / len;
      t->vec[i][2] = z / len;
    }
    for (int a = 0; a < 3; ++a) {
      for (int i = 0; i < 256; ++i) t->perm[a][i] = (uint8_t)i;
      for (int i = 255; i > 0; --i) {
        const int target = (int)(rng.next_f64() * (double)(i + 1));
        std::swap(t->perm[a][i], t->perm[a][target]);
      }
    }
    return t;
  }
  double noise(V3 p) const {
    const double fx = std::floor(p.x), fy = std::floor(p.y), fz = std::floor(p.z);
    const double u = p.x - fx, v = p.y - fy, w = p.z - fz;
    const int i = (int)fx, j = (int)fy, k = (int)fz;
    const double uu = u * u * (3.0 - 2.0 * u), vv = v * v * (3.0 - 2.0 * v), ww = w * w * (3.0 - 2.0 * w);
    double accum = 0.0;
    for (int di = 0; di < 2; ++di)
      for (int dj = 0; dj < 2; ++dj)
        for (int dk = 0; dk < 2; ++dk) {
          const int idx = perm[0][(i + di) & 255] ^ perm[1][(j + dj) & 255] ^ perm[2][(k + dk) & 255];
          const double wx = u - di, wy = v - dj, wz = w - dk;
          const double d = vec[idx][0] * wx + vec[idx][1] * wy + vec[idx][2] * wz;
          accum += (di ? uu : 1.0 - uu) * (dj ? vv : 1.0 - vv) * (dk ? ww : 1.0 - ww) * d;
        }
    return accum;
  }
  double turb(V3 p) const {
    double accum = 0.0, weight = 1.0;
    for (int o = 0; o < 7; ++o) {
      accum += weight * noise(p);
      weight *= 0.5;
      p = 2.0 * p;
    }
    return std::fabs(accum);
  }
};

// ---- texture.rs ------------------------------------------------------------------------
struct Texture {
  enum Kind { Solid, Checker, Image, Normal, UV, Noise } kind = Solid;
  std::shared_ptr<const NoiseTable> noise_table;  // Noise (extension)
  double noise_scale = 1.0;
  V3 color;                                   // Solid
  std::shared_ptr<const Texture> odd, even;   // Checker
  std::shared_ptr<const std::vector<uint8_t>> texels;  // Image: RGB8 row-major
  uint32_t width = 0, height = 0;

  static Texture solid(V3 c) {
    Texture t;
    t.kind = Solid;
    t.color = c;
    return t;
  }
  static Texture checker(Texture odd, Texture even) {  // texture.rs:17-19
    Texture t;
    t.kind = Checker;
    t.odd = std::make_shared<const Texture>(std::move(odd));
    t.even = std::make_shared<const Texture>(std::move(even));
    return t;
  }
  static Texture checker_color(V3 odd, V3 even) {  // texture.rs:21-23
    return checker(solid(odd), solid(even));
  }
  // texture.rs:25-31 minus the decode (image crate not available: texels are handed in
  // already decoded to RGB8; "parity unpinned" against jpeg-decoder 0.1.22).
  static Texture image(std::shared_ptr<const std::vector<uint8_t>> px, uint32_t w, uint32_t h) {
    Texture t;
    t.kind = Image;
    t.texels = std::move(px);
    t.width = w;
    t.height = h;
    return t;
  }
  static Texture normal() {
    Texture t;
    t.kind = Normal;
    return t;
  }
  static Texture uv() {
    Texture t;
    t.kind = UV;
    return t;
  }
  static Texture noise(std::shared_ptr<const NoiseTable> table, double scale) {  // extension
    Texture t;
    t.kind = Noise;
    t.noise_table = std::move(table);
    t.noise_scale = scale;
    return t;
  }

  // texture.rs:33-69
  V3 value(V2 uv, V3 p, V3 normal) const {
    switch (kind) {
      case Solid:
        return color;
      case Checker: {
        double scale = 10.0;
        double sines = std::sin(scale * p.x) * std::sin(scale * p.y) * std::sin(scale * p.z);
        if (sines < 0.0) return odd->value(uv, p, normal);
        return even->value(uv, p, normal);
      }
      case Image: {
        double u = std::fmin(std::fmax(uv.x, 0.0), 1.0);
        double v = std::fmin(std::fmax(uv.y, 0.0), 1.0);
        v = 1.0 - v;
        // Rust `as u32` saturates and maps NaN to 0.
        auto to_u32 = [](double f) -> uint32_t {
          if (!(f == f)) return 0u;
          if (f <= 0.0) return 0u;
          if (f >= 4294967295.0) return 4294967295u;
          return (uint32_t)f;
        };
        uint32_t x = to_u32(u * (double)width);
        uint32_t y = to_u32(v * (double)height);
        if (x > width - 1) x = width - 1;
        if (y > height - 1) y = height - 1;
        double color_scale = 1.0 / 255.0;
        const uint8_t* px = texels->data() + 3 * ((size_t)y * width + x);
        return {px[0] * color_scale, px[1] * color_scale, px[2] * color_scale};
      }
      case Normal:
        return {std::fabs(normal.x), std::fabs(normal.y), std::fabs(normal.z)};
      case UV:
        return {uv.x, uv.y, 0.0};
      case Noise: {  // marble: 0.5 (1 + sin(scale z + 10 turb(p)))
        const double m = 0.5 * (1.0 + std::sin(noise_scale * p.z + 10.0 * noise_table->turb(p)));
        return {m, m, m};
      }
    }
    return {};
  }
};

// ---- material.rs -----------------------------------------------------------------------
struct Material;

struct HitRecord {  // hittable.rs:8-40
  V3 p, normal;
  double t = 0;
  V2 uv;
  bool front_face = false;
  const Material* material = nullptr;
  uint32_t object_id = 0xFFFFFFFFu;  // oracle bookkeeping: index of the world-level object

  static HitRecord make(const Ray& r, V3 p, double t, V2 uv, V3 outward_normal,
                        const Material* m) {
    HitRecord h;
    h.front_face = dot(r.direction, outward_normal) < 0.0;
    h.p = p;
    h.t = t;
    h.uv = uv;
    h.normal = h.front_face ? outward_normal : -outward_normal;
    h.material = m;
    return h;
  }
};

struct ScatterRecord {
  enum Kind { None, Specular, Pdf } kind = None;
  Ray ray;          // Specular
  V3 attenuation;   // both
  ONB cosine;       // Pdf: PDF::Cosine(ONB)
};

// material.rs:128-133
inline double reflectance(double cosine, double ref_idx) {
  double r0 = (1.0 - ref_idx) / (1.0 + ref_idx);
  r0 = r0 * r0;
  double m = 1.0 - cosine;
  double m2 = m * m;  // powi(5) by repeated squaring, as llvm.powi expands it
  double m4 = m2 * m2;
  return r0 + (1.0 - r0) * (m4 * m);
}

struct Material {
  // AreaLight: EXTENSION, no counterpart in the reference.  DiffuseLight emits only on front faces
  // (material.rs:120) and a Cube's recorded normal points inward (hittable.rs:234-245, Q1), so a cube with a
  // DiffuseLight is black from outside; AreaLight emits on every hit.
  enum Kind { Lambertian, Dielectric, Metal, DiffuseLight, Isotropic, AreaLight } kind = Lambertian;
  Texture texture;  // Lambertian / DiffuseLight / Isotropic
  double ir = 1.0;  // Dielectric
  V3 albedo;        // Metal
  double fuzz = 0;  // Metal

  static Material lambertian(V3 albedo) { return lambertian_tex(Texture::solid(albedo)); }
  static Material lambertian_tex(Texture t) {
    Material m;
    m.kind = Lambertian;
    m.texture = std::move(t);
    return m;
  }
  static Material dielectric(double ir) {
    Material m;
    m.kind = Dielectric;
    m.ir = ir;
    return m;
  }
  static Material metal(V3 albedo, double fuzz) {
    Material m;
    m.kind = Metal;
    m.albedo = albedo;
    m.fuzz = fuzz;
    return m;
  }
  static Material diffuse_light(V3 emit) {
    Material m;
    m.kind = DiffuseLight;
    m.texture = Texture::solid(emit);
    return m;
  }
  static Material area_light(V3 emit) {
    Material m;
    m.kind = AreaLight;
    m.texture = Texture::solid(emit);
    return m;
  }
  static Material isotropic(V3 albedo) {
    Material m;
    m.kind = Isotropic;
    m.texture = Texture::solid(albedo);
    return m;
  }

  // material.rs:57-102
  ScatterRecord scatter(const Ray& r, const HitRecord& hit, Rng& rng) const {
    ScatterRecord s;
    switch (kind) {
      case Lambertian:
      case Isotropic: {  // :59-65 and :97-100 are the same computation
        s.kind = ScatterRecord::Pdf;
        s.attenuation = texture.value(hit.uv, hit.p, hit.normal);
        s.cosine = ONB::from_w(hit.normal);
        return s;
      }
      case Dielectric: {
        double refraction_ratio = hit.front_face ? 1.0 / ir : ir;
        V3 unit_direction = normalize(r.direction);
        double cos_theta = std::fmin(dot(-unit_direction, hit.normal), 1.0);
        double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
        bool cannot_refract = refraction_ratio * sin_theta > 1.0;
        V3 direction;
        // `||` short-circuits: the draw happens only when refraction is possible.
        if (cannot_refract || reflectance(cos_theta, refraction_ratio) > rng.next_f64())
          direction = reflect(unit_direction, hit.normal);
        else
          direction = refract(unit_direction, hit.normal, refraction_ratio);
        s.kind = ScatterRecord::Specular;
        s.ray = Ray{hit.p, direction, r.time};
        s.attenuation = V3(1.0, 1.0, 1.0);
        return s;
      }
      case Metal: {
        V3 reflected = reflect(normalize(r.direction), hit.normal);
        V3 direction = reflected + fuzz * random_in_unit_sphere(rng);
        s.kind = ScatterRecord::Specular;
        s.ray = Ray{hit.p, direction, r.time};
        s.attenuation = albedo;
        return s;
      }
      case DiffuseLight:
      case AreaLight:
        s.kind = ScatterRecord::None;
        return s;
    }
    return s;
  }

  // material.rs:104-116
  double scattering_pdf(const Ray&, const HitRecord& hit, const Ray& scattered) const {
    if (kind == Lambertian) {
      double cosine = dot(hit.normal, normalize(scattered.direction));
      return cosine <= 0.0 ? 0.0 : cosine / PI;
    }
    return 1.0;
  }

  // material.rs:118-125
  V3 emit(const HitRecord& hit) const {
    if (kind == DiffuseLight && hit.front_face) return texture.value(hit.uv, hit.p, hit.normal);
    if (kind == AreaLight) return texture.value(hit.uv, hit.p, hit.normal);  // extension: two-sided
    return V3();
  }
};

}  // namespace orc
