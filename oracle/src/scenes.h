// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Not part of the product path.
//
// Camera (cam.rs:6-72), the six reference scenes (scenes.rs:10-192), the fixed camera /
// background / lights fold of main.rs:125-150, and the three benchmark scenes that
// BASELINE.json's configs name but the reference does not contain:
// Cornell box, BVH stress, final-scene composition — built only from reference primitives.
#pragma once
#include "hittable.h"
#include "stdrng.h"

namespace orc {

// ---- cam.rs ------------------------------------------------------------------------------
struct Camera {
  V3 origin, lower_left_corner, horizontal, vertical, u, v, w;
  double lens_radius = 0, time_open = 0, time_close = 1;

  // cam.rs:26-61
  static Camera make(V3 lookfrom, V3 lookat, V3 vup, double vfov, double aspect_ratio,
                     double aperture, double focus_dist, double time_open, double time_close) {
    double theta = vfov * (PI / 180.0);  // f64::to_radians
    double h = std::tan(theta / 2.0);
    double viewport_height = 2.0 * h;
    double viewport_width = aspect_ratio * viewport_height;
    Camera c;
    c.w = normalize(lookfrom - lookat);
    c.u = normalize(cross(vup, c.w));
    c.v = cross(c.w, c.u);
    c.origin = lookfrom;
    c.horizontal = (focus_dist * viewport_width) * c.u;
    c.vertical = (focus_dist * viewport_height) * c.v;
    c.lower_left_corner = c.origin - c.horizontal / 2.0 - c.vertical / 2.0 - focus_dist * c.w;
    c.lens_radius = aperture / 2.0;
    c.time_open = time_open;
    c.time_close = time_close;
    return c;
  }

  // cam.rs:63-71.  One deliberate difference in DRAW ORDER (not in distribution): the shutter
  // time is drawn before the lens-disk rejection loop instead of after it.  The rand stream is
  // replaced anyway (north_star), the two quantities are independent, and the fixed position of
  // the time draw lets the GPU skip the (unused) disk draws of a pinhole camera.
  Ray get_ray(double s, double t, Rng& rng) const {
    const double time = rng.range(time_open, time_close);
    V3 rd = lens_radius * random_in_unit_disk(rng);
    V3 offset = u * rd.x + v * rd.y;
    Ray r;
    r.origin = origin + offset;
    r.direction = lower_left_corner + s * horizontal + t * vertical - origin - offset;
    r.time = time;
    return r;
  }
};

// f64::to_radians multiplies by (PI / 180.0) as a single rounded constant.
inline double to_radians(double deg) { return deg * (PI / 180.0); }

struct CameraSpec {
  V3 lookfrom, lookat, vup;
  double vfov, aperture, focus_dist, time_open, time_close;
};

struct SceneSpec {
  int scene_id = 3;
  uint64_t scene_seed = 5;  // echoes StdRng::seed_from_u64(5), scenes.rs:64
  uint32_t n = 0;           // scene 7: number of random spheres
  std::shared_ptr<const std::vector<uint8_t>> texels;  // earthmap RGB8 for scenes 2, 4, 8
  uint32_t tex_w = 0, tex_h = 0;
  // scenes that build a BVH of their own inside the scene function (an instance: scene 16) need what main.rs passes
  // to the world's new_bvh (main.rs:125): the axis stream's seed and the Rotate-bbox policy
  uint64_t bvh_seed = 7;
  int rotate_bbox_override = -1;
};

// BVH split-axis stream: axis of the k-th new_bvh call with >= 2 elements (stands in for gen_range(0..3), hittable.rs:161).
constexpr uint32_t BVH_KEY_HI = 0xB5A10000u;
inline int bvh_axis(uint64_t bvh_seed, uint32_t call) {
  Philox4 r = philox4x32_10(call, 0, 0, 0, (uint32_t)bvh_seed, (uint32_t)(bvh_seed >> 32) ^ BVH_KEY_HI);
  return (int)(((uint64_t)r.v[0] * 3u) >> 32);
}

// The axis stream of a BVH built INSIDE a scene function (hittable.rs:159-206 called from scenes.rs-style code) for the
// world-level object with index `world_index`: its own keyed stream, so that the world's stream (main.rs:125) is the
// same with or without instances.  The host layer (rth_scene.cpp: instance_bvh_seed) uses the same rule.
inline uint64_t instance_bvh_seed(uint64_t bvh_seed, uint32_t world_index) {
  return bvh_seed + 0x9E3779B97F4A7C15ull * (uint64_t)(world_index + 1u);
}

struct SceneDef {
  std::vector<Hittable> world, lights;
  CameraSpec cam;
  V3 background;
  RotateBBox rotate_bbox = RotateBBox::Reference;
};

// Scene-construction draw streams (a keyed Philox stream per purpose).
constexpr uint32_t SCENE_KEY_HI = 0x5CE4E5EEu;
inline Rng scene_rng(uint64_t seed, uint32_t stream) {
  Rng r;
  r.k0 = (uint32_t)seed;
  r.k1 = (uint32_t)(seed >> 32) ^ SCENE_KEY_HI;
  r.a = stream;
  return r;
}

// main.rs:134-150 — identical for every reference scene
inline CameraSpec reference_camera() {
  return {V3(13.0, 2.0, 3.0), V3(0.0, 0.0, 0.0), V3(0.0, 1.0, 0.0), 20.0, 0.1, 10.0, 0.0, 1.0};
}
inline V3 reference_background() { return V3(0.50, 0.70, 1.00); }

// scenes.rs:10-28
inline void two_spheres(SceneDef& s) {
  Texture checker = Texture::checker_color(V3(0.2, 0.3, 0.1), V3(0.9, 0.9, 0.9));
  s.world.push_back(Hittable::new_sphere(V3(0.0, -10.0, 0.0), 10.0, Material::lambertian_tex(checker)));
  s.world.push_back(Hittable::new_sphere(V3(0.0, 10.0, 0.0), 10.0, Material::lambertian_tex(checker)));
}

// scenes.rs:30-53
inline void small_scene(SceneDef& s) {
  Material ground = Material::lambertian(V3(0.8, 0.8, 0.0));
  Material center = Material::lambertian(V3(0.1, 0.2, 0.5));
  Material left = Material::dielectric(1.5);
  Material right = Material::metal(V3(0.8, 0.6, 0.2), 1.0);
  s.world.push_back(Hittable::new_sphere(V3(0.0, -100.5, -1.0), 100.0, ground));
  s.world.push_back(Hittable::new_sphere(V3(0.0, 0.0, -1.0), 0.5, center));
  s.world.push_back(Hittable::new_sphere(V3(-1.0, 0.0, -1.0), 0.5, left));
  s.world.push_back(Hittable::new_sphere(V3(-1.0, 0.0, -1.0), -0.4, left));
  s.world.push_back(Hittable::new_sphere(V3(1.0, 0.0, -1.0), 0.5, right));
}

// scenes.rs:55-123.  The seeded StdRng becomes scene stream 0; the unseeded thread_rng
// behind Vector::random_vector() (Q18) becomes scene stream 1.
// `rng` stands for scenes.rs:64's StdRng (scene 0: a keyed Philox stream, as north_star prescribes; scene 14: StdRng
// itself, stdrng.h); `trng` for the unseeded thread_rng behind Vector::random_vector (scenes.rs:77, alias.rs:40-42).
template <typename Main> inline void random_scene_with(SceneDef& s, Main& rng, Rng& trng) {
  Texture checker = Texture::checker_color(V3(0.2, 0.3, 0.1), V3(0.9, 0.9, 0.9));
  s.world.push_back(Hittable::new_sphere(V3(0.0, -1000.0, 0.0), 1000.0, Material::lambertian_tex(checker)));
  for (int a = -11; a < 11; ++a) {
    for (int b = -11; b < 11; ++b) {
      double choose_mat = rng.next_f64();
      double cx = (double)a + 0.9 * rng.next_f64();
      double cz = (double)b + 0.9 * rng.next_f64();
      V3 center(cx, 0.2, cz);
      if (length(center - V3(4.0, 0.2, 0.0)) > 0.9) {
        Material m;
        if (choose_mat < 0.8) {
          V3 r1 = random_vector(trng);
          V3 r2 = random_vector(trng);
          m = Material::lambertian(r1 * r2);
        } else if (choose_mat < 0.95) {
          double ar = rng.range(0.5, 1.0);
          double ag = rng.range(0.5, 1.0);
          double ab = rng.range(0.5, 1.0);
          double fuzz = rng.range(0.0, 0.5);
          m = Material::metal(V3(ar, ag, ab), fuzz);
        } else {
          m = Material::dielectric(1.5);
        }
        if (choose_mat < 0.8) {
          V3 center2 = center + V3(0.0, rng.range(0.0, 0.5), 0.0);
          s.world.push_back(Hittable::new_moving_sphere(center, center2, 0.0, 1.0, 0.2, m));
        } else {
          s.world.push_back(Hittable::new_sphere(center, 0.2, m));
        }
      }
    }
  }
  s.world.push_back(Hittable::new_sphere(V3(0.0, 1.0, 0.0), 1.0, Material::dielectric(1.5)));
  s.world.push_back(Hittable::new_sphere(V3(-4.0, 1.0, 0.0), 1.0, Material::lambertian(V3(0.4, 0.2, 0.1))));
  s.world.push_back(Hittable::new_sphere(V3(4.0, 1.0, 0.0), 1.0, Material::metal(V3(0.7, 0.6, 0.5), 0.0)));
}
inline void random_scene(SceneDef& s, uint64_t seed) {
  Rng rng = scene_rng(seed, 0);
  Rng trng = scene_rng(seed, 1);
  random_scene_with(s, rng, trng);
}
// scene 14: the reference's own geometry — positions, material classes, metal parameters and the moving spheres'
// second centres from StdRng::seed_from_u64(seed) (scenes.rs:64 uses 5); the Lambertian albedos, which the reference
// takes from the unseeded thread_rng, from the Philox stream as in scene 0.  Parity unpinned (stdrng.h).
inline void random_scene_stdrng(SceneDef& s, uint64_t seed) {
  StdRng rng = StdRng::seed_from_u64(seed);
  Rng trng = scene_rng(seed, 1);
  random_scene_with(s, rng, trng);
}

inline Texture earth_texture(const SceneSpec& spec) {
  if (!spec.texels || spec.tex_w == 0 || spec.tex_h == 0)
    throw std::runtime_error("image texture: no texels supplied (scenes.rs:126 unwrap would panic)");
  return Texture::image(spec.texels, spec.tex_w, spec.tex_h);
}

// scenes.rs:125-131
inline void earth(SceneDef& s, const SceneSpec& spec) {
  s.world.push_back(Hittable::new_sphere(V3(), 2.0, Material::lambertian_tex(earth_texture(spec))));
}

// scenes.rs:133-175
inline void simple_light(SceneDef& s) {
  s.lights.push_back(Hittable::new_sphere(V3(0.0, 2.0, 3.0), 1.0, Material::diffuse_light(V3(4.0, 4.0, 4.0))));
  s.lights.push_back(Hittable::new_sphere(V3(2.0, 0.3, 1.0), 0.2, Material::dielectric(1.5)));
  s.world.push_back(Hittable::new_sphere(V3(0.0, -1000.0, 0.0), 1000.0, Material::lambertian(V3(0.8, 0.8, 0.0))));
  s.world.push_back(Hittable::new_constant_medium(
      Hittable::new_sphere(V3(0.0, 2.0, 0.0), 2.0, Material::lambertian(V3(0.8, 0.0, 0.0))), 0.91,
      V3(0.0, 0.0, 0.0)));
  s.world.push_back(Hittable::new_rotate(
      Hittable::new_cube(V3(0.0, 1.0, -1.7), V3(4.0, 2.0, -1.6), Material::lambertian_tex(Texture::normal())),
      to_radians(5.0), V3(0.0, 0.0, 1.0)));
  for (const auto& l : s.lights) s.world.push_back(l.clone());  // world.append(lights.clone())
}

// scenes.rs:177-192
inline void cube_scene(SceneDef& s, const SceneSpec& spec) {
  s.world.push_back(Hittable::new_rotate(
      Hittable::new_cube(V3(-1.0, -1.0, -1.0), V3(1.0, 1.0, 1.0), Material::lambertian_tex(earth_texture(spec))),
      -to_radians(45.0), V3(0.0, 0.0, 1.0)));  // -45.0f64.to_radians() = -(45.0.to_radians())
}

// ---- scenes without a reference counterpart ------------------------------

// C2 / C5: Cornell box from thin Cubes, a Sphere light (the only light shape the
// reference can importance-sample, Q8) and two Translate(Rotate(Cube)) boxes.
inline void cornell_box(SceneDef& s) {
  Material red = Material::lambertian(V3(0.65, 0.05, 0.05));
  Material white = Material::lambertian(V3(0.73, 0.73, 0.73));
  Material green = Material::lambertian(V3(0.12, 0.45, 0.15));
  Material light = Material::diffuse_light(V3(15.0, 15.0, 15.0));
  s.world.push_back(Hittable::new_cube(V3(555.0, 0.0, 0.0), V3(556.0, 555.0, 555.0), green));
  s.world.push_back(Hittable::new_cube(V3(-1.0, 0.0, 0.0), V3(0.0, 555.0, 555.0), red));
  s.world.push_back(Hittable::new_cube(V3(0.0, -1.0, 0.0), V3(555.0, 0.0, 555.0), white));
  s.world.push_back(Hittable::new_cube(V3(0.0, 555.0, 0.0), V3(555.0, 556.0, 555.0), white));
  s.world.push_back(Hittable::new_cube(V3(0.0, 0.0, 555.0), V3(555.0, 555.0, 556.0), white));
  s.world.push_back(Hittable::new_sphere(V3(278.0, 480.0, 278.0), 60.0, light));
  // Q4: the object appears turned by −angle, so Shirley's +15° / −18° are passed negated.
  s.world.push_back(Hittable::new_translate(
      Hittable::new_rotate(Hittable::new_cube(V3(0.0, 0.0, 0.0), V3(165.0, 330.0, 165.0), white),
                           to_radians(-15.0), V3(0.0, 1.0, 0.0)),
      V3(265.0, 0.0, 295.0)));
  s.world.push_back(Hittable::new_translate(
      Hittable::new_rotate(Hittable::new_cube(V3(0.0, 0.0, 0.0), V3(165.0, 165.0, 165.0), white),
                           to_radians(18.0), V3(0.0, 1.0, 0.0)),
      V3(130.0, 0.0, 65.0)));
  s.lights.push_back(Hittable::new_sphere(V3(278.0, 480.0, 278.0), 60.0, light));
  s.cam = {V3(278.0, 278.0, -800.0), V3(278.0, 278.0, 0.0), V3(0.0, 1.0, 0.0), 40.0, 0.0, 10.0, 0.0, 1.0};
  s.background = V3(0.0, 0.0, 0.0);
  s.rotate_bbox = RotateBBox::Correct;
}

// EXTENSION scenes (a Cube as an importance-sampled area light; no counterpart in the reference).
// 12: the canonical Cornell box — the ceiling light is a thin box, everything else as scene 6.
inline void cornell_box_area(SceneDef& s) {
  Material red = Material::lambertian(V3(0.65, 0.05, 0.05));
  Material white = Material::lambertian(V3(0.73, 0.73, 0.73));
  Material green = Material::lambertian(V3(0.12, 0.45, 0.15));
  Material light = Material::area_light(V3(15.0, 15.0, 15.0));
  s.world.push_back(Hittable::new_cube(V3(555.0, 0.0, 0.0), V3(556.0, 555.0, 555.0), green));
  s.world.push_back(Hittable::new_cube(V3(-1.0, 0.0, 0.0), V3(0.0, 555.0, 555.0), red));
  s.world.push_back(Hittable::new_cube(V3(0.0, -1.0, 0.0), V3(555.0, 0.0, 555.0), white));
  s.world.push_back(Hittable::new_cube(V3(0.0, 555.0, 0.0), V3(555.0, 556.0, 555.0), white));
  s.world.push_back(Hittable::new_cube(V3(0.0, 0.0, 555.0), V3(555.0, 555.0, 556.0), white));
  s.world.push_back(Hittable::new_cube_light(V3(213.0, 553.0, 227.0), V3(343.0, 554.0, 332.0), light));
  s.world.push_back(Hittable::new_translate(
      Hittable::new_rotate(Hittable::new_cube(V3(0.0, 0.0, 0.0), V3(165.0, 330.0, 165.0), white),
                           to_radians(-15.0), V3(0.0, 1.0, 0.0)),
      V3(265.0, 0.0, 295.0)));
  s.world.push_back(Hittable::new_translate(
      Hittable::new_rotate(Hittable::new_cube(V3(0.0, 0.0, 0.0), V3(165.0, 165.0, 165.0), white),
                           to_radians(18.0), V3(0.0, 1.0, 0.0)),
      V3(130.0, 0.0, 65.0)));
  s.lights.push_back(Hittable::new_cube_light(V3(213.0, 553.0, 227.0), V3(343.0, 554.0, 332.0), light));
  s.cam = {V3(278.0, 278.0, -800.0), V3(278.0, 278.0, 0.0), V3(0.0, 1.0, 0.0), 40.0, 0.0, 10.0, 0.0, 1.0};
  s.background = V3(0.0, 0.0, 0.0);
  s.rotate_bbox = RotateBBox::Correct;
}
// 13: a grey floor under a box light against a black sky — the closed-form check of the extension (the irradiance
// of a point from a uniformly emitting polygon is Lambert's edge sum; tests/test_area_light.py).
inline void area_light_plate(SceneDef& s) {
  s.world.push_back(Hittable::new_cube(V3(-50.0, -1.0, -50.0), V3(50.0, 0.0, 50.0), Material::lambertian(V3(0.5, 0.5, 0.5))));
  Material light = Material::area_light(V3(4.0, 3.0, 2.0));
  s.world.push_back(Hittable::new_cube_light(V3(-1.0, 3.0, -1.5), V3(1.0, 3.25, 1.5), light));
  s.lights.push_back(Hittable::new_cube_light(V3(-1.0, 3.0, -1.5), V3(1.0, 3.25, 1.5), light));
  s.cam = {V3(0.0, 2.0, -9.0), V3(0.0, 0.0, 0.0), V3(0.0, 1.0, 0.0), 40.0, 0.0, 10.0, 0.0, 1.0};
  s.background = V3(0.0, 0.0, 0.0);
  s.rotate_bbox = RotateBBox::Correct;
}

// 15: scene 13 with a second, spherical lamp and a dielectric bead in `lights` (a List of three: the box light, a
// DiffuseLight sphere, a non-emitting sphere as scenes.rs:139-143 has one) — the box light inside List::pdf_value /
// List::random (hittable.rs:492-497, 510-512).
inline void area_light_list(SceneDef& s) {
  area_light_plate(s);
  Hittable lamp = Hittable::new_sphere(V3(-3.0, 2.0, 1.0), 0.5, Material::diffuse_light(V3(6.0, 6.0, 6.0)));
  Hittable bead = Hittable::new_sphere(V3(2.5, 0.4, -1.0), 0.4, Material::dielectric(1.5));
  s.world.push_back(lamp.clone());
  s.world.push_back(bead.clone());
  s.lights.push_back(std::move(lamp));
  s.lights.push_back(std::move(bead));
}

// C4: ground + n random static spheres, material mix as random_scene; scene stream 0.
// radius_scale 1: scene 7.  Scene 17 (bvh_stress_sparse) is the same draw sequence with every radius x 0.1: at 8 M spheres
// scene 7 fills a quarter of its volume (rays end within a few units: the working set is the visible shell), scene 17
// 0.03 % of it (a ray crosses the whole volume: the working set is the scene).
inline void bvh_stress(SceneDef& s, uint64_t seed, uint32_t n, double radius_scale = 1.0) {
  Texture checker = Texture::checker_color(V3(0.2, 0.3, 0.1), V3(0.9, 0.9, 0.9));
  s.world.push_back(Hittable::new_sphere(V3(0.0, -1000.0, 0.0), 1000.0, Material::lambertian_tex(checker)));
  Rng rng = scene_rng(seed, 0);
  for (uint32_t i = 0; i < n; ++i) {
    double choose_mat = rng.next_f64();
    double cx = rng.range(-40.0, 40.0);
    double cy = rng.range(0.1, 20.0);
    double cz = rng.range(-40.0, 40.0);
    double radius = rng.range(0.05, 0.15) * radius_scale;
    Material m;
    if (choose_mat < 0.8) {
      V3 r1 = random_vector(rng);
      V3 r2 = random_vector(rng);
      m = Material::lambertian(r1 * r2);
    } else if (choose_mat < 0.95) {
      double ar = rng.range(0.5, 1.0);
      double ag = rng.range(0.5, 1.0);
      double ab = rng.range(0.5, 1.0);
      double fuzz = rng.range(0.0, 0.5);
      m = Material::metal(V3(ar, ag, ab), fuzz);
    } else {
      m = Material::dielectric(1.5);
    }
    s.world.push_back(Hittable::new_sphere(V3(cx, cy, cz), radius, m));
  }
}

// C3: "Next Week" final-scene composition restricted to what the reference can express:
// a 20×20 floor of Cubes, a Sphere light, a MovingSphere, glass, metal, two
// ConstantMedium(Sphere) volumes, an earthmap sphere, a checker sphere in place of Perlin
// noise, and 1000 small white spheres each wrapped Translate(Rotate(Sphere)).
inline void final_scene(SceneDef& s, const SceneSpec& spec, bool perlin = false, bool instanced = false);
// scene 16: the same composition with the book's structure for the 1000 small spheres — ONE
// Translate(Rotate(Bvh(spheres))) instance (hittable.rs:81-91 over hittable.rs:93-98), built by Hittable::new_bvh inside
// the scene function as "The Next Week" does with its BvhNode
inline void final_scene_instanced(SceneDef& s, const SceneSpec& spec) { final_scene(s, spec, false, true); }
inline void final_scene(SceneDef& s, const SceneSpec& spec, bool perlin, bool instanced) {
  Rng rng = scene_rng(spec.scene_seed, 0);
  Material ground = Material::lambertian(V3(0.48, 0.83, 0.53));
  const int boxes_per_side = 20;
  for (int i = 0; i < boxes_per_side; ++i) {
    for (int j = 0; j < boxes_per_side; ++j) {
      double w = 100.0;
      double x0 = -1000.0 + i * w;
      double z0 = -1000.0 + j * w;
      double y0 = 0.0;
      double x1 = x0 + w;
      double y1 = rng.range(1.0, 101.0);
      double z1 = z0 + w;
      s.world.push_back(Hittable::new_cube(V3(x0, y0, z0), V3(x1, y1, z1), ground));
    }
  }
  Material light = Material::diffuse_light(V3(7.0, 7.0, 7.0));
  s.world.push_back(Hittable::new_sphere(V3(273.0, 654.0, 279.5), 100.0, light));
  s.lights.push_back(Hittable::new_sphere(V3(273.0, 654.0, 279.5), 100.0, light));
  V3 center1(400.0, 400.0, 200.0);
  V3 center2 = center1 + V3(30.0, 0.0, 0.0);
  s.world.push_back(Hittable::new_moving_sphere(center1, center2, 0.0, 1.0, 50.0, Material::lambertian(V3(0.7, 0.3, 0.1))));
  s.world.push_back(Hittable::new_sphere(V3(260.0, 150.0, 45.0), 50.0, Material::dielectric(1.5)));
  s.world.push_back(Hittable::new_sphere(V3(0.0, 150.0, 145.0), 50.0, Material::metal(V3(0.8, 0.8, 0.9), 1.0)));
  s.world.push_back(Hittable::new_sphere(V3(360.0, 150.0, 145.0), 70.0, Material::dielectric(1.5)));
  s.world.push_back(Hittable::new_constant_medium(
      Hittable::new_sphere(V3(360.0, 150.0, 145.0), 70.0, Material::dielectric(1.5)), 0.2, V3(0.2, 0.4, 0.9)));
  s.world.push_back(Hittable::new_constant_medium(
      Hittable::new_sphere(V3(0.0, 0.0, 0.0), 5000.0, Material::dielectric(1.5)), 0.0001, V3(1.0, 1.0, 1.0)));
  s.world.push_back(Hittable::new_sphere(V3(400.0, 200.0, 400.0), 100.0, Material::lambertian_tex(earth_texture(spec))));
  // scene 8 (BASELINE config 3) puts a checker here; scene 11 the book's Perlin marble (extension, scene stream 2)
  Rng noise_rng = scene_rng(spec.scene_seed, 2);
  s.world.push_back(Hittable::new_sphere(
      V3(220.0, 280.0, 300.0), 80.0,
      Material::lambertian_tex(perlin ? Texture::noise(NoiseTable::make(noise_rng), 0.1)
                                      : Texture::checker_color(V3(0.2, 0.3, 0.1), V3(0.9, 0.9, 0.9)))));
  Material white = Material::lambertian(V3(0.73, 0.73, 0.73));
  std::vector<Hittable> cluster;
  for (int j = 0; j < 1000; ++j) {
    double px = rng.range(0.0, 165.0);
    double py = rng.range(0.0, 165.0);
    double pz = rng.range(0.0, 165.0);
    if (instanced) {
      cluster.push_back(Hittable::new_sphere(V3(px, py, pz), 10.0, white));
      continue;
    }
    s.world.push_back(Hittable::new_translate(
        Hittable::new_rotate(Hittable::new_sphere(V3(px, py, pz), 10.0, white), to_radians(-15.0),
                             V3(0.0, 1.0, 0.0)),
        V3(-100.0, 270.0, 395.0)));
  }
  if (instanced) {
    BuildCtx ctx;
    ctx.time_start = 0.0;
    ctx.time_end = 1.0;
    ctx.rotate_bbox = spec.rotate_bbox_override == 0 ? RotateBBox::Reference : RotateBBox::Correct;
    const uint64_t seed = instance_bvh_seed(spec.bvh_seed, (uint32_t)s.world.size());
    ctx.axis_of_call = [seed](uint32_t k) { return bvh_axis(seed, k); };
    s.world.push_back(Hittable::new_translate(
        Hittable::new_rotate(Hittable::new_bvh(cluster, ctx), to_radians(-15.0), V3(0.0, 1.0, 0.0)),
        V3(-100.0, 270.0, 395.0)));
  }
  s.cam = {V3(478.0, 278.0, -600.0), V3(278.0, 278.0, 0.0), V3(0.0, 1.0, 0.0), 40.0, 0.0, 10.0, 0.0, 1.0};
  s.background = V3(0.0, 0.0, 0.0);
  s.rotate_bbox = RotateBBox::Correct;
}

// Scene 9, "mixed_scene": not a reference scene and not a benchmark config — a seeded generator of feature
// COMBINATIONS for the parity tests (every primitive kind under every wrapper nesting the flattener accepts,
// every material and texture kind (the image texture when texels are given), a negative-radius sphere, one or
// several lights incl. a non-emitting one, both Rotate-bbox modes).
// The host layer has its own copy (rth_scenes.cpp) written against the same draw list:
//   per object, in order: shape class, material class, x, y, z, size, angle, axis, a1, a2, a3, fuzz.
inline void mixed_scene(SceneDef& s, const SceneSpec& spec) {
  const uint64_t seed = spec.scene_seed;
  const uint32_t n = spec.n;
  Rng rng = scene_rng(seed, 0);
  Rng noise_rng = scene_rng(seed, 2);
  std::shared_ptr<const NoiseTable> noise_table = NoiseTable::make(noise_rng);  // one table for the scene's noise textures
  // the ground's top face is NOT at y = 0: a checker (sin(10 y)) on a plane at a zero of the sine picks its
  // colour from the rounding noise of p.y, differently in every arithmetic
  s.world.push_back(Hittable::new_cube(V3(-40.0, -1.1, -40.0), V3(40.0, -0.1, 40.0),
                                       Material::lambertian_tex(Texture::checker_color(V3(0.2, 0.3, 0.1), V3(0.9, 0.9, 0.9)))));
  for (uint32_t i = 0; i < n; ++i) {
    const int shape = (int)(rng.next_f64() * 14.0);
    const int mat = (int)(rng.next_f64() * 9.0);
    const double x = rng.range(-9.0, 9.0);
    const double y = rng.range(0.4, 3.5);
    const double z = rng.range(-9.0, 9.0);
    const double sz = rng.range(0.25, 0.9);
    const double angle = to_radians(rng.range(-60.0, 60.0));
    const int axis_k = (int)(rng.next_f64() * 3.0);
    const double a1 = rng.next_f64();
    const double a2 = rng.next_f64();
    const double a3 = rng.next_f64();
    const double fuzz = rng.range(0.0, 0.5);
    const V3 axis(axis_k == 0 ? 1.0 : 0.0, axis_k == 1 ? 1.0 : 0.0, axis_k == 2 ? 1.0 : 0.0);
    const V3 axis2(axis_k == 2 ? 1.0 : 0.0, axis_k == 0 ? 1.0 : 0.0, axis_k == 1 ? 1.0 : 0.0);  // the next axis
    const V3 c(x, y, z), col(a1, a2, a3), half(sz, sz, sz);
    Material m;
    switch (mat) {
      case 0: m = Material::lambertian(col); break;
      case 1: m = Material::lambertian_tex(Texture::checker_color(col, V3(0.9, 0.9, 0.9))); break;
      case 2: m = Material::lambertian_tex(Texture::normal()); break;
      case 3: m = Material::lambertian_tex(Texture::uv()); break;
      case 4: m = Material::metal(V3(0.5 + 0.5 * a1, 0.5 + 0.5 * a2, 0.5 + 0.5 * a3), fuzz); break;
      case 5: m = Material::dielectric(1.5); break;
      case 6: m = Material::isotropic(col); break;
      case 7: m = Material::lambertian_tex(spec.texels ? earth_texture(spec) : Texture::uv()); break;
      default: m = Material::lambertian_tex(Texture::noise(noise_table, 1.0 + 3.0 * a1)); break;  // extension
    }
    const double density = 0.5 + 2.0 * a2;
    const V3 zero(0.0, 0.0, 0.0);
    switch (shape) {
      case 0: s.world.push_back(Hittable::new_sphere(c, sz, m)); break;
      case 1: s.world.push_back(Hittable::new_moving_sphere(c, c + V3(0.0, 0.5 * a1, 0.0), 0.0, 1.0, sz, m)); break;
      case 2: s.world.push_back(Hittable::new_cube(c - half, c + half, m)); break;
      case 3: s.world.push_back(Hittable::new_rotate(Hittable::new_cube(c - half, c + half, m), angle, axis)); break;
      case 4:
        s.world.push_back(Hittable::new_translate(Hittable::new_rotate(Hittable::new_cube(zero - half, half, m), angle, axis), c));
        break;
      case 5:
        s.world.push_back(Hittable::new_translate(Hittable::new_rotate(Hittable::new_sphere(zero, sz, m), angle, axis), c));
        break;
      case 6: s.world.push_back(Hittable::new_constant_medium(Hittable::new_sphere(c, 1.5 * sz, m), density, col)); break;
      case 7:
        s.world.push_back(Hittable::new_constant_medium(
            Hittable::new_translate(Hittable::new_rotate(Hittable::new_cube(zero - half, half, m), angle, axis), c), density, col));
        break;
      case 8:
        s.world.push_back(
            Hittable::new_translate(Hittable::new_constant_medium(Hittable::new_sphere(zero, 1.5 * sz, m), density, col), c));
        break;
      case 9: s.world.push_back(Hittable::new_sphere(c, -sz, m)); break;  // negative radius: normals point inward (Q15)
      case 10: s.world.push_back(Hittable::new_rotate(Hittable::new_translate(Hittable::new_cube(zero - half, half, m), c), angle * 0.1, axis)); break;
      case 11:
        s.world.push_back(Hittable::new_translate(
            Hittable::new_rotate(Hittable::new_rotate(Hittable::new_cube(zero - half, half, m), angle, axis), 0.5 * angle, axis2), c));
        break;
      case 12:
        s.world.push_back(Hittable::new_constant_medium(
            Hittable::new_moving_sphere(c, c + V3(0.3 * a3, 0.0, 0.0), 0.0, 1.0, 1.5 * sz, m), density, col));
        break;
      default:
        s.world.push_back(Hittable::new_translate(
            Hittable::new_rotate(Hittable::new_constant_medium(Hittable::new_cube(zero - half, half, m), density, col), angle, axis), c));
        break;
    }
  }
  auto lamp_a = [] { return Hittable::new_sphere(V3(-3.0, 7.0, 2.0), 1.0, Material::diffuse_light(V3(6.0, 6.0, 6.0))); };
  auto lamp_b = [] { return Hittable::new_sphere(V3(4.0, 6.0, -3.0), 0.7, Material::diffuse_light(V3(9.0, 8.0, 7.0))); };
  s.world.push_back(lamp_a());
  s.world.push_back(lamp_b());
  s.lights.push_back(lamp_a());
  if (seed % 4 != 1) s.lights.push_back(lamp_b());  // else a single light: lights.pop() of main.rs:129
  if (seed % 3 == 0) {                               // a "light" that emits nothing, as in simple_light (scenes.rs:139-143)
    s.world.push_back(Hittable::new_sphere(V3(0.0, 1.0, 5.0), 0.5, Material::dielectric(1.5)));
    s.lights.push_back(Hittable::new_sphere(V3(0.0, 1.0, 5.0), 0.5, Material::dielectric(1.5)));
  }
  s.cam = {V3(0.0, 5.0, 20.0), V3(0.0, 1.5, 0.0), V3(0.0, 1.0, 0.0), 35.0, 0.1, 20.0, 0.0, 1.0};
  s.background = V3(0.05, 0.07, 0.1);
  s.rotate_bbox = (seed & 1) ? RotateBBox::Correct : RotateBBox::Reference;
}

// Scene 10 (extension): "two_perlin_spheres" of The Next Week under the reference's fixed camera and sky.
inline void perlin_spheres(SceneDef& s, uint64_t seed) {
  Rng noise_rng = scene_rng(seed, 2);
  Texture marble = Texture::noise(NoiseTable::make(noise_rng), 4.0);
  s.world.push_back(Hittable::new_sphere(V3(0.0, -1000.0, 0.0), 1000.0, Material::lambertian_tex(marble)));
  s.world.push_back(Hittable::new_sphere(V3(0.0, 2.0, 0.0), 2.0, Material::lambertian_tex(marble)));
}

// main.rs:117-124 plus the three extra ids
inline SceneDef build_scene_def(const SceneSpec& spec) {
  SceneDef s;
  s.cam = reference_camera();
  s.background = reference_background();
  switch (spec.scene_id) {
    case 0: random_scene(s, spec.scene_seed); break;
    case 1: two_spheres(s); break;
    case 2: earth(s, spec); break;
    case 3: simple_light(s); break;
    case 4: cube_scene(s, spec); break;
    case 6: cornell_box(s); break;
    case 7: bvh_stress(s, spec.scene_seed, spec.n); break;
    case 8: final_scene(s, spec); break;
    case 9: mixed_scene(s, spec); break;
    case 10: perlin_spheres(s, spec.scene_seed); break;
    case 11: final_scene(s, spec, true); break;
    case 12: cornell_box_area(s); break;
    case 13: area_light_plate(s); break;
    case 14: random_scene_stdrng(s, spec.scene_seed); break;
    case 15: area_light_list(s); break;
    case 16: final_scene_instanced(s, spec); break;
    case 17: bvh_stress(s, spec.scene_seed, spec.n, 0.1); break;
    default: small_scene(s); break;
  }
  return s;
}

}  // namespace orc
