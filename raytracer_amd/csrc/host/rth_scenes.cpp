// Scene builders: the six scenes of scenes.rs:10-192 with the fixed camera / background of
// main.rs:134-150, and the three benchmark scenes BASELINE.json names that the reference
// lacks (Cornell box, BVH stress, final-scene composition), assembled only from reference
// primitives.
#include <stdexcept>

#include "rth.hpp"
#include "std_rng.hpp"

namespace rth {

static SceneDef with_reference_view() {
  SceneDef s;
  s.cam = CameraSpec{Point(13.0, 2.0, 3.0), Point(0.0, 0.0, 0.0), Vector(0.0, 1.0, 0.0), 20.0, 0.1, 10.0, 0.0, 1.0};
  s.background = Color(0.50, 0.70, 1.00);
  return s;
}

static Vector random_vector(Stream& r) {  // alias.rs:40-42
  double x = r.next_f64(), y = r.next_f64(), z = r.next_f64();
  return Vector(x, y, z);
}
static Color mul(Color a, Color b) { return Color(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }

SceneDef two_spheres() {
  SceneDef s = with_reference_view();
  Texture checker = Texture::new_checker_color(Color(0.2, 0.3, 0.1), Color(0.9, 0.9, 0.9));
  s.world.push_back(Hittable::new_sphere(Point(0.0, -10.0, 0.0), 10.0, Material::new_lambertian_with_texture(checker)));
  s.world.push_back(Hittable::new_sphere(Point(0.0, 10.0, 0.0), 10.0, Material::new_lambertian_with_texture(checker)));
  return s;
}

SceneDef small_scene() {
  SceneDef s = with_reference_view();
  Material glass = Material::new_dielectric(1.5);
  s.world.push_back(Hittable::new_sphere(Point(0.0, -100.5, -1.0), 100.0, Material::new_lambertian(Color(0.8, 0.8, 0.0))));
  s.world.push_back(Hittable::new_sphere(Point(0.0, 0.0, -1.0), 0.5, Material::new_lambertian(Color(0.1, 0.2, 0.5))));
  s.world.push_back(Hittable::new_sphere(Point(-1.0, 0.0, -1.0), 0.5, glass));
  s.world.push_back(Hittable::new_sphere(Point(-1.0, 0.0, -1.0), -0.4, glass));  // hollow glass, Q15
  s.world.push_back(Hittable::new_sphere(Point(1.0, 0.0, -1.0), 0.5, Material::new_metal(Color(0.8, 0.6, 0.2), 1.0)));
  return s;
}

// Draw stream 0 replaces the seeded StdRng (scenes.rs:64), stream 1 the unseeded thread_rng
// behind Vector::random_vector() (scenes.rs:77, Q18).
// `main` = scenes.rs:64's StdRng (scene 0: a keyed Philox stream; scene 14: StdRng itself, std_rng.hpp)
template <typename Main> static SceneDef random_scene_from(Main& main, Stream& albedo) {
  SceneDef s = with_reference_view();
  s.world.push_back(Hittable::new_sphere(
      Point(0.0, -1000.0, 0.0), 1000.0,
      Material::new_lambertian_with_texture(Texture::new_checker_color(Color(0.2, 0.3, 0.1), Color(0.9, 0.9, 0.9)))));
  for (int a = -11; a < 11; ++a)
    for (int b = -11; b < 11; ++b) {
      const double choose_mat = main.next_f64();
      const double x = (double)a + 0.9 * main.next_f64();
      const double z = (double)b + 0.9 * main.next_f64();
      const Point center(x, 0.2, z);
      if (!(length(sub(center, Point(4.0, 0.2, 0.0))) > 0.9)) continue;
      if (choose_mat < 0.8) {  // diffuse, moving
        Vector v1 = random_vector(albedo), v2 = random_vector(albedo);
        Material m = Material::new_lambertian(mul(v1, v2));
        Point center2 = add(center, Vector(0.0, main.range(0.0, 0.5), 0.0));
        s.world.push_back(Hittable::new_moving_sphere(center, center2, 0.0, 1.0, 0.2, m));
      } else if (choose_mat < 0.95) {  // metal
        double r = main.range(0.5, 1.0), g = main.range(0.5, 1.0), bl = main.range(0.5, 1.0);
        double fuzz = main.range(0.0, 0.5);
        s.world.push_back(Hittable::new_sphere(center, 0.2, Material::new_metal(Color(r, g, bl), fuzz)));
      } else {  // glass
        s.world.push_back(Hittable::new_sphere(center, 0.2, Material::new_dielectric(1.5)));
      }
    }
  s.world.push_back(Hittable::new_sphere(Point(0.0, 1.0, 0.0), 1.0, Material::new_dielectric(1.5)));
  s.world.push_back(Hittable::new_sphere(Point(-4.0, 1.0, 0.0), 1.0, Material::new_lambertian(Color(0.4, 0.2, 0.1))));
  s.world.push_back(Hittable::new_sphere(Point(4.0, 1.0, 0.0), 1.0, Material::new_metal(Color(0.7, 0.6, 0.5), 0.0)));
  return s;
}
SceneDef random_scene(uint64_t seed) {
  Stream main = scene_stream(seed, 0), albedo = scene_stream(seed, 1);
  return random_scene_from(main, albedo);
}
// scene 14: positions, material classes, metal parameters and second centres from StdRng::seed_from_u64(seed), as
// scenes.rs:64-98 draws them (seed 5 there); the Lambertian albedos (thread_rng in the reference) as in scene 0
SceneDef random_scene_stdrng(uint64_t seed) {
  StdRng main(seed);
  Stream albedo = scene_stream(seed, 1);
  return random_scene_from(main, albedo);
}

static Texture earthmap(const SceneOpts& o) { return Texture::new_image_rgb8(o.texels, o.tex_w, o.tex_h); }

SceneDef earth(const SceneOpts& o) {
  SceneDef s = with_reference_view();
  s.world.push_back(Hittable::new_sphere(Point(), 2.0, Material::new_lambertian_with_texture(earthmap(o))));
  return s;
}

SceneDef simple_light() {
  SceneDef s = with_reference_view();
  s.lights.push_back(Hittable::new_sphere(Point(0.0, 2.0, 3.0), 1.0, Material::new_diffuse_light(Color(4.0, 4.0, 4.0))));
  s.lights.push_back(Hittable::new_sphere(Point(2.0, 0.3, 1.0), 0.2, Material::new_dielectric(1.5)));  // Q7
  s.world.push_back(Hittable::new_sphere(Point(0.0, -1000.0, 0.0), 1000.0, Material::new_lambertian(Color(0.8, 0.8, 0.0))));
  s.world.push_back(Hittable::new_constant_medium(
      Hittable::new_sphere(Point(0.0, 2.0, 0.0), 2.0, Material::new_lambertian(Color(0.8, 0.0, 0.0))), 0.91,
      Color(0.0, 0.0, 0.0)));
  s.world.push_back(Hittable::new_rotate(Hittable::new_cube(Point(0.0, 1.0, -1.7), Point(4.0, 2.0, -1.6),
                                                            Material::new_lambertian_with_texture(Texture::new_normal())),
                                         to_radians(5.0), Vector(0.0, 0.0, 1.0)));
  s.world.insert(s.world.end(), s.lights.begin(), s.lights.end());  // lights are part of the world too
  return s;
}

SceneDef cube_scene(const SceneOpts& o) {
  SceneDef s = with_reference_view();
  s.world.push_back(Hittable::new_rotate(
      Hittable::new_cube(Point(-1.0, -1.0, -1.0), Point(1.0, 1.0, 1.0), Material::new_lambertian_with_texture(earthmap(o))),
      -to_radians(45.0), Vector(0.0, 0.0, 1.0)));
  return s;
}

SceneDef cornell_box() {
  SceneDef s;
  const Material red = Material::new_lambertian(Color(0.65, 0.05, 0.05));
  const Material white = Material::new_lambertian(Color(0.73, 0.73, 0.73));
  const Material green = Material::new_lambertian(Color(0.12, 0.45, 0.15));
  const Material lamp = Material::new_diffuse_light(Color(15.0, 15.0, 15.0));
  // walls: Cubes one unit thick, outside the 0..555 cavity
  s.world.push_back(Hittable::new_cube(Point(555.0, 0.0, 0.0), Point(556.0, 555.0, 555.0), green));
  s.world.push_back(Hittable::new_cube(Point(-1.0, 0.0, 0.0), Point(0.0, 555.0, 555.0), red));
  s.world.push_back(Hittable::new_cube(Point(0.0, -1.0, 0.0), Point(555.0, 0.0, 555.0), white));
  s.world.push_back(Hittable::new_cube(Point(0.0, 555.0, 0.0), Point(555.0, 556.0, 555.0), white));
  s.world.push_back(Hittable::new_cube(Point(0.0, 0.0, 555.0), Point(555.0, 555.0, 556.0), white));
  // the only light shape the reference can importance-sample is a Sphere (Q8)
  const Hittable bulb = Hittable::new_sphere(Point(278.0, 480.0, 278.0), 60.0, lamp);
  s.world.push_back(bulb);
  // Rotate shows an object turned by MINUS its angle (Q4): +15 / -18 degrees are passed negated
  s.world.push_back(Hittable::new_translate(
      Hittable::new_rotate(Hittable::new_cube(Point(0.0, 0.0, 0.0), Point(165.0, 330.0, 165.0), white), to_radians(-15.0),
                           Vector(0.0, 1.0, 0.0)),
      Vector(265.0, 0.0, 295.0)));
  s.world.push_back(Hittable::new_translate(
      Hittable::new_rotate(Hittable::new_cube(Point(0.0, 0.0, 0.0), Point(165.0, 165.0, 165.0), white), to_radians(18.0),
                           Vector(0.0, 1.0, 0.0)),
      Vector(130.0, 0.0, 65.0)));
  s.lights.push_back(bulb);
  s.cam = CameraSpec{Point(278.0, 278.0, -800.0), Point(278.0, 278.0, 0.0), Vector(0.0, 1.0, 0.0), 40.0, 0.0, 10.0, 0.0, 1.0};
  s.background = Color(0.0, 0.0, 0.0);
  s.rotate_bbox = RotateBBox::Correct;
  return s;
}

// EXTENSION scenes (a Cube as an importance-sampled area light; no counterpart in the reference)
SceneDef cornell_box_area() {  // 12: the canonical Cornell box, the ceiling light a thin box
  SceneDef s = cornell_box();
  const Material lamp = Material::new_area_light(Color(15.0, 15.0, 15.0));
  const Hittable panel = Hittable::new_cube_light(Point(213.0, 553.0, 227.0), Point(343.0, 554.0, 332.0), lamp);
  s.world[5] = panel;  // in place of the bulb
  s.lights.assign(1, panel);
  return s;
}
SceneDef area_light_plate() {  // 13: a grey floor under a box light, black sky: the closed-form check
  SceneDef s;
  s.world.push_back(Hittable::new_cube(Point(-50.0, -1.0, -50.0), Point(50.0, 0.0, 50.0), Material::new_lambertian(Color(0.5, 0.5, 0.5))));
  const Hittable panel =
      Hittable::new_cube_light(Point(-1.0, 3.0, -1.5), Point(1.0, 3.25, 1.5), Material::new_area_light(Color(4.0, 3.0, 2.0)));
  s.world.push_back(panel);
  s.lights.push_back(panel);
  s.cam = CameraSpec{Point(0.0, 2.0, -9.0), Point(0.0, 0.0, 0.0), Vector(0.0, 1.0, 0.0), 40.0, 0.0, 10.0, 0.0, 1.0};
  s.background = Color(0.0, 0.0, 0.0);
  s.rotate_bbox = RotateBBox::Correct;
  return s;
}

SceneDef area_light_list() {  // 15: the box light of scene 13 in a List with a sphere lamp and a non-emitting bead
  SceneDef s = area_light_plate();
  const Hittable lamp = Hittable::new_sphere(Point(-3.0, 2.0, 1.0), 0.5, Material::new_diffuse_light(Color(6.0, 6.0, 6.0)));
  const Hittable bead = Hittable::new_sphere(Point(2.5, 0.4, -1.0), 0.4, Material::new_dielectric(1.5));
  s.world.push_back(lamp);
  s.world.push_back(bead);
  s.lights.push_back(lamp);
  s.lights.push_back(bead);
  return s;
}

SceneDef bvh_stress(uint64_t seed, uint32_t n, double radius_scale) {
  SceneDef s = with_reference_view();
  s.world.reserve((size_t)n + 1);
  s.world.push_back(Hittable::new_sphere(
      Point(0.0, -1000.0, 0.0), 1000.0,
      Material::new_lambertian_with_texture(Texture::new_checker_color(Color(0.2, 0.3, 0.1), Color(0.9, 0.9, 0.9)))));
  Stream r = scene_stream(seed, 0);
  for (uint32_t i = 0; i < n; ++i) {
    const double choose_mat = r.next_f64();
    const double x = r.range(-40.0, 40.0);
    const double y = r.range(0.1, 20.0);
    const double z = r.range(-40.0, 40.0);
    const double radius = r.range(0.05, 0.15) * radius_scale;
    Material m;
    if (choose_mat < 0.8) {
      Vector v1 = random_vector(r), v2 = random_vector(r);
      m = Material::new_lambertian(mul(v1, v2));
    } else if (choose_mat < 0.95) {
      double cr = r.range(0.5, 1.0), cg = r.range(0.5, 1.0), cb = r.range(0.5, 1.0);
      double fuzz = r.range(0.0, 0.5);
      m = Material::new_metal(Color(cr, cg, cb), fuzz);
    } else {
      m = Material::new_dielectric(1.5);
    }
    s.world.push_back(Hittable::new_sphere(Point(x, y, z), radius, m));
  }
  return s;
}

SceneDef final_scene(const SceneOpts& o, bool perlin, bool instanced) {
  SceneDef s;
  Stream r = scene_stream(o.scene_seed, 0);
  const Material ground = Material::new_lambertian(Color(0.48, 0.83, 0.53));
  for (int i = 0; i < 20; ++i)
    for (int j = 0; j < 20; ++j) {
      const double w = 100.0;
      const double x0 = -1000.0 + i * w, z0 = -1000.0 + j * w;
      const double y1 = r.range(1.0, 101.0);
      s.world.push_back(Hittable::new_cube(Point(x0, 0.0, z0), Point(x0 + w, y1, z0 + w), ground));
    }
  const Hittable lamp =
      Hittable::new_sphere(Point(273.0, 654.0, 279.5), 100.0, Material::new_diffuse_light(Color(7.0, 7.0, 7.0)));
  s.world.push_back(lamp);
  s.lights.push_back(lamp);
  const Point c1(400.0, 400.0, 200.0);
  s.world.push_back(Hittable::new_moving_sphere(c1, add(c1, Vector(30.0, 0.0, 0.0)), 0.0, 1.0, 50.0,
                                                Material::new_lambertian(Color(0.7, 0.3, 0.1))));
  s.world.push_back(Hittable::new_sphere(Point(260.0, 150.0, 45.0), 50.0, Material::new_dielectric(1.5)));
  s.world.push_back(Hittable::new_sphere(Point(0.0, 150.0, 145.0), 50.0, Material::new_metal(Color(0.8, 0.8, 0.9), 1.0)));
  const Hittable glass_ball = Hittable::new_sphere(Point(360.0, 150.0, 145.0), 70.0, Material::new_dielectric(1.5));
  s.world.push_back(glass_ball);
  s.world.push_back(Hittable::new_constant_medium(glass_ball, 0.2, Color(0.2, 0.4, 0.9)));
  s.world.push_back(Hittable::new_constant_medium(
      Hittable::new_sphere(Point(0.0, 0.0, 0.0), 5000.0, Material::new_dielectric(1.5)), 0.0001, Color(1.0, 1.0, 1.0)));
  s.world.push_back(Hittable::new_sphere(Point(400.0, 200.0, 400.0), 100.0, Material::new_lambertian_with_texture(earthmap(o))));
  Stream noise_stream = scene_stream(o.scene_seed, 2);  // scene 11 only: the book's marble instead of the checker
  s.world.push_back(Hittable::new_sphere(
      Point(220.0, 280.0, 300.0), 80.0,
      Material::new_lambertian_with_texture(perlin ? Texture::new_noise(Texture::make_noise_table(noise_stream), 0.1)
                                                   : Texture::new_checker_color(Color(0.2, 0.3, 0.1), Color(0.9, 0.9, 0.9)))));
  const Material white = Material::new_lambertian(Color(0.73, 0.73, 0.73));
  std::vector<Hittable> cluster;
  for (int j = 0; j < 1000; ++j) {
    const double x = r.range(0.0, 165.0), y = r.range(0.0, 165.0), z = r.range(0.0, 165.0);
    if (instanced)
      cluster.push_back(Hittable::new_sphere(Point(x, y, z), 10.0, white));
    else
      s.world.push_back(Hittable::new_translate(
          Hittable::new_rotate(Hittable::new_sphere(Point(x, y, z), 10.0, white), to_radians(-15.0), Vector(0.0, 1.0, 0.0)),
          Vector(-100.0, 270.0, 395.0)));
  }
  if (instanced)  // the book's structure: one box of spheres, turned and moved as a whole
    s.world.push_back(Hittable::new_translate(
        Hittable::new_rotate(Hittable::new_bvh(std::move(cluster)), to_radians(-15.0), Vector(0.0, 1.0, 0.0)),
        Vector(-100.0, 270.0, 395.0)));
  s.cam = CameraSpec{Point(478.0, 278.0, -600.0), Point(278.0, 278.0, 0.0), Vector(0.0, 1.0, 0.0), 40.0, 0.0, 10.0, 0.0, 1.0};
  s.background = Color(0.0, 0.0, 0.0);
  s.rotate_bbox = RotateBBox::Correct;
  return s;
}

// Scene 9: the parity tests' generator of feature combinations (not a reference scene, not a benchmark config).
// Draw list per object: shape class, material class, x, y, z, size, angle (degrees), axis, three colour
// components, fuzz — the oracle builds the same scene from the same list with its own code.
SceneDef mixed_scene(const SceneOpts& o) {
  SceneDef s;
  const uint64_t seed = o.scene_seed;
  const uint32_t n = o.n;
  Stream r = scene_stream(seed, 0);
  Stream noise_stream = scene_stream(seed, 2);
  const auto noise_table = Texture::make_noise_table(noise_stream);  // shared by the scene's noise textures
  const Color pale(0.9, 0.9, 0.9);
  s.world.push_back(Hittable::new_cube(Point(-40.0, -1.1, -40.0), Point(40.0, -0.1, 40.0),
                                       Material::new_lambertian_with_texture(Texture::new_checker_color(Color(0.2, 0.3, 0.1), pale))));
  for (uint32_t i = 0; i < n; ++i) {
    const int shape_class = (int)(r.next_f64() * 14.0);
    const int material_class = (int)(r.next_f64() * 9.0);
    const double px = r.range(-9.0, 9.0), py = r.range(0.4, 3.5), pz = r.range(-9.0, 9.0);
    const double size = r.range(0.25, 0.9);
    const double degrees = r.range(-60.0, 60.0);
    const int axis_index = (int)(r.next_f64() * 3.0);
    const double c1 = r.next_f64(), c2 = r.next_f64(), c3 = r.next_f64();
    const double fuzz = r.range(0.0, 0.5);
    Vector axis(0.0, 0.0, 0.0);
    axis[axis_index] = 1.0;
    const Point centre(px, py, pz), origin(0.0, 0.0, 0.0);
    const Point lo(-size, -size, -size), hi(size, size, size);
    const Color tint(c1, c2, c3);
    Material surface;
    if (material_class == 0) surface = Material::new_lambertian(tint);
    else if (material_class == 1) surface = Material::new_lambertian_with_texture(Texture::new_checker_color(tint, pale));
    else if (material_class == 2) surface = Material::new_lambertian_with_texture(Texture::new_normal());
    else if (material_class == 3) surface = Material::new_lambertian_with_texture(Texture::new_uv());
    else if (material_class == 4) surface = Material::new_metal(Color(0.5 + 0.5 * c1, 0.5 + 0.5 * c2, 0.5 + 0.5 * c3), fuzz);
    else if (material_class == 5) surface = Material::new_dielectric(1.5);
    else if (material_class == 6) surface = Material::new_isotropic(tint);
    else if (material_class == 7) surface = Material::new_lambertian_with_texture(o.texels ? earthmap(o) : Texture::new_uv());
    else surface = Material::new_lambertian_with_texture(Texture::new_noise(noise_table, 1.0 + 3.0 * c1));  // extension
    const double density = 0.5 + 2.0 * c2;
    const double turn = to_radians(degrees);
    Hittable h;
    if (shape_class == 0) {
      h = Hittable::new_sphere(centre, size, surface);
    } else if (shape_class == 1) {
      h = Hittable::new_moving_sphere(centre, add(centre, Vector(0.0, 0.5 * c1, 0.0)), 0.0, 1.0, size, surface);
    } else if (shape_class == 2) {
      h = Hittable::new_cube(add(centre, lo), add(centre, hi), surface);
    } else if (shape_class == 3) {
      h = Hittable::new_rotate(Hittable::new_cube(add(centre, lo), add(centre, hi), surface), turn, axis);
    } else if (shape_class == 4) {
      h = Hittable::new_translate(Hittable::new_rotate(Hittable::new_cube(lo, hi, surface), turn, axis), centre);
    } else if (shape_class == 5) {
      h = Hittable::new_translate(Hittable::new_rotate(Hittable::new_sphere(origin, size, surface), turn, axis), centre);
    } else if (shape_class == 6) {
      h = Hittable::new_constant_medium(Hittable::new_sphere(centre, 1.5 * size, surface), density, tint);
    } else if (shape_class == 7) {
      h = Hittable::new_constant_medium(
          Hittable::new_translate(Hittable::new_rotate(Hittable::new_cube(lo, hi, surface), turn, axis), centre), density, tint);
    } else if (shape_class == 8) {
      h = Hittable::new_translate(Hittable::new_constant_medium(Hittable::new_sphere(origin, 1.5 * size, surface), density, tint),
                                  centre);
    } else if (shape_class == 9) {
      h = Hittable::new_sphere(centre, -size, surface);  // Q15: a negative radius flips the normal
    } else if (shape_class == 10) {
      h = Hittable::new_rotate(Hittable::new_translate(Hittable::new_cube(lo, hi, surface), centre), turn * 0.1, axis);
    } else if (shape_class == 11) {
      Vector next_axis(0.0, 0.0, 0.0);
      next_axis[(axis_index + 1) % 3] = 1.0;
      h = Hittable::new_translate(
          Hittable::new_rotate(Hittable::new_rotate(Hittable::new_cube(lo, hi, surface), turn, axis), 0.5 * turn, next_axis), centre);
    } else if (shape_class == 12) {
      h = Hittable::new_constant_medium(
          Hittable::new_moving_sphere(centre, add(centre, Vector(0.3 * c3, 0.0, 0.0)), 0.0, 1.0, 1.5 * size, surface), density, tint);
    } else {
      h = Hittable::new_translate(
          Hittable::new_rotate(Hittable::new_constant_medium(Hittable::new_cube(lo, hi, surface), density, tint), turn, axis), centre);
    }
    s.world.push_back(h);
  }
  const Hittable lamp_a = Hittable::new_sphere(Point(-3.0, 7.0, 2.0), 1.0, Material::new_diffuse_light(Color(6.0, 6.0, 6.0)));
  const Hittable lamp_b = Hittable::new_sphere(Point(4.0, 6.0, -3.0), 0.7, Material::new_diffuse_light(Color(9.0, 8.0, 7.0)));
  s.world.push_back(lamp_a);
  s.world.push_back(lamp_b);
  s.lights.push_back(lamp_a);
  if (seed % 4 != 1) s.lights.push_back(lamp_b);
  if (seed % 3 == 0) {  // a non-emitting entry in the lights list, as simple_light has (scenes.rs:139-143)
    const Hittable bead = Hittable::new_sphere(Point(0.0, 1.0, 5.0), 0.5, Material::new_dielectric(1.5));
    s.world.push_back(bead);
    s.lights.push_back(bead);
  }
  s.cam = CameraSpec{Point(0.0, 5.0, 20.0), Point(0.0, 1.5, 0.0), Vector(0.0, 1.0, 0.0), 35.0, 0.1, 20.0, 0.0, 1.0};
  s.background = Color(0.05, 0.07, 0.1);
  s.rotate_bbox = (seed & 1) ? RotateBBox::Correct : RotateBBox::Reference;
  return s;
}

// Scene 10 (extension): the two Perlin spheres of "The Next Week", reference camera and sky.
SceneDef perlin_spheres(uint64_t seed) {
  SceneDef s = with_reference_view();
  Stream noise_stream = scene_stream(seed, 2);
  const Texture marble = Texture::new_noise(Texture::make_noise_table(noise_stream), 4.0);
  s.world.push_back(Hittable::new_sphere(Point(0.0, -1000.0, 0.0), 1000.0, Material::new_lambertian_with_texture(marble)));
  s.world.push_back(Hittable::new_sphere(Point(0.0, 2.0, 0.0), 2.0, Material::new_lambertian_with_texture(marble)));
  return s;
}

SceneDef scene_by_id(const SceneOpts& o) {
  switch (o.scene_id) {
    case 0: return random_scene(o.scene_seed);
    case 1: return two_spheres();
    case 2: return earth(o);
    case 3: return simple_light();
    case 4: return cube_scene(o);
    case 6: return cornell_box();
    case 7: return bvh_stress(o.scene_seed, o.n);
    case 8: return final_scene(o);
    case 9: return mixed_scene(o);
    case 10: return perlin_spheres(o.scene_seed);
    case 11: return final_scene(o, true);
    case 12: return cornell_box_area();
    case 13: return area_light_plate();
    case 14: return random_scene_stdrng(o.scene_seed);
    case 15: return area_light_list();
    case 16: return final_scene(o, false, true);
    case 17: return bvh_stress(o.scene_seed, o.n, 0.1);
    default: return small_scene();  // main.rs:123
  }
}

}  // namespace rth
