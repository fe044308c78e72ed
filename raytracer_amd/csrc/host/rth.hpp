// Host side of the drop-in: the reference's scene-construction API (hittable.rs:113-157,
// material.rs:32-55, texture.rs:16-31, cam.rs:26-61, scenes.rs, main.rs:107-150) mirrored
// in C++17 over VALUE types, plus what the reference never needed: a BVH built directly
// into a flat pre-order node array and a flattener into the tagged SoA of include/rt.h.
//
// Unlike the reference (and unlike oracle/), nothing here is a pointer tree and nothing
// here can intersect a ray: rays are traced only by the HIP library.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/rt.h"

namespace rth {

struct d3 {
  double v[3];
  d3() : v{0, 0, 0} {}
  d3(double x, double y, double z) : v{x, y, z} {}
  double operator[](int i) const { return v[i]; }
  double& operator[](int i) { return v[i]; }
};
using Color = d3;   // alias.rs:4
using Point = d3;   // alias.rs:5
using Vector = d3;  // alias.rs:6
struct Quaternion { // alias.rs:7 — (w, x, y, z)
  double q[4];
};

d3 add(d3 a, d3 b);
d3 sub(d3 a, d3 b);
d3 scale(double s, d3 a);
double dot(d3 a, d3 b);
d3 cross(d3 a, d3 b);
double length(d3 a);
d3 normalize(d3 a);
Quaternion new_quaternion(double angle, Vector axis);  // alias.rs:130-133
Quaternion invert(Quaternion q);                        // alias.rs:135-137
d3 rotate(d3 v, Quaternion q);                          // alias.rs:112-117
double to_radians(double degrees);

// Keyed Philox4x32-10 draw stream (host copy; see csrc/device/philox.hpp for the layout).
struct Stream {
  uint32_t k0, k1, a, b, c, block, w[4];
  int have;
  Stream(uint32_t k0, uint32_t k1, uint32_t a, uint32_t b, uint32_t c);
  uint32_t next_u32();
  double next_f64();
  double range(double lo, double hi);
};
Stream scene_stream(uint64_t scene_seed, uint32_t stream);  // replaces StdRng::seed_from_u64 / thread_rng in scenes.rs
int bvh_axis(uint64_t bvh_seed, uint32_t call);             // replaces gen_range(0..3) at hittable.rs:161

// ---- texture.rs:7-31 ------------------------------------------------------------------
struct Stream;
struct Texture {
  enum Kind { Solid, Checker, Image, Normal, UV, Noise } kind = Solid;
  Color color;
  // EXTENSION (no counterpart in the reference): Perlin marble of "The Next Week".  The table is the byte
  // image rt.h documents for RT_TEX_NOISE: 256 unit vectors (f64 xyz) followed by three 256-entry permutations.
  std::shared_ptr<const std::vector<uint8_t>> noise_table;
  double noise_scale = 1.0;
  static std::shared_ptr<const std::vector<uint8_t>> make_noise_table(Stream& r);
  static Texture new_noise(std::shared_ptr<const std::vector<uint8_t>> table, double scale);
  std::shared_ptr<const Texture> odd, even;
  std::shared_ptr<const std::vector<uint8_t>> texels;  // RGB8 row-major, already decoded
  uint32_t width = 0, height = 0;
  static Texture new_solid(Color c);
  static Texture new_checker(Texture odd, Texture even);
  static Texture new_checker_color(Color odd, Color even);
  static Texture new_image_rgb8(std::shared_ptr<const std::vector<uint8_t>> texels, uint32_t w, uint32_t h);
  static Texture new_normal();
  static Texture new_uv();
};

// ---- material.rs:23-55 -----------------------------------------------------------------
struct Material {
  enum Kind { Lambertian, Dielectric, Metal, DiffuseLight, Isotropic, AreaLight } kind = Lambertian;  // AreaLight: extension
  Texture texture;
  double ir = 1.0;
  Color albedo;
  double fuzz = 0.0;
  static Material new_lambertian(Color albedo);
  static Material new_lambertian_with_texture(Texture t);
  static Material new_dielectric(double ir);
  static Material new_metal(Color albedo, double fuzz);
  static Material new_diffuse_light(Color emit);
  static Material new_isotropic(Color albedo);
  static Material new_area_light(Color emit);  // EXTENSION (RT_MAT_AREA_LIGHT): emits on every hit
};

// ---- hittable.rs:42-157 ----------------------------------------------------------------
// A Hittable VALUE: one primitive under a chain of wrappers (outermost first).  List / Bvh
// are not values here — a world is a std::vector<Hittable>, a BVH is a FlatBvh.
struct Wrapper {
  enum Kind { Translate, Rotate, Medium } kind;
  double data[4];     // offset | quaternion | neg_inv_density
  Material material;  // Medium only: Isotropic(color), hittable.rs:148
};
struct Hittable {
  // Bvh: Hittable::new_bvh over a slice of PRIMITIVES (hittable.rs:93-98, 159-206) as a value that Translate / Rotate may
  // wrap — an INSTANCE (hittable.rs:81-91 hold a Box<Hittable> of any variant).  Its tree is built by the flattener
  // with the axis stream instance_bvh_seed(bvh_seed, world index).
  enum Kind { Empty, Sphere, MovingSphere, Cube, Bvh } kind = Empty;
  std::shared_ptr<const std::vector<Hittable>> items;  // Bvh: the slice (bare primitives)
  Point a, b;  // Sphere: a = centre; MovingSphere: a = centre_start, b = centre_end; Cube: p0, p1
  double radius = 0, time_start = 0, time_end = 1;
  Material material;
  std::vector<Wrapper> wrappers;  // outermost first
  bool area_light = false;        // Cube, EXTENSION: listed in `lights` it is importance-sampled (RT_LIGHT_CUBE)

  static Hittable new_cube(Point p0, Point p1, Material m);
  static Hittable new_cube_light(Point p0, Point p1, Material m);  // EXTENSION: a Cube that `lights` can sample
  static Hittable new_sphere(Point center, double radius, Material m);
  static Hittable new_moving_sphere(Point c0, Point c1, double t0, double t1, double radius, Material m);
  static Hittable new_constant_medium(Hittable boundary, double density, Color color);
  static Hittable new_rotate(Hittable h, double angle, Vector axis);
  static Hittable new_translate(Hittable h, Vector offset);  // hittable.rs:81-85 has no constructor (Q5)
  static Hittable new_bvh(std::vector<Hittable> items);      // hittable.rs:159: the slice must hold >= 2 bare primitives
};
uint64_t instance_bvh_seed(uint64_t bvh_seed, uint32_t world_index);  // axis stream of an instance's own new_bvh

struct Aabb {  // aabb.rs:6-10
  Point minimum, maximum;
};
enum class RotateBBox { Reference = 0, Correct = 1 };
bool bounding_box(const Hittable& h, double t0, double t1, RotateBBox mode, Aabb* out);  // hittable.rs:417-456
Aabb surrounding_box(const Aabb& a, const Aabb& b);                                      // aabb.rs:36-45

// Hittable::new_bvh (hittable.rs:159-206) straight into a flat pre-order array.  Leaves are
// 0x80000000 | world index until the flattener rewrites them to typed leaf refs.
struct FlatBvh {
  std::vector<double> box;  // [n][6]
  std::vector<uint32_t> left, right;
  uint32_t root = RT_REF_NONE;
  uint32_t height = 0;
  uint32_t calls = 0;  // new_bvh invocations with >= 2 elements = axis draws
};
FlatBvh new_bvh(const std::vector<Hittable>& world, double t0, double t1, uint64_t bvh_seed, RotateBBox mode);

// ---- cam.rs:26-61 ------------------------------------------------------------------------
rt_camera camera_new(Point lookfrom, Point lookat, Vector vup, double vfov, double aspect_ratio, double aperture,
                     double focus_dist, double time_open, double time_close);

struct CameraSpec {
  Point lookfrom, lookat;
  Vector vup;
  double vfov, aperture, focus_dist, time_open, time_close;
};
struct SceneDef {  // what a scenes.rs function returns, plus the per-scene constants of main.rs:134-150
  std::vector<Hittable> world, lights;
  CameraSpec cam;
  Color background;
  RotateBBox rotate_bbox = RotateBBox::Reference;
};
struct SceneOpts {
  int scene_id = 3;  // main.rs:35-36 default
  uint64_t scene_seed = 5;
  uint32_t n = 0;
  std::shared_ptr<const std::vector<uint8_t>> texels;
  uint32_t tex_w = 0, tex_h = 0;
};
SceneDef two_spheres();                          // scenes.rs:10-28
SceneDef small_scene();                          // scenes.rs:30-53
SceneDef random_scene(uint64_t seed);            // scenes.rs:55-123
SceneDef earth(const SceneOpts& o);              // scenes.rs:125-131
SceneDef simple_light();                         // scenes.rs:133-175
SceneDef cube_scene(const SceneOpts& o);         // scenes.rs:177-192
SceneDef cornell_box();                          // BASELINE.json configs[1], [4]
SceneDef bvh_stress(uint64_t seed, uint32_t n, double radius_scale = 1.0);  // BASELINE.json configs[3]; scene 17
                                                                            // (bvh_stress_sparse): every radius x 0.1
SceneDef final_scene(const SceneOpts& o, bool perlin = false, bool instanced = false);  // BASELINE.json configs[2]; perlin: scene 11
                                                                 // (extension); instanced: scene 16, the 1000 spheres as ONE
                                                                 // Translate(Rotate(Bvh)) instance
SceneDef scene_by_id(const SceneOpts& o);        // main.rs:117-124 (+ ids 6, 7, 8)

// The flat scene: owns every array an rt_scene_desc points into.
struct FlatScene {
  FlatBvh bvh;
  std::vector<double> sphere, msphere, cube, op_data, mat_params, tex_color, light_sphere, light_box;
  std::vector<uint32_t> sphere_material, sphere_object, msphere_material, msphere_object, cube_material, cube_object;
  std::vector<uint32_t> complex_body, complex_op_first, complex_op_count, complex_material, complex_object;
  std::vector<uint8_t> op_tag, mat_tag, tex_tag, light_tag, texels;
  std::vector<uint32_t> mat_texture, tex_odd, tex_even, tex_width, tex_height;
  std::vector<uint64_t> tex_texel_offset;
  uint32_t lights_kind = RT_LIGHTS_NONE;
  Color background;
  uint32_t n_objects = 0;
  rt_scene_desc desc() const;
};
// world BVH (main.rs:125) + lights fold (main.rs:126-131) + flatten.  Throws std::runtime_error
// where the reference would panic (None bounding box, NaN in the sort key).
FlatScene flatten(const SceneDef& def, uint64_t bvh_seed, RotateBBox mode);

}  // namespace rth
