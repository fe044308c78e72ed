// Host math, draw streams, value constructors, bounding boxes, the flat BVH builder, the
// camera and the flattener.  See rth.hpp for the reference lines each piece mirrors.
#include <algorithm>
#include <cmath>
#include <limits>
#include <stdexcept>

#include "rth.hpp"

namespace rth {

static const double kPi = 3.14159265358979323846;

d3 add(d3 a, d3 b) { return d3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
d3 sub(d3 a, d3 b) { return d3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
d3 scale(double s, d3 a) { return d3(s * a[0], s * a[1], s * a[2]); }
double dot(d3 a, d3 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
d3 cross(d3 a, d3 b) {
  return d3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
double length(d3 a) { return std::sqrt(dot(a, a)); }
d3 normalize(d3 a) {
  double l = length(a);
  return d3(a[0] / l, a[1] / l, a[2] / l);
}
double to_radians(double degrees) { return degrees * (kPi / 180.0); }

Quaternion new_quaternion(double angle, Vector axis) {
  d3 n = normalize(axis);
  double s = std::sin(angle / 2.0);
  return Quaternion{{std::cos(angle / 2.0), n[0] * s, n[1] * s, n[2] * s}};
}
Quaternion invert(Quaternion q) { return Quaternion{{q.q[0], -q.q[1], -q.q[2], -q.q[3]}}; }
static Quaternion hamilton(const Quaternion& l, const Quaternion& r) {
  const double* a = l.q;
  const double* b = r.q;
  return Quaternion{{a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                     a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                     a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]}};
}
d3 rotate(d3 v, Quaternion q) {
  Quaternion p{{0.0, v[0], v[1], v[2]}};
  Quaternion r = hamilton(hamilton(q, p), invert(q));
  return d3(r.q[1], r.q[2], r.q[3]);
}

// ---- Philox4x32-10 -------------------------------------------------------------------------
static void philox(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = 0xD2511F53ull * c[0];
    uint64_t p1 = 0xCD9E8D57ull * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[0] = n0;
    c[1] = (uint32_t)p1;
    c[2] = n2;
    c[3] = (uint32_t)p0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
Stream::Stream(uint32_t k0_, uint32_t k1_, uint32_t a_, uint32_t b_, uint32_t c_)
    : k0(k0_), k1(k1_), a(a_), b(b_), c(c_), block(0), w{0, 0, 0, 0}, have(0) {}
uint32_t Stream::next_u32() {
  if (have == 0) {
    uint32_t ctr[4] = {a, b, c, block++};
    philox(ctr, k0, k1);
    for (int i = 0; i < 4; ++i) w[i] = ctr[i];
    have = 4;
  }
  return w[4 - have--];
}
double Stream::next_f64() { return (double)next_u32() * (1.0 / 4294967296.0); }
double Stream::range(double lo, double hi) { return lo + (hi - lo) * next_f64(); }
Stream scene_stream(uint64_t seed, uint32_t stream) {
  return Stream((uint32_t)seed, (uint32_t)(seed >> 32) ^ 0x5CE4E5EEu, stream, 0, 0);
}
int bvh_axis(uint64_t bvh_seed, uint32_t call) {
  uint32_t ctr[4] = {call, 0, 0, 0};
  philox(ctr, (uint32_t)bvh_seed, (uint32_t)(bvh_seed >> 32) ^ 0xB5A10000u);
  return (int)(((uint64_t)ctr[0] * 3u) >> 32);
}

// ---- textures / materials --------------------------------------------------------------
Texture Texture::new_solid(Color c) {
  Texture t;
  t.kind = Solid;
  t.color = c;
  return t;
}
Texture Texture::new_checker(Texture odd, Texture even) {
  Texture t;
  t.kind = Checker;
  t.odd = std::make_shared<const Texture>(std::move(odd));
  t.even = std::make_shared<const Texture>(std::move(even));
  return t;
}
Texture Texture::new_checker_color(Color odd, Color even) { return new_checker(new_solid(odd), new_solid(even)); }
Texture Texture::new_image_rgb8(std::shared_ptr<const std::vector<uint8_t>> texels, uint32_t w, uint32_t h) {
  if (!texels || w == 0 || h == 0 || texels->size() < (size_t)3 * w * h)
    throw std::runtime_error("Texture::new_image: no texels (scenes.rs:126 unwrap would panic)");
  Texture t;
  t.kind = Image;
  t.texels = std::move(texels);
  t.width = w;
  t.height = h;
  return t;
}
Texture Texture::new_normal() {
  Texture t;
  t.kind = Normal;
  return t;
}
Texture Texture::new_uv() {
  Texture t;
  t.kind = UV;
  return t;
}
// Extension: the gradient table of the Perlin texture, laid out as the device reads it (rt.h, RT_TEX_NOISE).
// Draws: 256 x (x, y, z in [-1, 1)), then per permutation the Fisher-Yates targets floor(next * (i + 1)) for
// i = 255 .. 1 — the same list the oracle's NoiseTable::make consumes.
std::shared_ptr<const std::vector<uint8_t>> Texture::make_noise_table(Stream& r) {
  auto bytes = std::make_shared<std::vector<uint8_t>>(RT_NOISE_TABLE_BYTES);
  double* vec = reinterpret_cast<double*>(bytes->data());
  uint8_t* perm = bytes->data() + 256 * 3 * sizeof(double);
  for (int i = 0; i < 256; ++i) {
    const double x = r.range(-1.0, 1.0), y = r.range(-1.0, 1.0), z = r.range(-1.0, 1.0);
    const double len = std::sqrt(x * x + y * y + z * z);
    vec[3 * i] = x / len;
    vec[3 * i + 1] = y / len;
    vec[3 * i + 2] = z / len;
  }
  for (int axis = 0; axis < 3; ++axis) {
    uint8_t* p = perm + 256 * axis;
    for (int i = 0; i < 256; ++i) p[i] = (uint8_t)i;
    for (int i = 255; i > 0; --i) std::swap(p[i], p[(int)(r.next_f64() * (double)(i + 1))]);
  }
  return bytes;
}
Texture Texture::new_noise(std::shared_ptr<const std::vector<uint8_t>> table, double scale) {
  Texture t;
  t.kind = Noise;
  t.noise_table = std::move(table);
  t.noise_scale = scale;
  return t;
}

Material Material::new_lambertian(Color albedo) { return new_lambertian_with_texture(Texture::new_solid(albedo)); }
Material Material::new_lambertian_with_texture(Texture t) {
  Material m;
  m.kind = Lambertian;
  m.texture = std::move(t);
  return m;
}
Material Material::new_dielectric(double ir) {
  Material m;
  m.kind = Dielectric;
  m.ir = ir;
  return m;
}
Material Material::new_metal(Color albedo, double fuzz) {
  Material m;
  m.kind = Metal;
  m.albedo = albedo;
  m.fuzz = fuzz;
  return m;
}
Material Material::new_diffuse_light(Color emit) {
  Material m;
  m.kind = DiffuseLight;
  m.texture = Texture::new_solid(emit);
  return m;
}
Material Material::new_area_light(Color emit) {
  Material m;
  m.kind = AreaLight;
  m.texture = Texture::new_solid(emit);
  return m;
}
Material Material::new_isotropic(Color albedo) {
  Material m;
  m.kind = Isotropic;
  m.texture = Texture::new_solid(albedo);
  return m;
}

// ---- hittables ---------------------------------------------------------------------------
Hittable Hittable::new_cube(Point p0, Point p1, Material m) {
  Hittable h;
  h.kind = Cube;
  h.a = p0;
  h.b = p1;
  h.material = std::move(m);
  return h;
}
Hittable Hittable::new_cube_light(Point p0, Point p1, Material m) {
  Hittable h = new_cube(p0, p1, std::move(m));
  h.area_light = true;
  return h;
}
Hittable Hittable::new_sphere(Point center, double radius, Material m) {
  Hittable h;
  h.kind = Sphere;
  h.a = center;
  h.radius = radius;
  h.material = std::move(m);
  return h;
}
Hittable Hittable::new_moving_sphere(Point c0, Point c1, double t0, double t1, double radius, Material m) {
  Hittable h;
  h.kind = MovingSphere;
  h.a = c0;
  h.b = c1;
  h.time_start = t0;
  h.time_end = t1;
  h.radius = radius;
  h.material = std::move(m);
  return h;
}
Hittable Hittable::new_constant_medium(Hittable boundary, double density, Color color) {
  Wrapper w;
  w.kind = Wrapper::Medium;
  w.data[0] = -(1.0 / density);
  w.data[1] = w.data[2] = w.data[3] = 0.0;
  w.material = Material::new_isotropic(color);
  boundary.wrappers.insert(boundary.wrappers.begin(), std::move(w));
  return boundary;
}
Hittable Hittable::new_rotate(Hittable h, double angle, Vector axis) {
  Quaternion q = new_quaternion(angle, axis);
  Wrapper w;
  w.kind = Wrapper::Rotate;
  for (int i = 0; i < 4; ++i) w.data[i] = q.q[i];
  h.wrappers.insert(h.wrappers.begin(), std::move(w));
  return h;
}
Hittable Hittable::new_translate(Hittable h, Vector offset) {
  Wrapper w;
  w.kind = Wrapper::Translate;
  w.data[0] = offset[0];
  w.data[1] = offset[1];
  w.data[2] = offset[2];
  w.data[3] = 0.0;
  h.wrappers.insert(h.wrappers.begin(), std::move(w));
  return h;
}

Hittable Hittable::new_bvh(std::vector<Hittable> items) {
  if (items.size() < 2) throw std::runtime_error("Hittable::new_bvh: an instance needs a slice of two or more primitives");
  for (const Hittable& it : items)
    if (!it.wrappers.empty() || it.kind == Empty || it.kind == Bvh)
      throw std::runtime_error("Hittable::new_bvh: the slice of an instance must hold bare primitives (sphere, moving sphere, cube)");
  Hittable h;
  h.kind = Bvh;
  h.items = std::make_shared<const std::vector<Hittable>>(std::move(items));
  return h;
}
uint64_t instance_bvh_seed(uint64_t bvh_seed, uint32_t world_index) {
  return bvh_seed + 0x9E3779B97F4A7C15ull * (uint64_t)(world_index + 1u);
}

// ---- bounding boxes ----------------------------------------------------------------------
Aabb surrounding_box(const Aabb& a, const Aabb& b) {
  Aabb r;
  for (int i = 0; i < 3; ++i) {
    r.minimum[i] = std::fmin(a.minimum[i], b.minimum[i]);
    r.maximum[i] = std::fmax(a.maximum[i], b.maximum[i]);
  }
  return r;
}

static Point moving_center(const Hittable& h, double time) {  // hittable.rs:66-72
  double f = (time - h.time_start) / (h.time_end - h.time_start);
  return add(h.a, scale(f, sub(h.b, h.a)));
}

bool bounding_box(const Hittable& h, double t0, double t1, RotateBBox mode, Aabb* out) {
  Aabb box;
  double r = std::fabs(h.radius);
  d3 rv(r, r, r);
  switch (h.kind) {
    case Hittable::Empty:
      return false;  // hittable.rs:473
    case Hittable::Cube:
      box.minimum = h.a;
      box.maximum = h.b;
      break;
    case Hittable::Sphere:
      box.minimum = sub(h.a, rv);
      box.maximum = add(h.a, rv);
      break;
    case Hittable::MovingSphere: {
      Aabb b0{sub(moving_center(h, t0), rv), add(moving_center(h, t0), rv)};
      Aabb b1{sub(moving_center(h, t1), rv), add(moving_center(h, t1), rv)};
      box = surrounding_box(b0, b1);
      break;
    }
    case Hittable::Bvh: {  // hittable.rs:457: the root's box = the union of the slice's boxes (min / max are exact: any order)
      bool first = true;
      for (const Hittable& it : *h.items) {
        Aabb b;
        if (!bounding_box(it, t0, t1, mode, &b)) return false;
        box = first ? b : surrounding_box(box, b);
        first = false;
      }
      break;
    }
  }
  // wrappers from the innermost outwards (hittable.rs:435-456)
  for (size_t k = h.wrappers.size(); k-- > 0;) {
    const Wrapper& w = h.wrappers[k];
    if (w.kind == Wrapper::Medium) continue;
    if (w.kind == Wrapper::Translate) {
      d3 off(w.data[0], w.data[1], w.data[2]);
      box.minimum = add(box.minimum, off);
      box.maximum = add(box.maximum, off);
      continue;
    }
    Quaternion q{{w.data[0], w.data[1], w.data[2], w.data[3]}};
    if (mode == RotateBBox::Reference) {  // two corners through q, no re-sorting (Q3)
      box.minimum = rotate(box.minimum, q);
      box.maximum = rotate(box.maximum, q);
    } else {  // all eight corners through q⁻¹ (where the object appears, Q4), re-sorted
      Quaternion qi = invert(q);
      const double inf = std::numeric_limits<double>::infinity();
      d3 lo(inf, inf, inf), hi(-inf, -inf, -inf);
      for (int c = 0; c < 8; ++c) {
        d3 corner((c & 1) ? box.maximum[0] : box.minimum[0], (c & 2) ? box.maximum[1] : box.minimum[1],
                  (c & 4) ? box.maximum[2] : box.minimum[2]);
        d3 rc = rotate(corner, qi);
        for (int i = 0; i < 3; ++i) {
          lo[i] = std::fmin(lo[i], rc[i]);
          hi[i] = std::fmax(hi[i], rc[i]);
        }
      }
      box.minimum = lo;
      box.maximum = hi;
    }
  }
  *out = box;
  return true;
}

// ---- flat BVH ------------------------------------------------------------------------------
namespace {
struct Item {
  uint32_t index;
  Aabb box;
};
struct Built {
  uint32_t ref;
  uint32_t height;
  Aabb box;
};
struct Builder {
  FlatBvh& out;
  uint64_t seed;

  Built leaf(const Item& it) { return Built{RT_REF_LEAF | it.index, 0, it.box}; }

  Built node(const Built& l, const Built& r, uint32_t me) {
    Aabb b = surrounding_box(l.box, r.box);
    for (int i = 0; i < 3; ++i) {
      out.box[6 * (size_t)me + i] = b.minimum[i];
      out.box[6 * (size_t)me + 3 + i] = b.maximum[i];
    }
    out.left[me] = l.ref;
    out.right[me] = r.ref;
    return Built{me, 1 + std::max(l.height, r.height), b};
  }

  uint32_t reserve() {
    uint32_t me = (uint32_t)out.left.size();
    out.left.push_back(0);
    out.right.push_back(0);
    out.box.resize(out.box.size() + 6);
    return me;
  }

  Built build(Item* items, size_t n) {
    // n >= 2 here
    int axis = bvh_axis(seed, out.calls++);
    for (size_t i = 0; i < n; ++i)
      if (std::isnan(items[i].box.minimum[axis]))
        throw std::runtime_error("new_bvh: NaN bounding box (aabb.rs:47-51 unwrap would panic)");
    std::stable_sort(items, items + n,
                     [axis](const Item& x, const Item& y) { return x.box.minimum[axis] < y.box.minimum[axis]; });
    uint32_t me = reserve();  // pre-order: this node, then the left subtree, then the right
    if (n == 2) return node(leaf(items[0]), leaf(items[1]), me);
    size_t half = n / 2;
    Built l = half >= 2 ? build(items, half) : leaf(items[0]);
    Built r = (n - half) >= 2 ? build(items + half, n - half) : leaf(items[half]);
    return node(l, r, me);
  }
};
}  // namespace

FlatBvh new_bvh(const std::vector<Hittable>& world, double t0, double t1, uint64_t bvh_seed, RotateBBox mode) {
  FlatBvh out;
  if (world.empty()) return out;  // Hittable::Empty
  if (world.size() == 1) {        // returned bare, no box ever consulted (hittable.rs:172)
    out.root = RT_REF_LEAF | 0u;
    return out;
  }
  std::vector<Item> items(world.size());
  for (size_t i = 0; i < world.size(); ++i) {
    items[i].index = (uint32_t)i;
    if (!bounding_box(world[i], t0, t1, mode, &items[i].box))
      throw std::runtime_error("new_bvh: unwrap() on a None bounding box (hittable.rs:163-165)");
  }
  out.left.reserve(world.size());
  out.right.reserve(world.size());
  out.box.reserve(6 * world.size());
  Builder b{out, bvh_seed};
  Built root = b.build(items.data(), items.size());
  out.root = root.ref;
  out.height = root.height;
  return out;
}

// ---- camera ----------------------------------------------------------------------------------
rt_camera camera_new(Point lookfrom, Point lookat, Vector vup, double vfov, double aspect_ratio, double aperture,
                     double focus_dist, double time_open, double time_close) {
  double theta = to_radians(vfov);
  double h = std::tan(theta / 2.0);
  double viewport_height = 2.0 * h;
  double viewport_width = aspect_ratio * viewport_height;
  d3 w = normalize(sub(lookfrom, lookat));
  d3 u = normalize(cross(vup, w));
  d3 v = cross(w, u);
  d3 horizontal = scale(focus_dist * viewport_width, u);
  d3 vertical = scale(focus_dist * viewport_height, v);
  d3 half_h(horizontal[0] / 2.0, horizontal[1] / 2.0, horizontal[2] / 2.0);
  d3 half_v(vertical[0] / 2.0, vertical[1] / 2.0, vertical[2] / 2.0);
  d3 ll = sub(sub(sub(lookfrom, half_h), half_v), scale(focus_dist, w));
  rt_camera c;
  for (int i = 0; i < 3; ++i) {
    c.origin[i] = lookfrom[i];
    c.lower_left[i] = ll[i];
    c.horizontal[i] = horizontal[i];
    c.vertical[i] = vertical[i];
    c.u[i] = u[i];
    c.v[i] = v[i];
  }
  c.lens_radius = aperture / 2.0;
  c.time_open = time_open;
  c.time_close = time_close;
  return c;
}

// ---- flatten -----------------------------------------------------------------------------------
namespace {
struct Flattener {
  FlatScene& fs;

  uint32_t add_texture(const Texture& t) {
    uint32_t me = (uint32_t)fs.tex_tag.size();
    fs.tex_tag.push_back(0);
    fs.tex_color.insert(fs.tex_color.end(), {0.0, 0.0, 0.0});
    fs.tex_odd.push_back(RT_REF_NONE);
    fs.tex_even.push_back(RT_REF_NONE);
    fs.tex_texel_offset.push_back(0);
    fs.tex_width.push_back(0);
    fs.tex_height.push_back(0);
    switch (t.kind) {
      case Texture::Solid:
        fs.tex_tag[me] = RT_TEX_SOLID;
        for (int i = 0; i < 3; ++i) fs.tex_color[3 * (size_t)me + i] = t.color[i];
        break;
      case Texture::Checker: {
        fs.tex_tag[me] = RT_TEX_CHECKER;
        uint32_t o = add_texture(*t.odd);  // children get larger indices than their parent
        uint32_t e = add_texture(*t.even);
        fs.tex_odd[me] = o;
        fs.tex_even[me] = e;
        break;
      }
      case Texture::Image: {
        fs.tex_tag[me] = RT_TEX_IMAGE;
        fs.tex_width[me] = t.width;
        fs.tex_height[me] = t.height;
        // one copy per distinct image
        uint64_t off = ~0ull;
        for (auto& known : images)
          if (known.first == t.texels.get()) off = known.second;
        if (off == ~0ull) {
          off = fs.texels.size();
          fs.texels.insert(fs.texels.end(), t.texels->begin(), t.texels->begin() + (size_t)3 * t.width * t.height);
          images.emplace_back(t.texels.get(), off);
        }
        fs.tex_texel_offset[me] = off;
        break;
      }
      case Texture::Normal:
        fs.tex_tag[me] = RT_TEX_NORMAL;
        break;
      case Texture::UV:
        fs.tex_tag[me] = RT_TEX_UV;
        break;
      case Texture::Noise: {  // extension: the table travels in the texel blob, 8-byte aligned
        fs.tex_tag[me] = RT_TEX_NOISE;
        fs.tex_color[3 * (size_t)me] = t.noise_scale;
        uint64_t off = ~0ull;
        for (auto& known : images)
          if (known.first == t.noise_table.get()) off = known.second;
        if (off == ~0ull) {
          fs.texels.resize((fs.texels.size() + 7) / 8 * 8, 0);
          off = fs.texels.size();
          fs.texels.insert(fs.texels.end(), t.noise_table->begin(), t.noise_table->end());
          images.emplace_back(t.noise_table.get(), off);
        }
        fs.tex_texel_offset[me] = off;
        break;
      }
    }
    return me;
  }

  uint32_t add_material(const Material& m) {
    uint32_t me = (uint32_t)fs.mat_tag.size();
    fs.mat_tag.push_back(0);
    fs.mat_texture.push_back(RT_REF_NONE);
    fs.mat_params.insert(fs.mat_params.end(), {0.0, 0.0, 0.0, 0.0});
    double* p = &fs.mat_params[4 * (size_t)me];
    switch (m.kind) {
      case Material::Lambertian:
        fs.mat_tag[me] = RT_MAT_LAMBERTIAN;
        fs.mat_texture[me] = add_texture(m.texture);
        break;
      case Material::Dielectric:
        fs.mat_tag[me] = RT_MAT_DIELECTRIC;
        p[0] = m.ir;
        break;
      case Material::Metal:
        fs.mat_tag[me] = RT_MAT_METAL;
        p[0] = m.albedo[0];
        p[1] = m.albedo[1];
        p[2] = m.albedo[2];
        p[3] = m.fuzz;
        break;
      case Material::DiffuseLight:
        fs.mat_tag[me] = RT_MAT_DIFFUSE_LIGHT;
        fs.mat_texture[me] = add_texture(m.texture);
        break;
      case Material::Isotropic:
        fs.mat_tag[me] = RT_MAT_ISOTROPIC;
        fs.mat_texture[me] = add_texture(m.texture);
        break;
      case Material::AreaLight:
        fs.mat_tag[me] = RT_MAT_AREA_LIGHT;
        fs.mat_texture[me] = add_texture(m.texture);
        break;
    }
    return me;
  }

  // the bare primitive of a Hittable value -> typed leaf ref
  uint32_t add_primitive(const Hittable& h, uint32_t object) {
    uint32_t mat = add_material(h.material);
    switch (h.kind) {
      case Hittable::Sphere: {
        uint32_t i = (uint32_t)fs.sphere_material.size();
        fs.sphere.insert(fs.sphere.end(), {h.a[0], h.a[1], h.a[2], h.radius});
        fs.sphere_material.push_back(mat);
        fs.sphere_object.push_back(object);
        return RT_MAKE_LEAF(RT_LEAF_SPHERE, i);
      }
      case Hittable::MovingSphere: {
        uint32_t i = (uint32_t)fs.msphere_material.size();
        fs.msphere.insert(fs.msphere.end(),
                          {h.a[0], h.a[1], h.a[2], h.b[0], h.b[1], h.b[2], h.radius, h.time_start, h.time_end});
        fs.msphere_material.push_back(mat);
        fs.msphere_object.push_back(object);
        return RT_MAKE_LEAF(RT_LEAF_MSPHERE, i);
      }
      case Hittable::Cube: {
        uint32_t i = (uint32_t)fs.cube_material.size();
        fs.cube.insert(fs.cube.end(), {h.a[0], h.a[1], h.a[2], h.b[0], h.b[1], h.b[2]});
        fs.cube_material.push_back(mat);
        fs.cube_object.push_back(object);
        return RT_MAKE_LEAF(RT_LEAF_CUBE, i);
      }
      case Hittable::Empty:
      case Hittable::Bvh:
        break;
    }
    throw std::runtime_error("flatten: Hittable::Empty inside a world of two or more objects");
  }

  // An instance's own tree (Hittable::new_bvh over its slice), appended to the scene's node arrays behind the nodes
  // that exist so far (children still come after their parents); returns its root's node index.  The slice's
  // primitives carry the instance's object id (HitRecord keeps what the outermost world-level object is).
  uint32_t add_subtree(const Hittable& h, uint32_t object, uint64_t bvh_seed, RotateBBox mode) {
    const std::vector<Hittable>& items = *h.items;
    FlatBvh sub = new_bvh(items, 0.0, 1.0, instance_bvh_seed(bvh_seed, object), mode);
    const uint32_t base = (uint32_t)fs.bvh.left.size();
    std::vector<uint32_t> typed(items.size(), RT_REF_NONE);
    auto fix = [&](uint32_t ref) -> uint32_t {
      if (!(ref & RT_REF_LEAF)) return base + ref;
      const uint32_t k = ref & ~RT_REF_LEAF;
      if (typed[k] == RT_REF_NONE) typed[k] = add_primitive(items[k], object);
      return typed[k];
    };
    for (size_t i = 0; i < sub.left.size(); ++i) {  // pre-order = DFS order: the slice is laid out in leaf order
      const uint32_t l = fix(sub.left[i]), r = fix(sub.right[i]);
      fs.bvh.left.push_back(l);
      fs.bvh.right.push_back(r);
    }
    fs.bvh.box.insert(fs.bvh.box.end(), sub.box.begin(), sub.box.end());
    fs.bvh.calls += sub.calls;
    sub_height = std::max(sub_height, sub.height);
    return base + sub.root;  // (>= 2 items: the root is a node)
  }
  uint32_t sub_height = 0;

  uint32_t add_object(const Hittable& h, uint32_t object, uint64_t bvh_seed = 0, RotateBBox mode = RotateBBox::Correct) {
    if (h.kind == Hittable::Bvh) {
      for (const Wrapper& w : h.wrappers)
        if (w.kind == Wrapper::Medium)
          throw std::runtime_error("flatten: a constant medium around a Bvh is not representable (its boundary must be a primitive)");
    }
    uint32_t body = h.kind == Hittable::Bvh ? add_subtree(h, object, bvh_seed, mode) : add_primitive(h, object);
    if (h.wrappers.empty()) {
      if (h.kind == Hittable::Bvh) throw std::runtime_error("flatten: a bare Bvh inside a world (wrap it, or add its items to the world)");
      return body;
    }
    uint32_t me = (uint32_t)fs.complex_body.size();
    uint32_t medium_material = RT_REF_NONE;
    uint32_t first = (uint32_t)fs.op_tag.size();
    for (const Wrapper& w : h.wrappers) {
      uint8_t tag = w.kind == Wrapper::Translate ? RT_OP_TRANSLATE : (w.kind == Wrapper::Rotate ? RT_OP_ROTATE : RT_OP_MEDIUM);
      if (tag == RT_OP_MEDIUM) {
        if (medium_material != RT_REF_NONE)
          throw std::runtime_error("flatten: a constant medium inside a constant medium is not representable");
        medium_material = add_material(w.material);
      }
      fs.op_tag.push_back(tag);
      fs.op_data.insert(fs.op_data.end(), {w.data[0], w.data[1], w.data[2], w.data[3]});
    }
    fs.complex_body.push_back(body);
    fs.complex_op_first.push_back(first);
    fs.complex_op_count.push_back((uint32_t)h.wrappers.size());
    fs.complex_material.push_back(medium_material);
    fs.complex_object.push_back(object);
    return RT_MAKE_LEAF(RT_LEAF_COMPLEX, me);
  }

  std::vector<std::pair<const void*, uint64_t>> images;
};
}  // namespace

FlatScene flatten(const SceneDef& def, uint64_t bvh_seed, RotateBBox mode) {
  FlatScene fs;
  fs.background = def.background;
  fs.n_objects = (uint32_t)def.world.size();
  fs.bvh = new_bvh(def.world, 0.0, 1.0, bvh_seed, mode);  // main.rs:125
  Flattener fl{fs, 0, {}};
  // objects are laid out in BVH leaf (DFS) order so that siblings are neighbours in memory
  std::vector<uint32_t> typed(def.world.size(), RT_REF_NONE);
  auto leaf_ref = [&](uint32_t ref) -> uint32_t {
    uint32_t wi = ref & ~RT_REF_LEAF;
    if (typed[wi] == RT_REF_NONE) typed[wi] = fl.add_object(def.world[wi], wi, bvh_seed, mode);
    return typed[wi];
  };
  if (fs.bvh.root != RT_REF_NONE && (fs.bvh.root & RT_REF_LEAF)) {
    const Hittable& only = def.world[0];
    fs.bvh.root = only.kind == Hittable::Empty ? RT_REF_NONE : leaf_ref(fs.bvh.root);
  }
  const size_t n_world_nodes = fs.bvh.left.size();  // (instances append their own trees behind these)
  for (size_t i = 0; i < n_world_nodes; ++i) {  // pre-order index order = DFS order, left before right
    if (fs.bvh.left[i] & RT_REF_LEAF) {
      const uint32_t l = leaf_ref(fs.bvh.left[i]);  // (may grow the arrays: no reference into them across the call)
      fs.bvh.left[i] = l;
    }
    if (fs.bvh.right[i] & RT_REF_LEAF) {
      const uint32_t r = leaf_ref(fs.bvh.right[i]);
      fs.bvh.right[i] = r;
    }
  }
  if (fl.sub_height) fs.bvh.height += 1 + fl.sub_height;  // an upper bound (rt_scene_create computes the exact depth)

  // main.rs:126-131
  const std::vector<Hittable>& l = def.lights;
  bool none = l.empty() || (l.size() == 1 && l[0].kind == Hittable::Empty);
  fs.lights_kind = none ? RT_LIGHTS_NONE : (l.size() == 1 ? RT_LIGHTS_SINGLE : RT_LIGHTS_LIST);
  if (!none)
    for (const Hittable& h : l) {
      bool plain_sphere = h.kind == Hittable::Sphere && h.wrappers.empty();  // hittable.rs:479, 504
      // extension: a bare Cube built by new_cube_light; any other Cube is the reference's `_` arm
      bool box_light = h.kind == Hittable::Cube && h.wrappers.empty() && h.area_light;
      fs.light_tag.push_back(plain_sphere ? RT_LIGHT_SPHERE : (box_light ? RT_LIGHT_CUBE : RT_LIGHT_OTHER));
      if (plain_sphere)
        fs.light_sphere.insert(fs.light_sphere.end(), {h.a[0], h.a[1], h.a[2], h.radius});
      else
        fs.light_sphere.insert(fs.light_sphere.end(), {0.0, 0.0, 0.0, 0.0});
      if (box_light)
        fs.light_box.insert(fs.light_box.end(), {h.a[0], h.a[1], h.a[2], h.b[0], h.b[1], h.b[2]});
      else
        fs.light_box.insert(fs.light_box.end(), {0.0, 0.0, 0.0, 0.0, 0.0, 0.0});
    }
  return fs;
}

rt_scene_desc FlatScene::desc() const {
  rt_scene_desc d{};
  d.abi_version = RT_ABI_VERSION;
  d.bvh_height = bvh.height;
  d.n_nodes = (uint32_t)bvh.left.size();
  d.root = bvh.root;
  d.node_box = bvh.box.data();
  d.node_left = bvh.left.data();
  d.node_right = bvh.right.data();
  d.n_spheres = (uint32_t)sphere_material.size();
  d.sphere = sphere.data();
  d.sphere_material = sphere_material.data();
  d.sphere_object = sphere_object.data();
  d.n_mspheres = (uint32_t)msphere_material.size();
  d.msphere = msphere.data();
  d.msphere_material = msphere_material.data();
  d.msphere_object = msphere_object.data();
  d.n_cubes = (uint32_t)cube_material.size();
  d.cube = cube.data();
  d.cube_material = cube_material.data();
  d.cube_object = cube_object.data();
  d.n_complex = (uint32_t)complex_body.size();
  d.complex_body = complex_body.data();
  d.complex_op_first = complex_op_first.data();
  d.complex_op_count = complex_op_count.data();
  d.complex_material = complex_material.data();
  d.complex_object = complex_object.data();
  d.n_ops = (uint32_t)op_tag.size();
  d.op_tag = op_tag.data();
  d.op_data = op_data.data();
  d.n_materials = (uint32_t)mat_tag.size();
  d.mat_tag = mat_tag.data();
  d.mat_texture = mat_texture.data();
  d.mat_params = mat_params.data();
  d.n_textures = (uint32_t)tex_tag.size();
  d.tex_tag = tex_tag.data();
  d.tex_color = tex_color.data();
  d.tex_odd = tex_odd.data();
  d.tex_even = tex_even.data();
  d.tex_texel_offset = tex_texel_offset.data();
  d.tex_width = tex_width.data();
  d.tex_height = tex_height.data();
  d.n_texel_bytes = texels.size();
  d.texels = texels.data();
  d.lights_kind = lights_kind;
  d.n_lights = (uint32_t)light_tag.size();
  d.light_tag = light_tag.data();
  d.light_sphere = light_sphere.data();
  d.light_box = light_box.data();
  for (int i = 0; i < 3; ++i) d.background[i] = background[i];
  return d;
}

}  // namespace rth
