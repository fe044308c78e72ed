// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Not part of the product path.
//
// AxisAlignedBoundingBox (aabb.rs:6-52) and the Hittable enum with new_bvh / hit /
// bounding_box / pdf_value / random (hittable.rs:42-540), restated in C++17 / f64 as a
// pointer tree with recursive dispatch — the same shape as the reference, on purpose: the
// product flattens and iterates, the oracle does not.
#pragma once
#include <algorithm>
#include <functional>
#include <limits>
#include <memory>
#include <stdexcept>
#include <vector>

#include "material.h"

namespace orc {

constexpr double INF = std::numeric_limits<double>::infinity();
constexpr uint32_t NO_OBJECT = 0xFFFFFFFFu;

// Work counters (the reference counts nothing; SURVEY.md §6 defines these).
struct Counters {
  uint64_t rays = 0;           // ray_color invocations with depth > 0 (main.rs:77)
  uint64_t light_probes = 0;   // Sphere::hit inside pdf_value (hittable.rs:480-482)
  uint64_t medium_probes = 0;  // boundary hits inside ConstantMedium (hittable.rs:324-325)
  uint64_t nodes_visited = 0;  // Bvh arm box tests (hittable.rs:401)
  uint64_t sphere_tests = 0, sphere_hits = 0;  // Sphere + MovingSphere arms
  uint64_t cube_tests = 0, cube_hits = 0;
  uint64_t rotate_visits = 0, translate_visits = 0, medium_visits = 0;
  uint64_t closest_hits = 0;   // world.hit() calls that returned Some at main.rs:77
  uint64_t diffuse_vertices = 0, specular_vertices = 0, emitter_hits = 0, misses = 0;
  uint64_t nonfinite_samples = 0;
  void add(const Counters& o) {
    rays += o.rays;
    light_probes += o.light_probes;
    medium_probes += o.medium_probes;
    nodes_visited += o.nodes_visited;
    sphere_tests += o.sphere_tests;
    sphere_hits += o.sphere_hits;
    cube_tests += o.cube_tests;
    cube_hits += o.cube_hits;
    rotate_visits += o.rotate_visits;
    translate_visits += o.translate_visits;
    medium_visits += o.medium_visits;
    closest_hits += o.closest_hits;
    diffuse_vertices += o.diffuse_vertices;
    specular_vertices += o.specular_vertices;
    emitter_hits += o.emitter_hits;
    misses += o.misses;
    nonfinite_samples += o.nonfinite_samples;
  }
};
inline Counters*& tls_counters() {
  static thread_local Counters dummy;
  static thread_local Counters* cur = &dummy;
  return cur;
}
#ifdef ORC_NO_COUNT  // liboracle_nocount.so: the timed CPU baseline does not pay for the checker's counters
#define ORC_COUNT(field) ((void)0)
#else
#define ORC_COUNT(field) (++tls_counters()->field)
#endif

// ---- aabb.rs ---------------------------------------------------------------------------
struct AABB {
  V3 minimum, maximum;

  // aabb.rs:17-34
  bool hit(const Ray& r, double t_min, double t_max) const {
    for (int i = 0; i < 3; ++i) {
      double inv_d = 1.0 / r.direction[i];
      double t0 = (minimum[i] - r.origin[i]) * inv_d;
      double t1 = (maximum[i] - r.origin[i]) * inv_d;
      if (inv_d < 0.0) std::swap(t0, t1);
      t_min = std::fmax(t0, t_min);  // f64::max ignores a NaN operand, as fmax does
      t_max = std::fmin(t1, t_max);
      if (t_max <= t_min) return false;
    }
    return true;
  }
  // aabb.rs:36-45
  AABB surrounding_box(const AABB& o) const {
    AABB r;
    for (int i = 0; i < 3; ++i) {
      r.minimum[i] = std::fmin(minimum[i], o.minimum[i]);
      r.maximum[i] = std::fmax(maximum[i], o.maximum[i]);
    }
    return r;
  }
};

// hittable.rs:533-540
inline V2 get_sphere_uv(V3 p) {
  double theta = std::acos(-p.y);
  double phi = std::atan2(-p.z, p.x) + PI;
  return {phi / (2.0 * PI), theta / PI};
}

// Rotate::bounding_box policy.
enum class RotateBBox { Reference, Correct };

struct BuildCtx {
  double time_start = 0.0, time_end = 1.0;
  RotateBBox rotate_bbox = RotateBBox::Reference;
  // axis for the k-th new_bvh call that has >= 2 elements (pre-order, left before right);
  // stands in for thread_rng().gen_range(0..3) at hittable.rs:161.
  std::function<int(uint32_t)> axis_of_call;
  uint32_t calls = 0;
};

struct Hittable {
  enum Kind { Cube, Sphere, MovingSphere, ConstantMedium, Translate, Rotate, List, Bvh, Empty };
  Kind kind = Empty;
  uint32_t object_id = NO_OBJECT;
  uint32_t owner_id = NO_OBJECT;  // world-level object this node belongs to (keys the medium draw)

  V3 p0, p1;                         // Cube
  bool area_light = false;           // Cube, EXTENSION: importance-sampled when listed in `lights` (see pdf_value)
  V3 center;                         // Sphere
  V3 center_start, center_end;       // MovingSphere
  double time_start = 0, time_end = 1;
  double radius = 0;                 // Sphere / MovingSphere
  Material material;                 // Cube / Sphere / MovingSphere / ConstantMedium
  std::unique_ptr<Hittable> child;   // ConstantMedium.boundary / Translate / Rotate
  double neg_inv_density = 0;        // ConstantMedium
  V3 offset;                         // Translate
  Quat rotation;                     // Rotate
  std::vector<Hittable> list;        // List
  std::unique_ptr<Hittable> left, right;  // Bvh
  AABB box;                          // Bvh

  Hittable() = default;
  Hittable(Hittable&&) = default;
  Hittable& operator=(Hittable&&) = default;

  Hittable clone() const {
    Hittable h;
    h.kind = kind;
    h.object_id = object_id;
    h.owner_id = owner_id;
    h.p0 = p0;
    h.p1 = p1;
    h.area_light = area_light;
    h.center = center;
    h.center_start = center_start;
    h.center_end = center_end;
    h.time_start = time_start;
    h.time_end = time_end;
    h.radius = radius;
    h.material = material;
    if (child) h.child = std::make_unique<Hittable>(child->clone());
    h.neg_inv_density = neg_inv_density;
    h.offset = offset;
    h.rotation = rotation;
    for (const auto& e : list) h.list.push_back(e.clone());
    if (left) h.left = std::make_unique<Hittable>(left->clone());
    if (right) h.right = std::make_unique<Hittable>(right->clone());
    h.box = box;
    return h;
  }

  // ---- constructors: hittable.rs:114-157 ------------------------------------------------
  static Hittable new_cube(V3 p0, V3 p1, Material m) {
    Hittable h;
    h.kind = Cube;
    h.p0 = p0;
    h.p1 = p1;
    h.material = std::move(m);
    return h;
  }
  // EXTENSION (no counterpart in the reference, whose pdf_value / random know only Sphere and List,
  // hittable.rs:477-515): a Cube that `lights` can importance-sample.  Same geometry and hit() as new_cube.
  static Hittable new_cube_light(V3 p0, V3 p1, Material m) {
    Hittable h = new_cube(p0, p1, std::move(m));
    h.area_light = true;
    return h;
  }
  static Hittable new_sphere(V3 c, double r, Material m) {
    Hittable h;
    h.kind = Sphere;
    h.center = c;
    h.radius = r;
    h.material = std::move(m);
    return h;
  }
  static Hittable new_moving_sphere(V3 c0, V3 c1, double t0, double t1, double r, Material m) {
    Hittable h;
    h.kind = MovingSphere;
    h.center_start = c0;
    h.center_end = c1;
    h.time_start = t0;
    h.time_end = t1;
    h.radius = r;
    h.material = std::move(m);
    return h;
  }
  static Hittable new_constant_medium(Hittable boundary, double density, V3 color) {
    Hittable h;
    h.kind = ConstantMedium;
    h.child = std::make_unique<Hittable>(std::move(boundary));
    h.neg_inv_density = -(1.0 / density);
    h.material = Material::isotropic(color);
    return h;
  }
  static Hittable new_rotate(Hittable inner, double angle, V3 axis) {
    Hittable h;
    h.kind = Rotate;
    h.child = std::make_unique<Hittable>(std::move(inner));
    h.rotation = new_quaternion(angle, axis);
    return h;
  }
  // The reference has the Translate variant (hittable.rs:81-85, 364-373, 436-444) but no
  // constructor for it; this one fills exactly those fields.
  static Hittable new_translate(Hittable inner, V3 offset) {
    Hittable h;
    h.kind = Translate;
    h.child = std::make_unique<Hittable>(std::move(inner));
    h.offset = offset;
    return h;
  }
  static Hittable new_list(std::vector<Hittable> v) {
    Hittable h;
    h.kind = List;
    h.list = std::move(v);
    return h;
  }

  void set_owner(uint32_t id) {
    owner_id = id;
    if (child) child->set_owner(id);
    for (auto& e : list) e.set_owner(id);
    if (left) left->set_owner(id);
    if (right) right->set_owner(id);
  }

  // hittable.rs:66-72
  V3 moving_center(double time) const {
    return center_start +
           ((time - time_start) / (time_end - time_start)) * (center_end - center_start);
  }

  // ---- hittable.rs:417-475 --------------------------------------------------------------
  bool bounding_box(const BuildCtx& ctx, AABB& out) const {
    double t0 = ctx.time_start, t1 = ctx.time_end;
    switch (kind) {
      case Cube:
        out = AABB{p0, p1};
        return true;
      case Sphere: {
        V3 r(std::fabs(radius), std::fabs(radius), std::fabs(radius));
        out = AABB{center - r, center + r};
        return true;
      }
      case MovingSphere: {
        V3 r(std::fabs(radius), std::fabs(radius), std::fabs(radius));
        AABB b0{moving_center(t0) - r, moving_center(t0) + r};
        AABB b1{moving_center(t1) - r, moving_center(t1) + r};
        out = b0.surrounding_box(b1);
        return true;
      }
      case ConstantMedium:
        return child->bounding_box(ctx, out);
      case Translate: {
        AABB b;
        if (!child->bounding_box(ctx, b)) return false;
        out = AABB{b.minimum + offset, b.maximum + offset};
        return true;
      }
      case Rotate: {
        AABB b;
        if (!child->bounding_box(ctx, b)) return false;
        if (ctx.rotate_bbox == RotateBBox::Reference) {
          // hittable.rs:445-456, literally: the two corners through q, no re-sorting.
          out = AABB{rotate(b.minimum, rotation), rotate(b.maximum, rotation)};
        } else {
          // Scene-definition policy for the scenes that have no reference counterpart
          // (SURVEY.md §7.4-3): all 8 corners through q⁻¹ (where the object actually
          // appears, Q4), then min/max.
          Quat qi = invert(rotation);
          V3 lo(INF, INF, INF), hi(-INF, -INF, -INF);
          for (int k = 0; k < 8; ++k) {
            V3 c((k & 1) ? b.maximum.x : b.minimum.x, (k & 2) ? b.maximum.y : b.minimum.y,
                 (k & 4) ? b.maximum.z : b.minimum.z);
            V3 rc = rotate(c, qi);
            for (int i = 0; i < 3; ++i) {
              lo[i] = std::fmin(lo[i], rc[i]);
              hi[i] = std::fmax(hi[i], rc[i]);
            }
          }
          out = AABB{lo, hi};
        }
        return true;
      }
      case List: {
        bool have = false;
        AABB acc;
        for (const auto& item : list) {
          AABB b;
          if (!item.bounding_box(ctx, b)) return false;
          acc = have ? acc.surrounding_box(b) : b;
          have = true;
        }
        if (have) out = acc;
        return have;
      }
      case Bvh:
        out = box;
        return true;
      case Empty:
        return false;
    }
    return false;
  }

  // ---- hittable.rs:159-206 --------------------------------------------------------------
  struct Item {
    Hittable* h;
    AABB box;
  };
  static Hittable new_bvh_from_left_right(Hittable l, Hittable r, const BuildCtx& ctx) {
    AABB bl, br;
    if (!l.bounding_box(ctx, bl) || !r.bounding_box(ctx, br))
      throw std::runtime_error("new_bvh: unwrap() on a None bounding box (hittable.rs:197-200)");
    Hittable h;
    h.kind = Bvh;
    h.box = bl.surrounding_box(br);
    h.left = std::make_unique<Hittable>(std::move(l));
    h.right = std::make_unique<Hittable>(std::move(r));
    return h;
  }
  static Hittable new_bvh_items(Item* items, size_t n, BuildCtx& ctx) {
    if (n >= 2) {
      int axis = ctx.axis_of_call(ctx.calls++);
      // slice::sort_by is a stable merge sort; partial_cmp().unwrap() panics on NaN.
      for (size_t i = 0; i < n; ++i)
        if (!(items[i].box.minimum[axis] == items[i].box.minimum[axis]))
          throw std::runtime_error("new_bvh: NaN bounding box (aabb.rs:47-51 unwrap)");
      std::stable_sort(items, items + n, [axis](const Item& a, const Item& b) {
        return a.box.minimum[axis] < b.box.minimum[axis];
      });
    }
    if (n == 0) return Hittable();
    if (n == 1) return std::move(*items[0].h);
    if (n == 2)
      return new_bvh_from_left_right(std::move(*items[0].h), std::move(*items[1].h), ctx);
    size_t half = n / 2;
    Hittable l = new_bvh_items(items, half, ctx);
    Hittable r = new_bvh_items(items + half, n - half, ctx);
    return new_bvh_from_left_right(std::move(l), std::move(r), ctx);
  }
  static Hittable new_bvh(std::vector<Hittable>& hittables, BuildCtx& ctx) {
    // The reference recomputes bounding_box inside the comparator (hittable.rs:162-166);
    // the boxes of slice elements never change during the build, so computing them once
    // gives the same order.
    std::vector<Item> items(hittables.size());
    for (size_t i = 0; i < hittables.size(); ++i) {
      items[i].h = &hittables[i];
      if (hittables.size() >= 2 && !hittables[i].bounding_box(ctx, items[i].box))
        throw std::runtime_error("new_bvh: unwrap() on a None bounding box (hittable.rs:163-165)");
    }
    return new_bvh_items(items.data(), items.size(), ctx);
  }

  // ---- hittable.rs:208-415 --------------------------------------------------------------
  bool hit(const Ray& r, double t_min, double t_max, Rng& rng, HitRecord& out) const {
    bool ok = hit_inner(r, t_min, t_max, rng, out);
    if (ok && object_id != NO_OBJECT) out.object_id = object_id;
    return ok;
  }

  bool hit_inner(const Ray& r, double t_min, double t_max, Rng& rng, HitRecord& out) const {
    switch (kind) {
      case Cube: {
        ORC_COUNT(cube_tests);
        for (int i = 0; i < 3; ++i) {
          double inv_d = 1.0 / r.direction[i];
          double t0 = (p0[i] - r.origin[i]) * inv_d;
          double t1 = (p1[i] - r.origin[i]) * inv_d;
          if (inv_d < 0.0) std::swap(t0, t1);
          t_min = std::fmax(t0, t_min);
          t_max = std::fmin(t1, t_max);
          if (t_max <= t_min) return false;
        }
        ORC_COUNT(cube_hits);
        double t = t_min;
        V3 p = r.at(t);
        V3 rel_p = (p - p0) / (p1 - p0);
        double epsilon = 0.002;
        V3 normal;
        if (rel_p.x < epsilon)
          normal = V3(1.0, 0.0, 0.0);
        else if (rel_p.x > 1.0 - epsilon)
          normal = V3(-1.0, 0.0, 0.0);
        else if (rel_p.y < epsilon)
          normal = V3(0.0, 1.0, 0.0);
        else if (rel_p.y > 1.0 - epsilon)
          normal = V3(0.0, -1.0, 0.0);
        else if (rel_p.z < epsilon)
          normal = V3(0.0, 0.0, 1.0);
        else if (rel_p.z > 1.0 - epsilon)
          normal = V3(0.0, 0.0, -1.0);
        else
          normal = normalize(V3(0.5, 0.5, 0.5));
        double az = std::fabs(normal.z), ax = std::fabs(normal.x), ay = std::fabs(normal.y);
        V2 uv;
        uv.x = rel_p.x * az + rel_p.y * ax + rel_p.x * ay;
        uv.y = rel_p.y * az + rel_p.z * ax + rel_p.z * ay;
        out = HitRecord::make(r, p, t, uv, normal, &material);
        return true;
      }
      case Sphere:
      case MovingSphere: {
        ORC_COUNT(sphere_tests);
        V3 c = kind == Sphere ? center : moving_center(r.time);
        V3 oc = r.origin - c;
        double a = sqrlen(r.direction);
        double half_b = dot(oc, r.direction);
        double cc = sqrlen(oc) - radius * radius;
        double discriminant = half_b * half_b - a * cc;
        if (discriminant < 0.0) return false;
        double sqrtd = std::sqrt(discriminant);
        double root = (-half_b - sqrtd) / a;
        if (root < t_min || root > t_max) {
          root = (-half_b + sqrtd) / a;
          if (root < t_min || root > t_max) return false;
        }
        ORC_COUNT(sphere_hits);
        V3 p = r.at(root);
        V3 c2 = kind == Sphere ? center : moving_center(r.time);
        V3 outward_normal = (p - c2) / radius;
        out = HitRecord::make(r, p, root, get_sphere_uv(outward_normal), outward_normal,
                              &material);
        return true;
      }
      case ConstantMedium: {
        ORC_COUNT(medium_visits);
        HitRecord hit1, hit2;
        ORC_COUNT(medium_probes);
        if (!child->hit(r, -INF, INF, rng, hit1)) return false;
        ORC_COUNT(medium_probes);
        if (!child->hit(r, hit1.t + 0.0001, INF, rng, hit2)) return false;
        hit1.t = std::fmax(hit1.t, t_min);
        hit2.t = std::fmin(hit2.t, t_max);
        if (hit1.t >= hit2.t) return false;
        hit1.t = std::fmax(hit1.t, 0.0);
        double ray_length = length(r.direction);
        double distance_inside_boundary = (hit2.t - hit1.t) * ray_length;
        double hit_distance = neg_inv_density * std::log(rng.keyed_f64(owner_id));
        if (hit_distance > distance_inside_boundary) return false;
        double t = hit1.t + hit_distance / ray_length;
        V3 p = r.at(t);
        out = HitRecord::make(r, p, t, V2{0.0, 0.0}, V3(1.0, 0.0, 0.0), &material);
        return true;
      }
      case Translate: {
        ORC_COUNT(translate_visits);
        Ray moved{r.origin - offset, r.direction, r.time};
        if (!child->hit(moved, t_min, t_max, rng, out)) return false;
        out.p = out.p + offset;
        return true;
      }
      case Rotate: {
        ORC_COUNT(rotate_visits);
        Ray rotated{rotate(r.origin, rotation), rotate(r.direction, rotation), r.time};
        if (!child->hit(rotated, t_min, t_max, rng, out)) return false;
        Quat inv = invert(rotation);
        out.p = rotate(out.p, inv);
        out.normal = rotate(out.normal, inv);
        return true;
      }
      case List: {
        bool any = false;
        double closest_so_far = t_max;
        for (const auto& h : list) {
          HitRecord rec;
          if (h.hit(r, t_min, closest_so_far, rng, rec)) {
            if (rec.t < closest_so_far) {
              closest_so_far = rec.t;
              out = rec;
              any = true;
            }
          }
        }
        return any;
      }
      case Bvh: {
        ORC_COUNT(nodes_visited);
        if (!box.hit(r, t_min, t_max)) return false;
        HitRecord lrec;
        if (left->hit(r, t_min, t_max, rng, lrec)) {
          HitRecord rrec;
          if (right->hit(r, t_min, lrec.t, rng, rrec) && rrec.t < lrec.t)
            out = rrec;
          else
            out = lrec;
          return true;
        }
        return right->hit(r, t_min, t_max, rng, out);
      }
      case Empty:
        return false;
    }
    return false;
  }

  // ---- hittable.rs:477-500 --------------------------------------------------------------
  double pdf_value(V3 origin, V3 direction, Rng& rng) const {
    switch (kind) {
      case Sphere: {
        ORC_COUNT(light_probes);
        HitRecord rec;
        if (hit(Ray{origin, direction, 0.0}, 0.0, INF, rng, rec)) {
          double cos_theta_max = std::sqrt(1.0 - radius * radius / sqrlen(center - origin));
          double solid_angle = 2.0 * PI * (1.0 - cos_theta_max);
          return 1.0 / solid_angle;
        }
        return 0.0;
      }
      case List: {
        double weight = 1.0 / (double)list.size();
        double sum = 0.0;  // Iterator::sum::<f64>() starts from 0.0 and adds in order
        for (const auto& h : list) sum += weight * h.pdf_value(origin, direction, rng);
        return sum;
      }
      case Cube: {
        // EXTENSION.  random() below picks a point uniformly on the WHOLE surface (area A), so the density of a
        // direction is the sum over the surface points it meets — a ray crosses a box's surface where it enters
        // (if the origin is outside) and where it leaves — of distance^2 / (|cos| A).  With an unnormalised
        // direction v, the point at parameter t and an axis-aligned face normal e_k:
        // distance^2 = t^2 |v|^2, |cos| = |v_k| / |v|.  The slab loop is aabb.rs:17-34 with (0, inf), plus the axes.
        if (!area_light) return 0.0;
        ORC_COUNT(light_probes);
        ORC_COUNT(cube_tests);
        double t_in = 0.0, t_out = INF;
        int k_in = -1, k_out = -1;
        for (int i = 0; i < 3; ++i) {
          double inv_d = 1.0 / direction[i];
          double t0 = (p0[i] - origin[i]) * inv_d;
          double t1 = (p1[i] - origin[i]) * inv_d;
          if (inv_d < 0.0) std::swap(t0, t1);
          if (t0 > t_in) {
            t_in = t0;
            k_in = i;
          }
          if (t1 < t_out) {
            t_out = t1;
            k_out = i;
          }
          if (t_out <= t_in) return 0.0;
        }
        ORC_COUNT(cube_hits);
        const V3 e = p1 - p0;
        const double area = 2.0 * (e.y * e.z + e.x * e.z + e.x * e.y);
        const double len2 = sqrlen(direction), len3 = len2 * std::sqrt(len2);
        double sum = 0.0;
        if (k_in >= 0) sum += t_in * t_in * len3 / (std::fabs(direction[k_in]) * area);
        if (k_out >= 0) sum += t_out * t_out * len3 / (std::fabs(direction[k_out]) * area);
        return sum;
      }
      default:
        return 0.0;
    }
  }

  // ---- hittable.rs:502-515 --------------------------------------------------------------
  V3 random(V3 origin, Rng& rng) const {
    switch (kind) {
      case Sphere: {
        V3 direction = center - origin;
        double distance_squared = sqrlen(direction);
        ONB uvw = ONB::from_w(direction);
        return uvw.local(random_to_sphere(rng, radius, distance_squared));
      }
      case List:
        if (!list.empty()) return list[rng.below((uint32_t)list.size())].random(origin, rng);
        return V3(1.0, 0.0, 0.0);
      case Cube: {
        // EXTENSION: a point uniform on the surface — three draws: the face (by area: the two x faces, the two y
        // faces, the two z faces, low side first), then the two coordinates within it in axis order
        if (!area_light) return V3(1.0, 0.0, 0.0);
        const V3 e = p1 - p0;
        const double ax = e.y * e.z, ay = e.x * e.z, az = e.x * e.y;
        const double s = rng.next_f64() * (2.0 * (ax + ay + az));
        const double r1 = rng.next_f64(), r2 = rng.next_f64();
        V3 p;
        if (s < 2.0 * ax)
          p = V3(s < ax ? p0.x : p1.x, p0.y + r1 * e.y, p0.z + r2 * e.z);
        else if (s < 2.0 * ax + 2.0 * ay)
          p = V3(p0.x + r1 * e.x, s < 2.0 * ax + ay ? p0.y : p1.y, p0.z + r2 * e.z);
        else
          p = V3(p0.x + r1 * e.x, p0.y + r2 * e.y, s < 2.0 * ax + 2.0 * ay + az ? p0.z : p1.z);
        return p - origin;
      }
      default:
        return V3(1.0, 0.0, 0.0);
    }
  }
};

}  // namespace orc
