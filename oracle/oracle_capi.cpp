// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Not part of the product path.
//
// C entry points over the CPU restatement, for ctypes (tests/, __graft_entry__.smoke(),
// bench.py's cpu_baseline leg). Nothing under raytracer_amd/ may load this library.
#include <algorithm>
#include <cstring>
#include <string>

#include "src/render.h"

using namespace orc;

namespace {
thread_local std::string g_err;
}

extern "C" {

struct orc_scene {
  Scene scene;
};

struct orc_camera {  // same field order as rt_camera in include/rt.h
  double origin[3], lower_left[3], horizontal[3], vertical[3], u[3], v[3];
  double lens_radius, time_open, time_close;
};

struct orc_info {
  uint32_t n_objects, n_nodes, has_lights, n_lights, rotate_bbox_mode, root_is_leaf;
  double background[3];
  orc_camera camera;
};

struct orc_counters {
  uint64_t rays, light_probes, medium_probes, nodes_visited, sphere_tests, sphere_hits,
      cube_tests, cube_hits, rotate_visits, translate_visits, medium_visits, closest_hits,
      diffuse_vertices, specular_vertices, emitter_hits, misses, nonfinite_samples;
};

const char* orc_last_error() { return g_err.c_str(); }

orc_scene* orc_scene_create(int scene_id, double aspect_ratio, uint64_t scene_seed,
                            uint64_t bvh_seed, uint32_t n, int rotate_bbox_override,
                            const uint8_t* texels, uint32_t tex_w, uint32_t tex_h) {
  try {
    SceneSpec spec;
    spec.scene_id = scene_id;
    spec.scene_seed = scene_seed;
    spec.n = n;
    if (texels && tex_w && tex_h) {
      spec.texels = std::make_shared<const std::vector<uint8_t>>(texels, texels + (size_t)3 * tex_w * tex_h);
      spec.tex_w = tex_w;
      spec.tex_h = tex_h;
    }
    auto* s = new orc_scene;
    s->scene = build_scene(spec, aspect_ratio, bvh_seed, rotate_bbox_override);
    return s;
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}

void orc_scene_destroy(orc_scene* s) { delete s; }

// the Bvh behind a world-level object's Translate / Rotate wrappers (an instance), or null
static const Hittable* instance_tree(const Hittable& h) {
  const Hittable* c = &h;
  while ((c->kind == Hittable::Translate || c->kind == Hittable::Rotate) && c->child) c = c->child.get();
  return (c != &h && c->kind == Hittable::Bvh) ? c : nullptr;
}
static uint32_t count_nodes(const Hittable& h) {
  if (h.kind != Hittable::Bvh) {
    const Hittable* t = instance_tree(h);
    return t ? count_nodes(*t) : 0;
  }
  return 1 + count_nodes(*h.left) + count_nodes(*h.right);
}

static void fill3(double* d, V3 v) {
  d[0] = v.x;
  d[1] = v.y;
  d[2] = v.z;
}

void orc_scene_info(const orc_scene* s, orc_info* out) {
  const Scene& sc = s->scene;
  out->n_objects = sc.n_objects;
  out->n_nodes = count_nodes(sc.world);
  out->has_lights = sc.has_lights ? 1 : 0;
  out->n_lights = !sc.has_lights ? 0 : (sc.lights.kind == Hittable::List ? (uint32_t)sc.lights.list.size() : 1);
  out->rotate_bbox_mode = sc.rotate_bbox == RotateBBox::Correct ? 1 : 0;
  out->root_is_leaf = sc.world.kind != Hittable::Bvh ? 1 : 0;
  fill3(out->background, sc.background);
  fill3(out->camera.origin, sc.cam.origin);
  fill3(out->camera.lower_left, sc.cam.lower_left_corner);
  fill3(out->camera.horizontal, sc.cam.horizontal);
  fill3(out->camera.vertical, sc.cam.vertical);
  fill3(out->camera.u, sc.cam.u);
  fill3(out->camera.v, sc.cam.v);
  out->camera.lens_radius = sc.cam.lens_radius;
  out->camera.time_open = sc.cam.time_open;
  out->camera.time_close = sc.cam.time_close;
}

// Pre-order dump of the pointer tree: node k = (box, left ref, right ref); a ref is either a
// node's pre-order index or 0x80000000 | object_id for a leaf.  Returns the root ref
// (0x80000000|id for a bare 1-object world, 0xFFFFFFFF for an Empty world).
// An INSTANCE (a world-level Translate / Rotate chain over a Bvh, scene 16) is a leaf of the world's tree; its own tree
// is dumped behind the world's nodes, in the order the world's nodes reference the instances (node index order, a
// node's left child before its right); the leaves of an instance's tree carry the instance's object id (the owner).
struct DumpCtx {
  uint32_t next = 0;
  double* boxes;
  uint32_t *left, *right;
  std::vector<std::pair<uint32_t, const Hittable*>> node_leaves;  // (node, world-level leaf) in node order, left first
};
static uint32_t dump(const Hittable& h, DumpCtx& c, bool world_level) {
  if (h.kind == Hittable::Empty) return 0xFFFFFFFFu;
  if (h.kind != Hittable::Bvh) return 0x80000000u | (h.object_id != NO_OBJECT ? h.object_id : h.owner_id);
  uint32_t me = c.next++;
  fill3(c.boxes + 6 * (size_t)me, h.box.minimum);
  fill3(c.boxes + 6 * (size_t)me + 3, h.box.maximum);
  if (world_level) {
    if (h.left->kind != Hittable::Bvh) c.node_leaves.emplace_back(me, h.left.get());
    if (h.right->kind != Hittable::Bvh) c.node_leaves.emplace_back(me, h.right.get());
  }
  uint32_t l = dump(*h.left, c, world_level);
  uint32_t r = dump(*h.right, c, world_level);
  c.left[me] = l;
  c.right[me] = r;
  return me;
}
uint32_t orc_bvh_preorder(const orc_scene* s, double* boxes, uint32_t* left, uint32_t* right) {
  DumpCtx c;
  c.boxes = boxes;
  c.left = left;
  c.right = right;
  const Hittable& world = s->scene.world;
  const uint32_t root = dump(world, c, true);
  if (world.kind != Hittable::Bvh && world.kind != Hittable::Empty) c.node_leaves.emplace_back(0u, &world);
  std::stable_sort(c.node_leaves.begin(), c.node_leaves.end(),
                   [](const auto& a, const auto& b) { return a.first < b.first; });
  for (const auto& nl : c.node_leaves)
    if (const Hittable* t = instance_tree(*nl.second)) dump(*t, c, false);
  return root;
}

void orc_primary_hits(const orc_scene* s, uint32_t w, uint32_t h, uint64_t seed, double time,
                      uint32_t* object_id, double* t, double* normal, double* uv,
                      uint8_t* front_face) {
  for (uint32_t y = 0; y < h; ++y)
    for (uint32_t x = 0; x < w; ++x) {
      PrimaryHit ph = primary_hit(s->scene, w, h, x, y, seed, time);
      size_t i = (size_t)y * w + x;
      object_id[i] = ph.object_id;
      t[i] = ph.t;
      if (normal) fill3(normal + 3 * i, ph.normal);
      if (uv) {
        uv[2 * i] = ph.uv.x;
        uv[2 * i + 1] = ph.uv.y;
      }
      if (front_face) front_face[i] = ph.front_face ? 1 : 0;
    }
}

double orc_render(const orc_scene* s, uint32_t w, uint32_t h, uint32_t spp, uint32_t max_depth,
                  uint64_t seed, uint64_t sample_offset, uint32_t threads, double* accum,
                  orc_counters* counters) {
  RenderParams p;
  p.width = w;
  p.height = h;
  p.spp = spp;
  p.max_depth = max_depth;
  p.seed = seed;
  p.sample_offset = sample_offset;
  p.threads = threads;
  Counters c;
  double secs = render(s->scene, p, accum, &c);
  if (counters) {
    static_assert(sizeof(orc_counters) == sizeof(Counters), "counter layouts must match");
    std::memcpy(counters, &c, sizeof(c));
  }
  return secs;
}

// main.rs:167-168 — mean over spp, then pixel_from_color
void orc_tonemap(const double* accum, uint64_t n_px, uint32_t spp, uint8_t* rgb8) {
  for (uint64_t i = 0; i < 3 * n_px; ++i) rgb8[i] = quantise(accum[i] / (double)spp);
}

// ---- known-answer hooks (hittable.rs:547-556, alias.rs:188-215) ---------------------------
void orc_get_sphere_uv(const double* p, double* uv) {
  V2 r = get_sphere_uv(V3(p[0], p[1], p[2]));
  uv[0] = r.x;
  uv[1] = r.y;
}
void orc_rotate(double angle_rad, const double* axis, const double* v, double* out) {
  Quat q = new_quaternion(angle_rad, V3(axis[0], axis[1], axis[2]));
  fill3(out, rotate(V3(v[0], v[1], v[2]), q));
}
void orc_philox(const uint32_t* ctr, const uint32_t* key, uint32_t rounds, uint32_t* out) {
  Philox4 r = philox4x32(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], (int)rounds);
  std::memcpy(out, r.v, sizeof(r.v));
}
int orc_bvh_axis(uint64_t bvh_seed, uint32_t call) { return bvh_axis(bvh_seed, call); }
// stdrng.h known-answer hooks: the block function on a caller-made state; StdRng::seed_from_u64 streams
void orc_chacha_block(const uint32_t* state, uint32_t rounds, uint32_t* out) { chacha_block(state, (int)rounds, out); }
void orc_chacha_quarter(uint32_t* abcd) { chacha_quarter(abcd[0], abcd[1], abcd[2], abcd[3]); }
void orc_stdrng_draws(uint64_t seed, uint32_t n, uint64_t* words, double* unit, double* ranged, double lo, double hi) {
  StdRng a = StdRng::seed_from_u64(seed), b = a, c = a;
  for (uint32_t i = 0; i < n; ++i) {
    words[i] = a.next_u64();
    unit[i] = b.next_f64();
    ranged[i] = c.range(lo, hi);
  }
}
void orc_scene_draws(uint64_t seed, uint32_t stream, uint32_t n, double* out) {
  Rng r = scene_rng(seed, stream);
  for (uint32_t i = 0; i < n; ++i) out[i] = r.next_f64();
}
void orc_path_draws(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t slot, uint32_t n, double* out) {
  Rng r(seed, pixel, sample, slot);
  r.begin(slot);
  for (uint32_t i = 0; i < n; ++i) out[i] = r.next_f64();
}

}  // extern "C"
