// TEST INFRASTRUCTURE (uses the oracle): at every vertex of every path of a generated scene, does the exhaustive List walk
// over the world's leaves (hittable.rs:387-399) find the same hit as the BVH walk (hittable.rs:400-412)?  If the node
// boxes bound their leaves it must.  Prints the first few rays where it does not, with the leaf and the boxes above it.
// build: g++ -O2 -std=c++17 -I oracle/src tests/dbg/list_vs_bvh.cpp -o /tmp/list_vs_bvh -lpthread
// usage: list_vs_bvh <scene_id> <scene_seed> <n> <rotate_bbox -1|0|1> [w spp depth render_seed]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "render.h"

using namespace orc;

static std::vector<const Hittable*> g_leaves;
static void collect(const Hittable& h) {
  if (h.kind == Hittable::Bvh) {
    collect(*h.left);
    collect(*h.right);
  } else if (h.kind != Hittable::Empty) {
    g_leaves.push_back(&h);
  }
}
static bool path_to(const Hittable& h, const Hittable* leaf, std::vector<const Hittable*>& out) {
  if (&h == leaf) return true;
  if (h.kind != Hittable::Bvh) return false;
  out.push_back(&h);
  if (path_to(*h.left, leaf, out) || path_to(*h.right, leaf, out)) return true;
  out.pop_back();
  return false;
}
static int g_shown = 0;
static long g_vertices = 0, g_bad = 0;

static void check(const Scene& sc, const Ray& r, Rng rng /* copy: the draws are keyed */, uint32_t px, uint32_t py, uint32_t s,
                  uint32_t depth) {
  ++g_vertices;
  HitRecord a, b, tmp;
  Rng r1 = rng, r2 = rng;
  const bool ha = sc.world.hit(r, 0.001, INF, r1, a);
  bool hb = false;
  double closest = INF;
  const Hittable* winner = nullptr;
  for (const Hittable* leaf : g_leaves)
    if (leaf->hit(r, 0.001, closest, r2, tmp)) {
      if (!hb || tmp.t < closest) {
        hb = true;
        closest = tmp.t;
        b = tmp;
        winner = leaf;
      }
    }
  if (ha == hb && (!ha || (a.t == b.t && a.object_id == b.object_id))) return;
  ++g_bad;
  if (g_shown++ >= 4) return;
  printf("pixel (%u,%u) sample %u depth-left %u: BVH %s t=%.17g obj %u | list %s t=%.17g obj %u\n", px, py, s, depth,
         ha ? "hit" : "miss", ha ? a.t : 0.0, ha ? a.object_id : 0u, hb ? "hit" : "miss", hb ? b.t : 0.0, hb ? b.object_id : 0u);
  printf("  ray o=(%.17g, %.17g, %.17g) d=(%.17g, %.17g, %.17g) time %.17g\n", r.origin.x, r.origin.y, r.origin.z, r.direction.x,
         r.direction.y, r.direction.z, r.time);
  if (winner) {
    printf("  list winner kind %d, hit point (%.17g, %.17g, %.17g)\n", (int)winner->kind, b.p.x, b.p.y, b.p.z);
    std::vector<const Hittable*> up;
    path_to(sc.world, winner, up);
    for (const Hittable* n : up) {
      const bool bh = n->box.hit(r, 0.001, INF);
      const V3 lo = n->box.minimum, hi = n->box.maximum;
      const bool inside = b.p.x >= lo.x && b.p.x <= hi.x && b.p.y >= lo.y && b.p.y <= hi.y && b.p.z >= lo.z && b.p.z <= hi.z;
      printf("    node box [%.9g %.9g %.9g | %.9g %.9g %.9g] ray-hits-box(t_max=inf) %d hit-point-inside %d\n", lo.x, lo.y, lo.z, hi.x,
             hi.y, hi.z, (int)bh, (int)inside);
    }
  }
}

static V3 trace(const Ray& r, const Scene& sc, uint32_t depth, uint32_t slot, Rng& rng, uint32_t px, uint32_t py, uint32_t s) {
  if (depth == 0) return V3();
  rng.begin(slot);
  check(sc, r, rng, px, py, s, depth);
  HitRecord hit;
  if (!sc.world.hit(r, 0.001, INF, rng, hit)) return sc.background;
  V3 emitted = hit.material->emit(hit);
  ScatterRecord scatter = hit.material->scatter(r, hit, rng);
  if (scatter.kind == ScatterRecord::Specular) return scatter.attenuation * trace(scatter.ray, sc, depth - 1, slot + 1, rng, px, py, s);
  if (scatter.kind == ScatterRecord::Pdf) {
    V3 direction;
    if (sc.has_lights) {
      uint32_t pick = rng.below(2);
      direction = pick == 0 ? sc.lights.random(hit.p, rng) : scatter.cosine.local(random_cosine_direction(rng));
    } else {
      direction = scatter.cosine.local(random_cosine_direction(rng));
    }
    Ray scattered{hit.p, direction, r.time};
    if (sc.has_lights) (void)sc.lights.pdf_value(hit.p, scattered.direction, rng);
    return trace(scattered, sc, depth - 1, slot + 1, rng, px, py, s);
  }
  return emitted;
}

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  SceneSpec spec;
  spec.scene_id = atoi(argv[1]);
  spec.scene_seed = strtoull(argv[2], nullptr, 10);
  spec.n = (uint32_t)atoi(argv[3]);
  const int mode = atoi(argv[4]);
  const uint32_t w = argc > 5 ? atoi(argv[5]) : 64, spp = argc > 6 ? atoi(argv[6]) : 4, depth = argc > 7 ? atoi(argv[7]) : 10;
  const uint64_t seed = argc > 8 ? strtoull(argv[8], nullptr, 10) : spec.scene_seed;
  const double aspect = 1.5;
  const uint32_t h = (uint32_t)((double)w / aspect);
  Scene sc = build_scene(spec, aspect, 7, mode);
  collect(sc.world);
  printf("scene %d seed %llu n %u: %zu leaves, rotate_bbox %s\n", spec.scene_id, (unsigned long long)spec.scene_seed, spec.n,
         g_leaves.size(), sc.rotate_bbox == RotateBBox::Reference ? "reference" : "correct");
  Counters sink;
  tls_counters() = &sink;
  for (uint32_t y = 0; y < h; ++y)
    for (uint32_t x = 0; x < w; ++x)
      for (uint32_t s = 0; s < spp; ++s) {
        Rng rng(seed, y * w + x, s, 0);
        rng.begin(0);
        double ju = rng.next_f64(), jv = rng.next_f64();
        double u = ((double)x + ju) / (double)(w - 1), v = ((double)(h - y - 1) + jv) / (double)(h - 1);
        Ray r = sc.cam.get_ray(u, v, rng);
        (void)trace(r, sc, depth, 1, rng, x, y, s);
      }
  printf("%ld vertices, %ld where the List walk and the BVH walk disagree\n", g_vertices, g_bad);
  return 0;
}
