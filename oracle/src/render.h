// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md). Not part of the product path.
//
// pixel_from_color (main.rs:51-63), ray_color (main.rs:65-105), the lights fold and world
// BVH (main.rs:125-131), and the render loop (main.rs:152-170) with std::thread standing in
// for rayon's par_iter (main.rs:157).
#pragma once
#include <atomic>
#include <chrono>
#include <thread>

#include "scenes.h"

namespace orc {

struct Scene {
  Hittable world;
  bool has_lights = false;
  Hittable lights;
  Camera cam;
  CameraSpec cam_spec;
  V3 background;
  RotateBBox rotate_bbox = RotateBBox::Reference;
  uint32_t n_objects = 0;
  uint32_t bvh_calls = 0;
};

// (bvh_axis, the BVH split-axis stream, lives in scenes.h: scene 16 builds a BVH inside its scene function)

inline Scene build_scene(const SceneSpec& spec, double aspect_ratio, uint64_t bvh_seed,
                         int rotate_bbox_override /* -1 scene default, 0 reference, 1 correct */) {
  SceneSpec spec_with_build = spec;  // what a scene function that calls new_bvh itself needs (scene 16)
  spec_with_build.bvh_seed = bvh_seed;
  spec_with_build.rotate_bbox_override = rotate_bbox_override;
  SceneDef def = build_scene_def(spec_with_build);
  Scene sc;
  sc.background = def.background;
  sc.rotate_bbox = def.rotate_bbox;
  if (rotate_bbox_override == 0) sc.rotate_bbox = RotateBBox::Reference;
  if (rotate_bbox_override == 1) sc.rotate_bbox = RotateBBox::Correct;
  sc.n_objects = (uint32_t)def.world.size();
  for (uint32_t i = 0; i < def.world.size(); ++i) {
    def.world[i].object_id = i;
    def.world[i].set_owner(i);
  }

  // main.rs:125
  BuildCtx ctx;
  ctx.time_start = 0.0;
  ctx.time_end = 1.0;
  ctx.rotate_bbox = sc.rotate_bbox;
  ctx.axis_of_call = [bvh_seed](uint32_t k) { return bvh_axis(bvh_seed, k); };
  sc.world = Hittable::new_bvh(def.world, ctx);
  sc.bvh_calls = ctx.calls;

  // main.rs:126-131
  if (def.lights.empty()) {
    sc.has_lights = false;
  } else if (def.lights.size() == 1) {
    if (def.lights[0].kind == Hittable::Empty) {
      sc.has_lights = false;
    } else {
      sc.has_lights = true;
      sc.lights = std::move(def.lights[0]);
    }
  } else {
    sc.has_lights = true;
    sc.lights = Hittable::new_list(std::move(def.lights));
  }

  sc.cam_spec = def.cam;
  sc.cam = Camera::make(def.cam.lookfrom, def.cam.lookat, def.cam.vup, def.cam.vfov, aspect_ratio,
                        def.cam.aperture, def.cam.focus_dist, def.cam.time_open, def.cam.time_close);
  return sc;
}

// main.rs:51-63.  Rust `as u8` saturates and maps NaN to 0; f64::clamp(NaN) stays NaN.
inline uint8_t quantise(double c) {
  double g = std::sqrt(c);
  if (!(g == g)) return 0;
  if (g < 0.0) g = 0.0;
  if (g > 0.9999) g = 0.9999;
  return (uint8_t)(g * 256.0);
}

// main.rs:65-105.  `slot` = 1 + (bounce index): the draw stream of this invocation.
inline V3 ray_color(const Ray& r, V3 background, const Scene& sc, uint32_t depth, uint32_t slot,
                    Rng& rng) {
  if (depth == 0) return V3();
  rng.begin(slot);
  ORC_COUNT(rays);
  HitRecord hit;
  if (sc.world.hit(r, 0.001, INF, rng, hit)) {
    ORC_COUNT(closest_hits);
    V3 emitted = hit.material->emit(hit);
    ScatterRecord scatter = hit.material->scatter(r, hit, rng);
    if (scatter.kind == ScatterRecord::Specular) {
      ORC_COUNT(specular_vertices);
      return scatter.attenuation * ray_color(scatter.ray, background, sc, depth - 1, slot + 1, rng);
    }
    if (scatter.kind == ScatterRecord::Pdf) {
      ORC_COUNT(diffuse_vertices);
      // pdf = Mixture[Hittable(lights, hit.p), Cosine] when lights exist, else Cosine.
      V3 direction;
      if (sc.has_lights) {
        uint32_t pick = rng.below(2);  // pdfs.choose(): index 0 = light, 1 = cosine
        if (pick == 0)
          direction = sc.lights.random(hit.p, rng);
        else
          direction = scatter.cosine.local(random_cosine_direction(rng));
      } else {
        direction = scatter.cosine.local(random_cosine_direction(rng));
      }
      Ray scattered{hit.p, direction, r.time};
      // pdf.rs:13-31
      auto cosine_value = [&](V3 d) {
        double cosine = dot(normalize(d), scatter.cosine.w);
        return cosine <= 0.0 ? 0.0 : cosine / PI;
      };
      double pdf_value;
      if (sc.has_lights) {
        double probability = 1.0 / 2.0;
        pdf_value = 0.0;
        pdf_value += probability * sc.lights.pdf_value(hit.p, scattered.direction, rng);
        pdf_value += probability * cosine_value(scattered.direction);
      } else {
        pdf_value = cosine_value(scattered.direction);
      }
      double spdf = hit.material->scattering_pdf(r, hit, scattered);
      V3 rec = ray_color(scattered, background, sc, depth - 1, slot + 1, rng);
      return emitted + ((scatter.attenuation * spdf) * rec) / pdf_value;
    }
    ORC_COUNT(emitter_hits);
    return emitted;
  }
  ORC_COUNT(misses);
  return background;
}

struct RenderParams {
  uint32_t width = 0, height = 0, spp = 0, max_depth = 0;
  uint64_t seed = 1, sample_offset = 0;
  uint32_t threads = 1;
};

// main.rs:152-170.  accum is [h][w][3], the SUM over the spp samples (not the mean).
inline double render(const Scene& sc, const RenderParams& p, double* accum, Counters* total) {
  auto t0 = std::chrono::steady_clock::now();
  std::atomic<uint32_t> next_row{0};
  uint32_t nthreads = p.threads ? p.threads : 1;
  std::vector<Counters> per_thread(nthreads);
  auto worker = [&](uint32_t tid) {
    Counters* const saved = tls_counters();  // the calling thread keeps its own sink afterwards
    tls_counters() = &per_thread[tid];
    for (;;) {
      uint32_t y = next_row.fetch_add(1);
      if (y >= p.height) break;
      for (uint32_t x = 0; x < p.width; ++x) {
        uint32_t pixel = y * p.width + x;
        V3 pixel_color;
        for (uint32_t s = 0; s < p.spp; ++s) {
          uint32_t sample = (uint32_t)(p.sample_offset + s);
          Rng rng(p.seed, pixel, sample, 0);
          rng.begin(0);
          double ju = rng.next_f64();
          double jv = rng.next_f64();
          double u = ((double)x + ju) / (double)(p.width - 1);
          double v = ((double)(p.height - y - 1) + jv) / (double)(p.height - 1);
          Ray r = sc.cam.get_ray(u, v, rng);
          V3 c = ray_color(r, sc.background, sc, p.max_depth, 1, rng);
          if (!(std::isfinite(c.x) && std::isfinite(c.y) && std::isfinite(c.z)))
            ORC_COUNT(nonfinite_samples);
          pixel_color = pixel_color + c;
        }
        accum[3 * (size_t)pixel + 0] = pixel_color.x;
        accum[3 * (size_t)pixel + 1] = pixel_color.y;
        accum[3 * (size_t)pixel + 2] = pixel_color.z;
      }
    }
    tls_counters() = saved;
  };
  std::vector<std::thread> pool;
  for (uint32_t t = 1; t < nthreads; ++t) pool.emplace_back(worker, t);
  worker(0);
  for (auto& th : pool) th.join();
  if (total)
    for (auto& c : per_thread) total->add(c);
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// Deterministic primary rays (no jitter, no lens offset, fixed time): parity levels 1-2.
struct PrimaryHit {
  uint32_t object_id;
  double t;
  V3 normal;
  V2 uv;
  bool front_face;
};
inline PrimaryHit primary_hit(const Scene& sc, uint32_t w, uint32_t h, uint32_t x, uint32_t y,
                              uint64_t seed, double time) {
  uint32_t pixel = y * w + x;
  double u = ((double)x + 0.5) / (double)(w - 1);
  double v = ((double)(h - y - 1) + 0.5) / (double)(h - 1);
  Ray r;
  r.origin = sc.cam.origin;
  r.direction = sc.cam.lower_left_corner + u * sc.cam.horizontal + v * sc.cam.vertical - sc.cam.origin;
  r.time = time;
  Rng rng(seed, pixel, 0, 1);
  rng.begin(1);
  HitRecord hit;
  PrimaryHit out{NO_OBJECT, INF, V3(), V2(), false};
  if (sc.world.hit(r, 0.001, INF, rng, hit)) {
    out.object_id = hit.object_id;
    out.t = hit.t;
    out.normal = hit.normal;
    out.uv = hit.uv;
    out.front_face = hit.front_face;
  }
  return out;
}

}  // namespace orc
