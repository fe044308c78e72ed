// Included by rt_f32.hip and rt_f64.hip with RT_SUFFIX / RT_REAL / RT_EARLY_EXIT defined.
#include <algorithm>
#include <cstdlib>

#include "device/kernels.hpp"
#include "launch.hpp"

namespace rt {

// A launcher reports its own launch's error: hipGetLastError() is sticky per host thread, so an error some EARLIER call
// left behind (the caller's own HIP calls included — e.g. a refused hipSetDevice) is read and dropped first.
#ifndef RT_LAUNCH
#define RT_LAUNCH(...)               \
  do {                               \
    (void)hipGetLastError();         \
    hipLaunchKernelGGL(__VA_ARGS__); \
  } while (0)
#endif

#define RT_CAT_(a, b) a##b
#define RT_CAT(a, b) RT_CAT_(a, b)

hipError_t RT_CAT(launch_render_, RT_SUFFIX)(const SceneView<RT_REAL>& sc, const CameraView<RT_REAL>& cam,
                                             uint32_t width, uint32_t height, uint32_t sample_begin,
                                             uint32_t n_samples, uint32_t max_depth, uint64_t seed, int accumulate,
                                             bool stats, RT_REAL* accum, DeviceStats* dstats, LaunchShape shape,
                                             hipStream_t st) {
  RenderArgs a{};
  a.width = width;
  a.height = height;
  a.sample_begin = sample_begin;
  a.n_samples = n_samples;
  a.max_depth = max_depth;
  a.accumulate = accumulate ? 1u : 0u;
  a.splits = shape.splits;
  a.seed = seed;
  dim3 grid(shape.tiles, shape.splits, 1), block(BLOCK, 1, 1);
  size_t lds = (size_t)shape.stack_depth * BLOCK * sizeof(uint32_t);
  a.stack_depth = shape.stack_depth;
  a.refill_eighths = shape.refill_eighths;
  a.unit_samples = shape.unit_samples;
  a.unit_prefetch = shape.unit_prefetch;
  a.wave_tiles_x = shape.wave_tiles_x;
  a.wave_tiles = shape.wave_tiles;
  a.total_units = shape.total_units;
  a.leaf_shift = shape.leaf_shift;
  a.direct_accum = shape.direct_accum;
  a.xcd_chunk = shape.xcd_chunk;
  a.tile_band = shape.tile_band;
#if RT_HAS_LINEAR
  // render_queue_kernel (jobs not tied to lanes, camera rays a wave at a time) instead of render_persist_kernel
  static const bool use_queue = getenv("RT_QUEUE") != nullptr && atoi(getenv("RT_QUEUE")) != 0;
  if (shape.trav == TRAV_LINEAR && shape.unit_samples > 0 && use_queue && max_depth < 65536) {
    if (stats)
      RT_LAUNCH((render_queue_kernel<RT_REAL, true, TRAV_LINEAR, RT_EARLY_EXIT>), grid, block, 0, st, sc, cam, a, accum, dstats);
    else
      RT_LAUNCH((render_queue_kernel<RT_REAL, false, TRAV_LINEAR, RT_EARLY_EXIT>), grid, block, 0, st, sc, cam, a, accum, dstats);
    return hipGetLastError();
  }
  if (shape.trav == TRAV_LINEAR && shape.unit_samples > 0) {  // persistent form: grid = workers, not tiles
    if (stats)
      RT_LAUNCH((render_persist_kernel<RT_REAL, true>), grid, block, 0, st, sc, cam, a, accum, dstats);
    else
      RT_LAUNCH((render_persist_kernel<RT_REAL, false>), grid, block, 0, st, sc, cam, a, accum, dstats);
    return hipGetLastError();
  }
  if (shape.trav == TRAV_LINEAR) {
    // opt-in experiment ("megakernel vs queue"): measured 18 % slower than render_kernel
    static const bool sorted = getenv("RT_LINEAR_SORTED") != nullptr;
    if (sorted && max_depth > 0 && max_depth < 65536) {
      if (stats)
        RT_LAUNCH((render_sorted_kernel<true>), grid, block, 0, st, sc, cam, a, accum, dstats);
      else
        RT_LAUNCH((render_sorted_kernel<false>), grid, block, 0, st, sc, cam, a, accum, dstats);
      return hipGetLastError();
    }
    if (stats)
      RT_LAUNCH((render_kernel<RT_REAL, true, RT_EARLY_EXIT, TRAV_LINEAR>), grid, block, 0, st, sc, cam, a,
                         accum, dstats);
    else
      RT_LAUNCH((render_kernel<RT_REAL, false, RT_EARLY_EXIT, TRAV_LINEAR>), grid, block, 0, st, sc, cam, a,
                         accum, dstats);
    return hipGetLastError();
  }
  if (shape.trav == TRAV_ORDERED) {
    const size_t lds2 = lds + lds / 2;  // 32-bit refs + 16-bit entry distances
    static const bool no_refill = getenv("RT_ORDERED_NO_REFILL") != nullptr;
    if (!no_refill && shape.unit_samples > 0 && use_queue) {  // experiment: the queue kernel around the f32 ordered walk
      a.wave_tiles_x = (width + 7) / 8;  // (its units are 8x8 tiles in row-major order)
      a.wave_tiles = a.wave_tiles_x * ((height + 7) / 8);
      a.total_units = a.wave_tiles * ((n_samples + a.unit_samples - 1) / a.unit_samples);
      if (stats)
        RT_LAUNCH((render_queue_kernel<RT_REAL, true, TRAV_ORDERED, RT_EARLY_EXIT>), grid, block, lds2, st, sc, cam, a, accum, dstats);
      else
        RT_LAUNCH((render_queue_kernel<RT_REAL, false, TRAV_ORDERED, RT_EARLY_EXIT>), grid, block, lds2, st, sc, cam, a, accum, dstats);
      return hipGetLastError();
    }
    if (!no_refill && shape.unit_samples > 0) {  // persistent waves, work units from a counter: grid = workers
      const size_t lds3 = lds2 + (shape.direct_accum ? 0 : (BLOCK / 64) * UNITS_LDS_FLOATS * sizeof(float));
      if (stats)
        RT_LAUNCH((render_units_kernel<true>), grid, block, lds3, st, sc, cam, a, accum, dstats);
      else
        RT_LAUNCH((render_units_kernel<false>), grid, block, lds3, st, sc, cam, a, accum, dstats);
      return hipGetLastError();
    }
    if (!no_refill) {
      if (stats)
        RT_LAUNCH((render_refill_kernel<true>), grid, block, lds2, st, sc, cam, a, accum, dstats);
      else
        RT_LAUNCH((render_refill_kernel<false>), grid, block, lds2, st, sc, cam, a, accum, dstats);
      return hipGetLastError();
    }
    if (stats)
      RT_LAUNCH((render_kernel<RT_REAL, true, RT_EARLY_EXIT, TRAV_ORDERED>), grid, block, lds2, st, sc, cam,
                         a, accum, dstats);
    else
      RT_LAUNCH((render_kernel<RT_REAL, false, RT_EARLY_EXIT, TRAV_ORDERED>), grid, block, lds2, st, sc, cam,
                         a, accum, dstats);
    return hipGetLastError();
  }
#endif
  // The reference-order kernel is the one that carries the Perlin texture code (extension): always in the f64
  // audit build, and as a separate instantiation in f32, launched only for scenes that have a noise texture
  // (rt_api.hip routes those here) so that no other kernel pays registers for it.
  constexpr bool F64 = sizeof(RT_REAL) == 8;
  if constexpr (F64) {
    // leaf-list scenes (rt_api.hip: make_shape): the leaf list over the scene arrays, persistent form, without the
    // extension code (scenes that use an extension take the reference-order kernel below)
    static const bool use_queue64 = getenv("RT_QUEUE") == nullptr || atoi(getenv("RT_QUEUE")) != 0;  // the default in f64 (RT_QUEUE=0: render_persist_kernel, A/B)
    if (shape.trav == TRAV_LEAVES && shape.unit_samples > 0 && !sc.has_noise && use_queue64) {
      if (stats)
        RT_LAUNCH((render_queue_kernel<RT_REAL, true, TRAV_LEAVES, RT_EARLY_EXIT>), grid, block, 0, st, sc, cam, a, accum, dstats);
      else
        RT_LAUNCH((render_queue_kernel<RT_REAL, false, TRAV_LEAVES, RT_EARLY_EXIT>), grid, block, 0, st, sc, cam, a, accum, dstats);
      return hipGetLastError();
    }
    if (shape.trav == TRAV_LEAVES && shape.unit_samples > 0 && !sc.has_noise) {
      if (stats)
        RT_LAUNCH((render_persist_kernel<RT_REAL, true, TRAV_LEAVES, RT_EARLY_EXIT, false>), grid, block, 0, st, sc, cam, a, accum,
                  dstats);
      else
        RT_LAUNCH((render_persist_kernel<RT_REAL, false, TRAV_LEAVES, RT_EARLY_EXIT, false>), grid, block, 0, st, sc, cam, a, accum,
                  dstats);
      return hipGetLastError();
    }
    if (shape.trav == TRAV_ORDERED && shape.unit_samples > 0 && !sc.has_noise) {  // big scenes: the ordered walk in f64
      const size_t lds2 = lds + lds / 2;  // 32-bit refs + 16-bit entry distances
      // jobs shared by the wave's lanes, camera rays a wave at a time (RT_QUEUE=0: lane = pixel, A/B) — where the
      // queue's 28 KB of LDS next to the stacks still leave three workgroups per CU (measured in f64: random_scene
      // 7.6 -> 5.1 ms, final_scene 53 -> 31 ms; 10^6 spheres, 40 KB of stacks: 8.3 -> 11.4 ms, so deep trees keep the
      // lane = pixel form)
      if (use_queue64 && lds2 + (28u << 10) <= (160u << 10) / 3u) {
        if (stats)
          RT_LAUNCH((render_queue_kernel<RT_REAL, true, TRAV_ORDERED, RT_EARLY_EXIT>), grid, block, lds2, st, sc, cam, a, accum, dstats);
        else
          RT_LAUNCH((render_queue_kernel<RT_REAL, false, TRAV_ORDERED, RT_EARLY_EXIT>), grid, block, lds2, st, sc, cam, a, accum, dstats);
        return hipGetLastError();
      }
      if (stats)
        RT_LAUNCH((render_persist_kernel<RT_REAL, true, TRAV_ORDERED, RT_EARLY_EXIT, false>), grid, block, lds2, st, sc, cam, a,
                  accum, dstats);
      else
        RT_LAUNCH((render_persist_kernel<RT_REAL, false, TRAV_ORDERED, RT_EARLY_EXIT, false>), grid, block, lds2, st, sc, cam, a,
                  accum, dstats);
      return hipGetLastError();
    }
    if (shape.trav == TRAV_REFERENCE && shape.unit_samples > 0 && use_queue64 && lds + (28u << 10) <= (160u << 10) / 3u) {
      // the reference-order walk with the wave's jobs shared (as above; RT_QUEUE=0: lane = pixel)
      if (stats)
        RT_LAUNCH((render_queue_kernel<RT_REAL, true, TRAV_REFERENCE, RT_EARLY_EXIT, true>), grid, block, lds, st, sc, cam, a, accum, dstats);
      else
        RT_LAUNCH((render_queue_kernel<RT_REAL, false, TRAV_REFERENCE, RT_EARLY_EXIT, true>), grid, block, lds, st, sc, cam, a, accum, dstats);
      return hipGetLastError();
    }
    if (shape.trav == TRAV_REFERENCE && shape.unit_samples > 0) {  // persistent form of the audit kernel (rt_api.hip: make_shape)
      if (stats)
        RT_LAUNCH((render_persist_kernel<RT_REAL, true, TRAV_REFERENCE, RT_EARLY_EXIT, true>), grid, block, lds, st, sc, cam, a,
                  accum, dstats);
      else
        RT_LAUNCH((render_persist_kernel<RT_REAL, false, TRAV_REFERENCE, RT_EARLY_EXIT, true>), grid, block, lds, st, sc, cam, a,
                  accum, dstats);
      return hipGetLastError();
    }
  }
  if (F64 || sc.has_noise) {
    if (stats)
      RT_LAUNCH((render_kernel<RT_REAL, true, RT_EARLY_EXIT, TRAV_REFERENCE, true>), grid, block, lds, st, sc,
                         cam, a, accum, dstats);
    else
      RT_LAUNCH((render_kernel<RT_REAL, false, RT_EARLY_EXIT, TRAV_REFERENCE, true>), grid, block, lds, st, sc,
                         cam, a, accum, dstats);
    return hipGetLastError();
  }
  if constexpr (!F64) {
    if (shape.unit_samples > 0 && shape.trav == TRAV_REFERENCE && !sc.has_instances) {  // queue form (rt_api.hip: make_shape)
      if (stats)
        RT_LAUNCH((render_queue_kernel<RT_REAL, true, TRAV_REFERENCE, RT_EARLY_EXIT, false, 0>), grid, block, lds, st, sc, cam, a,
                  accum, dstats);
      else
        RT_LAUNCH((render_queue_kernel<RT_REAL, false, TRAV_REFERENCE, RT_EARLY_EXIT, false, 0>), grid, block, lds, st, sc, cam, a,
                  accum, dstats);
      return hipGetLastError();
    }
    if (sc.has_instances) {
      if (stats)
        RT_LAUNCH((render_kernel<RT_REAL, true, RT_EARLY_EXIT, TRAV_REFERENCE, false, 1>), grid, block, lds, st, sc, cam, a,
                  accum, dstats);
      else
        RT_LAUNCH((render_kernel<RT_REAL, false, RT_EARLY_EXIT, TRAV_REFERENCE, false, 1>), grid, block, lds, st, sc, cam, a,
                  accum, dstats);
    } else if (stats) {
      RT_LAUNCH((render_kernel<RT_REAL, true, RT_EARLY_EXIT, TRAV_REFERENCE, false, 0>), grid, block, lds, st, sc, cam, a,
                accum, dstats);
    } else {
      RT_LAUNCH((render_kernel<RT_REAL, false, RT_EARLY_EXIT, TRAV_REFERENCE, false, 0>), grid, block, lds, st, sc, cam, a,
                accum, dstats);
    }
  }
  return hipGetLastError();
}

hipError_t RT_CAT(launch_primary_, RT_SUFFIX)(const SceneView<RT_REAL>& sc, const CameraView<RT_REAL>& cam,
                                              uint32_t width, uint32_t height, uint64_t seed, RT_REAL time,
                                              uint32_t* prim_id, RT_REAL* t, RT_REAL* normal, RT_REAL* uv,
                                              uint8_t* front_face, LaunchShape shape, hipStream_t st) {
  dim3 grid(shape.tiles, 1, 1), block(BLOCK, 1, 1);
  size_t lds = (size_t)shape.stack_depth * BLOCK * sizeof(uint32_t);
#if RT_HAS_LINEAR
  if (shape.trav == TRAV_ORDERED) {
    RT_LAUNCH((primary_kernel<RT_REAL, TRAV_ORDERED>), grid, block, lds + lds / 2, st, sc, cam, width, height, seed,
                       time, prim_id, t, normal, uv, front_face, shape.stack_depth);
    return hipGetLastError();
  }
  if (shape.trav == TRAV_LINEAR) {
    RT_LAUNCH((primary_kernel<RT_REAL, TRAV_LINEAR>), grid, block, lds, st, sc, cam, width, height, seed, time,
                       prim_id, t, normal, uv, front_face, shape.stack_depth);
    return hipGetLastError();
  }
#endif
  RT_LAUNCH((primary_kernel<RT_REAL, TRAV_REFERENCE>), grid, block, lds, st, sc, cam, width, height, seed, time,
                     prim_id, t, normal, uv, front_face, shape.stack_depth);
  return hipGetLastError();
}

#if RT_HAS_LINEAR
hipError_t launch_primary_audited_f32(const SceneView<float>& sc, const SceneView<double>& sc64, const CameraView<double>& cam64,
                                      uint32_t width, uint32_t height, uint64_t seed, double time, uint32_t* prim_id, float* t,
                                      float* normal, float* uv, uint8_t* front_face, uint8_t* unsure, LaunchShape shape,
                                      hipStream_t st) {
  dim3 grid(shape.tiles, 1, 1), block(BLOCK, 1, 1);
  // the audit walks the reference's tree (4 B per level and lane); the ordered walk needs 6 B per level of its own tree
  const size_t lds_ref = (size_t)shape.ref_stack_depth * BLOCK * sizeof(uint32_t);
  const size_t lds_own = (size_t)shape.stack_depth * BLOCK * sizeof(uint32_t);
  if (shape.trav == TRAV_ORDERED) {
    RT_LAUNCH((primary_f32_kernel<TRAV_ORDERED>), grid, block, std::max(lds_ref, lds_own + lds_own / 2), st, sc, sc64,
                       cam64, width, height, seed, time, prim_id, t, normal, uv, front_face, unsure, shape.stack_depth);
  } else if (shape.trav == TRAV_LINEAR) {
    RT_LAUNCH((primary_f32_kernel<TRAV_LINEAR>), grid, block, lds_ref, st, sc, sc64, cam64, width, height, seed, time,
                       prim_id, t, normal, uv, front_face, unsure, shape.stack_depth);
  } else {
    RT_LAUNCH((primary_f32_kernel<TRAV_REFERENCE>), grid, block, lds_ref, st, sc, sc64, cam64, width, height, seed,
                       time, prim_id, t, normal, uv, front_face, unsure, shape.stack_depth);
  }
  return hipGetLastError();
}
#endif
#if RT_HAS_LINEAR
hipError_t launch_wf_init(const CameraView<float>& cam, const WfArgs& a, const WfPool& pool, WfCounters* ctr, hipStream_t st) {
  RT_LAUNCH(wf_init_kernel, dim3((pool.n_slots + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, cam, a, pool, ctr);
  return hipGetLastError();
}
hipError_t launch_wf_iteration(const SceneView<float>& sc, const CameraView<float>& cam, const WfArgs& a, const WfPool& pool,
                               WfCounters* ctr, uint32_t parity, float* accum, DeviceStats* dstats, bool stats,
                               uint32_t trace_blocks, hipStream_t st) {
  const size_t lds = (size_t)a.stack_depth * BLOCK * sizeof(uint32_t);
  const size_t lds2 = lds + lds / 2;  // 32-bit refs + 16-bit entry distances
  const dim3 shade_grid((pool.n_slots + BLOCK - 1) / BLOCK);
  if (stats) {
    RT_LAUNCH((wf_trace_kernel<true>), dim3(trace_blocks), dim3(BLOCK), lds2, st, sc, a, pool, ctr, parity, dstats);
    RT_LAUNCH((wf_shade_kernel<true>), shade_grid, dim3(BLOCK), 0, st, sc, cam, a, pool, ctr, parity, accum, dstats);
  } else {
    RT_LAUNCH((wf_trace_kernel<false>), dim3(trace_blocks), dim3(BLOCK), lds2, st, sc, a, pool, ctr, parity, dstats);
    RT_LAUNCH((wf_shade_kernel<false>), shade_grid, dim3(BLOCK), 0, st, sc, cam, a, pool, ctr, parity, accum, dstats);
  }
  return hipGetLastError();
}
hipError_t wf_trace_blocks_per_cu(size_t lds_bytes, int* out) {
  return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, wf_trace_kernel<false>, BLOCK, lds_bytes);
}
#endif
#ifdef RT_DEFINE_F64_FIX
hipError_t launch_tonemap_f64(const double* accum, uint64_t n, double spp, uint8_t* out, hipStream_t st) {
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks == 0) blocks = 1;
  RT_LAUNCH(tonemap_f64_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, accum, n, spp, out);
  return hipGetLastError();
}
hipError_t launch_primary_fix_f64(const SceneView<double>& sc, const CameraView<double>& cam, uint32_t width, uint32_t height,
                                  uint64_t seed, double time, const uint8_t* unsure, uint32_t* prim_id, float* t, float* normal,
                                  float* uv, uint8_t* front_face, uint32_t ref_stack_depth, hipStream_t st) {
  const uint32_t tiles = ((width + 15) / 16) * ((height + 15) / 16);
  RT_LAUNCH(primary_fix_kernel, dim3(tiles), dim3(BLOCK), (size_t)ref_stack_depth * BLOCK * sizeof(uint32_t), st, sc, cam,
                     width, height, seed, time, unsure, prim_id, t, normal, uv, front_face);
  return hipGetLastError();
}
#endif

hipError_t RT_CAT(launch_kat_sphere_uv_, RT_SUFFIX)(uint32_t n, const double* p, double* uv, hipStream_t st) {
  RT_LAUNCH((kat_sphere_uv_kernel<RT_REAL>), dim3((n + 63) / 64), dim3(64), 0, st, n, p, uv);
  return hipGetLastError();
}
hipError_t RT_CAT(launch_kat_rotate_, RT_SUFFIX)(uint32_t n, const double* angle, const double* axis,
                                                 const double* v, double* out, hipStream_t st) {
  RT_LAUNCH((kat_rotate_kernel<RT_REAL>), dim3((n + 63) / 64), dim3(64), 0, st, n, angle, axis, v, out);
  return hipGetLastError();
}

#ifdef RT_DEFINE_LIBM_KAT  // (in the f64 translation unit: built like the audit kernels, contraction off)
hipError_t launch_kat_libm(uint32_t n, uint32_t op, const double* a, const double* b, double* out, hipStream_t st) {
  RT_LAUNCH(kat_libm_kernel, dim3((n + 63) / 64), dim3(64), 0, st, n, op, a, b, out);
  return hipGetLastError();
}
#endif

#ifdef RT_DEFINE_COMMON
hipError_t launch_tonemap(const float* accum, uint64_t n, float inv_spp, uint8_t* out, hipStream_t st) {
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks == 0) blocks = 1;
  RT_LAUNCH(tonemap_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, accum, n, inv_spp, out);
  return hipGetLastError();
}
hipError_t launch_zero(void* dst, size_t bytes, hipStream_t st) {  // bytes: a multiple of 4
  const uint64_t n = bytes / 4;
  if (n == 0) return hipSuccess;
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  RT_LAUNCH(zero_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, reinterpret_cast<uint32_t*>(dst), n);
  return hipGetLastError();
}
hipError_t launch_accum_add(float* dst, const float* src, uint64_t n, uint32_t n_src, hipStream_t st) {
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks == 0) blocks = 1;
  RT_LAUNCH(accum_add_kernel<float>, dim3((uint32_t)blocks), dim3(256), 0, st, dst, src, n, n_src);
  return hipGetLastError();
}
hipError_t launch_accum_add_f64(double* dst, const double* src, uint64_t n, uint32_t n_src, hipStream_t st) {
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks == 0) blocks = 1;
  RT_LAUNCH(accum_add_kernel<double>, dim3((uint32_t)blocks), dim3(256), 0, st, dst, src, n, n_src);
  return hipGetLastError();
}
hipError_t launch_kat_philox(uint32_t n, const uint32_t* ctr, const uint32_t* key, int path_stream, uint32_t* out,
                             hipStream_t st) {
  RT_LAUNCH(kat_philox_kernel, dim3((n + 63) / 64), dim3(64), 0, st, n, ctr, key, path_stream, out);
  return hipGetLastError();
}
#endif

}  // namespace rt
