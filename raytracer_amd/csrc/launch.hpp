// Host-visible launchers of the gfx950 kernels, one set per device precision.  rt_f32.hip
// (production, contraction on) and rt_f64.hip (audit, -ffp-contract=off) each instantiate
// the templates in device/kernels.hpp; rt_api.hip calls them through these declarations.
#pragma once
#include <hip/hip_runtime.h>

#include "device/scene_view.hpp"
#include "device/wf_types.hpp"

namespace rt {

struct RenderArgs;

struct LaunchShape {
  uint32_t tiles;        // 16x16 pixel tiles = workgroups along grid.x
  uint32_t splits;       // sample splits along grid.y (atomic accumulation when > 1)
  uint32_t stack_depth;  // LDS traversal-stack entries per lane
  uint32_t ref_stack_depth = 1;  // ... of the reference's tree (the audit walk of rt_primary_hits always walks that one)
  uint32_t refill_eighths = 2;  // ordered walk: see RenderArgs::refill_eighths
  uint32_t leaf_shift = 2;      // ordered walk: see RenderArgs::leaf_shift
  uint32_t direct_accum = 0;    // units kernel: see RenderArgs::direct_accum
  uint32_t xcd_chunk = 0;       // refill kernel: see RenderArgs::xcd_chunk
  uint32_t tile_band = 1;       // refill kernel: see RenderArgs::tile_band
  // leaf-list scenes, persistent kernel (render_persist_kernel): 0 = off
  uint32_t unit_samples = 0, wave_tiles_x = 0, wave_tiles = 0, total_units = 0;
  uint32_t unit_prefetch = 0;  // render_persist_kernel: units a wave draws ahead of need per visit of the global counter
  int trav;              // 0 reference-order BVH walk, 1 leaf list, 2 ordered BVH walk (f32 only for 1, 2)
};

#define RT_DECLARE_LAUNCHERS(SUFFIX, REAL)                                                                  \
  hipError_t launch_render_##SUFFIX(const SceneView<REAL>& sc, const CameraView<REAL>& cam, uint32_t width, \
                                    uint32_t height, uint32_t sample_begin, uint32_t n_samples,             \
                                    uint32_t max_depth, uint64_t seed, int accumulate, bool stats,          \
                                    REAL* accum, DeviceStats* dstats, LaunchShape shape, hipStream_t st);   \
  hipError_t launch_primary_##SUFFIX(const SceneView<REAL>& sc, const CameraView<REAL>& cam, uint32_t width, \
                                     uint32_t height, uint64_t seed, REAL time, uint32_t* prim_id, REAL* t, \
                                     REAL* normal, REAL* uv, uint8_t* front_face, LaunchShape shape,        \
                                     hipStream_t st);                                                       \
  hipError_t launch_kat_sphere_uv_##SUFFIX(uint32_t n, const double* p, double* uv, hipStream_t st);        \
  hipError_t launch_kat_rotate_##SUFFIX(uint32_t n, const double* angle, const double* axis,                \
                                        const double* v, double* out, hipStream_t st);

RT_DECLARE_LAUNCHERS(f32, float)
RT_DECLARE_LAUNCHERS(f64, double)

// rt_primary_hits in f32: product walk + error-tracked audit (rt_f32.hip), then the f64 re-decision of the marked pixels
// (rt_f64.hip, contraction off)
hipError_t launch_primary_audited_f32(const SceneView<float>& sc, const SceneView<double>& sc64, const CameraView<double>& cam64,
                                      uint32_t width, uint32_t height, uint64_t seed, double time, uint32_t* prim_id, float* t,
                                      float* normal, float* uv, uint8_t* front_face, uint8_t* unsure, LaunchShape shape,
                                      hipStream_t st);
hipError_t launch_primary_fix_f64(const SceneView<double>& sc, const CameraView<double>& cam, uint32_t width, uint32_t height,
                                  uint64_t seed, double time, const uint8_t* unsure, uint32_t* prim_id, float* t, float* normal,
                                  float* uv, uint8_t* front_face, uint32_t ref_stack_depth, hipStream_t st);
// wavefront path (device/wavefront.hpp, rt_f32.hip)
hipError_t launch_wf_init(const CameraView<float>& cam, const WfArgs& a, const WfPool& pool, WfCounters* ctr, hipStream_t st);
hipError_t launch_wf_iteration(const SceneView<float>& sc, const CameraView<float>& cam, const WfArgs& a, const WfPool& pool,
                               WfCounters* ctr, uint32_t parity, float* accum, DeviceStats* dstats, bool stats,
                               uint32_t trace_blocks, hipStream_t st);
hipError_t wf_trace_blocks_per_cu(size_t lds_bytes, int* out);
hipError_t launch_tonemap(const float* accum, uint64_t n, float inv_spp, uint8_t* out, hipStream_t st);
hipError_t launch_tonemap_f64(const double* accum, uint64_t n, double spp, uint8_t* out, hipStream_t st);
hipError_t launch_zero(void* dst, size_t bytes, hipStream_t st);
hipError_t launch_accum_add(float* dst, const float* src, uint64_t n, uint32_t n_src, hipStream_t st);
hipError_t launch_accum_add_f64(double* dst, const double* src, uint64_t n, uint32_t n_src, hipStream_t st);
hipError_t launch_kat_libm(uint32_t n, uint32_t op, const double* a, const double* b, double* out, hipStream_t st);
hipError_t launch_kat_philox(uint32_t n, const uint32_t* ctr, const uint32_t* key, int path_stream, uint32_t* out,
                             hipStream_t st);

}  // namespace rt
