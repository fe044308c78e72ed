// Microbenchmark for the megakernel-vs-wavefront-queue question: what does it cost a
// wavefront design merely to STREAM its path state between bounce passes?  One pass reads a compacted array
// of path states (SoA, 12 or 16 dwords per path), does nothing with them but a survival test that matches
// Cornell's path-length statistics (59 % of paths continue), and writes the survivors, compacted (one
// atomic per 1024-path workgroup), to the next array — the floor under any per-bounce kernel, with the
// tracing, shading, hit records and material queues of a real wavefront tracer all left out.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/state_stream.hip -o tools/ubench/state_stream
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

template <int DW>
__global__ void pass(const float* __restrict__ in, float* __restrict__ out, uint32_t n_in, uint32_t cap,
                     uint32_t* __restrict__ n_out, uint32_t salt) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  float s[DW];
  bool live = false;
  if (i < n_in) {
#pragma unroll
    for (int k = 0; k < DW; ++k) s[k] = in[(size_t)k * cap + i];  // SoA: coalesced per component
    // survival: a hash of the state's "pixel/sample" words, 59 % continue (Cornell: 2.43 rays per sample)
    uint32_t h = (__float_as_uint(s[DW - 1]) ^ salt) * 0x9E3779B1u;
    h ^= h >> 15;
    live = (h % 100u) < 59u;
#pragma unroll
    for (int k = 0; k < DW - 1; ++k) s[k] = s[k] * 1.0009765625f + 0.5f;  // stand-in for "the state changed"
  }
  // workgroup-aggregated compaction: one global atomic per 1024 paths (a single queue counter serialises:
  // one atomic per WAVE measured 4x slower than this)
  __shared__ uint32_t s_wave_n[16], s_base;
  const unsigned long long m = __ballot(live);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (lane == 0) s_wave_n[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (uint32_t w = 0; w < blockDim.x / 64; ++w) {
      const uint32_t c = s_wave_n[w];
      s_wave_n[w] = tot;
      tot += c;
    }
    s_base = tot ? atomicAdd(n_out, tot) : 0u;
  }
  __syncthreads();
  if (live) {
    const uint32_t o = s_base + s_wave_n[wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
#pragma unroll
    for (int k = 0; k < DW; ++k) out[(size_t)k * cap + o] = s[k];
  }
}

template <int DW> int run(uint32_t pool, double rays_per_render) {
  float *a = nullptr, *b = nullptr;
  uint32_t* cnt = nullptr;
  CHECK(hipMalloc(&a, (size_t)pool * DW * 4));
  CHECK(hipMalloc(&b, (size_t)pool * DW * 4));
  CHECK(hipMalloc(&cnt, 64 * 4));
  std::vector<float> init((size_t)pool * DW);
  for (size_t i = 0; i < init.size(); ++i) init[i] = (float)(i * 2654435761u % 1000003u);
  CHECK(hipMemcpy(a, init.data(), init.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  double total_ms = 0, total_rays = 0, total_bytes = 0;
  for (int rep = 0; rep < 3; ++rep) {  // rep 0 = warm-up
    uint32_t n = pool;
    float *src = a, *dst = b;
    double ms_sum = 0, rays = 0, bytes = 0;
    for (int bounce = 0; bounce < 50 && n > 0; ++bounce) {
      CHECK(hipMemsetAsync(cnt, 0, 4, nullptr));
      CHECK(hipEventRecord(e0, nullptr));
      hipLaunchKernelGGL((pass<DW>), dim3((n + 1023) / 1024), dim3(1024), 0, nullptr, src, dst, n, pool, cnt,
                         (uint32_t)(bounce * 7919 + rep));
      CHECK(hipEventRecord(e1, nullptr));
      uint32_t n_next = 0;
      CHECK(hipMemcpy(&n_next, cnt, 4, hipMemcpyDeviceToHost));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      ms_sum += ms;
      rays += n;
      bytes += (double)n * DW * 4 + (double)n_next * DW * 4;
      n = n_next;
      float* t = src;
      src = dst;
      dst = t;
    }
    if (rep > 0) {
      total_ms += ms_sum;
      total_rays += rays;
      total_bytes += bytes;
    }
  }
  const double ms_per_ray = total_ms / total_rays;
  std::printf("state %2d dwords (%3d B), pool %9u paths (%6.1f MB x2): %.2f rays per path, %7.1f GB/s streamed, "
              "%.1f ms of pure state streaming per %.3g-ray render\n",
              DW, DW * 4, pool, (double)pool * DW * 4 / 1e6, total_rays / (2.0 * pool), total_bytes / total_ms * 1e-6,
              ms_per_ray * rays_per_render, rays_per_render);
  CHECK(hipFree(a));
  CHECK(hipFree(b));
  CHECK(hipFree(cnt));
  return 0;
}

int main() {
  const double rays_per_render = 1.0438e10;  // Cornell 1024^2 x 4096 spp x 2.43 rays per sample
  for (uint32_t pool : {1u << 20, 1u << 22, 1u << 24}) {
    if (run<12>(pool, rays_per_render)) return 1;
    if (run<16>(pool, rays_per_render)) return 1;
  }
  return 0;
}
