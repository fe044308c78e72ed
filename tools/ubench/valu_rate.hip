// Microbenchmark: issue rate of scalar-f32 FMA vs packed-f32 FMA vs f64 FMA vs v_mad_u64_u32 on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float float2v __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096, UNROLL = 16;

__global__ void k_fma32(float* out, float a, float b) {
  float x[UNROLL];
  for (int i = 0; i < UNROLL; ++i) x[i] = threadIdx.x + i;
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) x[i] = __builtin_fmaf(x[i], a, b);
  float s = 0;
  for (int i = 0; i < UNROLL; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_pkfma32(float* out, float a, float b) {
  float2v x[UNROLL];
  const float2v av = {a, a}, bv = {b, b + 1};
  for (int i = 0; i < UNROLL; ++i) x[i] = float2v{(float)threadIdx.x + i, (float)i};
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) x[i] = __builtin_elementwise_fma(x[i], av, bv);
  float s = 0;
  for (int i = 0; i < UNROLL; ++i) s += x[i].x + x[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma64(float* out, double a, double b) {
  double x[UNROLL];
  for (int i = 0; i < UNROLL; ++i) x[i] = threadIdx.x + i;
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) x[i] = __builtin_fma(x[i], a, b);
  double s = 0;
  for (int i = 0; i < UNROLL; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}
__global__ void k_mad64(float* out, uint32_t a) {
  uint64_t x[UNROLL];
  for (int i = 0; i < UNROLL; ++i) x[i] = threadIdx.x + i;
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) x[i] = (uint64_t)(uint32_t)x[i] * a + (x[i] >> 32);
  uint64_t s = 0;
  for (int i = 0; i < UNROLL; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}
template <typename F> double run(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  const int blocks = 256 * 8, threads = 512;  // 8 workgroups of 8 waves per CU: 16 waves per SIMD
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  const double n = (double)blocks * threads * ITERS * UNROLL;
  double ms;
  ms = run([&] { hipLaunchKernelGGL(k_fma32, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f); });
  printf("v_fma_f32      : %.3f ms  %.1f G lane-instr/s  %.1f TFLOP/s\n", ms, n / ms / 1e6, 2 * n / ms / 1e9);
  ms = run([&] { hipLaunchKernelGGL(k_pkfma32, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f, 0.5f); });
  printf("v_pk_fma_f32   : %.3f ms  %.1f G lane-instr/s  %.1f TFLOP/s\n", ms, n / ms / 1e6, 4 * n / ms / 1e9);
  ms = run([&] { hipLaunchKernelGGL(k_fma64, dim3(blocks), dim3(threads), 0, 0, out, 1.0001, 0.5); });
  printf("v_fma_f64      : %.3f ms  %.1f G lane-instr/s  %.1f TFLOP/s\n", ms, n / ms / 1e6, 2 * n / ms / 1e9);
  ms = run([&] { hipLaunchKernelGGL(k_mad64, dim3(blocks), dim3(threads), 0, 0, out, 0xD2511F53u); });
  printf("v_mad_u64_u32  : %.3f ms  %.1f G lane-instr/s\n", ms, n / ms / 1e6);
  hipFree(out);
  return 0;
}
