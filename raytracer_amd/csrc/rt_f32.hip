// Production precision: f32 kernels for gfx950 (-ffp-contract=on: fused per source expression, see vec.hpp).
#define RT_SUFFIX f32
#define RT_REAL float
#define RT_HAS_LINEAR 1
#ifndef RT_EARLY_EXIT
#define RT_EARLY_EXIT true
#endif
#define RT_DEFINE_COMMON 1
#include "launch_impl.hpp"

#ifdef RT_BLOCK_TRACE
extern "C" int rt_debug_block_trace(uint32_t* out, uint32_t n_waves) {
  if (n_waves > rt::BLOCK_TRACE_MAX) n_waves = rt::BLOCK_TRACE_MAX;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(rt::g_block_trace), (size_t)n_waves * 16) == hipSuccess ? 0 : 1;
}
#endif
