// Production precision: f32 kernels for gfx950 (-ffp-contract=on: fused per source expression, see vec.hpp).
#define RT_SUFFIX f32
#define RT_REAL float
#define RT_HAS_LINEAR 1
#ifndef RT_EARLY_EXIT
#define RT_EARLY_EXIT true
#endif
#define RT_DEFINE_COMMON 1
#include "launch_impl.hpp"
