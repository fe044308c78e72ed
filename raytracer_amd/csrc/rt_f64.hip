// Audit precision: the same kernels in f64, built with -ffp-contract=off so that they are
// draw-for-draw and branch-for-branch comparable with the CPU oracle.  Not a production path.
#define RT_SUFFIX f64
#define RT_REAL double
#define RT_HAS_LINEAR 0
#define RT_EARLY_EXIT false
#define RT_DEFINE_F64_FIX 1
#define RT_DEFINE_LIBM_KAT 1
#include "launch_impl.hpp"
