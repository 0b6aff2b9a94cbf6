// The exact-shape instances of k_curve_chi (K and M compile-time: kernels_curve.hip, "exact-shape instances") as a translation unit
// of their own, so that they compile next to the general instances instead of after them.  The diagnostic timeline build
// (-DBFMMM_TIMELINE) keeps everything in kernels_curve.hip: its per-workgroup trace arrays are device globals of that unit.
#ifndef BFMMM_TIMELINE
#define BFMMM_CURVE_EXACT_TU 1
#include "kernels_curve.hip"
#endif
