// ofl_splat_gather.hip -- the gather splat's round-6 kernel (splat_gather2_kernel: the same in-order sums as splat_gather_kernel on a
// register / LDS diet, three 512-thread blocks per CU instead of two) as a translation unit of its own: ofl_kernels.hip holds the
// kernel next to the round-5 one it replaces (they share every helper), this file instantiates it, so that the two compile side
// by side.  Exports one function to the other translation units: ofl_splat_launch_gather_diet.
#define OFL_SPLAT_TU 1
#include "ofl_kernels.hip"
