// ofl_warp_wide.hip -- the staged backward-warp kernels once more, on 64 x 16 output tiles (256-thread blocks, 52 KB of LDS, 3 blocks =
// 12 waves per CU).  Taken from here (the ofl_wide_launch_* functions, called by the launchers in ofl_kernels.hip): the ROW-TABLE kernels
// (warp_bwd_rows_kernel and its instantiations -- every lean launch of the bench runs on them -- and the channel loop's ROWS
// instantiation), and the four-tile column / channel-loop kernels on the sheared rectangle for large launches that are not lean.
// The tile shape is a set of file-scope constants of ofl_kernels.hip; compiling that file a
// second time with another shape costs nothing at run time and keeps the helpers free of a template parameter that every kernel
// but one would set to the same value.  Measured (profiles/r4_warp_16waves.txt): apply 't' -1.9 % at sigma 8, the pair kernel of
// mode 3 +5 % -- hence per kernel.  Same device code per pixel: bit-identical to the 32 x 16 kernels (tests compare them).
#define OFL_WIDE_TU 1
#define OFL_LDS_NT 256
#define OFL_LDS_TWQ 16
#define OFL_LDS_BYTES 53248
#include "ofl_kernels.hip"
