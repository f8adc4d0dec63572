// swb_wide.hip -- the step kernels of images wider than 64 columns (two or four output columns per lane).
//
// A translation unit of its own because it is compiled with another instruction scheduler than swb.hip:
// these kernels run at 2-3 waves per SIMD and gain 5 % from LLVM's ILP-first strategy
// (-amdgpu-sched-strategy=iterative-ilp), while the 64-column kernels at 4 waves per SIMD are as fast
// with the default strategy and spill 11 instead of 69 vector registers with it (scratch traffic that
// showed up as 80 MB of extra HBM writes per launch).  See spriteworld_amd/build.py for the flags.
#define SWB_WIDE_TU 1
#include "swb_kernels.hip.inc"
#include "swb_pow.hip.inc"

template __global__ void swb_step_kernel<4, 2, 8>(const swb_params);
template __global__ void swb_step_kernel<10, 2, 8>(const swb_params);
template __global__ void swb_step_kernel<20, 2, 6>(const swb_params);
template __global__ void swb_step_kernel<20, 2, 8>(const swb_params);
template __global__ void swb_step_kernel<20, 4, 8>(const swb_params);
// ... and their builds with live sprite overrides (swb_set_sprite_attr)
template __global__ void swb_step_kernel<4, 2, 8, true>(const swb_params);
template __global__ void swb_step_kernel<10, 2, 8, true>(const swb_params);
template __global__ void swb_step_kernel<20, 2, 6, true>(const swb_params);
template __global__ void swb_step_kernel<20, 2, 8, true>(const swb_params);
template __global__ void swb_step_kernel<20, 4, 8, true>(const swb_params);
