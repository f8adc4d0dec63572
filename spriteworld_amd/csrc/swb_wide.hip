// swb_wide.hip -- the cover kernels of canvases wider than 320 px (20 words of 32 pixels per canvas row).
//
// A translation unit of its own because it is compiled with another instruction scheduler than swb.hip
// (-amdgpu-sched-strategy=iterative-ilp, see spriteworld_amd/build.py for the flags and the measurements).
#define SWB_WIDE_TU 1
#include "swb_kernels.hip.inc"
#include "swb_pow.hip.inc"

template __global__ void swb_cover_kernel<20, false>(const swb_params);
// ... and the build with live sprite overrides (swb_set_sprite_attr)
template __global__ void swb_cover_kernel<20, true>(const swb_params);
