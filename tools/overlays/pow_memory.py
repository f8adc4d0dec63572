"""The libm pow's coefficients read from __constant__ memory instead of folded into 64-bit literals (experiment builds only).
As literals the compiler hoists them in front of the sub-task loop and keeps them in vector registers through the task evaluation:
the cover kernels of 64-px canvases (80 registers at six waves per SIMD) spill three of them -- 28 bytes of scratch per lane,
12.6 MB of HBM writes per launch of 8192 environments.  As __constant__ data: 79 registers, no scratch, and 92 instead of 96
registers in the 320-px kernel -- but 29 more spilled scalars, and measured (profiles/r06_pow_constants_ab.txt, five interleaved
rounds on one box) +0.8 % on the headline's cover kernel and +1.3 % / -0.8 % (two sessions) at anti_aliasing = 1.  Not adopted."""


def apply(files, arg, replace_once):
  del arg
  replace_once(files, 'swb_pow.hip.inc', '#define SWB_POW_CONST __device__ __constant__ const\n', '#define SWB_POW_CONST __device__ __constant__\n')
