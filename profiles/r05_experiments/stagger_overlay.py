"""The first round of cover waves staggered (experiment; results unchanged): the waves that share a SIMD begin k x ARG x 64 cycles
apart (k = their slot: the dispatcher deals consecutive blocks over the 1024 SIMDs), instead of all at once -- in lock step
they meet at the memory system (four dependent rounds of loads) and at the LDS atomic unit (P1b) together."""


def apply(files, arg, replace_once):
  k = 'swb_kernels.hip.inc'
  d = int(arg or 40)
  replace_once(files, k, '''  const uint32_t cycles0 = (uint32_t)__builtin_amdgcn_s_memtime();
  // the environment of this block: its own index, or (cover_order)''', '''  {
    const int slot = (int)(blockIdx.x >> 10);
    if (slot >= 1 && slot < 5) {
      for (int i = 0; i < slot; ++i) __builtin_amdgcn_s_sleep(%d);
    }
  }
  const uint32_t cycles0 = (uint32_t)__builtin_amdgcn_s_memtime();
  // the environment of this block: its own index, or (cover_order)''' % min(d, 127))
