"""Upper bound of what the filing atomics cost on the cover wave's critical path (experiment; frames unchanged, dispatch order
degraded; batches of one band only: band_tasks files several tasks per list): every (environment, column group) is filed in bucket 0 of its shard at a position derived from its index -- no
returning atomic at the end of the wave.  Run with SWB_NO_COVER_ORDER=1 (no second filing) on both sides of the comparison."""


def apply(files, arg, replace_once):
  k = 'swb_kernels.hip.inc'
  replace_once(files, k, '    const int key0 = cost_key(task_cost, (uint32_t)__builtin_amdgcn_readlane((int)filing_words_l, 0), (uint32_t)__builtin_amdgcn_readlane((int)filing_words_l, 1));\n',
               '    const int key0 = 0;\n')
  replace_once(files, k, '    if (file0) pos0 = atomicAdd(&p.cost_cnt[cost_row0(p.parity, sh) + key0], 1u);\n',
               '    if (file0) {\n'
               '      pos0 = (uint32_t)(env >> 3) * (uint32_t)p.ncg + (uint32_t)l;\n'
               '      if ((env >> 3) == 0 && l == 0) p.cost_cnt[cost_row0(p.parity, sh)] = (uint32_t)(p.ncg * ((p.N - sh + 7) >> 3));\n'
               '    }\n')
