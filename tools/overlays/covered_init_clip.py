"""P2 word loop without its per-word canvas-width clip (experiment; results unchanged): the pixels beyond the canvas' last
column start out as "already covered" (covered[] is initialised with them once per batch), so `vis = cov & ~covered[w]` drops
them by itself and the two wave-uniform conditionals per word -- `if (w == last_w) cov &= last_mask; if (w > last_w) cov = 0u`,
which the compiler turns into 3 loop-invariant scalar values PER WORD, 30 of them at NW = 10, spilled and re-read with
v_readlane + s_nop inside the loop (ISA of round 4) -- disappear."""


def apply(files, arg, replace_once):
  k = 'swb_kernels.hip.inc'
  replace_once(files, k, '''  uint32_t covered[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) covered[w] = 0u;
  rs.cnt = 0; rs.s0 = rs.s1 = rs.s2 = 0u;
  const uint32_t last_mask = (p.Wc & 31) ? ((1u << (p.Wc & 31)) - 1u) : 0xffffffffu;
  const int last_w = (p.Wc - 1) >> 5;
''', '''  const uint32_t last_mask = (p.Wc & 31) ? ((1u << (p.Wc & 31)) - 1u) : 0xffffffffu;
  const int last_w = (p.Wc - 1) >> 5;
  uint32_t covered[NW];                    // (pixels beyond the canvas' last column: covered from the start, never visible)
#pragma unroll
  for (int w = 0; w < NW; ++w) covered[w] = (w < last_w) ? 0u : ((w == last_w) ? ~last_mask : 0xffffffffu);
  rs.cnt = 0; rs.s0 = rs.s1 = rs.s2 = 0u;
''')
  replace_once(files, k, '''        uint32_t cov = t | px;
        if (w == last_w) cov &= last_mask;
        if (w > last_w) cov = 0u;
        const uint32_t vis = cov & ~covered[w];
''', '''        const uint32_t cov = t | px;
        const uint32_t vis = cov & ~covered[w];
''')
