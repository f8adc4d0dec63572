"""P2: the edge records of the NEXT sprite of a batch fetched while the current sprite's pass runs (experiment; results
unchanged).  In the edge-lane form every lane reads its edge's 16-byte record from LDS at the head of a pass and waits for
it (~130-200 cycles of a pass of ~4 100, wave timelines of round 4); here lane l issues the read for the next pass -- sprite
parameters of the next sprite through the two v_readlane the pass would do anyway -- right after taking over its own.

Measured (round 4, profiles/r04_experiments/queued_overlays_ab.txt): on top of scatter_unswitch it takes the whole gain back
(cover 0.0836 -> 0.0862 ms at 8192 envs, 0.0392 -> 0.0398 at 1024): the record held across the pass and the second set of
sprite parameters cost more (98 instead of 89 spilled SGPRs) than the LDS round trip they hide.  Kept as a record."""


def apply(files, arg, replace_once):
  k = 'swb_kernels.hip.inc'
  replace_once(files, k, '''  while (sm) {
    const int s = 63 - __builtin_clzll(sm);
    sm &= ~(1ull << s);
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)sp_a, s);
    const uint32_t e0m = (uint32_t)__builtin_amdgcn_readlane(sp_e0, s);      // first edge slot | ceil(2^16 / G) << 16
''', '''  // (edge-lane form) the lane's record of sprite s2: edge (l / G) of it, zeros beyond its edges
  uint32_t a_next = 0u, e0m_next = 0u;
  edge_rec ed_next;
  ed_next.x0 = ed_next.y0 = ed_next.y1 = 0; ed_next.dx = 0.f; ed_next.xtop = ed_next.xbot = SWB_NO_REPL; ed_next.pad = 0;
  auto prefetch = [&](unsigned long long left) __attribute__((always_inline)) {
    const int s2 = left ? 63 - __builtin_clzll(left) : 0;
    a_next = (uint32_t)__builtin_amdgcn_readlane((int)sp_a, s2);
    e0m_next = (uint32_t)__builtin_amdgcn_readlane(sp_e0, s2);
    const int e2 = (int)(((uint32_t)l * (e0m_next >> 16)) >> 16);
    edge_rec r;
    r.x0 = r.y0 = r.y1 = 0; r.dx = 0.f; r.xtop = r.xbot = SWB_NO_REPL; r.pad = 0;
    if (left != 0ull && e2 < (int)((a_next >> 10) & 127u)) r = edges[(int)(e0m_next & 0xffffu) + e2];
    ed_next = r;
  };
  prefetch(sm);
  while (sm) {
    const int s = 63 - __builtin_clzll(sm);
    sm &= ~(1ull << s);
    const uint32_t a = a_next;
    const uint32_t e0m = e0m_next;                                           // first edge slot | ceil(2^16 / G) << 16
    const edge_rec ed_cur = ed_next;
    prefetch(sm);                                                            // the next pass's records, in flight during this one
''')
  replace_once(files, k, '''        const bool have = e < ne;
        edge_rec ed;
        ed.x0 = ed.y0 = ed.y1 = 0; ed.dx = 0.f; ed.xtop = ed.xbot = SWB_NO_REPL; ed.pad = 0;
        if (have) ed = edges[e0 + e];
''', '''        const bool have = e < ne;
        const edge_rec ed = ed_cur;
''')
