"""Cover launch in cost order, dealt over the SIMDs in alternating directions (experiment; results unchanged): within the first
round of waves, tier t of 128 consecutive XCD-local blocks (one per SIMD of the XCD: the dispatcher deals consecutive blocks of an
XCD over its 32 CUs, then over the SIMDs of each) takes its 128 environments in descending order of cost when t is even and
in ascending order when t is odd -- plain descending order gives the first SIMD the heaviest environment of EVERY tier and the
last SIMD the lightest.  ARG: tiers dealt that way (default 5 = the first round at 5 waves per SIMD)."""


def apply(files, arg, replace_once):
  k = 'swb_kernels.hip.inc'
  tiers = int(arg or 5)
  replace_once(files, k, '''    env = cost_ordered_entry(p.cost_cnt + cost_row1(ph_prev, sh),
                             as_const(p.ccost_list) + (size_t)(ph_prev * SWB_COST_SHARDS + sh) * SWB_COST_BUCKETS * cap, cap,
                             (int)(blockIdx.x >> 3), &bucket);''', '''    int rank = (int)(blockIdx.x >> 3);
    {
      const int n_sh = (p.N - sh + SWB_COST_SHARDS - 1) / SWB_COST_SHARDS, tier = rank >> 7;
      if (tier < %d && (tier & 1) && ((tier + 1) << 7) <= n_sh) rank = (tier << 7) + 127 - (rank & 127);
    }
    env = cost_ordered_entry(p.cost_cnt + cost_row1(ph_prev, sh),
                             as_const(p.ccost_list) + (size_t)(ph_prev * SWB_COST_SHARDS + sh) * SWB_COST_BUCKETS * cap, cap,
                             rank, &bucket);''' % tiers)
