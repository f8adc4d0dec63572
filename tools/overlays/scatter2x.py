"""P2 edge-lane scatter with TWO rows per lane and iteration (experiment; results unchanged): the crossings of rows yy and
yy + G of a lane's edge are independent chains of ~35 dependent vector instructions each; in one basic block the scheduler
interleaves them.  Meant for what the wave timelines show (profiles/r04_timeline_1024.json, trace_p2): a cover wave alone on
its SIMD spends 44 % of its coverage cycles in the scatter at ~790 cycles per iteration, and the headline launch costs two
rounds of such waves.  Same instruction count when an even number of iterations is needed; the odd one runs the single form."""


def apply(files, arg, replace_once):
  k = 'swb_kernels.hip.inc'
  # a K-crossing form of scatter_crossing2 (K = 2): the uniform branches cover both crossings, the arithmetic is the same
  replace_once(files, k, '''template <int NW>
__device__ __forceinline__ void coverage_batch2(''', '''template <int NW>
__device__ __forceinline__ void scatter_crossing2x2(wave_lds<NW>* L, int xb, int npx, int rla, int rlb, bool acta, bool actb, int ya,
                                                    int yb2, float x0f, int y0, float dx, float xtop, float xbot, bool any_repl,
                                                    bool maybe_neg, int eymin, int eymax, int symax) {
  float xa = __fadd_rn(__fmul_rn((float)(ya - y0), dx), x0f), xb_ = __fadd_rn(__fmul_rn((float)(yb2 - y0), dx), x0f);
  const bool dupa = (ya == eymax) && (ya < symax), dupb = (yb2 == eymax) && (yb2 < symax);
  if (any_repl) {
    const bool repa = !dupa && dx != 0.0f, repb = !dupb && dx != 0.0f;
    if (repa && ya == eymin) { if (xtop == xtop) xa = xtop; }
    else if (repa && ya == eymax) { if (xbot == xbot) xa = xbot; }
    if (repb && yb2 == eymin) { if (xtop == xtop) xb_ = xtop; }
    else if (repb && yb2 == eymax) { if (xbot == xbot) xb_ = xbot; }
  }
  int rua, rda, rub, rdb;
  if (maybe_neg && __ballot((acta && xa < 0.0f) || (actb && xb_ < 0.0f))) {
    rda = pil_round_down(xa); rua = max(pil_round_up(xa), rda);
    rdb = pil_round_down(xb_); rub = max(pil_round_up(xb_), rdb);
  } else {
    rua = (int)floorf(__fadd_rn(xa, 0.5f)); rda = (int)ceilf(__fsub_rn(xa, 0.5f));
    rub = (int)floorf(__fadd_rn(xb_, 0.5f)); rdb = (int)ceilf(__fsub_rn(xb_, 0.5f));
  }
  uint32_t* Ta = &L->T[0][0] + rla; uint32_t* Pa = &L->Px[0][0] + rla;
  uint32_t* Tb = &L->T[0][0] + rlb; uint32_t* Pb = &L->Px[0][0] + rlb;
  const int ta = rda - xb + 1, tb = rdb - xb + 1;
  const bool toga = acta && !dupa && ta < npx, togb = actb && !dupb && tb < npx;
  const int tca = min(max(ta, 0), npx - 1), tcb = min(max(tb, 0), npx - 1);
  atomicXor(Ta + (tca >> 5) * SWB_WAVE, toga ? (1u << (tca & 31)) : 0u);
  atomicXor(Tb + (tcb >> 5) * SWB_WAVE, togb ? (1u << (tcb & 31)) : 0u);
  const int pua = rua - xb, pub = rub - xb;
  const bool oka = acta && rua <= rda && (uint32_t)pua < (uint32_t)npx, okb = actb && rub <= rdb && (uint32_t)pub < (uint32_t)npx;
  const int pca = min(max(pua, 0), npx - 1), pcb = min(max(pub, 0), npx - 1);
  atomicOr(Pa + (pca >> 5) * SWB_WAVE, oka ? (1u << (pca & 31)) : 0u);
  atomicOr(Pb + (pcb >> 5) * SWB_WAVE, okb ? (1u << (pcb & 31)) : 0u);
}

template <int NW>
__device__ __forceinline__ void coverage_batch2(''')
  replace_once(files, k, '''        for (int j = 0; j < bound; j += G) {
          const int yy = ylo + j0 + j;
          scatter_crossing2<NW>(L, xb, npx, (yy - yb) & 63, live && yy <= yhi, yy, x0f, ed.y0, ed.dx, xtop, xbot, any_repl, w0 == 0,
                                eymin, eymax, symax);
        }
''', '''        int j = 0;
        for (; j + G < bound; j += 2 * G) {
          const int ya = ylo + j0 + j, yb2 = ya + G;
          scatter_crossing2x2<NW>(L, xb, npx, (ya - yb) & 63, (yb2 - yb) & 63, live && ya <= yhi, live && yb2 <= yhi, ya, yb2, x0f, ed.y0,
                                  ed.dx, xtop, xbot, any_repl, w0 == 0, eymin, eymax, symax);
        }
        if (j < bound) {
          const int yy = ylo + j0 + j;
          scatter_crossing2<NW>(L, xb, npx, (yy - yb) & 63, live && yy <= yhi, yy, x0f, ed.y0, ed.dx, xtop, xbot, any_repl, w0 == 0,
                                eymin, eymax, symax);
        }
''')
