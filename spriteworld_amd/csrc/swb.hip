// swb.hip -- C-ABI host side of the batched Spriteworld engine (see include/swb.h).
//
// Owns the device copies of the constant tables, the reset pool, the live structure-of-arrays
// state and the run lists handed from the cover kernel to the resample / fill kernel; launches the
// two kernels of a step (swb_kernels.hip.inc).
// Built only for gfx950:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "swb_kernels.hip.inc"
#include "swb_pow.hip.inc"
#include "swb_sampler.hip.inc"

// the cover kernels of canvases wider than 320 px are instantiated in swb_wide.hip (other scheduler flags)
extern template __global__ void swb_cover_kernel<20, false>(const swb_params);
extern template __global__ void swb_cover_kernel<20, true>(const swb_params);

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) return fail(SWB_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

template <typename T>
int upload(T** dst, const T* src, size_t count) {
  if (*dst) { (void)hipFree(*dst); *dst = nullptr; }
  if (count == 0) count = 1;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(dst), count * sizeof(T)));
  if (src) HIP_TRY(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
  else HIP_TRY(hipMemset(*dst, 0, count * sizeof(T)));
  return 0;
}

}  // namespace

struct swb_engine {
  swb_config cfg;
  int device = 0;
  swb_params p;        // device pointers + config, passed by value to the kernel
  bool have_shapes = false, have_h = false, have_v = false, have_pool = false;
  int vslots = SWB_VSLOTS;
  int nbands = 1;                    // bands of output rows per (environment, column group) in the resample / fill kernel
  bool tables_dirty = true;          // band / break / column-group tables follow the resampling tables and nbands
  std::vector<int32_t> v_ymin_host, v_end_host, h_xmin_host, h_cnt_host;
  std::vector<int> shape_nverts;     // vertices per uploaded shape
  int ovf_lds_bytes = 0;             // LDS footprint the overflow slots were sized with
  void (*ovf_fn)(const swb_params) = nullptr;   // ... and the cover kernel (plain or override build)
  size_t lds_block = 0;
  // owned device buffers
  double* d_shape_verts = nullptr;
  double* d_shape_dmin = nullptr;
  int32_t* d_shape_off = nullptr;
  int32_t *d_h_xmin = nullptr, *d_h_cnt = nullptr, *d_h_tbl = nullptr, *d_h_pfx = nullptr;
  int32_t *d_v_tab = nullptr, *d_v_end = nullptr, *d_v_pfx = nullptr;
  int32_t* d_p_n = nullptr;
  double *d_p_x = nullptr, *d_p_y = nullptr, *d_p_xv = nullptr, *d_p_yv = nullptr, *d_p_scale = nullptr,
         *d_p_ca = nullptr, *d_p_sa = nullptr;
  int32_t* d_p_shape = nullptr;
  uint32_t* d_p_rgb = nullptr;
  int8_t* d_p_label = nullptr;
  int8_t* d_p_cell_label = nullptr;  // swb_pool::cell_label (tasks that key on position)
  bool keyed = false;                // some task's filter keys on position (swb_task::n_xcuts / n_ycuts)
  uint8_t* d_p_attr = nullptr;       // swb_pool::attr_f32
  int32_t *d_pool_base = nullptr, *d_pool_len = nullptr;
  double *d_p_angle = nullptr, *d_p_color = nullptr;
  swb_sampler* d_sampler = nullptr;
  int pool_entries = 0;
  bool pool_sampled = false, pool_uniform = false;   // pool came from swb_sample_pool / env-major fixed-length layout
  double *d_x = nullptr, *d_y = nullptr;
  int32_t *d_nspr = nullptr, *d_entry = nullptr, *d_step_count = nullptr, *d_episode = nullptr;
  uint8_t* d_reset_next = nullptr;
  uint32_t *d_ovf = nullptr, *d_ovf_bitmap = nullptr;
  int ovf_slots = 0;
  // cost-ordered dispatch (swb_params::cost_cnt)
  uint32_t* d_cost_cnt = nullptr;
  int32_t* d_cost_list = nullptr;
  int32_t* d_ccost_list = nullptr;   // ... and the environments in order of the cover kernel's cost (swb_params::cover_order)
  bool cover_lists_filed = false;    // the previous launch filed every environment (it rendered)
  int launch_phase = 0;              // launch count % 3 (swb_params::cphase)
  int launch_parity = 0;
  // hand-off cover -> resample
  uint32_t *d_runs = nullptr, *d_rhdr = nullptr, *d_arena_head = nullptr;
  int32_t* d_env_state = nullptr;    // swb_get_env_state's 20-byte record
  int arena_override = -1;           // SWB_ARENA_UNITS (tests): units of the shared arena; -1: sized from the batch
  int run_cap_worst = 0;             // (max(4, S + 1) canvas heights + 1: what a list reserves until swb_trim_run_lists)
  bool lists_trimmed = false;        // the lists have been cut down to what the launches so far needed
  bool lists_valid = false;          // the last launch wrote them (it rendered through the second kernel)
  int32_t *d_band_y0 = nullptr, *d_band_first = nullptr, *d_band_lo = nullptr, *d_cg_lo = nullptr, *d_cg_hi = nullptr;
  uint32_t* d_v_break = nullptr;
  // live sprite overrides (swb_set_sprite_attr), allocated at the first call
  uint8_t* d_ov_flag = nullptr;
  int32_t* d_ov_shape = nullptr;
  double *d_ov_scale = nullptr, *d_ov_angle = nullptr, *d_ov_cpath = nullptr;
  int8_t* d_ov_label = nullptr;
  int8_t* d_ov_cell_label = nullptr;
  // timing
  bool timing = false;
  struct step_events { hipEvent_t e0, e1, e2; };     // before cover, between the kernels, after resample / fill
  std::vector<step_events> events;
  std::vector<step_events> event_pool;               // events of flushed steps, reused (no hipEventCreate per step)
  // read once at swb_create (never per launch): the device's compute units and the test / A-B switches of the environment
  int cus = 0;
  bool no_paint_in_cover = false, force_cover_order = false;
  double timed_ms = 0.0, timed_cover_ms = 0.0;
  int64_t timed_launches = 0;
};

namespace {

typedef void (*kernel_fn)(const swb_params);

// cover kernel by canvas width (NW 32-pixel words per canvas row); resample kernel by the output rows a canvas
// row can feed at once (VS)
struct variant { int nw; kernel_fn fn, fn_ov, fn_paint, fn_paint_ov; size_t lds_fixed, outrow_bytes; };

template <int NW>
variant make_variant() {
  const size_t outrow = 528;          // wave_lds::outrow is build_all_edges' scratch (132 dwords)
  kernel_fn paint = nullptr, paint_ov = nullptr;       // the builds that paint the frame themselves: canvases of up to 64 px only
  if constexpr (NW == 2) { paint = swb_cover_kernel<NW, false, true>; paint_ov = swb_cover_kernel<NW, true, true>; }
  return {NW, swb_cover_kernel<NW>, swb_cover_kernel<NW, true>, paint, paint_ov, (sizeof(wave_lds<NW>) + outrow + 15) & ~(size_t)15, outrow};
}

const variant kVariants[] = {make_variant<2>(), make_variant<4>(), make_variant<5>(), make_variant<10>(), make_variant<20>()};

const variant* pick_variant(int Wc) {
  for (const variant& v : kVariants)
    if (32 * v.nw >= Wc) return &v;
  return nullptr;
}

kernel_fn pick_resample(int AA, int vslots, int* vs_out) {
  if (AA == 1) { if (vs_out) *vs_out = 0; return swb_fill_kernel; }
  if (vslots <= 6) { if (vs_out) *vs_out = 6; return swb_resample_kernel<6>; }
  if (vs_out) *vs_out = 8;
  return swb_resample_kernel<8>;
}

// LDS bytes of one wave (= one environment) of the cover kernel: wave_lds + edge records + span lists.  The
// centred paths (16 B per vertex) borrow the idle mask arrays, or the edge records' storage.
size_t lds_per_wave(const swb_engine* h, const variant* v, int* cpath_in_masks) {
  const swb_params& p = h->p;
  const size_t cpath_bytes = (size_t)p.max_edges * 16;
  const int in_masks = (2 * (size_t)SWB_NWA(v->nw) * SWB_WAVE * 4 >= cpath_bytes) ? 1 : 0;
  if (cpath_in_masks) *cpath_in_masks = in_masks;
  return (v->lds_fixed + (((size_t)p.max_edges * sizeof(edge_rec) + 15) & ~(size_t)15) +
          (size_t)p.max_spans * SWB_WAVE * 4 + 15) & ~(size_t)15;
}

// Overflow slots for the span lists of the cover kernel: one per wave that can be resident at once (occupancy of
// the kernel that will be launched with its LDS footprint x CUs, capped by the batch), never one per environment.
// (fn_alt: the other build of the same family -- the one that paints the frame itself / the one that does not --, which a
// launch with / without an observation buffer switches to: the slots are sized once for the more demanding of the two)
int ensure_overflow_slots(swb_engine* h, kernel_fn fn, kernel_fn fn_alt, size_t lds_bytes) {
  if (h->d_ovf) return 0;
  int per_cu = 0, cus = 0;
  for (kernel_fn f : {fn, fn_alt}) {
    int n = 0;
    if (!f) continue;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(f), SWB_WAVE, lds_bytes) != hipSuccess || n < 1)
      n = 32;
    per_cu = std::max(per_cu, n);
  }
  HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
  long long slots = (long long)per_cu * cus;
  slots = std::min<long long>(2 * slots, (long long)h->p.N);              // the bitmap stays at most half full: few retries
  slots = (slots + 31) / 32 * 32;
  h->ovf_slots = (int)slots;
  if (upload(&h->d_ovf, (const uint32_t*)nullptr, (size_t)slots * 64 * h->p.ovf_cap)) return SWB_ERR_HIP;
  if (upload(&h->d_ovf_bitmap, (const uint32_t*)nullptr, (size_t)slots / 32)) return SWB_ERR_HIP;
  h->p.ovf = h->d_ovf; h->p.ovf_bitmap = h->d_ovf_bitmap; h->p.ovf_slots = h->ovf_slots;
  return 0;
}

// Bands of output rows, the rows at which a run must end, and the canvas columns each group of 64 output columns
// can see -- from the resampling tables (anti_aliasing > 1) or the identity (anti_aliasing = 1).
// need_lists = false: a handle whose cover kernel paints the frame itself (anti_aliasing = 1, an image of up to 64 columns)
// never hands anything to a second kernel and reserves no lists.
int ensure_handoff_tables(swb_engine* h, bool need_lists = true) {
  if (!h->tables_dirty && (h->d_runs || !need_lists)) return 0;
  swb_params& p = h->p;
  int nb = h->nbands;
  p.ncg = (p.Wo + 63) / 64;
  // Bands: about Ho / nb rows each; for the resample kernel the first row of a band is moved so that the number of
  // output rows in flight at its first canvas row is a multiple of VS (the band's oldest row then sits in slot 0).
  int vs = 0;
  (void)pick_resample(p.AA, h->vslots, &vs);
  auto first_in_flight = [&](int o_lo) {
    int f = o_lo;
    while (f > 0 && h->v_end_host[f - 1] >= h->v_ymin_host[o_lo]) --f;
    return f;
  };
  std::vector<int32_t> blo(1, 0);
  for (int b = 1; b < nb; ++b) {
    int want = (int)((long long)b * p.Ho / nb), best = -1;
    if (p.AA == 1) best = want;
    else
      for (int d = 0; d < p.Ho && best < 0; ++d)
        for (int c : {want + d, want - d})
          if (c > blo.back() && c < p.Ho && first_in_flight(c) % vs == 0) { best = c; break; }
    if (best > blo.back() && best < p.Ho) blo.push_back(best);
  }
  nb = (int)blo.size();
  blo.push_back(p.Ho);
  p.nbands = nb;
  std::vector<int32_t> y0(nb, 0), first(nb, 0), lo(p.ncg, 0), hi(p.ncg, 0);
  std::vector<uint32_t> brk((size_t)(p.Hc + 31) / 32 + 3, 0u);      // (+3: a batch reads three words from its first row's word)
  auto set_break = [&](int y) { if (y >= 0 && y < p.Hc) brk[y >> 5] |= 1u << (y & 31); };
  for (int b = 0; b < nb; ++b) {
    const int o_lo = blo[b];
    if (p.AA == 1) { y0[b] = o_lo; first[b] = o_lo; }
    else { y0[b] = h->v_ymin_host[o_lo]; first[b] = first_in_flight(o_lo); }
    set_break(y0[b]);
  }
  if (p.AA != 1)
    for (int o = 0; o < p.Ho; ++o) set_break(h->v_end_host[o] + 1);
  for (int g = 0; g < p.ncg; ++g) {
    if (p.AA == 1) { lo[g] = 64 * g; hi[g] = std::min(64 * g + 64, p.Wo); continue; }
    lo[g] = 1 << 30; hi[g] = 0;
    for (int o = 64 * g; o < std::min(64 * g + 64, p.Wo); ++o) {
      lo[g] = std::min(lo[g], h->h_xmin_host[o]);
      hi[g] = std::max(hi[g], h->h_xmin_host[o] + h->h_cnt_host[o]);
    }
  }
  if (upload(&h->d_band_y0, y0.data(), y0.size()) || upload(&h->d_band_first, first.data(), first.size()) ||
      upload(&h->d_band_lo, blo.data(), blo.size()) ||
      upload(&h->d_v_break, brk.data(), brk.size()) || upload(&h->d_cg_lo, lo.data(), lo.size()) ||
      upload(&h->d_cg_hi, hi.data(), hi.size()))
    return SWB_ERR_HIP;
  p.band_lo = h->d_band_lo;
  p.band_y0 = h->d_band_y0; p.band_first = h->d_band_first; p.v_break = h->d_v_break; p.cg_lo = h->d_cg_lo; p.cg_hi = h->d_cg_hi;
  // run lists: run_cap units of 8 bytes per (environment, column group), then the shared arena their overflow segments
  // come from, then 4 units of slack (the resample kernel reads a run's first 16 bytes in one go)
  if (!h->d_runs && need_lists) {
    const size_t fixed = (size_t)p.N * p.ncg * p.run_cap;
    // arena (a list that outgrows its own part moves there, into a segment twice as large): half of the own parts together once
    // they have been trimmed (before that they hold any scene of convex sprites by themselves), and at least eight worst-case
    // lists -- sixteen times their size, a moved list's segment being a power of two times its own part
    size_t arena = std::max(h->lists_trimmed ? fixed / 2 : (size_t)0, (size_t)16 * h->run_cap_worst);
    if (h->arena_override >= 0) arena = (size_t)h->arena_override;
    const size_t units = fixed + arena + 4;
    if (units * 8 >= ((size_t)1 << 32))                                  // (list positions are 32-bit byte offsets)
      return fail(SWB_ERR_INVALID, "run lists of %zu MB (%d environments x %d column groups x %d units + an arena of %zu units) exceed "
                  "the 4 GB a list position can address: step fewer environments per engine", units * 8 >> 20, p.N, p.ncg, p.run_cap, arena);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && units * 8 + (size_t)p.N * SWB_RHDR_DWORDS * 4 > free_b)
      return fail(SWB_ERR_HIP, "run lists need %zu MB of device memory (%d environments x %d column groups x %d units of 8 bytes + an "
                  "arena of %zu units), %zu MB are free", units * 8 >> 20, p.N, p.ncg, p.run_cap, arena, free_b >> 20);
    if (upload(&h->d_runs, (const uint32_t*)nullptr, units * 2)) return SWB_ERR_HIP;
    if (upload(&h->d_rhdr, (const uint32_t*)nullptr, (size_t)p.N * SWB_RHDR_DWORDS)) return SWB_ERR_HIP;
    if (upload(&h->d_arena_head, (const uint32_t*)nullptr, 2)) return SWB_ERR_HIP;
    p.runs = h->d_runs; p.rhdr = h->d_rhdr;
    p.arena_head = h->d_arena_head;
    p.arena_base = (int64_t)fixed;
    p.arena_units = (int32_t)std::min(arena, (size_t)0x7fffffff);
  }
  h->tables_dirty = false;
  return 0;
}

// Frees the hand-off lists; the next launch allocates them again with the capacity h->p.run_cap holds by then.
int drop_run_lists(swb_engine* h) {
  if (!h->d_runs) return 0;
  HIP_TRY(hipDeviceSynchronize());
  (void)hipFree(h->d_runs); (void)hipFree(h->d_rhdr); (void)hipFree(h->d_arena_head);
  h->d_runs = nullptr; h->d_rhdr = nullptr; h->d_arena_head = nullptr;
  h->p.runs = nullptr; h->p.rhdr = nullptr; h->p.arena_head = nullptr;
  h->lists_valid = false;
  h->tables_dirty = true;
  return 0;
}

// A new pool may hold denser scenes than the one the lists were trimmed to: back to the full reservation.
int restore_run_list_reservation(swb_engine* h) {
  if (!h->lists_trimmed) return 0;
  if (int rc = drop_run_lists(h)) return rc;
  h->lists_trimmed = false;
  if (!getenv("SWB_RUN_CAP")) h->p.run_cap = h->run_cap_worst;
  return 0;
}

int flush_timing(swb_engine* h) {
  for (auto& ev : h->events) {
    HIP_TRY(hipEventSynchronize(ev.e2));
    float ms = 0.f, ms_cover = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ev.e0, ev.e2));
    HIP_TRY(hipEventElapsedTime(&ms_cover, ev.e0, ev.e1));
    h->timed_ms += ms;
    h->timed_cover_ms += ms_cover;
    h->timed_launches += 1;
    h->event_pool.push_back(ev);
  }
  h->events.clear();
  return 0;
}

int launch(swb_engine* h, const void* actions, const swb_outputs* out, int render_only, hipStream_t stream) {
  if (!h->have_shapes) return fail(SWB_ERR_STATE, "swb_upload_shapes has not been called");
  if (!h->have_pool) return fail(SWB_ERR_STATE, "swb_set_pool has not been called");
  const swb_config& c = h->cfg;
  if (c.anti_aliasing != 1 && !(h->have_h && h->have_v))
    return fail(SWB_ERR_STATE, "swb_upload_resample (both axes) is required when anti_aliasing > 1");
  const variant* v = pick_variant(h->p.Wc);
  if (!v) return fail(SWB_ERR_INVALID, "canvas %dx%d not supported", h->p.Wc, h->p.Hc);
  const bool paints_itself = h->p.AA == 1 && (h->p.Wo + 63) / 64 == 1 && !h->no_paint_in_cover && v->fn_paint;
  if (int rc = ensure_handoff_tables(h, !paints_itself)) return rc;
  int vs = 0;
  const kernel_fn fn2 = pick_resample(h->p.AA, h->vslots, &vs);
  swb_params p = h->p;
  if (p.v_tab && vs == 8) {                                           // slot tables of this VS
    p.v_tab += (size_t)p.Hc * SWB_VSLOTS;
    p.v_pfx += (size_t)(p.Hc + 1) * SWB_VSLOTS;
  }
  p.actions = actions;
  p.obs = out ? out->obs : nullptr;
  p.reward = out ? out->reward : nullptr;
  p.discount = out ? out->discount : nullptr;
  p.step_type = out ? out->step_type : nullptr;
  p.success = out ? out->success : nullptr;
  p.error = out ? out->error : nullptr;
  p.render_only = render_only;
  if (!render_only && actions == nullptr) return fail(SWB_ERR_INVALID, "actions is NULL");
  int cpath_in_masks = 0;
  const size_t lds = lds_per_wave(h, v, &cpath_in_masks);
  p.cpath_in_masks = cpath_in_masks;
  p.lds_per_wave = (int32_t)lds;
  p.outrow_bytes = (int32_t)v->outrow_bytes;
  if (lds > 160 * 1024) return fail(SWB_ERR_INVALID, "LDS request %zu B exceeds 160 KiB", lds);
  // engines on which a sprite setter has been called run the build that reads the per-environment overrides
  const bool paint = h->p.AA == 1 && h->p.ncg == 1 && out && out->obs && !h->no_paint_in_cover && v->fn_paint;
  const kernel_fn fn = paint ? (h->d_ov_flag ? v->fn_paint_ov : v->fn_paint) : (h->d_ov_flag ? v->fn_ov : v->fn);
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // first launch, a new pool that shrank the LDS footprint (more waves resident), or the switch to the override build -- not
  // the switch between the painting and the plain build of a family, which alternating launches with and without an
  // observation buffer make at every launch (the cache is keyed on the family's plain build)
  const kernel_fn fn_family = h->d_ov_flag ? v->fn_ov : v->fn;
  if (!h->d_ovf || (int)lds < h->ovf_lds_bytes || fn_family != h->ovf_fn) {
    if (h->d_ovf) {
      HIP_TRY(hipDeviceSynchronize());
      (void)hipFree(h->d_ovf); (void)hipFree(h->d_ovf_bitmap);
      h->d_ovf = nullptr; h->d_ovf_bitmap = nullptr;
    }
    if (int rc = ensure_overflow_slots(h, fn_family, h->d_ov_flag ? v->fn_paint_ov : v->fn_paint, lds)) return rc;
    h->ovf_lds_bytes = (int)lds;
    h->ovf_fn = fn_family;
    p.ovf = h->p.ovf; p.ovf_bitmap = h->p.ovf_bitmap; p.ovf_slots = h->p.ovf_slots;
  }
  swb_engine::step_events ev = {nullptr, nullptr, nullptr};
  if (h->timing) {
    if (!h->event_pool.empty()) { ev = h->event_pool.back(); h->event_pool.pop_back(); }
    else {
      HIP_TRY(hipEventCreate(&ev.e0));
      HIP_TRY(hipEventCreate(&ev.e1));
      HIP_TRY(hipEventCreate(&ev.e2));
    }
    HIP_TRY(hipEventRecord(ev.e0, stream));
  }
  const size_t lds2 = p.AA == 1 ? 0 : (((size_t)p.h_pfx_len * 4 + 15) & ~(size_t)15);
  p.parity = h->launch_parity;
  p.cphase = h->launch_phase;
  // anti_aliasing = 1, one column group: the cover kernel paints the frame itself, there is no second kernel
  p.paint_in_cover = paint ? 1 : 0;
  // environments in order of what their cover wave cost in the previous launch -- if that launch filed them all
  // (and the launch is more than one round of cover waves but not many: one round needs no order, and from about a dozen
  // rounds on the plain order was measured 1.6 % faster)
  {
    const long long slots = (long long)std::max(h->cus, 1) * 4 * SWB_COVER_WAVES(v->nw);
    p.cover_order = (h->cover_lists_filed && p.ccost_list && ((p.N > slots && p.N <= 4 * slots) || h->force_cover_order)) ? 1 : 0;
    if (p.N > 2 * slots) p.prio_levels &= ~2;          // (cover waves' priorities: measured +1.3 % at three rounds of waves)
  }
  // (cost-ordered: block b serves rank b / 8 of shard b % 8; a shard holds up to cost_cap environments)
  auto launch_cover = [&](int e0, int e1) {
    hipLaunchKernelGGL(fn, dim3(e1 - e0), dim3(SWB_WAVE), lds, stream, p);
  };
  auto launch_resample = [&](int e0, int e1, hipStream_t st) {
    const int blocks_x = p.cost_cnt ? SWB_COST_SHARDS * ((p.cost_cap + SWB_RS_WAVES_PER_BLOCK - 1) / SWB_RS_WAVES_PER_BLOCK)
                                    : (e1 - e0 + SWB_RS_WAVES_PER_BLOCK - 1) / SWB_RS_WAVES_PER_BLOCK;
    // (cost-ordered: a list entry names its column group -- and, for small batches, its band: swb_params::band_tasks)
    const dim3 grid(blocks_x, (p.cost_cnt && p.band_tasks) ? 1 : p.nbands, p.cost_cnt ? 1 : p.ncg);
    hipLaunchKernelGGL(fn2, grid, dim3(SWB_WAVE * SWB_RS_WAVES_PER_BLOCK), lds2, st, p);
  };
  launch_cover(0, c.n_envs);
  HIP_TRY(hipGetLastError());
  if (h->timing && !p.paint_in_cover) HIP_TRY(hipEventRecord(ev.e1, stream));
  if (p.obs && !p.paint_in_cover) launch_resample(0, c.n_envs, stream);
  HIP_TRY(hipGetLastError());
  if (!p.obs && p.cost_cnt && render_only != 2)   // no second kernel to clear the next launch's bucket counters (kind 0; the cover kernel keeps kind 1)
    HIP_TRY(hipMemsetAsync(h->d_cost_cnt + (size_t)(p.parity ^ 1) * SWB_COST_SET, 0, SWB_COST_SET * sizeof(uint32_t), stream));
  if (render_only == 2) {
    // swb_evaluate files nothing and lists nothing: the dispatch state (parity, phase, what the previous launch filed) stays as
    // the last step left it -- an evaluation between two steps does not cost the next one its cost order (round-5 advice)
    if (h->timing) {
      HIP_TRY(hipEventRecord(ev.e1, stream));
      HIP_TRY(hipEventRecord(ev.e2, stream));
      h->event_pool.push_back(ev);
    }
    return 0;
  }
  h->cover_lists_filed = p.obs && p.ccost_list;
  h->lists_valid = p.obs && !p.paint_in_cover;
  h->launch_phase = (h->launch_phase + 1) % 3;
  h->launch_parity ^= 1;
  if (h->timing) {
    if (p.paint_in_cover) HIP_TRY(hipEventRecord(ev.e1, stream));      // (no second kernel: the whole step is the cover kernel)
    HIP_TRY(hipEventRecord(ev.e2, stream));
    h->events.push_back(ev);
    if (h->events.size() >= 4096) return flush_timing(h);
  }
  return 0;
}

}  // namespace

extern "C" {

const char* swb_last_error(void) { return g_err.c_str(); }
int swb_version(void) { return 1; }

int swb_create(const swb_config* cfg, int device, swb_handle* out) {
  if (!cfg || !out) return fail(SWB_ERR_INVALID, "null argument");
  if (cfg->n_envs < 1) return fail(SWB_ERR_INVALID, "n_envs must be >= 1");
  if (cfg->max_sprites < 1 || cfg->max_sprites > SWB_MAX_SPRITES)
    return fail(SWB_ERR_INVALID, "max_sprites must be in [1, %d]", SWB_MAX_SPRITES);
  if (cfg->n_tasks < 1 || cfg->n_tasks > SWB_MAX_TASKS) return fail(SWB_ERR_INVALID, "n_tasks must be in [1, %d]", SWB_MAX_TASKS);
  if (cfg->anti_aliasing < 1) return fail(SWB_ERR_INVALID, "anti_aliasing must be >= 1");
  if (cfg->image_h < 1 || cfg->image_w < 1 || (cfg->image_h % 4) != 0)
    return fail(SWB_ERR_INVALID, "image_size[0] must be a positive multiple of 4");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SWB_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= ndev) return fail(SWB_ERR_NO_DEVICE, "device %d out of range (%d devices)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(SWB_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
  swb_engine* h = new swb_engine();
  h->cfg = *cfg;
  h->device = device;
  (void)hipDeviceGetAttribute(&h->cus, hipDeviceAttributeMultiprocessorCount, device);
  h->no_paint_in_cover = getenv("SWB_NO_PAINT_IN_COVER") != nullptr;
  h->force_cover_order = getenv("SWB_COVER_ORDER") != nullptr;
  swb_params& p = h->p;
  memset(&p, 0, sizeof(p));
  p.N = cfg->n_envs; p.S = cfg->max_sprites; p.AA = cfg->anti_aliasing;
  // The reference makes the PIL canvas of *size* (AA*image_size[0], AA*image_size[1]) = (width, height)
  // (pil_renderer.py:50-51,64); np.array(image) is then [image_size[1], image_size[0], 3].
  p.Wo = cfg->image_h; p.Ho = cfg->image_w;
  p.Wc = p.AA * p.Wo; p.Hc = p.AA * p.Ho;
  p.bg = (uint32_t)cfg->bg_rgb[0] | ((uint32_t)cfg->bg_rgb[1] << 8) | ((uint32_t)cfg->bg_rgb[2] << 16);
  p.action_space = cfg->action_space; p.keep_in_frame = cfg->keep_in_frame;
  p.max_episode_length = cfg->max_episode_length; p.pos_is_f32 = cfg->pos_is_f32;
  p.action_is_f32 = cfg->action_is_f32;
  p.action_scale = cfg->action_scale; p.motion_cost = cfg->motion_cost;
  p.n_tasks = cfg->n_tasks; p.is_meta = cfg->is_meta; p.meta_aggregator = cfg->meta_aggregator;
  p.meta_termination = cfg->meta_termination; p.meta_terminate_bonus = cfg->meta_terminate_bonus;
  memcpy(p.tasks, cfg->tasks, sizeof(p.tasks));
  for (int t = 0; t < cfg->n_tasks; ++t) {
    const swb_task& tk = cfg->tasks[t];
    if (tk.n_xcuts < 0 || tk.n_xcuts > SWB_MAX_CUTS || tk.n_ycuts < 0 || tk.n_ycuts > SWB_MAX_CUTS) {
      delete h;
      return fail(SWB_ERR_INVALID, "task %d: between 0 and %d position thresholds per axis supported", t, SWB_MAX_CUTS);
    }
    for (int k = 1; k < tk.n_xcuts; ++k) if (!(tk.xcuts[k - 1] < tk.xcuts[k])) { delete h; return fail(SWB_ERR_INVALID, "task %d: xcuts must ascend", t); }
    for (int k = 1; k < tk.n_ycuts; ++k) if (!(tk.ycuts[k - 1] < tk.ycuts[k])) { delete h; return fail(SWB_ERR_INVALID, "task %d: ycuts must ascend", t); }
    if (tk.n_xcuts + tk.n_ycuts > 0) h->keyed = true;
  }
  if (p.Wc > 1023 || p.Hc > 65535) { delete h; return fail(SWB_ERR_INVALID, "canvas %dx%d too large", p.Wc, p.Hc); }
  if (!pick_variant(p.Wc)) { delete h; return fail(SWB_ERR_INVALID, "canvas width %d not supported", p.Wc); }
  if (p.Wo > 64 * SWB_MAX_CG) { delete h; return fail(SWB_ERR_INVALID, "image width %d not supported (max %d)", p.Wo, 64 * SWB_MAX_CG); }
  // Bands of output rows per (environment, column group) in the second kernel: a band repeats the 25 canvas rows it
  // shares with the band above, so there are only as many as it takes to give every SIMD its eight waves (small
  // batches), in bands of at least 16 rows -- of 8 rows where those fill the SIMDs no more than once (measured, 64-px images:
  // 512 environments 0.0311 ms in 8 bands against 0.0354 in 4; 1024 environments, every band a task of its own (band_tasks
  // below), 0.0342 against 0.0366; 2048 environments 0.0498 against 0.0422: profiles/r06_experiments/bands_small_batches.txt).
  {
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    const long long resident = (long long)std::max(cus, 1) * 4 * SWB_RS_WAVES_PER_SIMD;
    const long long tasks = (long long)p.N * ((p.Wo + 63) / 64);
    int nb = 1;
    while (nb < SWB_MAX_BANDS && tasks * nb < resident && (p.Ho / (2 * nb) >= 16 || (p.Ho / (2 * nb) >= 8 && tasks * nb * 2 <= resident))) nb *= 2;
    if (const char* x = getenv("SWB_BANDS")) nb = std::max(1, std::min(atoi(x), (int)SWB_MAX_BANDS));
    h->nbands = std::min(nb, p.Ho);
  }
  // cost buckets of the second kernel's tasks: 32 of them over the cost of a run list (3 per run + 2 per 8-byte unit: a canvas row
  // costs 1 unit (one span), 2 (two or three) or more, and rows that repeat the row above nothing); the range they span is
  // fitted by every launch (cost_housekeeping), to begin with it is [0, 8 * canvas height)
  p.cost_range0 = std::max(8 * p.Hc, SWB_KEY_BUCKETS_FITTED);      // (a run and its units cost 5 .. 9; a launch later the range is a measured one)
  if (!getenv("SWB_NO_COST_ORDER") && p.N < (1 << 24)) {
    // Small batches in several bands: a task of the second kernel is one BAND of a list, filed under the band's own cost, once
    // the launch has four waves or more per SIMD to deal (measured, resample / fill kernel: 2048 environments in 4 bands -9.5 %,
    // 1024 -2 %, a 128x128 image at anti_aliasing = 1 -22 %; two waves per SIMD, 256 environments in 8 bands, +3 %:
    // profiles/r06_experiments/band_tasks_ab.txt).
    {
      int cus3 = 0;
      (void)hipDeviceGetAttribute(&cus3, hipDeviceAttributeMultiprocessorCount, device);
      const long long waves3 = (long long)p.N * ((p.Wo + 63) / 64) * h->nbands;
      p.band_tasks = (h->nbands > 1 && waves3 >= 4ll * std::max(cus3, 1) * 4 && !getenv("SWB_NO_BAND_TASKS")) ? 1 : 0;
      if (const char* x = getenv("SWB_BAND_TASKS")) p.band_tasks = (h->nbands > 1 && atoi(x) != 0) ? 1 : 0;      // tests
    }
    p.cost_cap = ((p.Wo + 63) / 64) * ((p.N + SWB_COST_SHARDS - 1) / SWB_COST_SHARDS) * (p.band_tasks ? h->nbands : 1);
    std::vector<uint32_t> cnt0(5 * SWB_COST_SET + SWB_COST_WORDS, 0u);
    for (int ph = 0; ph < 3; ++ph) cnt0[SWB_COST_WORD_COVER_SHIFT(ph)] = 12;   // bucket width of the cover kernel's cycle counts: 2^12 to begin with
    for (int par = 0; par < 2; ++par) {                                       // buckets of the second kernel's tasks: [0, cost_range0) to begin with
      cnt0[SWB_COST_WORD_KEY_LO(par)] = 0u;
      cnt0[SWB_COST_WORD_KEY_RANGE(par)] = (uint32_t)p.cost_range0;
      cnt0[SWB_COST_WORD_KEY_SCALE(par)] = (uint32_t)(((unsigned long long)SWB_KEY_BUCKETS_FITTED << 16) / (uint32_t)p.cost_range0);
    }
    p.prio_div = 4;              // (measured: levels of a quarter of a mean wave 2 % better than of half a wave, at 8192 environments)
    if (const char* x = getenv("SWB_PRIO_DIV")) p.prio_div = std::max(1, atoi(x));
    // (priorities when the launch is at most two rounds of resample waves: in steady state they cost 0.8 %)
    {
      int cus2 = 0;
      (void)hipDeviceGetAttribute(&cus2, hipDeviceAttributeMultiprocessorCount, device);
      const long long waves = (long long)p.N * ((p.Wo + 63) / 64) * h->nbands;
      p.prio_levels = (!getenv("SWB_NO_PRIO") && waves <= 2ll * std::max(cus2, 1) * 4 * SWB_RS_WAVES_PER_SIMD) ? 1 : 0;
      if (!getenv("SWB_NO_COVER_PRIO")) p.prio_levels |= 2;          // (cover waves: only ever used with cover_order)
    }
    const size_t ccap = (size_t)(p.N + SWB_COST_SHARDS - 1) / SWB_COST_SHARDS;
    if (upload(&h->d_cost_cnt, cnt0.data(), cnt0.size()) ||
        upload(&h->d_ccost_list, (const int32_t*)nullptr, (size_t)3 * SWB_COST_SHARDS * SWB_COST_BUCKETS * ccap) ||
        upload(&h->d_cost_list, (const int32_t*)nullptr, (size_t)2 * SWB_COST_SHARDS * SWB_COST_BUCKETS * p.cost_cap)) {
      swb_destroy(h);
      return SWB_ERR_HIP;
    }
    p.cost_cnt = h->d_cost_cnt; p.cost_list = h->d_cost_list;
    if (!getenv("SWB_NO_COVER_ORDER")) p.ccost_list = h->d_ccost_list;
    // rounds of the dealing = the compute units of an XCD (8 XCDs; a power of two on this part: 32)
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    const int per_xcd = cus / SWB_COST_SHARDS;
    // (only when every wave of the launch is resident at once: with several rounds of waves the heaviest tasks must simply
    // come first -- measured on two rounds: +1...2 % with alternating rounds)
    const long long waves = (long long)p.N * ((p.Wo + 63) / 64) * h->nbands;
    if (per_xcd >= 2 && (per_xcd & (per_xcd - 1)) == 0 && waves <= (long long)cus * 4 * SWB_RS_WAVES_PER_SIMD && !getenv("SWB_NO_SNAKE"))
      while ((1 << p.deal_shift) < per_xcd) ++p.deal_shift;
    if (const char* x = getenv("SWB_DEAL_SHIFT")) p.deal_shift = std::max(0, atoi(x));      // tests: short rounds on small batches
  }
  // Run lists, in 8-byte units.  A canvas row of s >= 2 visible spans costs 1 + s / 2 units and S convex sprites leave at most
  // 2 S - 1 spans in a row: (S + 1) units per row hold any scene of convex sprites even if no two rows fold into a run.  That is
  // what every list reserves to BEGIN with (round 5: ten sprites on a 60-row canvas needed 4.5 units per row) -- 133 KB per
  // environment on 12 sprites at 128x128, a tenth of it used.  swb_trim_run_lists() then cuts the lists' own (fixed) parts down to
  // 1.25 x the longest list the launches so far wrote and adds a shared ARENA from which a list that outgrows its part continues
  // by atomic bump (swb_params::run_cap); only a scene that finds the arena exhausted too is flagged (SWB_ENV_ERR_SPAN_OVERFLOW,
  // its frame short of a batch of rows) -- never silently.  A new pool (swb_set_pool / swb_sample_pool) restores the reservation.
  h->run_cap_worst = std::max(4, p.S + 1) * p.Hc + 1;
  p.run_cap = h->run_cap_worst;
  if (const char* x = getenv("SWB_RUN_CAP")) p.run_cap = std::max(8, atoi(x));      // tests
  if (const char* x = getenv("SWB_ARENA_UNITS")) h->arena_override = std::max(0, atoi(x));   // tests (0: no arena)
  const size_t NS = (size_t)p.N * p.S;
  int rc = 0;
  rc |= upload(&h->d_x, (const double*)nullptr, NS);
  rc |= upload(&h->d_y, (const double*)nullptr, NS);
  rc |= upload(&h->d_nspr, (const int32_t*)nullptr, p.N);
  rc |= upload(&h->d_entry, (const int32_t*)nullptr, p.N);
  rc |= upload(&h->d_step_count, (const int32_t*)nullptr, p.N);
  rc |= upload(&h->d_episode, (const int32_t*)nullptr, p.N);
  rc |= upload(&h->d_reset_next, (const uint8_t*)nullptr, p.N);
  if (rc) { swb_destroy(h); return SWB_ERR_HIP; }
  p.max_spans = SWB_MIN_SPANS;
  if (const char* ms = getenv("SWB_MAX_SPANS")) p.max_spans = atoi(ms) < SWB_MIN_SPANS ? SWB_MIN_SPANS : atoi(ms);
  p.span_cap = p.max_spans;
  if (const char* sc = getenv("SWB_LDS_SPAN_CAP")) p.span_cap = std::max(0, std::min(atoi(sc), p.max_spans));   // tests
  // visible-span lists: 3 per canvas row in registers, M in LDS, up to ovf_cap more in an HBM slot that a wave
  // takes only when a row of its batch needs it (allocated at the first launch, see ensure_overflow_slots)
  p.ovf_cap = std::min(p.Wc / 2 + 1, 32);
  p.ovf = nullptr;
  p.x = h->d_x; p.y = h->d_y; p.nspr = h->d_nspr; p.entry = h->d_entry; p.step_count = h->d_step_count;
  p.episode = h->d_episode; p.reset_next = h->d_reset_next;
  *out = h;
  return SWB_OK;
}

int swb_destroy(swb_handle h) {
  if (!h) return SWB_OK;
  (void)hipSetDevice(h->device);
  for (auto* list : {&h->events, &h->event_pool})
    for (auto& ev : *list) { (void)hipEventDestroy(ev.e0); (void)hipEventDestroy(ev.e1); (void)hipEventDestroy(ev.e2); }
  void* bufs[] = {h->d_shape_verts, h->d_shape_dmin, h->d_shape_off, h->d_h_xmin, h->d_h_cnt, h->d_h_tbl, h->d_h_pfx, h->d_v_tab,
                  h->d_v_pfx, h->d_v_end, h->d_p_n, h->d_p_x, h->d_p_y, h->d_p_xv, h->d_p_yv, h->d_p_scale, h->d_p_ca, h->d_p_sa,
                  h->d_p_shape, h->d_p_rgb, h->d_p_label, h->d_p_cell_label, h->d_ov_cell_label, h->d_p_attr, h->d_pool_base, h->d_pool_len, h->d_x, h->d_y, h->d_nspr,
                  h->d_entry, h->d_step_count, h->d_episode, h->d_reset_next, h->d_ovf, h->d_ovf_bitmap, h->d_p_angle, h->d_p_color, h->d_sampler,
                  h->d_ov_flag, h->d_ov_shape, h->d_ov_scale, h->d_ov_angle, h->d_ov_cpath, h->d_ov_label,
                  h->d_cost_cnt, h->d_cost_list, h->d_ccost_list, h->d_runs, h->d_rhdr, h->d_arena_head, h->d_env_state, h->d_band_y0, h->d_band_first, h->d_band_lo, h->d_cg_lo, h->d_cg_hi, h->d_v_break};
  for (void* b : bufs) if (b) (void)hipFree(b);
  delete h;
  return SWB_OK;
}

int swb_upload_shapes(swb_handle h, const double* verts, const int32_t* offsets, int32_t n_shapes) {
  if (!h || !verts || !offsets) return fail(SWB_ERR_INVALID, "null argument");
  if (n_shapes < 1 || n_shapes > SWB_MAX_SHAPES) return fail(SWB_ERR_INVALID, "n_shapes must be in [1, %d]", SWB_MAX_SHAPES);
  HIP_TRY(hipSetDevice(h->device));
  int maxv = 0;
  for (int i = 0; i < n_shapes; ++i) {
    const int n = offsets[i + 1] - offsets[i];
    if (n < 3 || n > SWB_MAX_SHAPE_VERTS) return fail(SWB_ERR_INVALID, "shape %d has %d vertices (3..%d supported)", i, n, SWB_MAX_SHAPE_VERTS);
    if (n > maxv) maxv = n;
  }
  // smallest distance between two vertices of each shape: the kernel's test for "no two vertices
  // of this sprite can land on one canvas pixel" (swb_kernels.hip.inc, build_all_edges)
  std::vector<double> dmin(n_shapes, 0.0);
  for (int i = 0; i < n_shapes; ++i) {
    double best = 1e300;
    for (int a = offsets[i]; a < offsets[i + 1]; ++a)
      for (int b = a + 1; b < offsets[i + 1]; ++b)
        best = std::min(best, std::hypot(verts[2 * a] - verts[2 * b], verts[2 * a + 1] - verts[2 * b + 1]));
    dmin[i] = best;
  }
  // (both small tables are padded to SWB_MAX_SHAPES entries: a cover wave loads them whole, lane i = entry i, beside its first
  // round of loads and looks a sprite's shape up with a cross-lane read -- not with a third dependent round trip to memory)
  dmin.resize(SWB_MAX_SHAPES, 0.0);
  std::vector<int32_t> off_padded(offsets, offsets + n_shapes + 1);
  off_padded.resize(SWB_MAX_SHAPES + 1, offsets[n_shapes]);
  if (upload(&h->d_shape_dmin, dmin.data(), dmin.size())) return SWB_ERR_HIP;
  h->p.shape_dmin = h->d_shape_dmin;
  if (upload(&h->d_shape_verts, verts, (size_t)offsets[n_shapes] * 2)) return SWB_ERR_HIP;
  if (upload(&h->d_shape_off, off_padded.data(), off_padded.size())) return SWB_ERR_HIP;
  h->p.shape_verts = h->d_shape_verts;
  h->p.shape_off = h->d_shape_off;
  h->p.max_verts = maxv;
  h->p.max_edges = maxv * h->p.S;                 // until a pool is installed (then: the pool's largest episode)
  h->shape_nverts.resize(n_shapes);
  for (int i = 0; i < n_shapes; ++i) h->shape_nverts[i] = offsets[i + 1] - offsets[i];
  h->have_shapes = true;
  return SWB_OK;
}

int swb_upload_resample(swb_handle h, int32_t axis, int32_t out_size, int32_t ksize, const int32_t* bounds,
                        const int32_t* coeffs) {
  if (!h || !bounds || !coeffs) return fail(SWB_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  const swb_params& p = h->p;
  if (axis == 0) {
    if (out_size != p.Wo) return fail(SWB_ERR_INVALID, "horizontal table has %d outputs, image width is %d", out_size, p.Wo);
    // prefix sums, de-duplicated: the interior columns share one vector
    std::vector<int32_t> xmin(out_size), cnt(out_size), tbl(out_size), pfx;
    std::vector<std::vector<int32_t>> uniq;
    std::vector<int32_t> uniq_off;
    for (int o = 0; o < out_size; ++o) {
      xmin[o] = bounds[2 * o]; cnt[o] = bounds[2 * o + 1];
      if (cnt[o] < 0 || cnt[o] > ksize || xmin[o] < 0 || xmin[o] + cnt[o] > p.Wc)
        return fail(SWB_ERR_INVALID, "bad horizontal bounds at %d", o);
      std::vector<int32_t> v(cnt[o] + 1, 0);
      for (int j = 0; j < cnt[o]; ++j) v[j + 1] = v[j] + coeffs[(size_t)o * ksize + j];
      int found = -1;
      for (size_t u = 0; u < uniq.size(); ++u) if (uniq[u] == v) { found = (int)u; break; }
      if (found < 0) { found = (int)uniq.size(); uniq.push_back(v); uniq_off.push_back((int32_t)pfx.size()); pfx.insert(pfx.end(), v.begin(), v.end()); }
      tbl[o] = uniq_off[found];
    }
    if (pfx.size() * 4 > 32 * 1024) return fail(SWB_ERR_INVALID, "horizontal prefix table too large (%zu entries)", pfx.size());
    if (upload(&h->d_h_xmin, xmin.data(), xmin.size()) || upload(&h->d_h_cnt, cnt.data(), cnt.size()) ||
        upload(&h->d_h_tbl, tbl.data(), tbl.size()) || upload(&h->d_h_pfx, pfx.data(), pfx.size()))
      return SWB_ERR_HIP;
    h->p.h_xmin = h->d_h_xmin; h->p.h_cnt = h->d_h_cnt; h->p.h_tbl = h->d_h_tbl; h->p.h_pfx = h->d_h_pfx;
    h->p.h_pfx_len = (int32_t)pfx.size();
    h->h_xmin_host = xmin; h->h_cnt_host = cnt;
    h->tables_dirty = true;
    h->have_h = true;
  } else if (axis == 1) {
    if (out_size != p.Ho) return fail(SWB_ERR_INVALID, "vertical table has %d outputs, image height is %d", out_size, p.Ho);
    // v_tab holds two tables of [Hc][SWB_VSLOTS]: output row r accumulates in slot r % 6 (first table,
    // kernels with VS = 6) or r % 8 (second table, VS = 8), so the in-flight rows never move between
    // accumulators; the rows in flight at a canvas row are consecutive, hence in distinct slots.
    const size_t tab_len = (size_t)p.Hc * SWB_VSLOTS;
    std::vector<int32_t> vend(out_size), vtab(2 * tab_len, 0);
    for (int r = 0; r < out_size; ++r) {
      const int ymin = bounds[2 * r], c = bounds[2 * r + 1];
      if (c < 1 || c > ksize || ymin < 0 || ymin + c > p.Hc) return fail(SWB_ERR_INVALID, "bad vertical bounds at %d", r);
      vend[r] = ymin + c - 1;
      if (r > 0 && (vend[r] < vend[r - 1] || ymin < bounds[2 * (r - 1)])) return fail(SWB_ERR_INVALID, "vertical windows are not monotone");
    }
    int used = 1;
    for (int y = 0; y < p.Hc; ++y) {
      int rf = 0;
      while (rf < out_size && vend[rf] < y) ++rf;     // rows completed before canvas row y
      for (int r = 0; r < out_size; ++r) {
        const int ymin = bounds[2 * r];
        if (y < ymin || y > vend[r]) continue;
        const int k = r - rf;
        if (k < 0 || k >= SWB_VSLOTS) return fail(SWB_ERR_INVALID, "more than %d output rows in flight at canvas row %d", SWB_VSLOTS, y);
        const int32_t cf = coeffs[(size_t)r * ksize + (y - ymin)];
        vtab[(size_t)y * SWB_VSLOTS + (r % 6)] = cf;           // only used when used <= 6
        vtab[tab_len + (size_t)y * SWB_VSLOTS + (r % 8)] = cf;
        if (k + 1 > used) used = k + 1;
      }
    }
    // prefix sums over canvas rows of both slot tables: the coefficient sums of a run of rows
    // [y1, y2] that lies between two row completions are pfx[y2 + 1] - pfx[y1]
    const size_t pfx_len = (size_t)(p.Hc + 1) * SWB_VSLOTS;
    std::vector<int32_t> vpfx(2 * pfx_len, 0);
    for (int t = 0; t < 2; ++t)
      for (int y = 0; y < p.Hc; ++y)
        for (int k = 0; k < SWB_VSLOTS; ++k)
          vpfx[t * pfx_len + (size_t)(y + 1) * SWB_VSLOTS + k] =
              vpfx[t * pfx_len + (size_t)y * SWB_VSLOTS + k] + vtab[t * tab_len + (size_t)y * SWB_VSLOTS + k];
    std::vector<int32_t> vend_padded(vend);
    vend_padded.push_back(0x7fffffff); vend_padded.push_back(0x7fffffff);       // the resample kernel loads one row ahead
    if (upload(&h->d_v_tab, vtab.data(), vtab.size()) || upload(&h->d_v_end, vend_padded.data(), vend_padded.size()) ||
        upload(&h->d_v_pfx, vpfx.data(), vpfx.size()))
      return SWB_ERR_HIP;
    h->p.v_tab = h->d_v_tab; h->p.v_end = h->d_v_end; h->p.v_pfx = h->d_v_pfx;
    h->vslots = used;
    h->v_end_host = vend;
    h->v_ymin_host.resize(out_size);
    for (int r = 0; r < out_size; ++r) h->v_ymin_host[r] = bounds[2 * r];
    h->tables_dirty = true;
    h->have_v = true;
  } else {
    return fail(SWB_ERR_INVALID, "axis must be 0 or 1");
  }
  return SWB_OK;
}

int swb_set_pool(swb_handle h, const swb_pool* pool) {
  if (!h || !pool) return fail(SWB_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  const int P = pool->n_entries, S = h->p.S, T = h->p.n_tasks, N = h->p.N;
  if (P < 1) return fail(SWB_ERR_INVALID, "pool is empty");
  for (int i = 0; i < P; ++i)
    if (pool->n_sprites[i] < 0 || pool->n_sprites[i] > S) return fail(SWB_ERR_INVALID, "pool entry %d has %d sprites (max %d)", i, pool->n_sprites[i], S);
  for (int i = 0; i < N; ++i)
    if (pool->pool_len[i] < 1 || pool->pool_base[i] < 0 || pool->pool_base[i] + pool->pool_len[i] > P)
      return fail(SWB_ERR_INVALID, "env %d: pool range [%d, +%d) outside the pool of %d", i, pool->pool_base[i], pool->pool_len[i], P);
  const size_t PS = (size_t)P * S;
  std::vector<uint32_t> rgb(PS);
  for (size_t i = 0; i < PS; ++i)
    rgb[i] = (uint32_t)pool->rgb[4 * i] | ((uint32_t)pool->rgb[4 * i + 1] << 8) | ((uint32_t)pool->rgb[4 * i + 2] << 16);
  for (size_t i = 0; i < PS; ++i)
    if (pool->shape[i] < 0 || (h->have_shapes && pool->shape[i] >= SWB_MAX_SHAPES)) return fail(SWB_ERR_INVALID, "bad shape index in pool");
  int rc = 0;
  rc |= upload(&h->d_p_n, pool->n_sprites, P);
  rc |= upload(&h->d_p_x, pool->x, PS); rc |= upload(&h->d_p_y, pool->y, PS);
  rc |= upload(&h->d_p_xv, pool->x_vel, PS); rc |= upload(&h->d_p_yv, pool->y_vel, PS);
  rc |= upload(&h->d_p_scale, pool->scale, PS); rc |= upload(&h->d_p_ca, pool->cos_a, PS); rc |= upload(&h->d_p_sa, pool->sin_a, PS);
  rc |= upload(&h->d_p_shape, pool->shape, PS);
  rc |= upload(&h->d_p_rgb, rgb.data(), PS);
  rc |= upload(&h->d_p_label, pool->label, (size_t)P * T * S);
  if (h->keyed) {
    if (!pool->cell_label) return fail(SWB_ERR_INVALID, "a task of this handle keys on position (swb_task::n_xcuts / n_ycuts): swb_pool::cell_label is required");
    rc |= upload(&h->d_p_cell_label, pool->cell_label, (size_t)P * T * S * SWB_MAX_CELLS);
  }
  rc |= upload(&h->d_p_attr, pool->attr_f32, PS);                      // (NULL: zeros = Python numbers)
  rc |= upload(&h->d_pool_base, pool->pool_base, N);
  rc |= upload(&h->d_pool_len, pool->pool_len, N);
  if (pool->angle) rc |= upload(&h->d_p_angle, pool->angle, PS);
  if (pool->color) rc |= upload(&h->d_p_color, pool->color, PS * 3);
  if (rc) return SWB_ERR_HIP;
  h->p.p_angle = pool->angle ? h->d_p_angle : nullptr;
  h->p.p_color = pool->color ? h->d_p_color : nullptr;
  swb_params& p = h->p;
  p.p_n = h->d_p_n; p.p_x = h->d_p_x; p.p_y = h->d_p_y; p.p_xv = h->d_p_xv; p.p_yv = h->d_p_yv;
  p.p_scale = h->d_p_scale; p.p_ca = h->d_p_ca; p.p_sa = h->d_p_sa; p.p_shape = h->d_p_shape; p.p_rgb = h->d_p_rgb;
  p.p_label = h->d_p_label; p.pool_base = h->d_pool_base; p.pool_len = h->d_pool_len;
  p.p_cell_label = h->keyed ? h->d_p_cell_label : nullptr;
  h->pool_entries = P;
  h->pool_sampled = false;
  // polygon vertices of the largest episode: sizes the per-wave edge records and centred paths in LDS
  if (h->have_shapes) {
    int most = 1;
    for (int e = 0; e < P; ++e) {
      int tot = 0;
      for (int s2 = 0; s2 < pool->n_sprites[e]; ++s2) {
        const int sh = pool->shape[(size_t)e * S + s2];
        if (sh >= (int)h->shape_nverts.size()) return fail(SWB_ERR_INVALID, "pool entry %d uses shape %d, %zu shapes uploaded", e, sh, h->shape_nverts.size());
        tot += h->shape_nverts[sh];
      }
      most = std::max(most, tot);
    }
    h->p.max_edges = (most + 3) & ~3;
  }
  HIP_TRY(hipMemset(h->d_reset_next, 1, N));      // environment.py:70
  HIP_TRY(hipMemset(h->d_episode, 0, sizeof(int32_t) * N));
  HIP_TRY(hipMemset(h->d_step_count, 0, sizeof(int32_t) * N));
  h->have_pool = true;
  if (int rc2 = restore_run_list_reservation(h)) return rc2;
  return SWB_OK;
}

int swb_sample_pool(swb_handle h, const swb_sampler* spec, int32_t n_entries, const int32_t* pool_base_host,
                    const int32_t* pool_len_host, uint64_t seed, uint64_t first_entry, void* stream) {
  if (!h || !spec || !pool_base_host || !pool_len_host) return fail(SWB_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  const int P = n_entries, S = h->p.S, T = h->p.n_tasks, N = h->p.N;
  if (P < 1) return fail(SWB_ERR_INVALID, "pool is empty");
  if (h->keyed) return fail(SWB_ERR_INVALID, "a task of this handle keys on position: its labels depend on where a sprite is drawn -- sample such "
                            "generators on the host (swb_set_pool with swb_pool::cell_label)");
  if (spec->n_groups < 1 || spec->n_groups > SWB_MAX_GROUPS) return fail(SWB_ERR_INVALID, "n_groups must be in [1, %d]", SWB_MAX_GROUPS);
  if (spec->shuffle < 0) return fail(SWB_ERR_INVALID, "shuffle must be >= 0");
  if (spec->n_alternatives < 0 || spec->n_alternatives > SWB_MAX_ALTERNATIVES)
    return fail(SWB_ERR_INVALID, "n_alternatives must be in [0, %d]", SWB_MAX_ALTERNATIVES);
  for (int i = 0; i < spec->n_alternatives; ++i) {
    const swb_alternative& alt = spec->alternatives[i];
    if (alt.n < 1 || alt.n > SWB_MAX_GROUPS) return fail(SWB_ERR_INVALID, "alternative %d: 1..%d groups", i, SWB_MAX_GROUPS);
    for (int g = 0; g < alt.n; ++g)
      if (alt.group[g] < 0 || alt.group[g] >= spec->n_groups) return fail(SWB_ERR_INVALID, "alternative %d: bad group index", i);
  }
  int max_total = 0;
  for (int g = 0; g < spec->n_groups; ++g) {
    const swb_sprite_group& grp = spec->groups[g];
    if (grp.count_min < 0 || grp.count_max < grp.count_min) return fail(SWB_ERR_INVALID, "group %d: bad sprite count range", g);
    if (grp.n_shapes < 1 || grp.n_shapes > SWB_MAX_CANDIDATES)
      return fail(SWB_ERR_INVALID, "group %d: 1..%d shape candidates", g, SWB_MAX_CANDIDATES);
    for (const swb_factor* f = grp.factors; f != grp.factors + SWB_N_FACTORS; ++f) {
      if (f->kind < SWB_FACTOR_UNIFORM_F32 || f->kind > SWB_FACTOR_DISCRETE) return fail(SWB_ERR_INVALID, "group %d: bad factor kind", g);
      if (f->kind == SWB_FACTOR_DISCRETE && (f->n < 1 || f->n > SWB_MAX_CANDIDATES))
        return fail(SWB_ERR_INVALID, "group %d: Discrete factors take 1..%d candidates", g, SWB_MAX_CANDIDATES);
    }
    if (grp.factors[SWB_F_X].kind != SWB_FACTOR_UNIFORM_F32 || grp.factors[SWB_F_Y].kind != SWB_FACTOR_UNIFORM_F32)
      return fail(SWB_ERR_INVALID, "group %d: x and y must be float32 Continuous factors", g);
    const swb_factor& ang = grp.factors[SWB_F_ANGLE];
    if (ang.kind == SWB_FACTOR_UNIFORM_F32 ||
        (ang.kind == SWB_FACTOR_UNIFORM_INT && !(ang.lo >= 0.0 && ang.hi <= 360.0 && ang.lo <= ang.hi)))
      return fail(SWB_ERR_INVALID, "group %d: angle must be Discrete or integer degrees within [0, 360]", g);
    if (grp.n_holdouts < 0 || grp.n_holdouts > SWB_MAX_HOLDOUTS) return fail(SWB_ERR_INVALID, "group %d: at most %d hold-outs", g, SWB_MAX_HOLDOUTS);
    for (int i = 0; i < grp.n_holdouts; ++i)
      if (grp.holdouts[i].box_mask == 0 || (grp.holdouts[i].box_mask & ~grp.holdouts[i].redraw_mask) ||
          (grp.holdouts[i].redraw_mask >> SWB_N_FACTORS))
        return fail(SWB_ERR_INVALID, "group %d: hold-out keys must be a non-empty subset of the redrawn keys", g);
    if (spec->color_map == 1 && (grp.factors[SWB_F_C0].kind == SWB_FACTOR_UNIFORM_INT || grp.factors[SWB_F_C1].kind == SWB_FACTOR_UNIFORM_INT ||
                                 grp.factors[SWB_F_C2].kind == SWB_FACTOR_UNIFORM_INT))
      return fail(SWB_ERR_INVALID, "group %d: hsv colours must be float factors", g);
    for (int i = 0; i < grp.n_shapes; ++i)
      if (grp.shapes[i] < 0 || grp.shapes[i] >= SWB_MAX_SHAPES) return fail(SWB_ERR_INVALID, "group %d: bad shape index", g);
    max_total += grp.count_max;
  }
  if (spec->n_alternatives > 0) {
    max_total = 0;
    for (int i = 0; i < spec->n_alternatives; ++i) {
      int tot = 0;
      for (int g = 0; g < spec->alternatives[i].n; ++g) tot += spec->groups[spec->alternatives[i].group[g]].count_max;
      if (tot > max_total) max_total = tot;
    }
  }
  if (max_total > S) return fail(SWB_ERR_INVALID, "sampler can emit %d sprites (max_sprites %d)", max_total, S);
  if (h->have_shapes) {       // the most polygon vertices an episode of this sampler can have (LDS sizing)
    auto group_verts = [&](const swb_sprite_group& grp) {
      int mv = 0;
      for (int i = 0; i < grp.n_shapes; ++i)
        if (grp.shapes[i] < (int)h->shape_nverts.size()) mv = std::max(mv, h->shape_nverts[grp.shapes[i]]);
      return mv * grp.count_max;
    };
    int most = 0;
    if (spec->n_alternatives > 0) {
      for (int i = 0; i < spec->n_alternatives; ++i) {
        int tot = 0;
        for (int g = 0; g < spec->alternatives[i].n; ++g) tot += group_verts(spec->groups[spec->alternatives[i].group[g]]);
        most = std::max(most, tot);
      }
    } else {
      for (int g = 0; g < spec->n_groups; ++g) most += group_verts(spec->groups[g]);
    }
    h->p.max_edges = (std::max(most, 1) + 3) & ~3;
  }
  for (int i = 0; i < N; ++i)
    if (pool_len_host[i] < 1 || pool_base_host[i] < 0 || pool_base_host[i] + pool_len_host[i] > P)
      return fail(SWB_ERR_INVALID, "env %d: pool range [%d, +%d) outside the pool of %d", i, pool_base_host[i], pool_len_host[i], P);
  const size_t PS = (size_t)P * S;
  int rc = 0;
  if (h->pool_entries != P || !h->d_p_angle || !h->d_p_color) {
    rc |= upload<int32_t>(&h->d_p_n, nullptr, P);
    rc |= upload<double>(&h->d_p_x, nullptr, PS); rc |= upload<double>(&h->d_p_y, nullptr, PS);
    rc |= upload<double>(&h->d_p_xv, nullptr, PS); rc |= upload<double>(&h->d_p_yv, nullptr, PS);
    rc |= upload<double>(&h->d_p_scale, nullptr, PS); rc |= upload<double>(&h->d_p_ca, nullptr, PS);
    rc |= upload<double>(&h->d_p_sa, nullptr, PS);
    rc |= upload<int32_t>(&h->d_p_shape, nullptr, PS); rc |= upload<uint32_t>(&h->d_p_rgb, nullptr, PS);
    rc |= upload<int8_t>(&h->d_p_label, nullptr, (size_t)P * T * S);
    rc |= upload<uint8_t>(&h->d_p_attr, nullptr, PS);
    rc |= upload<double>(&h->d_p_angle, nullptr, PS); rc |= upload<double>(&h->d_p_color, nullptr, PS * 3);
  }
  rc |= upload(&h->d_pool_base, pool_base_host, N);
  rc |= upload(&h->d_pool_len, pool_len_host, N);
  rc |= upload(&h->d_sampler, spec, 1);
  if (rc) return SWB_ERR_HIP;
  h->pool_entries = P;
  swb_params& p = h->p;
  p.p_n = h->d_p_n; p.p_x = h->d_p_x; p.p_y = h->d_p_y; p.p_xv = h->d_p_xv; p.p_yv = h->d_p_yv;
  p.p_scale = h->d_p_scale; p.p_ca = h->d_p_ca; p.p_sa = h->d_p_sa; p.p_shape = h->d_p_shape; p.p_rgb = h->d_p_rgb;
  p.p_label = h->d_p_label; p.pool_base = h->d_pool_base; p.pool_len = h->d_pool_len;
  p.p_angle = h->d_p_angle; p.p_color = h->d_p_color;
  swb_sampler_args a{h->d_sampler, P, S, T, seed, first_entry, nullptr, nullptr, nullptr, N, h->d_p_n, h->d_p_x, h->d_p_y, h->d_p_xv, h->d_p_yv, h->d_p_scale,
                     h->d_p_ca, h->d_p_sa, h->d_p_angle, h->d_p_color, h->d_p_shape, h->d_p_rgb, h->d_p_label, h->d_p_attr};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(swb_sample_pool_kernel, dim3((P + 255) / 256), dim3(256), 0, st, a);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemsetAsync(h->d_reset_next, 1, N, st));      // environment.py:70
  HIP_TRY(hipMemsetAsync(h->d_episode, 0, sizeof(int32_t) * N, st));
  HIP_TRY(hipMemsetAsync(h->d_step_count, 0, sizeof(int32_t) * N, st));
  h->have_pool = true;
  if (int rc2 = restore_run_list_reservation(h)) return rc2;
  h->pool_sampled = true;
  h->pool_uniform = true;
  for (int i = 0; i < N; ++i)
    if (pool_len_host[i] != pool_len_host[0] || pool_base_host[i] != i * pool_len_host[0]) h->pool_uniform = false;
  return SWB_OK;
}

int swb_resample_pool(swb_handle h, uint64_t seed, uint64_t first_entry, void* stream) {
  if (!h) return fail(SWB_ERR_INVALID, "null handle");
  if (!h->have_pool || !h->pool_sampled) return fail(SWB_ERR_STATE, "swb_sample_pool has not been called");
  if (!h->pool_uniform)
    return fail(SWB_ERR_STATE, "swb_resample_pool needs the env-major layout pool_base[n] = n * pool_len, equal pool_len");
  HIP_TRY(hipSetDevice(h->device));
  const int P = h->pool_entries, S = h->p.S, T = h->p.n_tasks, N = h->p.N;
  swb_sampler_args a{h->d_sampler, P, S, T, seed, first_entry, h->d_entry, h->d_pool_base, h->d_pool_len, N, h->d_p_n,
                     h->d_p_x, h->d_p_y, h->d_p_xv, h->d_p_yv, h->d_p_scale, h->d_p_ca, h->d_p_sa, h->d_p_angle,
                     h->d_p_color, h->d_p_shape, h->d_p_rgb, h->d_p_label, h->d_p_attr};
  hipLaunchKernelGGL(swb_sample_pool_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return SWB_OK;
}

int swb_get_pool(swb_handle h, const swb_pool* pool) {
  if (!h || !pool) return fail(SWB_ERR_INVALID, "null argument");
  if (!h->have_pool) return fail(SWB_ERR_STATE, "no pool on the device");
  if (pool->n_entries != h->pool_entries) return fail(SWB_ERR_INVALID, "pool has %d entries, not %d", h->pool_entries, pool->n_entries);
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipDeviceSynchronize());
  const int P = h->pool_entries, S = h->p.S, T = h->p.n_tasks, N = h->p.N;
  const size_t PS = (size_t)P * S;
#define SWB_D2H(dst, src, count) \
  if (dst) HIP_TRY(hipMemcpy((void*)(dst), (src), (count) * sizeof(*(src)), hipMemcpyDeviceToHost))
  SWB_D2H(pool->n_sprites, h->d_p_n, (size_t)P);
  SWB_D2H(pool->x, h->d_p_x, PS); SWB_D2H(pool->y, h->d_p_y, PS);
  SWB_D2H(pool->x_vel, h->d_p_xv, PS); SWB_D2H(pool->y_vel, h->d_p_yv, PS);
  SWB_D2H(pool->scale, h->d_p_scale, PS); SWB_D2H(pool->cos_a, h->d_p_ca, PS); SWB_D2H(pool->sin_a, h->d_p_sa, PS);
  SWB_D2H(pool->shape, h->d_p_shape, PS);
  SWB_D2H(pool->label, h->d_p_label, (size_t)P * T * S);
  SWB_D2H(pool->attr_f32, h->d_p_attr, PS);
  SWB_D2H(pool->pool_base, h->d_pool_base, (size_t)N); SWB_D2H(pool->pool_len, h->d_pool_len, (size_t)N);
  if (h->p.p_angle) SWB_D2H(pool->angle, h->d_p_angle, PS);
  if (h->p.p_color) SWB_D2H(pool->color, h->d_p_color, PS * 3);
  if (pool->rgb) {
    std::vector<uint32_t> rgb(PS);
    HIP_TRY(hipMemcpy(rgb.data(), h->d_p_rgb, PS * 4, hipMemcpyDeviceToHost));
    uint8_t* dst = const_cast<uint8_t*>(pool->rgb);
    for (size_t i = 0; i < PS; ++i) {
      dst[4 * i] = rgb[i] & 255; dst[4 * i + 1] = (rgb[i] >> 8) & 255; dst[4 * i + 2] = (rgb[i] >> 16) & 255; dst[4 * i + 3] = 0;
    }
  }
#undef SWB_D2H
  return SWB_OK;
}

int swb_reset_all(swb_handle h, void* stream) {
  if (!h) return fail(SWB_ERR_INVALID, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipMemsetAsync(h->d_reset_next, 1, h->p.N, (hipStream_t)stream));
  return SWB_OK;
}

int swb_step(swb_handle h, const void* actions_dev, const swb_outputs* out, void* stream) {
  if (!h) return fail(SWB_ERR_INVALID, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  return launch(h, actions_dev, out, 0, (hipStream_t)stream);
}

int swb_render(swb_handle h, uint8_t* obs_dev, void* stream) {
  if (!h || !obs_dev) return fail(SWB_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  swb_outputs out;
  memset(&out, 0, sizeof(out));
  out.obs = obs_dev;
  return launch(h, nullptr, &out, 1, (hipStream_t)stream);
}

int swb_evaluate(swb_handle h, uint8_t* success_dev, void* stream) {
  if (!h || !success_dev) return fail(SWB_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  swb_outputs out;
  memset(&out, 0, sizeof(out));
  out.success = success_dev;
  return launch(h, nullptr, &out, 2, (hipStream_t)stream);
}

int swb_trim_run_lists(swb_handle h, int32_t* run_cap_out, void* stream) {
  if (!h) return fail(SWB_ERR_INVALID, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  if (run_cap_out) *run_cap_out = h->p.run_cap;
  if (h->lists_trimmed || getenv("SWB_RUN_CAP") || getenv("SWB_NO_TRIM")) return SWB_OK;   // (done already / a test pinned the capacity)
  if (!h->d_runs && h->p.AA == 1 && (h->p.Wo + 63) / 64 == 1) return SWB_OK;      // (the cover kernel paints the frame: no lists at all)
  if (!h->d_runs || !h->lists_valid) return fail(SWB_ERR_STATE, "swb_trim_run_lists: the last launch listed no runs (step or render first)");
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  // the longest list of the last launch: its end is in the header (no list has left its fixed part: it holds any scene)
  const swb_params& p = h->p;
  std::vector<uint32_t> hdr((size_t)p.N * SWB_RHDR_DWORDS);
  HIP_TRY(hipMemcpy(hdr.data(), h->d_rhdr, hdr.size() * 4, hipMemcpyDeviceToHost));
  uint32_t longest = 0;
  for (int n = 0; n < p.N; ++n)
    for (int g = 0; g < p.ncg; ++g) longest = std::max(longest, hdr[(size_t)n * SWB_RHDR_DWORDS + SWB_RHDR_GROUPS + g * SWB_RHDR_GSTRIDE]);
  if (longest >= (uint32_t)p.run_cap) return SWB_OK;                 // (cannot be: keep what there is)
  const int want = std::max(p.Hc / 2, (int)(longest + longest / 4 + 16)) + 1;
  if (want >= p.run_cap) return SWB_OK;                              // nothing to gain
  if (int rc = drop_run_lists(h)) return rc;
  h->p.run_cap = want;
  h->lists_trimmed = true;
  if (int rc = ensure_handoff_tables(h)) return rc;
  if (run_cap_out) *run_cap_out = h->p.run_cap;
  return SWB_OK;
}

int swb_factors(swb_handle h, double* factors_dev, void* stream) {
  if (!h || !factors_dev) return fail(SWB_ERR_INVALID, "null argument");
  if (!h->have_pool) return fail(SWB_ERR_STATE, "swb_set_pool has not been called");
  HIP_TRY(hipSetDevice(h->device));
  const int total = h->p.N * h->p.S;
  hipLaunchKernelGGL(swb_factors_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->p, factors_dev);
  HIP_TRY(hipGetLastError());
  return SWB_OK;
}

int swb_get_state(swb_handle h, const swb_state* st, void* stream) {
  if (!h || !st) return fail(SWB_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  const size_t N = h->p.N, NS = N * h->p.S;
  if (st->x) HIP_TRY(hipMemcpy(st->x, h->d_x, NS * 8, hipMemcpyDeviceToHost));
  if (st->y) HIP_TRY(hipMemcpy(st->y, h->d_y, NS * 8, hipMemcpyDeviceToHost));
  if (st->n_sprites) HIP_TRY(hipMemcpy(st->n_sprites, h->d_nspr, N * 4, hipMemcpyDeviceToHost));
  if (st->pool_entry) HIP_TRY(hipMemcpy(st->pool_entry, h->d_entry, N * 4, hipMemcpyDeviceToHost));
  if (st->step_count) HIP_TRY(hipMemcpy(st->step_count, h->d_step_count, N * 4, hipMemcpyDeviceToHost));
  if (st->reset_next) HIP_TRY(hipMemcpy(st->reset_next, h->d_reset_next, N, hipMemcpyDeviceToHost));
  if (st->episode) HIP_TRY(hipMemcpy(st->episode, h->d_episode, N * 4, hipMemcpyDeviceToHost));
  return SWB_OK;
}

int swb_get_env_state(swb_handle h, int32_t env, int32_t* out5, void* stream) {
  if (!h || !out5) return fail(SWB_ERR_INVALID, "null argument");
  if (env < 0 || env >= h->p.N) return fail(SWB_ERR_INVALID, "environment %d out of range", env);
  HIP_TRY(hipSetDevice(h->device));
  // (the N = 1 drop-in asks for this several times per step: one gather kernel and ONE 20-byte copy, stream-ordered behind
  // the steps, instead of a stream synchronisation and five 4-byte copies)
  if (!h->d_env_state && upload(&h->d_env_state, (const int32_t*)nullptr, 8)) return SWB_ERR_HIP;
  hipLaunchKernelGGL(swb_env_state_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, h->p, env, h->d_env_state);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(out5, h->d_env_state, 5 * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return SWB_OK;
}

int swb_get_sprite_types(swb_handle h, int32_t env, int32_t sprite, int32_t* flags, void* stream) {
  if (!h || !flags) return fail(SWB_ERR_INVALID, "null argument");
  if (!h->have_pool) return fail(SWB_ERR_STATE, "no pool on the device");
  if (env < 0 || env >= h->p.N) return fail(SWB_ERR_INVALID, "environment %d out of range", env);
  if (sprite < 0 || sprite >= h->p.S) return fail(SWB_ERR_INVALID, "sprite %d out of range", sprite);
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  int32_t en = 0;
  uint8_t f = 0;
  HIP_TRY(hipMemcpy(&en, h->d_entry + env, 4, hipMemcpyDeviceToHost));
  if (en < 0 || en >= h->pool_entries) return fail(SWB_ERR_STATE, "environment %d has not been reset yet", env);
  HIP_TRY(hipMemcpy(&f, h->d_p_attr + (size_t)en * h->p.S + sprite, 1, hipMemcpyDeviceToHost));
  *flags = f;
  return SWB_OK;
}

int swb_set_positions(swb_handle h, const double* x_host, const double* y_host, void* stream) {
  if (!h || !x_host || !y_host) return fail(SWB_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  const size_t NS = (size_t)h->p.N * h->p.S;
  HIP_TRY(hipMemcpy(h->d_x, x_host, NS * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->d_y, y_host, NS * 8, hipMemcpyHostToDevice));
  return SWB_OK;
}

// ---------------------------------------------------------------------------------------------------
// Sprite attribute setters (sprite.py:152-175).  The arithmetic is matplotlib's, restated on the host:
//   Affine2D.rotate(theta)   -> [[a, -b, 0], [b, a, 0]] with a = cos(theta), b = sin(theta)   (transforms.py)
//   Affine2D.scale(s)        -> [[s, 0, 0], [0, s, 0]]
//   transform_path           -> _path.h affine_transform_2d: t0 = sx*x; t1 = shx*y; x' = t0 + t1 + tx  (no FMA;
//                               this file is compiled with -ffp-contract=off)
// ---------------------------------------------------------------------------------------------------
static void affine2(double sx, double shx, double shy, double sy, int n, const double* in, double* out) {
  for (int i = 0; i < n; ++i) {
    const double x = in[2 * i], y = in[2 * i + 1];
    double t0 = sx * x, t1 = shx * y;
    const double ox = t0 + t1 + 0.0;
    t0 = shy * x; t1 = sy * y;
    out[2 * i] = ox;
    out[2 * i + 1] = t0 + t1 + 0.0;
  }
}
static double deg2rad(double d) { return d * (3.14159265358979323846 / 180.0); }   // math.radians

int swb_sprite_path_op(int32_t attr, double a, double b, int32_t n, const double* in_xy, double* out_xy) {
  if (!in_xy || !out_xy || n < 0) return fail(SWB_ERR_INVALID, "bad argument");
  if (attr == SWB_ATTR_SHAPE) {          // _reset_centered_path: scale(a) then rotate_deg(b): M = R * S
    const double th = deg2rad(b), ca = std::cos(th), sa = std::sin(th);
    affine2(ca * a, -sa * a, sa * a, ca * a, n, in_xy, out_xy);
  } else if (attr == SWB_ATTR_ANGLE) {   // rotate_deg(a - b)
    const double th = deg2rad(a - b), ca = std::cos(th), sa = std::sin(th);
    affine2(ca, -sa, sa, ca, n, in_xy, out_xy);
  } else if (attr == SWB_ATTR_SCALE) {   // scale(a - b): the reference's setter scales by the DIFFERENCE (sprite.py:173)
    const double f = a - b;
    affine2(f, 0.0, 0.0, f, n, in_xy, out_xy);
  } else {
    return fail(SWB_ERR_INVALID, "unknown sprite attribute %d", attr);
  }
  return SWB_OK;
}

namespace {

// Host copy of one environment's override record.
struct env_override {
  std::vector<int32_t> shape;
  std::vector<double> scale, angle, cpath;   // cpath: [S][SWB_MAX_SHAPE_VERTS][2]
  std::vector<int8_t> label;                 // [T][S]
};

int ov_allocate(swb_engine* h) {
  if (h->d_ov_flag) return 0;
  const size_t N = h->p.N, NS = N * h->p.S, T = h->p.n_tasks;
  int rc = 0;
  rc |= upload<int32_t>(&h->d_ov_shape, nullptr, NS);
  rc |= upload<double>(&h->d_ov_scale, nullptr, NS);
  rc |= upload<double>(&h->d_ov_angle, nullptr, NS);
  rc |= upload<int8_t>(&h->d_ov_label, nullptr, NS * T);
  rc |= upload<double>(&h->d_ov_cpath, nullptr, NS * SWB_MAX_SHAPE_VERTS * 2);
  if (h->keyed) rc |= upload<int8_t>(&h->d_ov_cell_label, nullptr, NS * T * SWB_MAX_CELLS);
  rc |= upload<uint8_t>(&h->d_ov_flag, nullptr, N);       // last: its presence switches the engine to the OV kernels
  if (rc) return SWB_ERR_HIP;
  swb_params& p = h->p;
  p.ov_flag = h->d_ov_flag; p.ov_shape = h->d_ov_shape; p.ov_scale = h->d_ov_scale; p.ov_angle = h->d_ov_angle;
  p.ov_label = h->d_ov_label; p.ov_cpath = h->d_ov_cpath;
  p.ov_cell_label = h->d_ov_cell_label;
  return 0;
}

// The record of environment `env`: its override arrays when the flag is set, else built from its pool entry
// (fresh centred paths, sprite.py:96-101).  `n` receives the episode's sprite count.
// `for_write`: a setter needs a running episode; reading the sprites of an episode that has just ended (the
// reference's sprites stay readable until the next reset) is fine.
int ov_fetch(swb_engine* h, int env, env_override* r, int* n, bool* was_set, bool for_write = true) {
  const int S = h->p.S, T = h->p.n_tasks;
  r->shape.assign(S, 0); r->scale.assign(S, 1.0); r->angle.assign(S, 0.0);
  r->cpath.assign((size_t)S * SWB_MAX_SHAPE_VERTS * 2, 0.0); r->label.assign((size_t)T * S, 0);
  uint8_t flag = 0, rn = 0;
  int32_t en = 0, ns = 0;
  if (h->d_ov_flag) HIP_TRY(hipMemcpy(&flag, h->d_ov_flag + env, 1, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(&rn, h->d_reset_next + env, 1, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(&en, h->d_entry + env, 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(&ns, h->d_nspr + env, 4, hipMemcpyDeviceToHost));
  int32_t ep = 0;
  HIP_TRY(hipMemcpy(&ep, h->d_episode + env, 4, hipMemcpyDeviceToHost));
  if (ep == 0) return fail(SWB_ERR_STATE, "environment %d has not been reset yet: it has no sprites", env);
  if (rn && for_write)
    return fail(SWB_ERR_STATE, "environment %d is about to reset (its episode ended): its sprites are gone at the next step", env);
  *n = ns;
  *was_set = flag != 0;
  const size_t o = (size_t)env * S, pe = (size_t)en * S;
  if (flag) {
    HIP_TRY(hipMemcpy(r->shape.data(), h->d_ov_shape + o, S * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(r->scale.data(), h->d_ov_scale + o, S * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(r->angle.data(), h->d_ov_angle + o, S * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(r->label.data(), h->d_ov_label + o * T, (size_t)T * S, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(r->cpath.data(), h->d_ov_cpath + o * SWB_MAX_SHAPE_VERTS * 2, (size_t)S * SWB_MAX_SHAPE_VERTS * 16,
                      hipMemcpyDeviceToHost));
    return 0;
  }
  if (!h->p.p_angle) return fail(SWB_ERR_STATE, "the pool carries no angle column (swb_pool.angle)");
  std::vector<double> ca(S), sa(S);
  HIP_TRY(hipMemcpy(r->shape.data(), h->d_p_shape + pe, S * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(r->scale.data(), h->d_p_scale + pe, S * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(r->angle.data(), h->d_p_angle + pe, S * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(ca.data(), h->d_p_ca + pe, S * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(sa.data(), h->d_p_sa + pe, S * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(r->label.data(), h->d_p_label + pe * T, (size_t)T * S, hipMemcpyDeviceToHost));
  std::vector<double> sv((size_t)h->p.max_verts * 2 + 2);
  std::vector<int32_t> off(h->shape_nverts.size() + 1);
  HIP_TRY(hipMemcpy(off.data(), h->d_shape_off, off.size() * 4, hipMemcpyDeviceToHost));
  for (int s2 = 0; s2 < ns; ++s2) {
    const int sh = r->shape[s2];
    if (sh < 0 || sh >= (int)h->shape_nverts.size()) return fail(SWB_ERR_STATE, "pool entry %d uses shape %d", en, sh);
    const int nv = h->shape_nverts[sh];
    HIP_TRY(hipMemcpy(sv.data(), h->d_shape_verts + 2 * (size_t)off[sh], (size_t)nv * 16, hipMemcpyDeviceToHost));
    // the kernel's centered_vertex with the pool's cos / sin (math.cos(math.radians(angle)), computed by the caller)
    affine2(ca[s2] * r->scale[s2], -sa[s2] * r->scale[s2], sa[s2] * r->scale[s2], ca[s2] * r->scale[s2], nv, sv.data(),
            r->cpath.data() + (size_t)s2 * SWB_MAX_SHAPE_VERTS * 2);
  }
  return 0;
}

int ov_store(swb_engine* h, int env, const env_override& r) {
  const int S = h->p.S, T = h->p.n_tasks;
  const size_t o = (size_t)env * S;
  HIP_TRY(hipMemcpy(h->d_ov_shape + o, r.shape.data(), S * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->d_ov_scale + o, r.scale.data(), S * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->d_ov_angle + o, r.angle.data(), S * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->d_ov_label + o * T, r.label.data(), (size_t)T * S, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->d_ov_cpath + o * SWB_MAX_SHAPE_VERTS * 2, r.cpath.data(), (size_t)S * SWB_MAX_SHAPE_VERTS * 16,
                    hipMemcpyHostToDevice));
  const uint8_t one = 1;
  HIP_TRY(hipMemcpy(h->d_ov_flag + env, &one, 1, hipMemcpyHostToDevice));
  return 0;
}

}  // namespace

int swb_set_sprite_attr(swb_handle h, int32_t env, int32_t sprite, int32_t attr, double value, const double* delta,
                        const int8_t* label, void* stream) {
  if (!h) return fail(SWB_ERR_INVALID, "null handle");
  if (!h->have_pool || !h->have_shapes) return fail(SWB_ERR_STATE, "no pool / shape table installed");
  if (env < 0 || env >= h->p.N) return fail(SWB_ERR_INVALID, "environment %d out of range", env);
  if (attr < SWB_ATTR_SHAPE || attr > SWB_ATTR_SCALE) return fail(SWB_ERR_INVALID, "unknown sprite attribute %d", attr);
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  env_override r;
  int n = 0;
  bool was_set = false;
  if (int rc = ov_fetch(h, env, &r, &n, &was_set)) return rc;
  if (sprite < 0 || sprite >= n) return fail(SWB_ERR_INVALID, "environment %d has %d sprites: sprite %d out of range", env, n, sprite);
  const int S = h->p.S, T = h->p.n_tasks;
  double* path = r.cpath.data() + (size_t)sprite * SWB_MAX_SHAPE_VERTS * 2;
  const int nv_old = h->shape_nverts[r.shape[sprite]];
  std::vector<double> out((size_t)SWB_MAX_SHAPE_VERTS * 2, 0.0);
  if (attr == SWB_ATTR_SHAPE) {                        // sprite.py:152-155: _shape = s; _reset_centered_path()
    const int sh = (int)value;
    if ((double)sh != value || sh < 0 || sh >= (int)h->shape_nverts.size()) return fail(SWB_ERR_INVALID, "bad shape index %g", value);
    const int nv = h->shape_nverts[sh];
    std::vector<int32_t> off(h->shape_nverts.size() + 1);
    HIP_TRY(hipMemcpy(off.data(), h->d_shape_off, off.size() * 4, hipMemcpyDeviceToHost));
    std::vector<double> sv((size_t)nv * 2);
    HIP_TRY(hipMemcpy(sv.data(), h->d_shape_verts + 2 * (size_t)off[sh], (size_t)nv * 16, hipMemcpyDeviceToHost));
    if (int rc = swb_sprite_path_op(SWB_ATTR_SHAPE, r.scale[sprite], r.angle[sprite], nv, sv.data(), out.data())) return rc;
    r.shape[sprite] = sh;
  } else if (attr == SWB_ATTR_ANGLE) {                 // :161-165: rotate_deg(a - _angle) applied to the current path
    const double a = delta ? *delta : value, b = delta ? 0.0 : r.angle[sprite];
    if (int rc = swb_sprite_path_op(SWB_ATTR_ANGLE, a, b, nv_old, path, out.data())) return rc;
    r.angle[sprite] = value;
  } else {                                             // :171-175: scale(s - _scale) applied to the current path
    const double a = delta ? *delta : value, b = delta ? 0.0 : r.scale[sprite];
    if (int rc = swb_sprite_path_op(SWB_ATTR_SCALE, a, b, nv_old, path, out.data())) return rc;
    r.scale[sprite] = value;
  }
  memcpy(path, out.data(), out.size() * sizeof(double));
  if (label) for (int t = 0; t < T; ++t) r.label[(size_t)t * S + sprite] = label[t];
  // LDS of a wave is sized by the most polygon vertices an episode can have: a new shape may raise it
  int tot = 0;
  for (int s2 = 0; s2 < n; ++s2) tot += h->shape_nverts[r.shape[s2]];
  h->p.max_edges = std::max(h->p.max_edges, (tot + 3) & ~3);
  if (int rc = ov_allocate(h)) return rc;
  if (!was_set && h->keyed) {                          // the episode's per-cell labels: from its pool entry, until
    int32_t en = 0;                                    // swb_set_sprite_cell_labels replaces the sprite's
    HIP_TRY(hipMemcpy(&en, h->d_entry + env, 4, hipMemcpyDeviceToHost));
    const size_t blk = (size_t)T * S * SWB_MAX_CELLS;
    HIP_TRY(hipMemcpy(h->d_ov_cell_label + (size_t)env * blk, h->d_p_cell_label + (size_t)en * blk, blk, hipMemcpyDeviceToDevice));
  }
  return ov_store(h, env, r);
}

int swb_set_sprite_cell_labels(swb_handle h, int32_t env, int32_t sprite, const int8_t* cells, void* stream) {
  if (!h || !cells) return fail(SWB_ERR_INVALID, "null argument");
  if (!h->keyed) return fail(SWB_ERR_INVALID, "no task of this handle keys on position");
  if (env < 0 || env >= h->p.N || sprite < 0 || sprite >= h->p.S) return fail(SWB_ERR_INVALID, "environment %d / sprite %d out of range", env, sprite);
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  uint8_t flag = 0;
  if (h->d_ov_flag) HIP_TRY(hipMemcpy(&flag, h->d_ov_flag + env, 1, hipMemcpyDeviceToHost));
  if (!flag) return fail(SWB_ERR_STATE, "swb_set_sprite_cell_labels follows swb_set_sprite_attr on the same environment");
  const int S = h->p.S, T = h->p.n_tasks;
  for (int t = 0; t < T; ++t)
    HIP_TRY(hipMemcpy(h->d_ov_cell_label + (((size_t)env * T + t) * S + sprite) * SWB_MAX_CELLS, cells + (size_t)t * SWB_MAX_CELLS,
                      SWB_MAX_CELLS, hipMemcpyHostToDevice));
  return SWB_OK;
}

int swb_get_sprite(swb_handle h, int32_t env, int32_t sprite, int32_t* shape, double* angle, double* scale, int32_t* n_verts,
                   double* path_xy, void* stream) {
  if (!h) return fail(SWB_ERR_INVALID, "null handle");
  if (!h->have_pool || !h->have_shapes) return fail(SWB_ERR_STATE, "no pool / shape table installed");
  if (env < 0 || env >= h->p.N) return fail(SWB_ERR_INVALID, "environment %d out of range", env);
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  env_override r;
  int n = 0;
  bool was_set = false;
  if (int rc = ov_fetch(h, env, &r, &n, &was_set, false)) return rc;
  if (sprite < 0 || sprite >= n) return fail(SWB_ERR_INVALID, "environment %d has %d sprites: sprite %d out of range", env, n, sprite);
  const int nv = h->shape_nverts[r.shape[sprite]];
  if (shape) *shape = r.shape[sprite];
  if (angle) *angle = r.angle[sprite];
  if (scale) *scale = r.scale[sprite];
  if (n_verts) *n_verts = nv;
  if (path_xy) memcpy(path_xy, r.cpath.data() + (size_t)sprite * SWB_MAX_SHAPE_VERTS * 2, (size_t)nv * 16);
  return SWB_OK;
}

int swb_variant(swb_handle h, swb_variant_info* out) {
  if (!h || !out) return fail(SWB_ERR_INVALID, "null argument");
  const variant* v = pick_variant(h->p.Wc);
  if (!v) return fail(SWB_ERR_INVALID, "canvas %dx%d not supported", h->p.Wc, h->p.Hc);
  int vs = 0;
  (void)pick_resample(h->p.AA, h->vslots, &vs);
  out->nw = v->nw; out->ncol = 1; out->vs = vs;
  out->lds_bytes_per_wave = (int32_t)lds_per_wave(h, v, nullptr);
  out->waves_per_simd = SWB_COVER_WAVES(v->nw);
  out->resample_waves_per_simd = SWB_RS_WAVES_PER_SIMD;
  out->n_bands = h->p.nbands ? h->p.nbands : h->nbands;
  out->n_column_groups = (h->p.Wo + 63) / 64;
  out->run_cap = h->p.run_cap;
  out->paint_in_cover = (h->p.AA == 1 && (h->p.Wo + 63) / 64 == 1 && !h->no_paint_in_cover) ? 1 : 0;
  out->arena_units = h->d_runs ? h->p.arena_units : 0;
  out->reserved_ = 0;
  out->run_list_bytes = h->d_runs ? ((int64_t)h->p.arena_base + h->p.arena_units + 4) * 8 : 0;
  return SWB_OK;
}

#ifndef SWB_BUILD_ID
#define SWB_BUILD_ID "unknown"
#endif
const char* swb_build_id(void) { return SWB_BUILD_ID; }

int swb_timing_enable(swb_handle h, int32_t enable) {
  if (!h) return fail(SWB_ERR_INVALID, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  if (flush_timing(h)) return SWB_ERR_HIP;
  h->timing = enable != 0;
  h->timed_ms = 0.0;
  h->timed_cover_ms = 0.0;
  h->timed_launches = 0;
  return SWB_OK;
}

int swb_step_time_ms(swb_handle h, double* total_ms, int64_t* launches) {
  if (!h) return fail(SWB_ERR_INVALID, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  if (flush_timing(h)) return SWB_ERR_HIP;
  if (total_ms) *total_ms = h->timed_ms;
  if (launches) *launches = h->timed_launches;
  return SWB_OK;
}

int swb_kernel_times_ms(swb_handle h, double* cover_ms, double* resample_ms, int64_t* launches) {
  if (!h) return fail(SWB_ERR_INVALID, "null handle");
  HIP_TRY(hipSetDevice(h->device));
  if (flush_timing(h)) return SWB_ERR_HIP;
  if (cover_ms) *cover_ms = h->timed_cover_ms;
  if (resample_ms) *resample_ms = h->timed_ms - h->timed_cover_ms;
  if (launches) *launches = h->timed_launches;
  return SWB_OK;
}

}  // extern "C"
