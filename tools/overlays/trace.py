"""Wave timeline of both kernels of a step (experiment only; results unchanged).

Adds `exp_trace` (u64[2][N * bands][12]: realtime at wave start / end (100 MHz), shader cycles of the wave, HW_ID | XCC_ID << 32)
to swb_params and the export swb_exp_set(handle, trace).  Row 0 = cover kernel (index = environment), row 1 = resample kernel
(index = environment * bands + band)."""


def apply(files, arg, replace_once):
  k = 'swb_kernels.hip.inc'
  replace_once(files, k, '  const double* ov_cpath;      // [N][S][SWB_MAX_SHAPE_VERTS][2] centred paths as the setters left them\n',
               '  const double* ov_cpath;\n  unsigned long long* exp_trace;\n')
  begin = ('  const unsigned long long exp_rt0 = __builtin_amdgcn_s_memrealtime();\n'
           '  const unsigned long long exp_c0 = __builtin_amdgcn_s_memtime();\n')

  def end(index):
    return ('  if (p.exp_trace && l == 0) {\n'
            '    unsigned long long* t = p.exp_trace + (size_t)(%s) * 12;\n'
            '    t[0] = exp_rt0; t[1] = __builtin_amdgcn_s_memrealtime(); t[2] = __builtin_amdgcn_s_memtime() - exp_c0;\n'
            '    t[3] = (unsigned)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)(unsigned)__builtin_amdgcn_s_getreg(63508) << 32);\n'
            '  }\n' % index)
  # cover kernel: phase stamps through the phase hook (cycles since wave start at the end of phase k -> t[4 + slot])
  replace_once(files, k, '#ifndef SWB_HOOK_PHASE_END\n#define SWB_HOOK_PHASE_END(k)\n#endif\n',
               '#define SWB_HOOK_PHASE_END(k) { if (p.exp_trace && l == 0) p.exp_trace[(size_t)env * 12 + 4 + ((k) == 3 ? 0 : (k) == 4 ? 1 : (k) == 5 ? 2 : 3)] = __builtin_amdgcn_s_memtime() - exp_c0; }\n')
  replace_once(files, k, '    row_spans rs;\n    coverage_batch2<NW>(p, L, edges, spans, ovf, n, yb, sp_ymin, sp_ymax, sp_a, sp_e0, rs, err);\n    SWB_RESCAN_WITH_OVERFLOW_SLOT()\n    emit_runs(p, spans, ovf, rs, yb, env, runs_env, hdr, bst, base_l, band_l, cost_l, err);\n',
               '    row_spans rs;\n    const unsigned long long exp_a = __builtin_amdgcn_s_memtime();\n    coverage_batch2<NW>(p, L, edges, spans, ovf, n, yb, sp_ymin, sp_ymax, sp_a, sp_e0, rs, err);\n    SWB_RESCAN_WITH_OVERFLOW_SLOT()\n    const unsigned long long exp_b = __builtin_amdgcn_s_memtime();\n    emit_runs(p, spans, ovf, rs, yb, env, runs_env, hdr, bst, base_l, band_l, cost_l, err);\n    exp_cov += exp_b - exp_a; exp_emit += __builtin_amdgcn_s_memtime() - exp_b; exp_nb += 1;\n')
  replace_once(files, k, '  int env = blockIdx.x;\n  // ... and, as in the resample kernel, the wave\'s priority follows',
               begin + '  unsigned long long exp_cov = 0, exp_emit = 0, exp_nb = 0;\n  int env = blockIdx.x;\n  // ... and, as in the resample kernel, the wave\'s priority follows')
  replace_once(files, k, '  if (l == 0) ovf_slot_release(p, ovf);\n  if (p.error) {\n    uint32_t e = err & ~SWB_INT_NEED_SLOT;',
               '  if (l == 0) ovf_slot_release(p, ovf);\n' + end('env') +
               '  if (p.exp_trace && l == 0) { p.exp_trace[(size_t)env * 12 + 8] = exp_cov; p.exp_trace[(size_t)env * 12 + 9] = exp_emit; p.exp_trace[(size_t)env * 12 + 10] = exp_nb; }\n'
               '  if (p.error) {\n    uint32_t e = err & ~SWB_INT_NEED_SLOT;')
  # resample kernel: the end of its row loop is the end of the kernel
  replace_once(files, k, '  const int o_lo = as_const(p.band_lo)[band], o_hi = as_const(p.band_lo)[band + 1];\n  if (o_lo >= o_hi) return;\n',
               begin + '  const int o_lo = as_const(p.band_lo)[band], o_hi = as_const(p.band_lo)[band + 1];\n  if (o_lo >= o_hi) return;\n')
  replace_once(files, k, '      if (r_first >= o_hi) { more = false; break; }\n    }\n  }\n}\n',
               '      if (r_first >= o_hi) { more = false; break; }\n    }\n  }\n' +
               end('(size_t)p.N * p.nbands + (size_t)env * p.nbands + band') +
               '  if (p.exp_trace && l == 0) {   // where the task ran: block, wave of the block; and what it cost: units of its list\n'
               '    unsigned long long* t = p.exp_trace + ((size_t)p.N * p.nbands + (size_t)env * p.nbands + band) * 12;\n'
               '    t[4] = blockIdx.x | ((unsigned long long)(threadIdx.x >> 6) << 32);\n'
               '    t[5] = hdr[SWB_RHDR_GROUPS + g * SWB_RHDR_GSTRIDE];\n'
               '    t[6] = exp_runs; t[7] = exp_spans;\n'
               '  }\n' + '}\n')
  replace_once(files, k, '        const int ns = (int)(rec.x >> 24);\n        const swb_i8 ps =',
               '        const int ns = (int)(rec.x >> 24);\n        exp_runs += 1; exp_spans += (unsigned)ns;\n        const swb_i8 ps =')
  replace_once(files, k, '  swb_u4 rec = *reinterpret_cast<cptr<swb_u4>>(runs + uo);       // { row',
               '  unsigned exp_runs = 0, exp_spans = 0;\n  swb_u4 rec = *reinterpret_cast<cptr<swb_u4>>(runs + uo);       // { row')
  h = 'swb.hip'
  replace_once(files, h, 'const char* swb_last_error(void) { return g_err.c_str(); }\n',
               'const char* swb_last_error(void) { return g_err.c_str(); }\n'
               'int swb_exp_set(swb_handle h, unsigned long long* trace) { h->p.exp_trace = trace; return 0; }\n')
