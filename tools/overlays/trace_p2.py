"""Where the cycles of a coverage pass go (experiment only; results unchanged): the wave timeline of `trace` with the phase
stamps of the cover kernel replaced by cycle sums INSIDE P2 -- per wave, over all its (batch, sprite) passes:
  t[4] head (sprite parameters, edge records)   t[5] edge scatter   t[6] the fence after it   t[7] word loop (masks ->
  coverage -> spans) incl. its closing fence     t[11] rest of the pass (the run that reaches the sprite's last word).
s_memtime waits for the scalar-memory AND LDS counter, so every stamp is also a fence: the sums are upper bounds of what the
unstamped kernel spends there."""
import importlib.util
import os


def apply(files, arg, replace_once):
  here = os.path.dirname(os.path.abspath(__file__))
  spec = importlib.util.spec_from_file_location('trace', os.path.join(here, 'trace.py'))
  base = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(base)
  base.apply(files, arg, replace_once)
  k = 'swb_kernels.hip.inc'
  # no phase stamps in slots 4..7
  replace_once(files, k, '#define SWB_HOOK_PHASE_END(k) { if (p.exp_trace && l == 0)', '#define SWB_HOOK_PHASE_END_UNUSED(k) { if (p.exp_trace && l == 0)')
  replace_once(files, k, '#define SWB_HOOK_PHASE_END_UNUSED(k)', '#define SWB_HOOK_PHASE_END(k)\n#define SWB_HOOK_PHASE_END_UNUSED(k)')
  T = '__builtin_amdgcn_s_memtime()'
  replace_once(files, k, '                                                int sp_e0, row_spans& rs, uint32_t& err) {\n  constexpr int NWA = SWB_NWA(NW);',
               '                                                int sp_e0, row_spans& rs, uint32_t& err, unsigned long long (&xs)[5]) {\n  constexpr int NWA = SWB_NWA(NW);')
  replace_once(files, k, '    sm &= ~(1ull << s);\n    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)sp_a, s);',
               '    sm &= ~(1ull << s);\n    unsigned long long xprev = %s;\n    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)sp_a, s);' % T)
  replace_once(files, k, '      if (packed) {\n        // the edges of this chunk',
               '      const unsigned long long x1 = %s; xs[0] += x1 - xprev;\n      if (packed) {\n        // the edges of this chunk' % T)
  replace_once(files, k, '      // masks -> coverage -> visible spans\n',
               '      const unsigned long long x2 = %s; xs[1] += x2 - x1;\n' % T)
  replace_once(files, k, '      const uint32_t wbits = ((1u << nwords) - 1u) << wc;      // canvas words of this chunk\n      wave_sync();\n',
               '      const uint32_t wbits = ((1u << nwords) - 1u) << wc;\n      wave_sync();\n      const unsigned long long x3 = %s; xs[2] += x3 - x2;\n' % T)
  replace_once(files, k, '      }\n      }\n      wave_sync();\n    }\n    if (open_start >= 0) {',
               '      }\n      }\n      wave_sync();\n      xprev = %s; xs[3] += xprev - x3;\n    }\n    if (open_start >= 0) {' % T)
  replace_once(files, k, '      put_span(p, spans, ovf, rs, l, (uint32_t)open_start | ((uint32_t)(32 * (w1 + 1)) << 10) | sprite_bits, err);\n    }\n  }\n',
               '      put_span(p, spans, ovf, rs, l, (uint32_t)open_start | ((uint32_t)(32 * (w1 + 1)) << 10) | sprite_bits, err);\n    }\n    xs[4] += %s - xprev;\n  }\n' % T)
  # the kernel: accumulators, both call sites, the store
  files[k] = files[k].replace('sp_ymin, sp_ymax, sp_a, sp_e0, rs, err);', 'sp_ymin, sp_ymax, sp_a, sp_e0, rs, err, exp_xs);')
  replace_once(files, k, '  unsigned long long exp_cov = 0, exp_emit = 0, exp_nb = 0;\n',
               '  unsigned long long exp_cov = 0, exp_emit = 0, exp_nb = 0;\n  unsigned long long exp_xs[5] = {0, 0, 0, 0, 0};\n')
  replace_once(files, k, '  if (p.exp_trace && l == 0) { p.exp_trace[(size_t)env * 12 + 8] = exp_cov;',
               '  if (p.exp_trace && l == 0) { for (int i = 0; i < 4; ++i) p.exp_trace[(size_t)env * 12 + 4 + i] = exp_xs[i]; p.exp_trace[(size_t)env * 12 + 11] = exp_xs[4]; }\n'
               '  if (p.exp_trace && l == 0) { p.exp_trace[(size_t)env * 12 + 8] = exp_cov;')
