"""The state kernel cut short after phase ARG (timing experiment only: no frames).

  python tools/overlay_build.py phase3 phase_cut:3      # 3 = loads + centred paths, 4 = + hit-test / move, 5 = + task /
                                                         # termination, 1 = + canvas edges (P1b; = the whole state kernel but its hand-off)
The shipped kernel carries an empty hook macro (SWB_HOOK_PHASE_END) at those four points and nothing else of this.  (Since the
round-4 split the cover and resample kernels are timed on their own: swb_kernel_times3_ms.)"""


def apply(files, arg, replace_once):
  cut = int(arg)
  assert cut in (3, 4, 5, 1), cut
  replace_once(files, 'swb_kernels.hip.inc', '#ifndef SWB_HOOK_PHASE_END\n#define SWB_HOOK_PHASE_END(k)\n#endif\n',
               '#define SWB_HOOK_PHASE_END(k) { if ((k) == %d) return; }\n' % cut)
  # no hand-off is written: neither the cover nor the second kernel may run
  replace_once(files, 'swb.hip', '  if (p.obs) launch_cover(0, c.n_envs);', '  (void)launch_cover;')
  replace_once(files, 'swb.hip', '  if (p.obs && !p.paint_in_cover) launch_resample(0, c.n_envs, stream);\n', '  (void)launch_resample;\n')
  replace_once(files, 'swb.hip', '  h->cover_lists_filed = p.obs && p.ccost_list;\n', '  h->cover_lists_filed = false;\n')
