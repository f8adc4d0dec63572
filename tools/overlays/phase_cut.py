"""The cover kernel cut short after phase ARG (timing experiment only: the frames are wrong or missing).

  python tools/overlay_build.py phase3 phase_cut:3      # 3 = loads + centred paths, 4 = + hit-test / move, 5 = + task /
                                                         # termination, 1 = + canvas edges (P1b)
The shipped kernel carries an empty hook macro (SWB_HOOK_PHASE_END) at those four points and nothing else of this."""


def apply(files, arg, replace_once):
  cut = int(arg)
  assert cut in (3, 4, 5, 1), cut
  replace_once(files, 'swb_kernels.hip.inc', '#ifndef SWB_HOOK_PHASE_END\n#define SWB_HOOK_PHASE_END(k)\n#endif\n',
               '#define SWB_HOOK_PHASE_END(k) { if ((k) == %d) { if (lane_id() == 0) ovf_slot_release(p, ovf); return; } }\n' % cut)
  # no run lists are written: the second kernel must not run
  replace_once(files, 'swb.hip', '  if (p.obs && !p.paint_in_cover) launch_resample(0, c.n_envs, stream);\n', '  (void)launch_resample;\n')
  # ... and no environment is filed for the order of the next cover launch: plain order
  replace_once(files, 'swb.hip', '  h->cover_lists_filed = p.obs && p.ccost_list;\n', '  h->cover_lists_filed = false;\n')
