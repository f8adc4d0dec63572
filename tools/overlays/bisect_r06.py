"""Round-6 regression hunt (experiment builds only): removes, one by one, what round 6 added to the cover kernels of canvases up to
320 px, to see which of them moved the headline's cover kernel.  ARG: comma-separated subset of
  cells     the per-step cell lookup of tasks that key on position (P0)
  grow      the run lists' move to the arena (emit_runs: room check as in round 5, no list_room_or_move)"""


def apply(files, arg, replace_once):
  k = 'swb_kernels.hip.inc'
  what = set(arg.split(','))
  if 'cells' in what:
    a = files[k].index('    if (p.p_cell_label != nullptr) {                     // (wave-uniform; no shipped configuration)')
    b = files[k].index('    const bool oof_l = (l < n) && !(px >= 0. && py >= 0. && px <= 1. && py <= 1.);')
    files[k] = files[k][:a] + files[k][b:]
  if 'grow' in what:
    a = files[k].index('    if ((meta_g >> 4) != 0u || (!room && p.arena_units > 0)) {          // (wave-uniform, rare)')
    b = files[k].index('    if (!room) err |= SWB_ENV_ERR_SPAN_OVERFLOW;\n    uint32_t* dst')
    files[k] = files[k][:a] + files[k][b:]
