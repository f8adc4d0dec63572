"""P2 row-lane edge loop unswitched on the pass-level flag "the sprite reaches the canvas' first word" (experiment; results
unchanged): only then can a crossing be negative and need Draw.c's float64 rounding -- every other sprite runs a loop whose
crossing code has no `maybe_neg && ballot` branch (see scatter_unswitch.py for the edge-lane loop, measured -1.5 ... -2.4 %)."""


def apply(files, arg, replace_once):
  k = 'swb_kernels.hip.inc'
  old_loop = '''        for (int e = 0; e < ne; ++e) {
          const edge_rec ed = edges[e0 + e];
          const int x0 = rfl(ed.x0), y0 = rfl(ed.y0), y1 = rfl(ed.y1);
          const float dx = rflf(ed.dx);
          if (y0 == y1) {
            if (y0 < yb || y0 > yb + 63) continue;
            const int x1 = __float_as_int(dx);
            if (y == y0) scatter_hline<NW>(L, xb, npx, l, min(x0, x1), max(x0, x1));
            continue;
          }
          const int eymin = min(y0, y1), eymax = max(y0, y1);
          if (eymax < yb || eymin > yb + 63) continue;
          const int it = rfl(ed.xtop), ib = rfl(ed.xbot);
          const bool act = (y >= eymin) && (y <= eymax) && (y >= symin) && (y <= symax);
          scatter_crossing2<NW>(L, xb, npx, l, act, y, (float)x0, y0, dx, repl_to_float(it), repl_to_float(ib),
                                it != SWB_NO_REPL || ib != SWB_NO_REPL, w0 == 0, eymin, eymax, symax);
        }
'''
  assert files[k].count(old_loop) == 1
  body = old_loop.replace('w0 == 0, eymin, eymax, symax);', 'MAYBE_NEG, eymin, eymax, symax);')
  new = ('        if (w0 != 0) {\n' + body.replace('MAYBE_NEG', 'false').replace('\n        ', '\n          ').replace('        for (int e', '          for (int e', 1) +
         '        } else {\n' + body.replace('MAYBE_NEG', 'true').replace('\n        ', '\n          ').replace('        for (int e', '          for (int e', 1) + '        }\n')
  files[k] = files[k].replace(old_loop, new)
