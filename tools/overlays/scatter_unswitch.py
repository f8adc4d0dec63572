"""P2 edge-lane scatter with the loop unswitched on its two pass-level flags (experiment; results unchanged): a sprite without
corner replacements that does not reach the canvas' first word -- the common case -- runs a loop whose body is straight-line
code (the compiled general loop takes four to five branches per iteration: the structurised form of `if (any_repl)` and of
`if (maybe_neg && ballot)`)."""


def apply(files, arg, replace_once):
  replace_once(files, 'swb_kernels.hip.inc', '''        for (int j = 0; j < bound; j += G) {
          const int yy = ylo + j0 + j;
          scatter_crossing2<NW>(L, xb, npx, (yy - yb) & 63, live && yy <= yhi, yy, x0f, ed.y0, ed.dx, xtop, xbot, any_repl, w0 == 0,
                                eymin, eymax, symax);
        }
''', '''        if (!any_repl && w0 != 0) {
          for (int j = 0; j < bound; j += G) {
            const int yy = ylo + j0 + j;
            scatter_crossing2<NW>(L, xb, npx, (yy - yb) & 63, live && yy <= yhi, yy, x0f, ed.y0, ed.dx, xtop, xbot, false, false,
                                  eymin, eymax, symax);
          }
        } else {
          for (int j = 0; j < bound; j += G) {
            const int yy = ylo + j0 + j;
            scatter_crossing2<NW>(L, xb, npx, (yy - yb) & 63, live && yy <= yhi, yy, x0f, ed.y0, ed.dx, xtop, xbot, any_repl, w0 == 0,
                                  eymin, eymax, symax);
          }
        }
''')
