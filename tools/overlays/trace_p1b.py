"""Where the cycles of P1b (canvas edges) go (experiment only; results unchanged): the wave timeline of `trace` plus three stamps
inside build_all_edges -- after sweep 1 (vertices -> canvas ints), after sweep 2 (edges, extents through LDS atomics) and before
sweep 3 (the corner rule) -- packed in t[11] as three 21-bit cycle counts since the wave began (s1 | s2 << 21 | s3 << 42).  The
stamps pass through wave_lds::pad_ (three free dwords)."""
import importlib.util
import os


def apply(files, arg, replace_once):
  here = os.path.dirname(os.path.abspath(__file__))
  spec = importlib.util.spec_from_file_location('trace', os.path.join(here, 'trace.py'))
  base = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(base)
  base.apply(files, arg, replace_once)
  k = 'swb_kernels.hip.inc'
  T = '(int)(uint32_t)__builtin_amdgcn_s_memtime()'
  replace_once(files, k, '''      edges[k] = e;
    }
  }
  wave_sync();
''', '''      edges[k] = e;
    }
  }
  wave_sync();
  if (l == 0) L->pad_[0] = %s;
''' % T)
  replace_once(files, k, '''  wave_sync();
  // per sprite: scan range (Draw.c: ymin starts at ysize-1, ymax at 0), word extent, corner-rule mode
''', '''  wave_sync();
  if (l == 0) L->pad_[1] = %s;
  // per sprite: scan range (Draw.c: ymin starts at ysize-1, ymax at 0), word extent, corner-rule mode
''' % T)
  replace_once(files, k, '''  wave_sync();
  // ---- sweep 3: "connect discontiguous corners" for the top end''', '''  wave_sync();
  if (l == 0) L->pad_[2] = %s;
  // ---- sweep 3: "connect discontiguous corners" for the top end''' % T)
  replace_once(files, k, '''  build_all_edges<NW>(p, L, edges, cpath, spans, n, vtotal, voff_l, nv_l, px, py, dmin_l, scale_l, rgb_reg, err);
''', '''  build_all_edges<NW>(p, L, edges, cpath, spans, n, vtotal, voff_l, nv_l, px, py, dmin_l, scale_l, rgb_reg, err);
  if (p.exp_trace && l == 0) {
    const unsigned long long c0 = (uint32_t)exp_c0;
    const unsigned long long s1 = ((uint32_t)L->pad_[0] - (uint32_t)c0) & 0x1fffffu, s2 = ((uint32_t)L->pad_[1] - (uint32_t)c0) & 0x1fffffu,
                             s3 = ((uint32_t)L->pad_[2] - (uint32_t)c0) & 0x1fffffu;
    p.exp_trace[(size_t)env * 12 + 11] = s1 | (s2 << 21) | (s3 << 42);
  }
''')
