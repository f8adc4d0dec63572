"""Builds tests/emu/_build/libswb_emu.so: the sources of spriteworld_amd/csrc compiled for the HOST against the
emulation shim (tests/emu/shim/hip/hip_runtime.h) -- TEST INFRASTRUCTURE ONLY, see tests/emu/README.md.

The product sources are not edited for this.  Two constructs cannot be compiled for x86 and are rewritten in a
temporary copy (each rewrite must match, else the build fails):
  * the four inline-assembly statements (v_med3_i32, three v_mad_i32_i24) -> their C meaning;
  * the constant-address-space pointer alias (address_space(4), scalar loads on the GPU) -> a plain pointer.
"""
import hashlib
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
# SWB_EMU_CSRC: emulate another copy of the kernel sources (tools/try_patch.py: a patched copy on its way to the GPU)
CSRC = os.path.abspath(os.environ.get('SWB_EMU_CSRC') or os.path.join(ROOT, 'spriteworld_amd', 'csrc'))
OUT_DIR = os.path.join(HERE, '_build') if not os.environ.get('SWB_EMU_CSRC') else \
    os.path.join(HERE, '_build', 'alt_' + hashlib.sha256(CSRC.encode()).hexdigest()[:8])
if os.environ.get('SWB_EMU_STATS'):
  OUT_DIR = os.path.join(OUT_DIR, 'stats')
LIB = os.path.join(OUT_DIR, 'libswb_emu.so')
CLANG = '/opt/rocm/lib/llvm/bin/clang++'
SOURCES = ('swb.hip', 'swb_wide.hip', 'swb_kernels.hip.inc', 'swb_pow.hip.inc', 'swb_pow_tables.inc', 'swb_sampler.hip.inc')


def _rewrite(text, name):
  if name != 'swb_kernels.hip.inc':
    return text
  n_total = 0
  text, n = re.subn(r'asm\("v_med3_i32 %0, %1, %2, %3" : "=v"\((\w+)\) : "s"\((\w+)\), "v"\((\w+)\), "v"\((\w+)\)\);',
                    r'\1 = std::max(std::min(\2, \3), std::min(std::max(\2, \3), \4));   /* emu: median of three */', text)
  assert n == 1, 'v_med3_i32 statement not found'
  n_total += n
  text, n = re.subn(r'asm\("v_mad_i32_i24 %0, %1, %2, %3" : "=v"\(([^)]+)\) : "v"\((\w+)\), "s"\((\w+)\), "v"\(([^)]+)\)\);',
                    r'\1 = __mul24(\2, \3) + \4;   /* emu: v_mad_i32_i24 */', text)
  assert n == 3, 'v_mad_i32_i24 statements not found'
  n_total += n
  text, n = re.subn(r'using cptr = const T __attribute__\(\(address_space\(4\)\)\)\*;', 'using cptr = const T*;   /* emu */', text)
  assert n == 1, 'constant address space alias not found'
  assert 'asm(' not in text.replace('asm("")', ''), 'an inline-assembly statement is left'
  # bounds checks on the hand-off lists (counted by emu_violations(), see emu_runtime.cc): a run record is read at a unit of
  # the list's own fixed part -- the sentinel / jump unit is the last one a segment can hold -- or inside the shared arena
  text = text.replace('#define SWB_WAVE 64', '#define SWB_WAVE 64\nextern "C" void emu_check(int ok);', 1)
  ok_fn = ('\n// emu: is unit `u` (relative to the first unit of the list of (env, g)) one a run record may be read at?\n'
           'static inline int emu_list_unit_ok(const swb_params& p, int env, int g, long long u) {\n'
           '  if (u >= 0 && u < p.run_cap) return 1;\n'
           '  const long long a = ((long long)env * p.ncg + g) * p.run_cap + u - p.arena_base;\n'
           '  return a >= 0 && a < p.arena_units;\n}\n')
  anchor = '#define SWB_MAX_CG 4 '
  assert text.count(anchor) == 1, anchor
  text = text.replace(anchor, ok_fn + anchor)
  anchor = 'rec = *reinterpret_cast<cptr<swb_u4>>(runs + uo);'
  assert text.count(anchor) == 2, anchor
  text = text.replace(anchor, anchor + ' emu_check(emu_list_unit_ok(p, env, g, uo >> 3));')
  anchor = 'hd = runs[2 * u]; s0 = runs[2 * u + 1];'
  assert text.count(anchor) == 1, anchor
  text = text.replace(anchor, 'emu_check(emu_list_unit_ok(p, env, g, u)); ' + anchor)
  if os.environ.get('SWB_EMU_STATS'):
    text = _instrument(text)
  return text


# SWB_EMU_STATS=1: event counters at a few anchor points of the kernel source (counted by lane 0 of each wave), read back
# through emu_stats() -- exact dynamic figures for the cost model in DESIGN.md (tools/emu_stats.py).
_COUNTERS = ('p3_row_runs', 'p3_rows_in_runs', 'p3_spans', 'p3_completed_rows', 'p3_clean_rows', 'p2_batches',
             'p2_sprite_passes', 'p2_edge_iterations', 'p2_transition_steps', 'run_units', 'p2_chunks', 'p2_packed_passes', 'p2_words',
             'p1b_general_iterations', 'p1b_sweeps')


def _instrument(text):
  def hook(counter, amount='1'):
    return ' emu_count(%d, %s);' % (_COUNTERS.index(counter), amount)

  def after(anchor, code, count=1):
    nonlocal text
    assert text.count(anchor) == count, (anchor, text.count(anchor))
    text = text.replace(anchor, anchor + code)

  text = text.replace('#define SWB_WAVE 64', '#define SWB_WAVE 64\nextern "C" void emu_count(int counter, long amount);', 1)
  # resample kernel: runs, the canvas rows they stand for, their spans; finished output rows and the untouched ones
  after('      while ((int)(rec.x & 0xffffu) <= next_end) {',
        hook('p3_row_runs') + hook('p3_rows_in_runs', '(long)((rec.x >> 16) & 63u) + 1') + hook('p3_spans', '(long)std::max(1u, rec.x >> 24)'))
  after('        const bool untouched = black && mark[k] == uo;\n', '       ' + hook('p3_completed_rows') + ' if (untouched)' + hook('p3_clean_rows') + '\n')
  # cover kernel: coverage passes and what they emit
  after('    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)sp_a, s);', hook('p2_sprite_passes'))
  after('  rs.cnt = 0; rs.s0 = rs.s1 = rs.s2 = 0u;', hook('p2_batches'))
  after('        for (int e = 0; e < ne; ++e) {', hook('p2_edge_iterations'))
  after('          for (int j = 0; j < bound; j += G) {', hook('p2_edge_iterations'), count=2)      # (the loop exists unswitched: two copies)
  after('            for (int j = 0; j < pk_iters; ++j) {', hook('p2_edge_iterations'), count=2)   # ("active edges" form)
  after('    const int total = __builtin_amdgcn_readlane(incl, SWB_WAVE - 1);', hook('run_units', 'total'))
  after('    for (int wc = w0; wc <= w1; wc += NWA) {', hook('p2_chunks'))
  after('        if (!((wbits >> w) & 1u)) continue;         // uniform: not in this chunk\n', '       ' + hook('p2_words') + '\n')
  after('        pk_hlines = __ballot(have && horiz && ed.y0 >= yb && ed.y0 <= yb + 63 && ed.y0 < p.Hc) != 0ull;', hook('p2_packed_passes'))
  after('      while (__ballot(cand != 0ull)) {', hook('p1b_general_iterations'))
  after('    const bool base_live = have && (cy0 != cy1) && (cdx != 0.0f);', hook('p1b_sweeps'))
  anchor = '        auto take = [&]() __attribute__((always_inline)) {'
  assert text.count(anchor) == 1, anchor
  text = text.replace(anchor, anchor + hook('p2_transition_steps') + ' ')
  return text


def source_hash():
  h = hashlib.sha256()
  for path in [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(ROOT, 'include', 'swb.h'), os.path.join(HERE, 'emu_runtime.cc'),
                                                          os.path.join(HERE, 'shim', 'hip', 'hip_runtime.h'), os.path.abspath(__file__)]:
    with open(path, 'rb') as f:
      h.update(f.read())
  return h.hexdigest()


def build(force=False):
  """Builds (or reuses) the library; an exclusive file lock makes concurrent callers (pytest-xdist workers) wait for one build."""
  import fcntl
  os.makedirs(OUT_DIR, exist_ok=True)
  with open(os.path.join(OUT_DIR, '.lock'), 'w') as lock:
    fcntl.flock(lock, fcntl.LOCK_EX)
    return _build_locked(force)


def _build_locked(force):
  stamp = LIB + '.hash'
  want = source_hash()
  if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == want:
    return LIB
  src_dir = os.path.join(OUT_DIR, 'src', 'csrc')
  shutil.rmtree(os.path.join(OUT_DIR, 'src'), ignore_errors=True)
  os.makedirs(src_dir)
  os.makedirs(OUT_DIR, exist_ok=True)
  os.makedirs(os.path.join(OUT_DIR, 'include'), exist_ok=True)
  shutil.copy(os.path.join(ROOT, 'include', 'swb.h'), os.path.join(OUT_DIR, 'include', 'swb.h'))
  for name in SOURCES:
    with open(os.path.join(CSRC, name)) as f:
      text = _rewrite(f.read(), name)
    with open(os.path.join(src_dir, name), 'w') as f:
      f.write(text)
  # the kernels include "../../include/swb.h" relative to csrc/: OUT_DIR/src/csrc -> OUT_DIR/include
  flags = ['-x', 'c++', '-std=c++17', '-O1', '-g', '-fPIC', '-ffp-contract=off', '-fno-fast-math', '-fno-strict-aliasing',
           '-I', os.path.join(HERE, 'shim'), '-DSWB_BUILD_ID="emulated"', '-Wno-unused-value', '-Wno-ignored-attributes',
           '-Wno-unknown-attributes']
  objs, procs = [], []
  for unit in ('swb.hip', 'swb_wide.hip'):
    obj = os.path.join(OUT_DIR, unit + '.o')
    procs.append(subprocess.Popen([CLANG] + flags + ['-c', '-o', obj, os.path.join(src_dir, unit)]))
    objs.append(obj)
  obj = os.path.join(OUT_DIR, 'emu_runtime.o')
  procs.append(subprocess.Popen([CLANG] + flags + ['-c', '-o', obj, os.path.join(HERE, 'emu_runtime.cc')]))
  objs.append(obj)
  for proc in procs:
    if proc.wait() != 0:
      raise RuntimeError('emulator build failed')
  subprocess.check_call([CLANG, '-shared', '-fPIC', '-o', LIB] + objs + ['-lm'])
  with open(stamp, 'w') as f:
    f.write(want + '\n')
  return LIB


if __name__ == '__main__':
  print(build(force=True))
