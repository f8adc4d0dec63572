#!/usr/bin/env python
"""Turns what tools/final_evidence.sh left under gpurun_out/TAG into the committed evidence under profiles/ (ROUND = r04 ...):
  ROUND_rocprofv3_summary_{default,aa1,embodied_s12_128}.md   kernel trace + the PMC tables of both kernels
  ROUND_counters.json                                         what bench.py quotes, keyed by the build id of the run
  ROUND_bench_default.json, ROUND_bench_driver_cmd.json, ROUND_launch_convergence.json
usage: python tools/assemble_evidence.py ROUND gpurun_out/TAG"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RUNS = (('default', 'cluster_s5', 5), ('aa1', 'cluster_s5', 1), ('embodied_s12_128', 'embodied_s12', 5))


def load_pmc(d):
  out = {}
  for name in ('insts', 'active', 'write', 'fetch', 'wait', 'lds'):
    path = os.path.join(d, 'pmc_%s.json' % name)
    if os.path.exists(path) and os.path.getsize(path) > 2:
      for kernel, c in json.load(open(path)).items():
        out.setdefault(kernel, {}).update(c)
  return out


def model_min(workload, aa):
  """Cost-model minimum of the resample kernel's vector instructions (exact event counts from the emulated kernel source)."""
  if aa == 1:
    return None, None
  code = ('import sys; sys.argv=["emu_stats", "%s", "32", "10", "%d", "1"]; sys.path.insert(0, "tools"); import emu_stats as e; '
          'v = e.main(); import json; print("JSON" + json.dumps({"model": e.resample_valu_model(v), "events": v}))' % (workload, aa))
  try:
    out = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True, timeout=1500).stdout
    d = json.loads([l for l in out.splitlines() if l.startswith('JSON')][-1][4:])
    return d['model'], d['events']
  except Exception as e:  # pylint: disable=broad-except
    print('no cost model for %s: %r' % (workload, e))
    return None, None


def main():
  rnd, src = sys.argv[1], sys.argv[2]
  records = []
  for tag, workload, aa in RUNS:
    d = os.path.join(src, tag)
    if not os.path.exists(os.path.join(d, 'bench_unprofiled.json')):
      continue
    bench = json.loads(open(os.path.join(d, 'bench_unprofiled.json')).readlines()[-1])
    pmc = load_pmc(d)
    envs = bench['config']['envs_per_gpu']
    by = {}
    for kernel, c in pmc.items():
      role = 'cover' if 'cover' in kernel else ('resample' if ('resample' in kernel or 'fill' in kernel) else None)
      if role:
        by[role] = dict(c, kernel=kernel)
    lines = open(os.path.join(d, 'summary.md')).read().rstrip().split('\n') if os.path.exists(os.path.join(d, 'summary.md')) else []
    kms = {('cover' if 'cover' in k['name'] else 'resample'): k['ms'] for k in bench['roofline']['kernels']}
    if by:
      lines += ['', '## Counters per kernel (rocprofv3 --pmc, one process per counter set, tools/pmc_sets.sh; averages per dispatch, '
                'per environment where divided)', '',
                '| kernel | waves | VALU / env | SALU / env | LDS / env | SMEM / env | wave quad-cycles / wave | vector ALU busy (SQ_ACTIVE_INST_VALU x 4 cycles '
                '/ 1024 SIMDs / 2.4 GHz / kernel time) | WRITE_SIZE MB | 2 x FETCH_SIZE MB |', '|---|---|---|---|---|---|---|---|---|---|']
      for role in ('cover', 'resample'):
        c = by.get(role)
        if not c:
          continue
        waves = c.get('SQ_WAVES', envs)
        lines.append('| `%s` | %d | %.0f | %.0f | %.0f | %.0f | %.0f | %.2f | %.1f | %.1f |' % (
            c['kernel'], waves, c.get('SQ_INSTS_VALU', 0) / envs, c.get('SQ_INSTS_SALU', 0) / envs, c.get('SQ_INSTS_LDS', 0) / envs,
            c.get('SQ_INSTS_SMEM', 0) / envs, c.get('SQ_WAVE_CYCLES', 0) / max(waves, 1),
            c.get('SQ_ACTIVE_INST_VALU', 0) * 4.0 / 1024.0 / 2.4e9 / max(kms.get(role, 0.0) * 1e-3, 1e-12),
            c.get('WRITE_SIZE', 0) / 1024.0, 2 * c.get('FETCH_SIZE', 0) / 1024.0))
    a_bytes = bench['roofline']['algorithmic_bytes_per_env_step'] * envs
    traffic = sum((c.get('WRITE_SIZE', 0) + 2.0 * c.get('FETCH_SIZE', 0)) * 1024 for c in by.values())
    have_traffic = all('WRITE_SIZE' in c and 'FETCH_SIZE' in c for c in by.values()) and len(by) == (1 if 'paints the frame' in bench['roofline']['kernel'] else 2)
    model, events = model_min(workload, aa)
    rec = {
        'build_id': bench['roofline']['build_id'], 'workload': workload, 'envs': envs, 'anti_aliasing': aa,
        'kernel': bench['roofline']['kernel'],
        'insts_valu_per_env': sum(c.get('SQ_INSTS_VALU', 0) for c in by.values()) / envs,
        'insts_salu_per_env': sum(c.get('SQ_INSTS_SALU', 0) for c in by.values()) / envs,
        'insts_valu_per_env_by_kernel': {r: c.get('SQ_INSTS_VALU', 0) / envs for r, c in by.items()},
        'insts_salu_per_env_by_kernel': {r: c.get('SQ_INSTS_SALU', 0) / envs for r, c in by.items()},
        'wave_cycles_per_wave_by_kernel': {r: c.get('SQ_WAVE_CYCLES', 0) / max(c.get('SQ_WAVES', envs), 1) for r, c in by.items()},
        # vector ALU busy: SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / 1024 SIMDs / 2.4 GHz / the kernel's own duration
        'valu_busy_by_kernel': {r: c['SQ_ACTIVE_INST_VALU'] * 4.0 / 1024.0 / 2.4e9 / max(kms.get(r, 0.0) * 1e-3, 1e-12)
                                for r, c in by.items() if 'SQ_ACTIVE_INST_VALU' in c} if by else None,
        'hbm_traffic_bytes_per_launch': int(traffic) if have_traffic else None,
        'hbm_traffic_by_kernel': {r: int((c.get('WRITE_SIZE', 0) + 2.0 * c.get('FETCH_SIZE', 0)) * 1024) for r, c in by.items()} if have_traffic else None,
        'fetch_correction': 2.0, 'algorithmic_bytes_per_launch': a_bytes,
        'resample_valu_model_min_per_env': model, 'events_per_env': events,
        'kernel_ms_unprofiled': bench['roofline']['kernel_ms'], 'kernels_ms_unprofiled': {k['name']: k['ms'] for k in bench['roofline']['kernels']},
        'source': 'profiles/' + rnd + '_rocprofv3_summary_%s.md (rocprofv3 --pmc, one pass per counter set, 10 measured steps each; SQ cycle counters in '
                  'quad-cycles; FETCH_SIZE doubled per MI355X_MICROARCH.md; traffic = both kernels of a step)' % tag,
    }
    if by:
      records.append(rec)
      if have_traffic:
        raw = sum((c.get('WRITE_SIZE', 0) + c.get('FETCH_SIZE', 0)) * 1024 for c in by.values())
        lines += ['', 'HBM traffic per step = WRITE_SIZE + 2 x FETCH_SIZE (gfx950 correction of MI355X_MICROARCH.md, calibrated there on wide streaming reads) over '
                  'both kernels = %.1f MB; algorithmic bytes = %.1f MB (x %.2f).  Uncorrected (WRITE_SIZE + FETCH_SIZE): %.1f MB (x %.2f) -- the reads here are scalar '
                  'loads and 4-8-byte vector loads, whose FETCH_SIZE about equals the bytes the kernels ask for (the run lists and headers the cover kernel '
                  'writes and the resample kernel reads: 1.6 KB per environment on the headline scene).' % (traffic / 1e6, a_bytes / 1e6, traffic / a_bytes, raw / 1e6, raw / a_bytes)]
      if model:
        lines += ['', 'Resample kernel, vector instructions per environment: measured %.0f, cost-model minimum %.0f (x %.2f; tools/emu_stats.py).' % (
            rec['insts_valu_per_env_by_kernel'].get('resample', 0), model, rec['insts_valu_per_env_by_kernel'].get('resample', 0) / model)]
    if lines:
      open(os.path.join(ROOT, 'profiles', '%s_rocprofv3_summary_%s.md' % (rnd, tag)), 'w').write('\n'.join(lines) + '\n')
  if records:
    json.dump({'records': records}, open(os.path.join(ROOT, 'profiles', rnd + '_counters.json'), 'w'), indent=1)
  for name in ('bench_default', 'bench_driver_cmd'):
    b = os.path.join(src, name + '.json')
    if os.path.exists(b) and os.path.getsize(b) > 10:
      open(os.path.join(ROOT, 'profiles', '%s_%s.json' % (rnd, name)), 'w').write(open(b).read())
  conv = {}
  for name in ('convergence_headline', 'convergence_embodied'):
    b = os.path.join(src, name + '.json')
    if os.path.exists(b) and os.path.getsize(b) > 10:
      conv[name] = json.load(open(b))
  if conv:
    json.dump(conv, open(os.path.join(ROOT, 'profiles', rnd + '_launch_convergence.json'), 'w'), indent=1)
  print('records:', [(r['workload'], r['anti_aliasing'], r['build_id']) for r in records])


if __name__ == '__main__':
  main()
