#!/usr/bin/env python
"""DESIGN.md section 4's roofline table from the committed bench line and PMC records (profiles/rNN_bench_default.json,
rNN_bench_driver_cmd.json, rNN_counters.json): rewrites the block between the ROOFLINE_TABLE markers.
usage: python tools/design_numbers.py r06"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
  rnd = sys.argv[1] if len(sys.argv) > 1 else 'r06'
  d = json.loads(open(os.path.join(ROOT, 'profiles', rnd + '_bench_default.json')).readlines()[-1])
  dc = json.loads(open(os.path.join(ROOT, 'profiles', rnd + '_bench_driver_cmd.json')).readlines()[-1])
  c = {(r['workload'], r['anti_aliasing']): r for r in json.load(open(os.path.join(ROOT, 'profiles', rnd + '_counters.json')))['records']}
  ex = d['extra']

  def row(name, ms, rate, frac, traffic='', busy=''):
    return '| %s | %.4f | %.1f M | %.0f | **%.3f** | %s | %s |' % (name, ms, rate / 1e6, frac * 8000, frac, traffic, busy)

  def ratio(r):
    return '%.2f ×' % (r['hbm_traffic_bytes_per_launch'] / r['algorithmic_bytes_per_launch']) if r.get('hbm_traffic_bytes_per_launch') else ''

  def busy(r):
    b = r.get('valu_busy_by_kernel') or {}
    return ' / '.join('%.2f' % b[k] for k in ('cover', 'resample') if k in b)

  h, a, e = c[('cluster_s5', 5)], c[('cluster_s5', 1)], c[('embodied_s12', 5)]
  t = ['| workload (8192 environments unless said) | ms per step (one pair of HIP events / steps) | env-steps/s | A·N / t (GB/s) | of 8 TB/s | HBM traffic (PMC) / algorithmic | vector ALU busy, cover / second kernel |',
       '|---|---|---|---|---|---|---|',
       row('**headline** cluster_s5, anti_aliasing 5 (`python bench.py`: %d steps after %d)' % (d['steps'], d['warmup']), d['ms_per_step'], d['value'], d['roofline']['frac'], ratio(h), busy(h)),
       row("the same, the driver's `--steps %d --warmup %d`" % (dc['steps'], dc['warmup']), dc['ms_per_step'], dc['value'], dc['roofline']['frac'])]
  for label, key, n, rec in (('cluster_s5, anti_aliasing 1 (the cover kernel paints the frame)', 'cluster_s5_aa1', 8192, a),
                             ('configs[1]: goal_s5, 1024 environments, anti_aliasing 5', 'goal_s5_1024_aa5', 1024, None),
                             ('configs[1], anti_aliasing 1', 'goal_s5_1024_aa1', 1024, None),
                             ('configs[4]: embodied_s12, 12 sprites, 128×128, anti_aliasing 5', 'embodied_s12_128_aa5', 8192, e),
                             ('configs[4], anti_aliasing 1', 'embodied_s12_128_aa1', 8192, None),
                             ('cluster_s5, 65 536 environments in one launch', 'cluster_s5_65536_aa5', 65536, None),
                             ('the same, anti_aliasing 1', 'cluster_s5_65536_aa1', 65536, None)):
    if key not in ex:
      continue
    x = ex[key]
    t.append(row(label, x['kernel_ms'], n / x['kernel_ms'] * 1e3, x['hbm_frac'], ratio(rec) if rec else '', busy(rec) if rec else ''))
  t.append('| the headline batch as two groups of 4096 on two HIP streams (`EnvironmentGroups`; host wall time) | | %.1f M | | | | |' % (
      ex['cluster_s5_2_groups_2_streams']['env_steps_per_s'] / 1e6))
  t += ['', '(build `%s`; rates of the extra rows = environments / kernel time; `traffic` = WRITE_SIZE + 2 × FETCH_SIZE of both kernels, separate `--pmc` '
        'passes, `profiles/%s_counters.json`; box-to-box and run-to-run the headline moves by ± 1 %%.)' % (d['roofline']['build_id'], rnd)]
  block = '<!-- ROOFLINE_TABLE_BEGIN -->\n' + '\n'.join(t) + '\n<!-- ROOFLINE_TABLE_END -->'
  p = os.path.join(ROOT, 'DESIGN.md')
  s = open(p).read()
  assert '<!-- ROOFLINE_TABLE_BEGIN -->' in s
  s = re.sub(r'<!-- ROOFLINE_TABLE_BEGIN -->.*?<!-- ROOFLINE_TABLE_END -->', lambda m: block, s, flags=re.S)
  open(p, 'w').write(s)
  print(block)


if __name__ == '__main__':
  main()
