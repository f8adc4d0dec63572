#!/usr/bin/env python
"""Summarises rocprofv3 (rocpd .db) outputs into a markdown file for profiles/.

usage: rocprof_summary.py OUT.md TITLE trace.db [pmc1.db pmc2.db ...]
"""
import sqlite3
import sys


def kernel_stats(db):
  cur = sqlite3.connect(db).cursor()
  rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
  return rows


def pmc_stats(db):
  cur = sqlite3.connect(db).cursor()
  q = ('select kernel_name, counter_name, avg(value), count(*), max(vgpr_count), max(sgpr_count), '
       'max(lds_block_size), max(grid_size), max(workgroup_size) from counters_collection '
       'group by kernel_name, counter_name')
  return list(cur.execute(q))


def main():
  out, title, trace = sys.argv[1], sys.argv[2], sys.argv[3]
  lines = ['# ' + title, '', '## Kernel trace (`rocprofv3 --kernel-trace --stats`)', '',
           '| kernel | calls | total (us) | average (us) | % |', '|---|---|---|---|---|']
  for name, calls, total, avg, pct in kernel_stats(trace)[:8]:
    lines.append('| `%s` | %d | %.1f | %.3f | %.2f |' % (name[:90], calls, total, avg, pct))
  for db in sys.argv[4:]:
    rows = [r for r in pmc_stats(db) if 'swb_' in r[0]]
    if not rows:
      continue
    lines += ['', '## PMC pass `%s` (per-dispatch averages)' % db.split('/')[-1], '',
              '| kernel | counter | avg per dispatch | dispatches |', '|---|---|---|---|']
    for r in rows:
      lines.append('| `%s` | %s | %.6g | %d |' % (r[0][:60], r[1], r[2], r[3]))
    r = rows[0]
    lines.append('')
    lines.append('dispatch: grid %s, workgroup %s, VGPRs %s, SGPRs %s, LDS %s B/workgroup' % (r[7], r[8], r[4], r[5], r[6]))
  open(out, 'w').write('\n'.join(lines) + '\n')
  print('wrote', out)


if __name__ == '__main__':
  main()
