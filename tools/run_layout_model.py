#!/usr/bin/env python
"""Cost model of the resample kernel under other RUN LAYOUTS (round-3 verdict, item 4) -- from the exact canvases of the headline
scene, rendered by the unmodified reference (TEST / ANALYSIS TOOL: needs /root/reference or oracle/_ref; nothing of the product
imports it).

A run = consecutive canvas rows with identical visible spans; the kernel pays, per run, the horizontal pass (9 vector
instructions for the first span, 10 per further span, 6 for the clip) and 3 multiply-adds per accumulator slot in flight.  Today
(layout A) a run also ends wherever an OUTPUT row ends (`v_break`), because the slot of the finished row is restarted there: VS = 6
slots, 18 multiply-adds.  Layouts compared:
  A  break at every output-row end (shipped)                 6 slots -> 33 per single-span run
  B  a run may straddle ONE output-row end                   7 slots -> 36 per single-span run
  C  a run may straddle TWO output-row ends                  8 slots -> 39 per single-span run
  D  no breaks at all (lower bound on the number of runs; needs 6 + (rows of the longest run) / 5 slots -- not realisable)
usage: python tools/run_layout_model.py [ENVS] [STEPS]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import numpy as np  # noqa: E402


def row_signature(row):
  """Visible spans of a canvas row (black background): tuple of (start, end, colour)."""
  nz = row.any(axis=1)
  if not nz.any():
    return ()
  key = (row[:, 0].astype(np.int64) << 16) | (row[:, 1].astype(np.int64) << 8) | row[:, 2]
  change = np.flatnonzero(np.diff(key)) + 1
  starts = np.concatenate(([0], change))
  ends = np.concatenate((change, [len(key)]))
  return tuple((int(a), int(b), int(key[a])) for a, b in zip(starts, ends) if key[a] != 0)


def count(canvases, row_ends, max_crossings):
  """(runs, spans summed over runs, rows in runs) per canvas on average; max_crossings: output-row ends a run may straddle."""
  runs = spans = rows = 0
  ends = set(int(e) for e in row_ends)
  for canvas in canvases:
    prev, crossed = None, 0
    for y in range(canvas.shape[0]):
      sig = row_signature(canvas[y])
      if not sig:
        prev = None
        continue
      rows += 1
      boundary = (y - 1) in ends                 # an output row ended on the row above
      if prev is not None and sig == prev and (not boundary or crossed < max_crossings):
        crossed += 1 if boundary else 0
      else:
        runs += 1
        spans += len(sig)
        crossed = 0
      prev = sig
  n = float(len(canvases))
  return runs / n, spans / n, rows / n


def main():
  envs = int(sys.argv[1]) if len(sys.argv) > 1 else 96
  steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
  from oracle import ref_harness
  ref = ref_harness.load_reference()
  import reference_cpu_baseline as rcb
  from spriteworld_amd import lanczos
  bounds, _ = lanczos.resample_tables(320, 64)
  bounds = np.asarray(bounds).reshape(-1, 2)
  row_ends = bounds[:, 0] + bounds[:, 1] - 1
  canvases = []
  for i in range(envs):
    env = rcb.make_env(ref, i)
    env.reset()
    rng = np.random.RandomState(2000 + i)
    for _ in range(steps):
      env.step(rng.uniform(size=4))
    rend = env._renderers['image']
    canvases.append(np.array(rend._canvas)[:, :, :3].copy())
  print('| layout | accumulator slots | runs / env | spans / run | rows / run | vector instructions / run (1 span) | resample VALU model / env |')
  print('|---|---|---|---|---|---|---|')
  finished = 64 - 19            # finished output rows that received something (r03_counters: 64 rows, 18.9 untouched)
  for name, crossings, slots in (('A: break at every output-row end (shipped)', 0, 6), ('B: may straddle one end', 1, 7),
                                 ('C: may straddle two ends', 2, 8), ('D: no breaks (lower bound on runs)', 10 ** 6, None)):
    runs, spans, rows = count(canvases, row_ends, crossings)
    if slots is None:
      print('| %s | -- | %.1f | %.2f | %.2f | -- | -- |' % (name, runs, spans / runs, rows / runs))
      continue
    per_run = 15 + 3 * slots
    model = per_run * runs + 10 * (spans - runs) + (11 + slots) * finished + 19 + 150
    print('| %s | %d | %.1f | %.2f | %.2f | %d | %.0f |' % (name, slots, runs, spans / runs, rows / runs, per_run, model))


if __name__ == '__main__':
  main()
