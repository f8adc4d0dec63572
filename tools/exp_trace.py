#!/usr/bin/env python
"""Wave timelines of the two kernels of a step (experiment; needs the library built by
`python tools/overlay_build.py trace trace`, loaded through SWB_LIBRARY).
usage: SWB_LIBRARY=spriteworld_amd/csrc/exp_trace.so python tools/exp_trace.py [WORKLOAD] [N_ENVS] [AA] [OUT.json]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from spriteworld_amd import engine, workloads  # noqa: E402


def pct(a, qs=(0, 5, 25, 50, 75, 95, 99, 100)):
  return {str(q): round(float(np.percentile(a, q)), 2) for q in qs}


def timeline(t, label):
  rt0, rt1, cyc, hw = [t[:, i] for i in range(4)]
  ok = rt1 > 0
  if not ok.any():                                  # (no second kernel: anti_aliasing = 1 painted by the cover kernel)
    print(label, 'no waves recorded', flush=True)
    return {'waves': 0}
  rt0, rt1, cyc, hw = rt0[ok], rt1[ok], cyc[ok], hw[ok]
  t0 = rt0.min()
  start_us, end_us = (rt0 - t0) / 100.0, (rt1 - t0) / 100.0
  hwid, xcc = hw & 0xffffffff, hw >> 32
  key = ((xcc & 15) << 12) | (((hwid >> 13) & 7) << 8) | (((hwid >> 8) & 15) << 4) | ((hwid >> 4) & 3)
  uniq, counts = np.unique(key, return_counts=True)
  res = {'waves': int(ok.sum()), 'span_us': round(float(end_us.max()), 2), 'wave_us': pct(end_us - start_us), 'wave_cycles': pct(cyc),
         'start_us': pct(start_us), 'end_us': pct(end_us), 'simds_used': int(len(uniq)), 'waves_per_simd': pct(counts),
         'simd_last_end_us': pct(np.array([end_us[key == u].max() for u in uniq])),
         'simd_sum_cycles': pct(np.array([cyc[key == u].sum() for u in uniq]))}
  bins = np.arange(0, end_us.max() + 5, 5.0)
  res['active_waves_per_5us'] = [int(((start_us < b + 5) & (end_us > b)).sum()) for b in bins]
  print(label, json.dumps(res), flush=True)
  return res


def main():
  name = sys.argv[1] if len(sys.argv) > 1 else 'cluster_s5'
  n_envs = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
  aa = int(sys.argv[3]) if len(sys.argv) > 3 else 5
  out_path = sys.argv[4] if len(sys.argv) > 4 else None
  cfg, pool, sample = workloads.build(name, n_envs, episodes_per_env=4, seed=0, anti_aliasing=aa)
  eng = engine.Engine(cfg, pool, device=0)
  lib = eng.lib
  lib.swb_exp_set.argtypes = [C.c_void_p, C.c_void_p]
  rng = np.random.default_rng(2000)
  acts = [torch.as_tensor(sample(rng), device=eng.device) for _ in range(16)]
  for i in range(8):
    eng.step(acts[i])
  nb = eng.variant()['n_bands']
  trace = torch.zeros((2, n_envs * nb, 12), dtype=torch.int64, device=eng.device)
  lib.swb_exp_set(eng._h, C.c_void_p(trace.data_ptr()))
  for i in range(4):
    eng.step(acts[8 + i])
  torch.cuda.synchronize()
  t = trace.cpu().numpy().astype(np.int64)
  res = {'workload': name, 'n_envs': n_envs, 'aa': aa, 'bands': nb}
  res['cover'] = timeline(t[0, :n_envs], 'cover')
  c = t[0, :n_envs]
  names = ['loads+paths', 'hit/move', 'task', 'edges(P1b)']
  prev = np.zeros(n_envs)
  if os.environ.get('EXP_P2'):      # the library of tools/overlays/trace_p2.py: cycle sums inside P2 instead of the phase stamps
    for i, nm in enumerate(['head', 'scatter', 'fence', 'words']):
      res['cover_p2_cycles_' + nm] = pct(c[:, 4 + i])
    res['cover_p2_cycles_tail'] = pct(c[:, 11])
    for k in sorted(res):
      if k.startswith('cover_p2'):
        print(k, res[k])
    names = []
  for i, nm in enumerate(names):
    res['cover_phase_cycles_' + nm] = pct(c[:, 4 + i] - prev)
    prev = c[:, 4 + i]
  res['cover_phase_cycles_coverage'] = pct(c[:, 8])
  res['cover_phase_cycles_emit'] = pct(c[:, 9])
  res['cover_batches'] = pct(c[:, 10])
  if names:
    res['cover_phase_cycles_rest'] = pct(c[:, 2] - prev - c[:, 8] - c[:, 9])
  for k in sorted(res):
    if k.startswith('cover_phase') or k == 'cover_batches':
      print(k, res[k])
  res['resample'] = timeline(t[1], 'resample')
  if out_path:
    np.save(out_path.replace('.json', '') + '_trace.npy', t)
    json.dump(res, open(out_path, 'w'), indent=1)
  eng.close()


if __name__ == '__main__':
  main()
