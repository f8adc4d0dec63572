#!/bin/bash
# round 5, GPU session 13: instructions of the cover kernel by phase (PMC insts pass over the phase-cut builds)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
C=spriteworld_amd/csrc
for aa in 5 1; do
for k in 3 4 5 1; do
  SWB_LIBRARY=$PWD/$C/exp_phase$k.so bash tools/pmc_sets.sh r05n/aa$aa/phase$k cluster_s5 8192 $aa insts > /dev/null 2>&1
done
bash tools/pmc_sets.sh r05n/aa$aa/full cluster_s5 8192 $aa insts > /dev/null 2>&1
done
python - <<'PY'
import json, glob
for aa in (5, 1):
  prev = {}
  for k in ('phase3', 'phase4', 'phase5', 'phase1', 'full'):
    d = json.load(open('gpurun_out/r05n/aa%d/%s/pmc_insts.json' % (aa, k)))
    c = [v for kk, v in d.items() if 'cover' in kk][0]
    cur = {n: c[n] / 8192 for n in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_SMEM', 'SQ_WAVE_CYCLES')}
    print('AA=%d %-7s' % (aa, k), ' '.join('%s %7.0f (+%6.0f)' % (n[8:] if n.startswith('SQ_INSTS') else 'WCYC', cur[n], cur[n] - prev.get(n, 0)) for n in cur))
    prev = cur
PY
