#!/bin/bash
# round 5, GPU session 6: cover launch order between one and two rounds of waves (heaviest + lightest first, the middle second)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05g; mkdir -p $OUT
SPECS="cluster_s5:8192:5 cluster_s5:8192:1 goal_s5:8192:5 cluster_s5:6144:5 cluster_s5:10240:5"
for r in 1 2 3; do
  echo "== plain round $r";      SWB_NO_COVER_MIX=1 python tools/quick_bench.py $SPECS 2>&1 | grep -v amdgpu.ids
  echo "== mixed round $r";      python tools/quick_bench.py $SPECS 2>&1 | grep -v amdgpu.ids
  echo "== mixed_noprio round $r"; SWB_NO_COVER_PRIO=1 python tools/quick_bench.py $SPECS 2>&1 | grep -v amdgpu.ids
  echo "== plain_noprio round $r"; SWB_NO_COVER_MIX=1 SWB_NO_COVER_PRIO=1 python tools/quick_bench.py $SPECS 2>&1 | grep -v amdgpu.ids
done > $OUT/mix.txt 2>&1
python - <<'PY'
import re, collections
d = collections.defaultdict(list); lib=None
for line in open('gpurun_out/r05g/mix.txt'):
  m = re.match(r'== (\S+) round', line)
  if m: lib = m.group(1); continue
  m = re.match(r'(\S+)\s+N=(\d+)\s+AA=(\d).*step ([\d.]+) ms\s+cover ([\d.]+)', line)
  if m: d[(m.group(1), m.group(2), m.group(3), lib)].append((float(m.group(4)), float(m.group(5))))
for k in sorted(d):
  v = d[k]; print('%-12s N=%-6s AA=%s %-14s step %s cover %s' % (k[0], k[1], k[2], k[3], ' '.join('%.4f' % x[0] for x in v), ' '.join('%.4f' % x[1] for x in v)))
PY
C=spriteworld_amd/csrc
SWB_LIBRARY=$PWD/$C/exp_trace.so python tools/exp_trace.py cluster_s5 8192 5 $OUT/timeline_8192.json > $OUT/timeline_8192.log 2>&1
