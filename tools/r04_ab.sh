#!/bin/bash
# A/B session: interleaved quick benches of several builds, phase profile and PMC passes of the shipped one.
# usage: tools/r04_ab.sh TAG ROUNDS "SPECS" lib...      (then: phase profile + PMC when PHASES=1 / PMC=1 are set)
TAG=$1; ROUNDS=$2; SPECS=$3; shift 3
OUT=gpurun_out/$TAG; mkdir -p $OUT
bash tools/bench_libs.sh $ROUNDS "$SPECS" "$@" > $OUT/ab.txt 2>&1
python - <<PY
import re, collections
d = collections.defaultdict(list)
lib = None
for line in open('$OUT/ab.txt'):
  m = re.match(r'== (\S+) round', line)
  if m: lib = m.group(1); continue
  m = re.match(r'(\S+)\s+N=(\d+)\s+AA=(\d).*step ([\d.]+) ms\s+cover ([\d.]+)\s+\S+(?: \S+)* ([\d.]+)\s+\(', line)
  if m: d[(m.group(1), m.group(2), m.group(3), lib)].append((float(m.group(4)), float(m.group(5)), float(m.group(6))))
for k in sorted(d):
  v = d[k]
  print('%-14s N=%-6s AA=%s %-14s step %s cover %s second %s' % (k[0], k[1], k[2], k[3], ' '.join('%.4f' % x[0] for x in v), ' '.join('%.4f' % x[1] for x in v), ' '.join('%.4f' % x[2] for x in v)))
PY
if [ -n "$PHASES" ]; then rm -f $OUT/phases.md; python tools/phase_profile.py $OUT/phases.md cluster_s5:5 cluster_s5:1 embodied_s12:5 2>&1 | tail -4; fi
if [ -n "$PMC" ]; then
  bash tools/pmc_sets.sh $TAG/pmc_headline cluster_s5 8192 5 active insts wait > $OUT/pmc.txt 2>&1
  bash tools/pmc_sets.sh $TAG/pmc_aa1 cluster_s5 8192 1 active insts >> $OUT/pmc.txt 2>&1
  tail -c 3000 $OUT/pmc.txt
fi
