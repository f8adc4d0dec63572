#!/bin/bash
# round 5, GPU session 7: staggered start of the first round of cover waves; the reordered default bench line against the driver's command
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05h; mkdir -p $OUT
C=spriteworld_amd/csrc
bash tools/r04_ab.sh r05h 3 "cluster_s5:8192:5 cluster_s5:8192:1 embodied_s12:8192:5" $C/libswb.so $C/exp_stagger20.so $C/exp_stagger50.so $C/exp_stagger100.so
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_cmd.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench_driver_cmd_noextra.json 2>> $OUT/bench.err
python bench.py --no-cpu-baseline > $OUT/bench_default.json 2>> $OUT/bench.err
python - <<'PY'
import json
for f in ('bench_driver_cmd','bench_driver_cmd_noextra','bench_default'):
  d=json.loads(open('gpurun_out/r05h/%s.json'%f).readlines()[-1])
  print(f, round(d['value']/1e6,2), 'M', d['ms_per_step'], [round(k['ms'],4) for k in d['roofline']['kernels']], d.get('mismatches'), d.get('order','-')[:30])
PY
