#!/bin/bash
# round 5, GPU session 8: the default bench line with every engine built first and the extra workloads run back to back before the
# headline (driver's command and default); P1b inner timeline
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05i; mkdir -p $OUT
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_cmd.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/bench_driver_cmd_noextra.json 2>> $OUT/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_cmd2.json 2>> $OUT/bench.err
python bench.py --no-cpu-baseline > $OUT/bench_default.json 2>> $OUT/bench.err
python - <<'PY'
import json
for f in ('bench_driver_cmd','bench_driver_cmd_noextra','bench_driver_cmd2','bench_default'):
  try:
    d=json.loads(open('gpurun_out/r05i/%s.json'%f).readlines()[-1])
    print(f, round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), [round(k['ms'],4) for k in d['roofline']['kernels']], d.get('mismatches'), {k: round(v['env_steps_per_s']/1e6,2) for k,v in d.get('extra',{}).items()})
  except Exception as e: print(f, 'failed', e)
PY
tail -5 $OUT/bench.err
C=spriteworld_amd/csrc
SWB_LIBRARY=$PWD/$C/exp_trace.so python tools/exp_trace.py cluster_s5 8192 5 $OUT/timeline_8192.json > $OUT/timeline_8192.log 2>&1
SWB_LIBRARY=$PWD/$C/exp_trace.so python tools/exp_trace.py cluster_s5 8192 1 $OUT/timeline_8192_aa1.json > $OUT/timeline_8192_aa1.log 2>&1
