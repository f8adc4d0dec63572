#!/bin/bash
# round 5, GPU session 10: GPU suite on the current build; the bench line with the event-bracketed timed region
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05k; mkdir -p $OUT
timeout 600 python -m pytest tests -q -m gpu -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_cmd.json 2> $OUT/bench.err
python bench.py --no-cpu-baseline > $OUT/bench_default.json 2>> $OUT/bench.err
python - <<'PY'
import json
for f in ('bench_driver_cmd','bench_default'):
  try:
    d=json.loads(open('gpurun_out/r05k/%s.json'%f).readlines()[-1])
    r=d['roofline']
    print(f, round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), 'kernel_ms', round(r['kernel_ms'],4), 'frac', round(r['frac'],4), [round(k['ms'],4) for k in r['kernels']], d.get('mismatches'), {k: round(v['env_steps_per_s']/1e6,2) for k,v in d.get('extra',{}).items()})
  except Exception as e: print(f, 'failed', e)
PY
tail -3 $OUT/bench.err
