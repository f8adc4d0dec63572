#!/bin/bash
# One gpurun session of round 4: versions, GPU tests, bench lines, convergence.  usage: tools/r04_session.sh TAG [skip-tests]
TAG=${1:-r04a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
{
  python -c "import PIL, matplotlib, sklearn, numpy, scipy; print('numpy', numpy.__version__, 'pillow', PIL.__version__, 'matplotlib', matplotlib.__version__, 'sklearn', sklearn.__version__)"
  nproc; grep -m1 'model name' /proc/cpuinfo; cat /sys/fs/cgroup/cpu.max 2>/dev/null; free -g | head -2
  ls oracle/_ref | head
} > $OUT/host.txt 2>&1
if [ -z "$2" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
  tail -3 $OUT/pytest.log
fi
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $OUT/bench_driver_cmd.json 2>> $OUT/bench_default.err
python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline > $OUT/bench_long.json 2>> $OUT/bench_default.err
python tools/launch_convergence.py cluster_s5 8192 5 48 > $OUT/convergence_headline.json 2> $OUT/convergence.err
python tools/launch_convergence.py embodied_s12 8192 5 32 > $OUT/convergence_embodied.json 2>> $OUT/convergence.err
python - <<PY
import json
for f in ('bench_default','bench_driver_cmd','bench_long'):
  try:
    d=json.load(open('$OUT/%s.json'%f))
    print(f, round(d['value']/1e6,2), 'M', d['ms_per_step'], [round(k['ms'],4) for k in d['roofline']['kernels']], d.get('verified_envs'), d.get('mismatches'), (d.get('cpu_baseline') or {}).get('kind'), (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('cores'))
    for k,v in (d.get('extra') or {}).items(): print('   ', k, round(v['env_steps_per_s']/1e6,2), v.get('kernel_ms'), v.get('cover_ms'), v.get('resample_ms'))
  except Exception as e: print(f, 'ERR', e)
for f in ('convergence_headline','convergence_embodied'):
  try:
    d=json.load(open('$OUT/%s.json'%f))
    for e in d['engines']: print(f, e['trial'], e['step_ms'][:12], e['mean_last_16'], e['first_launch_within_1pct'])
  except Exception as e: print(f, 'ERR', e)
PY
