#!/bin/bash
# round 5, GPU session 11 (closing): GPU suite, the two bench lines with the committed counters in place, the round-4 build timed
# without per-step events, a fuzz sweep on the final build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05l; mkdir -p $OUT
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2>> $OUT/bench.err
python - <<'PY'
import json
for f in ('bench_default','bench_driver_cmd'):
  d=json.loads(open('gpurun_out/r05l/%s.json'%f).readlines()[-1]); r=d['roofline']
  print(f, round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), 'kernel_ms', round(r['kernel_ms'],4), 'frac', round(r['frac'],4), 'traffic', r['traffic'], 'instr', bool(r.get('instructions')), d.get('mismatches'), d['cpu_baseline']['kind'], round(d['cpu_baseline']['value']))
PY
C=spriteworld_amd/csrc
echo "== round-4 final build, no per-step events" > $OUT/timing_overhead.txt
SWB_LIBRARY=$PWD/$C/exp_r4final.so python tools/exp_timing_overhead.py 2>&1 | grep -v amdgpu.ids >> $OUT/timing_overhead.txt
echo "== this build" >> $OUT/timing_overhead.txt
python tools/exp_timing_overhead.py 2>&1 | grep -v amdgpu.ids >> $OUT/timing_overhead.txt
cat $OUT/timing_overhead.txt
timeout 420 python tools/fuzz_sweep.py 1200 1800 $OUT/fuzz_1200_1800.txt --big-polygons; tail -2 $OUT/fuzz_1200_1800.txt
