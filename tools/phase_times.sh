# Kernel time with the step kernel cut short after each phase (SWB_DEBUG_PHASE), from bench.py's HIP events.
for ph in 3 4 5 1 2 0; do
  echo -n "phase $ph: "
  SWB_DEBUG_PHASE=$ph python bench.py --steps 50 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline'].get('kernel_ms'))"
done
