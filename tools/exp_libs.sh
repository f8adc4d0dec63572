# A/B of library builds on one box: SWB_LIBRARY selects the build (see spriteworld_amd/_lib.py)
for rep in 1 2; do
for lib in libswb.so $(cd spriteworld_amd/csrc && ls exp_*.so 2>/dev/null); do
  echo -n "$lib: "
  SWB_LIBRARY=$PWD/spriteworld_amd/csrc/$lib python bench.py --steps 200 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['roofline']['kernel_ms'], d['env_errors'])"
done
done
