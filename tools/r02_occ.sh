#!/bin/bash
# occupancy sensitivity: extra dynamic LDS lowers the resident waves per CU
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/occ
for extra in 0 2000 3500 5000 8000 12000; do
  echo -n "extra_lds=$extra: " | tee -a gpurun_out/occ/occ.txt
  SWB_EXTRA_LDS=$extra python bench.py --steps 100 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), d['roofline']['kernel_ms'], d['roofline']['lds_bytes_per_wave'])" | tee -a gpurun_out/occ/occ.txt
done
