#!/bin/bash
# GPU box: bench-only A/B of exp_corner.so (tools/next_round/b_corner_rule_candidate_lists.patch) + the GPU parity tests it touches.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02corner; mkdir -p $OUT
b() { python bench.py --steps 100 --no-extra --no-cpu-baseline --workload $1 --aa $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), d['roofline']['kernel_ms'], d['env_errors'])"; }
for spec in "cluster_s5 1" "cluster_s5 5"; do
  set -- $spec
  echo -n "exp_corner $1 aa$2: " | tee -a $OUT/exp.txt; SWB_LIBRARY=$PWD/spriteworld_amd/csrc/exp_corner.so b $1 $2 | tee -a $OUT/exp.txt
  echo -n "libswb     $1 aa$2: " | tee -a $OUT/exp.txt; b $1 $2 | tee -a $OUT/exp.txt
done
SWB_LIBRARY=$PWD/spriteworld_amd/csrc/exp_corner.so timeout 40 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "aa1 or tiny or goal_s5_aa5" 2>&1 | tail -1 | tee -a $OUT/exp.txt
