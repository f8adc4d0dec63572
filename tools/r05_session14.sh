#!/bin/bash
# round 5, GPU session 14: more parity on the final build -- a long run (3000 timed steps, 64 sampled environments replayed by the
# oracle) on three workloads, and 1200 more fuzz seeds (every third with a 33 .. 64-gon)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05p; mkdir -p $OUT
for spec in "cluster_s5 5" "goal_s5 5" "embodied_s12 5" "cluster_s5 1"; do
  set -- $spec
  python bench.py --steps 3000 --warmup 20 --no-extra --no-cpu-baseline --workload $1 --aa $2 > $OUT/long_$1_aa$2.json 2>> $OUT/long.err
  python - <<PY
import json
d=json.loads(open('$OUT/long_$1_aa$2.json').readlines()[-1])
print('$1 aa$2', round(d['value']/1e6,2),'M', 'verified', d.get('verified_envs'), 'mismatches', d.get('mismatches'), 'frame bytes differing', d.get('frame_bytes_differing'), 'steps replayed', d.get('steps_replayed'), 'env_errors', d.get('env_errors'))
PY
done
timeout 600 python tools/fuzz_sweep.py 1800 3000 $OUT/fuzz_1800_3000.txt --big-polygons; tail -2 $OUT/fuzz_1800_3000.txt
