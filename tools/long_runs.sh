#!/bin/bash
# GPU box: long-run parity of the shipped build -- 3000 timed steps on four workloads at 8192 environments, the last step of 64
# sampled environments compared with the oracle replayed through the same actions (bench.py's in-run check).  usage: tools/long_runs.sh OUT.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/long_runs.txt}
mkdir -p $(dirname $OUT)
echo "# long runs (python bench.py --steps 3000 --warmup 20 --no-extra --no-cpu-baseline --workload W --aa A): 8192 environments, 3020 steps each, the last step of 64 sampled environments compared with the oracle replayed through the same 3020 actions" > $OUT
for spec in "cluster_s5 1" "cluster_s5 5" "embodied_s12 5" "goal_s5 5"; do
  set -- $spec
  python bench.py --steps 3000 --warmup 20 --no-extra --no-cpu-baseline --workload $1 --aa $2 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('%s | build %s | %.2f M env-steps/s | verified %d mismatches %d frame bytes differing %d max abs diff %d steps replayed %d env_errors %d' % (
    d['config']['workload'][:60], d['roofline']['build_id'], d['value'] / 1e6, d['verified_envs'], d['mismatches'], d['frame_bytes_differing'], d['frame_max_abs_diff'], d['steps_replayed'], d['env_errors']))" >> $OUT
done
cat $OUT
