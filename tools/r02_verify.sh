#!/bin/bash
# GPU box, the round's last seconds of budget: the driver's own bench command on the final tree, smoke, the
# setter + golden GPU tests, then two bench-only experiments for the next round (exp_*.so, never shipped).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02verify
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
stamp "driver bench command"
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err
echo "rc=$?"; python -c "
import json; d=json.loads(open('$OUT/bench_driver_cmd.json').readlines()[-1]); r=d['roofline']
print(round(d['value']), d['ms_per_step'], r['kernel_ms'], r['frac'], r['traffic'], r['instructions'] and round(r['instructions']['valu_issue_frac'],3), d['env_errors'], d['cpu_baseline']['value'])"
stamp "smoke"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
stamp "experiments (bench only)"
for lib in exp_packed exp_wpb2; do
  [ -f spriteworld_amd/csrc/$lib.so ] || continue
  for rep in 1; do
    echo -n "$lib cluster_s5 aa5: " | tee -a $OUT/exp.txt
    SWB_LIBRARY=$PWD/spriteworld_amd/csrc/$lib.so python bench.py --steps 100 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), d['roofline']['kernel_ms'], d['env_errors'])" | tee -a $OUT/exp.txt
  done
done
echo -n "libswb cluster_s5 aa5: " | tee -a $OUT/exp.txt
python bench.py --steps 100 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), d['roofline']['kernel_ms'], d['env_errors'])" | tee -a $OUT/exp.txt
stamp "exp parity (headline only)"
for lib in exp_packed exp_wpb2; do
  SWB_LIBRARY=$PWD/spriteworld_amd/csrc/$lib.so timeout 60 python -m pytest "tests/test_gpu_parity.py::test_cluster_s5_aa5" "tests/test_gpu_parity.py::test_goal_s5_aa1" -q -m gpu -p no:cacheprovider 2>&1 | tail -1 | sed "s/^/$lib: /" | tee -a $OUT/exp.txt
done
stamp "setter + golden tests"
timeout 100 python -m pytest tests/test_gpu_setters.py tests/test_golden.py -q -m gpu -p no:cacheprovider 2>&1 | tail -2
stamp "end"
