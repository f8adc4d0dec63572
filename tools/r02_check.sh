#!/bin/bash
# GPU box: parity tests of the default build (variants only when it fails), A/B bench of all builds,
# phase split of the default build.  usage: tools/r02_check.sh TAG [full|quick] [phase]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r02b}; MODE=${2:-quick}; PHASE=${3:-nophase}
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ "$MODE" = full ]; then TESTS="tests"; else TESTS="tests/test_gpu_parity.py tests/test_golden.py"; fi
timeout 900 python -m pytest $TESTS -x -q -m gpu > $OUT/pytest_default.log 2>&1
echo "default pytest rc=$?" | tee $OUT/status.txt
tail -5 $OUT/pytest_default.log
if ! grep -q " passed" $OUT/pytest_default.log || grep -q "failed" $OUT/pytest_default.log; then
  for lib in $(ls spriteworld_amd/csrc/exp_*.so 2>/dev/null); do
    SWB_LIBRARY=$PWD/$lib timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $OUT/pytest_$(basename $lib .so).log 2>&1
    echo "$lib pytest rc=$?" | tee -a $OUT/status.txt
    tail -3 $OUT/pytest_$(basename $lib .so).log
  done
fi
for rep in 1 2; do
for lib in spriteworld_amd/csrc/libswb.so $(ls spriteworld_amd/csrc/exp_*.so 2>/dev/null); do
  for wl in "cluster_s5 5" "cluster_s5 1" "embodied_s12 5"; do
    set -- $wl
    if [ -n "${SWB_ALT_ENV:-}" ]; then
      echo -n "$(basename $lib) [$SWB_ALT_ENV] $1 aa$2: " | tee -a $OUT/bench.txt
      env $SWB_ALT_ENV SWB_LIBRARY=$PWD/$lib python bench.py --steps 100 --workload $1 --aa $2 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), d['roofline']['kernel_ms'], d['env_errors'])" | tee -a $OUT/bench.txt
    fi
    echo -n "$(basename $lib) $1 aa$2: " | tee -a $OUT/bench.txt
    SWB_LIBRARY=$PWD/$lib python bench.py --steps 100 --workload $1 --aa $2 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), d['roofline']['kernel_ms'], d['env_errors'])" | tee -a $OUT/bench.txt
  done
done
done
if [ "$PHASE" = phase ]; then
  echo "# Phase split of swb_step_kernel ($TAG)" > $OUT/phase.md; echo >> $OUT/phase.md
  python tools/phase_profile.py time $OUT/phase.md > $OUT/phase_time.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY -d $OUT/pmc -o p -- python tools/phase_profile.py pmc-run $OUT/order.json > $OUT/pmc_run.log 2>&1
  python tools/phase_profile.py pmc-report $OUT/order.json $(find $OUT/pmc -name "*.db") >> $OUT/phase.md 2> $OUT/pmc_report.err
  find $OUT -name "*.db" -delete
  cat $OUT/phase.md
fi
