#!/bin/bash
# GPU box: timing of experimental builds (no parity) + an extra PMC set on the default build.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r02e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for rep in 1 2; do
for lib in spriteworld_amd/csrc/libswb.so $(ls spriteworld_amd/csrc/exp_*.so 2>/dev/null); do
  echo -n "$(basename $lib) cluster_s5 aa5: " | tee -a $OUT/bench.txt
  SWB_LIBRARY=$PWD/$lib python bench.py --steps 100 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), d['roofline']['kernel_ms'], d['env_errors'])" | tee -a $OUT/bench.txt
done
done
export PHASE_WORKLOADS="cluster_s5:5" PHASE_LIST="2,0"
echo "# extra counters ($TAG)" > $OUT/pmc2.md
rocprofv3 --pmc SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES -d $OUT/pmcA -o p -- python tools/phase_profile.py pmc-run $OUT/orderA.json > $OUT/pmcA.log 2>&1
python tools/phase_profile.py pmc-report $OUT/orderA.json $(find $OUT/pmcA -name "*.db") >> $OUT/pmc2.md 2>> $OUT/pmc_report.err
rocprofv3 --pmc SQ_WAVES SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INSTS_LDS SQ_IFETCH SQ_BUSY_CYCLES SQ_INST_CYCLES_SMEM -d $OUT/pmcB -o p -- python tools/phase_profile.py pmc-run $OUT/orderB.json > $OUT/pmcB.log 2>&1
python tools/phase_profile.py pmc-report $OUT/orderB.json $(find $OUT/pmcB -name "*.db") >> $OUT/pmc2.md 2>> $OUT/pmc_report.err
find $OUT -name "*.db" -delete
cat $OUT/pmc2.md
