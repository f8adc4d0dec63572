#!/bin/bash
# Round-2 baseline on the GPU box: instruction micro-benchmark + phase split (time and PMC).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r02a}
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O2 -o /tmp/ubench_valu tools/ubench_valu.hip 2> $OUT/ubench_build.err && /tmp/ubench_valu > $OUT/ubench.md 2>&1
echo "# Phase split of swb_step_kernel ($(date -u +%F))" > $OUT/phase.md
echo >> $OUT/phase.md
python tools/phase_profile.py time $OUT/phase.md > $OUT/phase_time.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY -d $OUT/pmc -o p -- python tools/phase_profile.py pmc-run $OUT/order.json > $OUT/pmc_run.log 2>&1
python tools/phase_profile.py pmc-report $OUT/order.json $(find $OUT/pmc -name "*.db") >> $OUT/phase.md 2> $OUT/pmc_report.err
find $OUT -name "*.db" -delete
cat $OUT/ubench.md; cat $OUT/phase.md
