#!/bin/bash
# GPU box, end of round: full GPU test suite; if green, the rocprofv3 summaries + counters of the three judged
# workloads (tools/r02_profile.sh), the phase split (time + PMC) and the default bench line.
# usage: tools/r02_final.sh TAG
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r02final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
rc=$?
tail -4 $OUT/pytest_gpu.log
if [ $rc != 0 ]; then echo "GPU tests failed (rc=$rc): no profile"; exit 1; fi
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
bash tools/r02_profile.sh > $OUT/profile.log 2>&1
echo "# Phase split of swb_step_kernel ($TAG)" > $OUT/phase.md; echo >> $OUT/phase.md
python tools/phase_profile.py time $OUT/phase.md > $OUT/phase_time.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY -d $OUT/pmc -o p -- python tools/phase_profile.py pmc-run $OUT/order.json > $OUT/pmc_run.log 2>&1
python tools/phase_profile.py pmc-report $OUT/order.json $(find $OUT/pmc -name "*.db") >> $OUT/phase.md 2> $OUT/pmc_report.err
find $OUT -name "*.db" -delete
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -1 $OUT/bench_default.json
