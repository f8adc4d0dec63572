#!/bin/bash
# GPU box, last call of round 2 (a few minutes of budget): in order of priority
#   1. the whole GPU test suite on the shipped build                          -> $OUT/pytest_gpu.log
#   2. headline workload: bench line + rocprofv3 kernel trace                  -> $OUT/default/
#   3. PMC passes (4 separate processes, each running the three judged workloads once)  -> $OUT/pmc_*.json
#   4. the default `python bench.py` line (extras + cpu baseline)             -> $OUT/bench_default.json
#   5. instruction micro-benchmark, round-2 additions                          -> $OUT/ubench_new.md
#   6. kernel traces of the AA=1 and the 12-sprite 128x128 workloads, phase times
# Every step writes its result at once (the call may be cut by the budget).   usage: tools/r02_last.sh [TAG]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r02last}
OUT=gpurun_out/$TAG
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }

stamp "pytest -m gpu"
timeout 420 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/timeline.txt
tail -5 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log

trace() {   # tag workload aa
  local D=$OUT/$1; mkdir -p $D
  local ARGS="--steps 40 --warmup 5 --no-extra --no-cpu-baseline --workload $2 --aa $3"
  python bench.py $ARGS > $D/bench_unprofiled.json 2> $D/bench.err
  rocprofv3 --kernel-trace --stats -d $D/trace -o t -- python bench.py $ARGS > $D/bench_trace.json 2> $D/trace.err
  python tools/rocprof_summary.py $D/summary.md "rocprofv3 summary (round 2 final, $1): python bench.py $ARGS" $(find $D/trace -name "*.db" | head -1) > /dev/null 2>> $D/trace.err
  echo >> $D/summary.md; echo '```' >> $D/summary.md; cat $D/bench_unprofiled.json >> $D/summary.md; echo '```' >> $D/summary.md
  find $D -name "*.db" -delete
  python -c "
import json,sys; d=json.loads(open('$D/bench_unprofiled.json').readlines()[-1]); print('$1', round(d['value']), d['roofline']['kernel'], d['roofline']['kernel_ms'], d['env_errors'])"
}
stamp "headline bench + trace"
trace default cluster_s5 5

export PHASE_WORKLOADS="cluster_s5:5,cluster_s5:1,embodied_s12:5" PHASE_LIST="0"
pmc() {   # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" -d $OUT/pmc_$name -o p -- python tools/phase_profile.py pmc-run $OUT/order_$name.json > $OUT/pmc_$name.log 2>&1
  python tools/phase_profile.py pmc-json $OUT/order_$name.json $(find $OUT/pmc_$name -name "*.db") > $OUT/pmc_$name.json 2>> $OUT/pmc_$name.log
  find $OUT/pmc_$name -name "*.db" -delete
  head -c 400 $OUT/pmc_$name.json; echo
}
stamp "pmc insts"
pmc insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
stamp "pmc write"
pmc write WRITE_SIZE
stamp "pmc fetch"
pmc fetch FETCH_SIZE
stamp "pmc active"
pmc active SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM
unset PHASE_WORKLOADS PHASE_LIST

stamp "default bench line"
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.json; echo

stamp "ubench"
if [ -x tools/_build/ubench_valu ]; then timeout 60 tools/_build/ubench_valu 200000 new > $OUT/ubench_new.md 2>&1; tail -32 $OUT/ubench_new.md; fi

stamp "aa1 / embodied traces"
trace aa1 cluster_s5 1
trace embodied_s12_128 embodied_s12 5

stamp "phase times"
echo "# Phase split of swb_step_kernel ($TAG)" > $OUT/phase.md; echo >> $OUT/phase.md
PHASE_WORKLOADS="cluster_s5:5,cluster_s5:1,embodied_s12:5" timeout 200 python tools/phase_profile.py time $OUT/phase.md > $OUT/phase_time.log 2>&1
cat $OUT/phase.md
stamp "done"
stamp "extra fuzz seeds (only if budget remains)"
timeout 120 python tools/fuzz_sweep.py 224 284 $OUT/fuzz_extra.txt; tail -2 $OUT/fuzz_extra.txt
stamp "end"
