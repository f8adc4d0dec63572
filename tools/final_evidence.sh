#!/bin/bash
# GPU box: the evidence of a round (ROUND=r04 ...), in order of priority (every step writes its result at once; the call may be cut by the budget)
#   1. the whole GPU test suite + smoke                                        -> $OUT/pytest_gpu.log
#   2. headline workload: bench line + rocprofv3 kernel trace                   -> $OUT/default/
#   3. PMC passes over the three judged workloads (one process per counter set) -> $OUT/<workload>/pmc_*.json
#   4. the default `python bench.py` line (extras + cpu baseline)              -> $OUT/bench_default.json
#   5. kernel traces of the AA = 1 and the 12-sprite 128x128 workloads
#   6. wave timelines of both kernels (experiment build)                       -> $OUT/timeline_*.json
#   7. kernel time by phase (experiment builds cut short after a phase)        -> $OUT/phase_times.md
# usage: ROUND=r04 tools/final_evidence.sh [TAG]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ROUND=${ROUND:-r06}
TAG=${1:-${ROUND}final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }

stamp "pytest -m gpu"
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/timeline.txt
tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log

trace() {   # tag workload aa
  local D=$OUT/$1; mkdir -p $D
  local ARGS="--steps 40 --warmup 5 --no-extra --no-cpu-baseline --workload $2 --aa $3"
  python bench.py $ARGS > $D/bench_unprofiled.json 2> $D/bench.err
  rocprofv3 --kernel-trace --stats -d $D/trace -o t -- python bench.py $ARGS > $D/bench_trace.json 2> $D/trace.err
  python tools/rocprof_summary.py $D/summary.md "rocprofv3 summary ($ROUND, $1): python bench.py $ARGS" $(find $D/trace -name "*.db" | head -1) > /dev/null 2>> $D/trace.err
  echo >> $D/summary.md; echo '```' >> $D/summary.md; cat $D/bench_unprofiled.json >> $D/summary.md; echo '```' >> $D/summary.md
  find $D -name "*.db" -delete
  python -c "
import json; d=json.loads(open('$D/bench_unprofiled.json').readlines()[-1]); print('$1', round(d['value']), d['roofline']['kernel'], [k['ms'] for k in d['roofline']['kernels']], d['env_errors'])"
}
stamp "headline bench + trace"
trace default cluster_s5 5

pmcs() {   # tag workload aa
  stamp "pmc $1"
  tools/pmc_sets.sh $TAG/$1 $2 8192 $3 insts active write fetch wait lds > $OUT/$1/pmc.log 2>&1
}
pmcs default cluster_s5 5

stamp "default bench line, the driver's command (device clocks sampled beside both), launch convergence"
# (round-5 review: is the gap between the 20-step and the 200-step line the device's clocks?  rocm-smi sampled every 50 ms)
clocks() { while true; do echo "$(date +%s.%N) $(rocm-smi --showclocks 2>/dev/null | grep -E 'sclk|mclk' | tr -s ' ' | tr '\n' ';')"; sleep 0.05; done; }
clocks > $OUT/clocks_default.txt & CPID=$!
python bench.py --no-cpu-baseline > $OUT/bench_default_nocpu.json 2> $OUT/bench_default.err
kill $CPID; wait $CPID 2>/dev/null
clocks > $OUT/clocks_driver_cmd.txt & CPID=$!
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_cmd_nocpu.json 2>> $OUT/bench_default.err
kill $CPID; wait $CPID 2>/dev/null
python bench.py > $OUT/bench_default.json 2>> $OUT/bench_default.err
tail -c 700 $OUT/bench_default.json; echo
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2>> $OUT/bench_default.err
python tools/launch_convergence.py cluster_s5 8192 5 64 > $OUT/convergence_headline.json 2> $OUT/convergence.err
python tools/launch_convergence.py embodied_s12 8192 5 24 > $OUT/convergence_embodied.json 2>> $OUT/convergence.err
rocm-smi --showclocks --showperflevel > $OUT/rocm_smi.txt 2>&1

stamp "traces aa1 / embodied"
trace aa1 cluster_s5 1
trace embodied_s12_128 embodied_s12 5
pmcs aa1 cluster_s5 1
pmcs embodied_s12_128 embodied_s12 5

stamp "wave timelines"
if [ -f spriteworld_amd/csrc/exp_trace.so ]; then
  SWB_LIBRARY=$PWD/spriteworld_amd/csrc/exp_trace.so python tools/exp_trace.py cluster_s5 8192 5 $OUT/timeline_8192.json > $OUT/timeline_8192.log 2>&1
  SWB_LIBRARY=$PWD/spriteworld_amd/csrc/exp_trace.so python tools/exp_trace.py cluster_s5 65536 5 $OUT/timeline_65536.json > $OUT/timeline_65536.log 2>&1
  rm -f $OUT/*_trace.npy
fi
stamp "phase profile"
if [ -f spriteworld_amd/csrc/exp_phase1.so ]; then
  timeout 200 python tools/phase_profile.py $OUT/phase_times.md cluster_s5:5 cluster_s5:1 embodied_s12:5 > $OUT/phase_profile.log 2>&1
fi
stamp "done"
