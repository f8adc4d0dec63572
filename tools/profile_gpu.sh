#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + separate PMC passes of the bench command.
# usage: tools/profile_gpu.sh TAG [bench args...]
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
ARGS="--steps 40 --warmup 5 --no-extra --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d $OUT/pmc1 -o p -- python bench.py $ARGS > /dev/null 2> $OUT/pmc1.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM -d $OUT/pmc2 -o p -- python bench.py $ARGS > /dev/null 2> $OUT/pmc2.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o p -- python bench.py $ARGS > /dev/null 2> $OUT/pmc3.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o p -- python bench.py $ARGS > /dev/null 2> $OUT/pmc4.err
python bench.py $ARGS > $OUT/bench_unprofiled.json 2>/dev/null
python tools/rocprof_summary.py $OUT/summary.md "rocprofv3 summary ($TAG): python bench.py $ARGS" $(find $OUT/trace -name "*.db" | head -1) $(find $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 -name "*.db")
cat $OUT/bench_unprofiled.json >> $OUT/summary.md
find $OUT -name "*.db" -delete
cat $OUT/summary.md | head -60
