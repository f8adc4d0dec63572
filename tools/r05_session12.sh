#!/bin/bash
# round 5, GPU session 12: cover launch dealt over the SIMDs in alternating directions (snake) within the first round
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05m; mkdir -p $OUT
C=spriteworld_amd/csrc
bash tools/r04_ab.sh r05m 3 "cluster_s5:8192:5 cluster_s5:8192:1 goal_s5:8192:5 embodied_s12:8192:5 cluster_s5:6144:5" $C/libswb.so $C/exp_snake5.so $C/exp_snake8.so
SWB_LIBRARY=$PWD/$C/exp_tracesnake.so python tools/exp_trace.py cluster_s5 8192 5 $OUT/timeline_snake.json > $OUT/timeline_snake.log 2>&1
grep -o '"simd_last_end_us": {[^}]*}' $OUT/timeline_snake.json | head -1
SWB_LIBRARY=$PWD/$C/exp_snake5.so timeout 300 python -m pytest tests/test_gpu_full_size.py -q -m gpu -p no:cacheprovider 2>&1 | tail -2
