#!/bin/bash
# round 5, GPU session 4: dispatch keys of the second kernel by run-list cost (fitted buckets) against the list length
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05e; mkdir -p $OUT
C=spriteworld_amd/csrc
bash tools/r04_ab.sh r05e 3 "cluster_s5:8192:5 cluster_s5:1024:5 embodied_s12:8192:5 cluster_s5:65536:5 goal_s5:8192:5" $C/exp_r5b.so $C/libswb.so
SWB_LIBRARY=$PWD/$C/exp_trace.so python tools/exp_trace.py cluster_s5 8192 5 $OUT/timeline_8192.json > $OUT/timeline_8192.log 2>&1
tail -1 $OUT/timeline_8192.log | cut -c1-300
