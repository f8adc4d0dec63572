#!/bin/bash
# round 5, GPU session 5: cost keys with the filing words preloaded; one-ballot cost formula
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05f; mkdir -p $OUT
C=spriteworld_amd/csrc
bash tools/r04_ab.sh r05f 3 "cluster_s5:8192:5 cluster_s5:1024:5 embodied_s12:8192:5 cluster_s5:65536:5 cluster_s5:8192:1" $C/exp_r5b.so $C/libswb.so $C/exp_cheapcost.so
SWB_LIBRARY=$PWD/$C/exp_trace.so python tools/exp_trace.py cluster_s5 8192 5 $OUT/timeline_8192.json > $OUT/timeline_8192.log 2>&1
SWB_LIBRARY=$PWD/$C/exp_trace.so python tools/exp_trace.py embodied_s12 8192 5 $OUT/timeline_emb.json > $OUT/timeline_emb.log 2>&1
