#!/bin/bash
# round 5, GPU session 2: A/B against the round-4 final build; the filing atomics' share of the cover wave; raw wave timeline with run / span counts
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05c; mkdir -p $OUT
C=spriteworld_amd/csrc
bash tools/r04_ab.sh r05c 3 "cluster_s5:8192:5 cluster_s5:8192:1 cluster_s5:1024:5 embodied_s12:8192:5" $C/exp_r4final.so $C/exp_r5a.so $C/libswb.so
mv $OUT/ab.txt $OUT/ab_builds.txt
export SWB_NO_COVER_ORDER=1
bash tools/r04_ab.sh r05c 3 "cluster_s5:8192:5 cluster_s5:8192:1 cluster_s5:1024:5" $C/libswb.so $C/exp_nofile.so
mv $OUT/ab.txt $OUT/ab_nofile.txt
unset SWB_NO_COVER_ORDER
SWB_LIBRARY=$PWD/$C/exp_trace.so python tools/exp_trace.py cluster_s5 8192 5 $OUT/timeline_8192.json > $OUT/timeline_8192.log 2>&1
tail -1 $OUT/timeline_8192.log | cut -c1-300
