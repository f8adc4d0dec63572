#!/bin/bash
# round 5, GPU session 9: resample workgroups of 2 / 8 waves; row-lane edge loop unswitched (12 sprites)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
C=spriteworld_amd/csrc
bash tools/r04_ab.sh r05j 3 "cluster_s5:8192:5 cluster_s5:1024:5 embodied_s12:8192:5 cluster_s5:65536:5" $C/libswb.so $C/exp_rsblock8.so $C/exp_rsblock2.so $C/exp_rowlane.so
