#!/bin/bash
# round 5, GPU session 3: split launch (cover(B) beside resample(A)) at several first-half sizes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r05d; mkdir -p $OUT
for r in 1 2; do
for s in 0 4096 5120 3072 6144 2048; do
  echo "== SWB_SPLIT=$s round $r"
  SWB_SPLIT=$s python tools/quick_bench.py cluster_s5:8192:5 embodied_s12:8192:5 2>&1 | grep -v amdgpu.ids
done
done > $OUT/split.txt 2>&1
for s in 0 32768 16384; do
  echo "== SWB_SPLIT=$s 65536 envs"
  SWB_SPLIT=$s python tools/quick_bench.py cluster_s5:65536:5 2>&1 | grep -v amdgpu.ids
done >> $OUT/split.txt 2>&1
for s in 0 512 1024; do
  echo "== SWB_SPLIT=$s 2048 / 1024 envs"
  SWB_SPLIT=$s python tools/quick_bench.py cluster_s5:2048:5 cluster_s5:1024:5 2>&1 | grep -v amdgpu.ids
done >> $OUT/split.txt 2>&1
cat $OUT/split.txt
