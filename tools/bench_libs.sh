#!/bin/bash
# A/B of engine builds on one box, interleaved (box-to-box and run-to-run differences are of the order of the effects looked for):
#   tools/bench_libs.sh ROUNDS "SPECS" lib1.so lib2.so ...      SPECS as for tools/quick_bench.py
ROUNDS=$1; SPECS=$2; shift 2
for r in $(seq 1 $ROUNDS); do
  for lib in "$@"; do
    echo "== $(basename $lib) round $r"
    SWB_LIBRARY=$PWD/$lib python tools/quick_bench.py $SPECS 2>&1 | grep -v amdgpu.ids
  done
done
