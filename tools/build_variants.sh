#!/bin/bash
# Builds experimental variants exp_<name>.so (extra -D flags) of libswb.so in spriteworld_amd/csrc
# (both translation units, flags as in spriteworld_amd/build.py).
# usage: tools/build_variants.sh name1:"-DFOO=0 -DBAR=1" name2:"..."
set -e
cd "$(dirname "$0")/../spriteworld_amd/csrc"
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC"
rm -f exp_*.so
pids=()
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  ( hipcc $COMMON $defs -c -o /tmp/exp_${name}_a.o swb.hip &
    hipcc $COMMON -mllvm -amdgpu-sched-strategy=iterative-ilp $defs -c -o /tmp/exp_${name}_b.o swb_wide.hip &
    wait
    hipcc --offload-arch=gfx950 -shared -fPIC -o exp_$name.so /tmp/exp_${name}_a.o /tmp/exp_${name}_b.o ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
ls -la *.so
