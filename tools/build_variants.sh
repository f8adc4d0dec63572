#!/bin/bash
# Builds experimental variants exp_<name>.so (extra -D flags) of libswb.so in spriteworld_amd/csrc.
# usage: tools/build_variants.sh name1:"-DFOO=0 -DBAR=1" name2:"..."
set -e
cd "$(dirname "$0")/../spriteworld_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -mllvm -amdgpu-sched-strategy=iterative-ilp -shared -fPIC"
rm -f exp_*.so
pids=()
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  hipcc $FLAGS $defs -o exp_$name.so swb.hip &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
ls -la *.so
