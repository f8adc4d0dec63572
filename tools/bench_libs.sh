cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/sched2
for rep in 1 2; do
for lib in spriteworld_amd/csrc/libswb.so $(ls spriteworld_amd/csrc/exp_*.so 2>/dev/null); do
  for wl in "cluster_s5 5" "cluster_s5 1" "embodied_s12 5"; do
    set -- $wl
    echo -n "$(basename $lib) $1 aa$2: " | tee -a gpurun_out/sched2/bench.txt
    SWB_LIBRARY=$PWD/$lib python bench.py --steps 100 --workload $1 --aa $2 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), d['roofline']['kernel_ms'], d['env_errors'])" | tee -a gpurun_out/sched2/bench.txt
  done
done
done
