cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06o; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider -k "parity or dispatch or golden or full_size or band" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
SPECS="goal_s5:1024:5 cluster_s5:1024:5 goal_s5:512:5 goal_s5:2048:5 goal_s5:4096:5 goal_s5:256:5 embodied_s12:1024:5 geom_128x128:1024:1"
for r in 1 2 3; do
  echo "== band tasks round $r" | tee -a $O/ab.txt
  python tools/quick_bench.py $SPECS 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
  echo "== SWB_NO_BAND_TASKS round $r" | tee -a $O/ab.txt
  SWB_NO_BAND_TASKS=1 python tools/quick_bench.py $SPECS 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
