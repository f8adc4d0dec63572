cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r06i; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "team" > $OUT/gputests_team.log 2>&1; tail -3 $OUT/gputests_team.log
SPECS="goal_s5:1024:5 cluster_s5:1024:5 goal_s5:512:5 goal_s5:256:5 sorting_s4:1024:5"
for r in 1 2 3 4; do
  echo "== plain round $r"; SWB_NO_TEAM=1 python tools/quick_bench.py $SPECS 2>&1 | grep -v amdgpu.ids
  echo "== team round $r"; python tools/quick_bench.py $SPECS 2>&1 | grep -v amdgpu.ids
  echo "== team8 round $r"; SWB_TEAM=1 SWB_BANDS=8 python tools/quick_bench.py goal_s5:512:5 goal_s5:256:5 2>&1 | grep -v amdgpu.ids
done > $OUT/ab.txt 2>&1
tail -14 $OUT/ab.txt
python -m pytest tests -m gpu -q > $OUT/gputests.log 2>&1; tail -3 $OUT/gputests.log
