#!/bin/bash
# kernel trace of the default bench (per-kernel times) for each build
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-trace}; mkdir -p $OUT
for lib in spriteworld_amd/csrc/libswb.so $(ls spriteworld_amd/csrc/exp_*.so 2>/dev/null); do
  name=$(basename $lib .so)
  SWB_LIBRARY=$PWD/$lib rocprofv3 --kernel-trace --stats -d $OUT/t_$name -o t -- python bench.py --steps 60 --warmup 5 --no-extra --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/err_$name.log
  echo "== $name"; python - <<PY
import sqlite3,glob
db=glob.glob("$OUT/t_$name/**/*.db",recursive=True)[0]
cur=sqlite3.connect(db).cursor()
for r in list(cur.execute('select name, total_calls, average from top_kernels'))[:3]: print(r[0][:70], r[1], round(r[2]/1000,2),'us')
PY
  python -c "
import json; d=json.loads(open('$OUT/bench_$name.json').readlines()[-1]); print('bench', round(d['value']), d['roofline']['kernel_ms'])"
done
find $OUT -name "*.db" -delete
