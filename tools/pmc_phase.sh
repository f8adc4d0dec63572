cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ph in 1 2 0; do
  mkdir -p gpurun_out/ph$ph
  SWB_DEBUG_PHASE=$ph rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_ANY -d gpurun_out/ph$ph -o p -- python bench.py --steps 20 --warmup 3 --no-extra --no-cpu-baseline > /dev/null 2>&1
  python - <<PY
import sqlite3,glob
db=glob.glob("gpurun_out/ph$ph/**/*.db",recursive=True)[0]
cur=sqlite3.connect(db).cursor()
print("phase $ph", {r[0]: round(r[1]/8192) for r in cur.execute("select counter_name, avg(value) from counters_collection where kernel_name like '%swb_step%' group by counter_name")})
PY
  rm -rf gpurun_out/ph$ph
done
