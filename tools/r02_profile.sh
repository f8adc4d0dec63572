#!/bin/bash
# GPU box: rocprofv3 kernel trace + separate PMC passes of `python bench.py` for the three judged workloads;
# writes gpurun_out/r02prof/{summary_<tag>.md, counters.json}.  Copy them to profiles/ afterwards.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02prof
mkdir -p $OUT
echo '{"records": [' > $OUT/counters.json
first=1
for spec in "default:cluster_s5:5" "aa1:cluster_s5:1" "embodied_s12_128:embodied_s12:5"; do
  IFS=: read tag wl aa <<< "$spec"
  D=$OUT/$tag; mkdir -p $D
  ARGS="--steps 40 --warmup 5 --no-extra --no-cpu-baseline --workload $wl --aa $aa"
  rocprofv3 --kernel-trace --stats -d $D/trace -o t -- python bench.py $ARGS > $D/bench_trace.json 2> $D/trace.err
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d $D/pmc1 -o p -- python bench.py $ARGS > /dev/null 2> $D/pmc1.err
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM -d $D/pmc2 -o p -- python bench.py $ARGS > /dev/null 2> $D/pmc2.err
  rocprofv3 --pmc FETCH_SIZE -d $D/pmc3 -o p -- python bench.py $ARGS > /dev/null 2> $D/pmc3.err
  rocprofv3 --pmc WRITE_SIZE -d $D/pmc4 -o p -- python bench.py $ARGS > /dev/null 2> $D/pmc4.err
  python bench.py $ARGS > $D/bench_unprofiled.json 2>/dev/null
  python tools/rocprof_summary.py $OUT/summary_$tag.md "rocprofv3 summary (round 2, $tag): python bench.py $ARGS" $(find $D/trace -name "*.db" | head -1) $(find $D/pmc1 $D/pmc2 $D/pmc3 $D/pmc4 -name "*.db")
  echo >> $OUT/summary_$tag.md; echo '```' >> $OUT/summary_$tag.md; cat $D/bench_unprofiled.json >> $OUT/summary_$tag.md; echo '```' >> $OUT/summary_$tag.md
  [ $first = 1 ] || echo ',' >> $OUT/counters.json
  first=0
  python - >> $OUT/counters.json <<PY
import sqlite3, glob, json
def avg(db, name):
  cur = sqlite3.connect(db).cursor()
  r = list(cur.execute("select avg(value) from counters_collection where kernel_name like '%swb_step%' and counter_name=?", (name,)))
  return r[0][0]
d = "$D"
g = lambda sub: glob.glob(d + "/" + sub + "/**/*.db", recursive=True)[0]
b = json.loads(open(d + "/bench_unprofiled.json").readlines()[-1])
n = b["config"]["envs_per_gpu"]
fetch, write = avg(g("pmc3"), "FETCH_SIZE"), avg(g("pmc4"), "WRITE_SIZE")
rec = {
  "build_id": b["roofline"]["build_id"], "workload": "$wl", "envs": n, "anti_aliasing": $aa, "kernel": b["roofline"]["kernel"],
  "insts_valu_per_wave": avg(g("pmc1"), "SQ_INSTS_VALU") / n, "insts_salu_per_wave": avg(g("pmc1"), "SQ_INSTS_SALU") / n,
  "insts_lds_per_wave": avg(g("pmc1"), "SQ_INSTS_LDS") / n, "wave_cycles_per_wave": avg(g("pmc1"), "SQ_WAVE_CYCLES") / n,
  "active_inst_valu_per_wave": avg(g("pmc2"), "SQ_ACTIVE_INST_VALU") / n, "active_inst_sca_per_wave": avg(g("pmc2"), "SQ_ACTIVE_INST_SCA") / n,
  "wait_any_per_wave": avg(g("pmc2"), "SQ_WAIT_ANY") / n, "wait_inst_any_per_wave": avg(g("pmc2"), "SQ_WAIT_INST_ANY") / n,
  "resident_waves_per_simd": b["roofline"]["waves_per_simd"],
  "fetch_size_kb": fetch, "write_size_kb": write, "fetch_correction": 2.0,
  "hbm_traffic_bytes_per_launch": int((2.0 * fetch + write) * 1024),
  "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_env_step"] * n,
  "kernel_ms_unprofiled": b["roofline"]["kernel_ms"],
  "source": "profiles/r02_rocprofv3_summary_$tag.md (rocprofv3 --pmc, separate passes; quad-cycle counters per wave; FETCH_SIZE doubled per MI355X_MICROARCH.md)",
}
print(json.dumps(rec, indent=1))
PY
  find $D -name "*.db" -delete
done
echo ']}' >> $OUT/counters.json
cat $OUT/counters.json | head -80
