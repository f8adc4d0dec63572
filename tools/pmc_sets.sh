#!/bin/bash
# PMC passes over one workload (each counter set in its own rocprofv3 process).  usage: tools/pmc_sets.sh TAG WORKLOAD N AA [sets...]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; W=$2; N=$3; AA=$4; shift 4
OUT=gpurun_out/$TAG; mkdir -p $OUT
pmc() {
  local name=$1; shift
  rocprofv3 --pmc "$@" -d $OUT/pmc_$name -o p -- python tools/pmc_kernels.py run $W $N $AA > $OUT/pmc_$name.log 2>&1
  python tools/pmc_kernels.py report $(find $OUT/pmc_$name -name "*.db") > $OUT/pmc_$name.json 2>> $OUT/pmc_$name.log
  find $OUT/pmc_$name -name "*.db" -delete
  cat $OUT/pmc_$name.json | tr -d '\n' | head -c 1500; echo
}
for s in "$@"; do
  case $s in
    insts) pmc insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR ;;
    active) pmc active SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAVES ;;
    sqc) pmc sqc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_REQ SQC_ICACHE_REQ SQC_ICACHE_MISSES ;;
    write) pmc write WRITE_SIZE ;;
    fetch) pmc fetch FETCH_SIZE ;;
    lds) pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES ;;
    misc) pmc misc SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAVE_CYCLES SQ_WAVES ;;
    wait) pmc wait SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_WR SQ_IFETCH SQ_WAVE_CYCLES SQ_WAVES ;;
  esac
done
