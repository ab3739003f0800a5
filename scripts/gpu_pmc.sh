#!/usr/bin/env bash
# PMC passes over the bench (one counter group per rocprofv3 run, as the
# MI355X guide prescribes: FETCH_SIZE and WRITE_SIZE cannot share a pass).
# Usage: gpu_pmc.sh [bench args...]   -> gpurun_out/pmc/<group>/..._counter_collection.csv
set -u
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
ARGS=${*:-"--steps 5 --warmup 2 --no-cpu-baseline"}
run() {  # run <name> <counters...>
  local name=$1; shift
  rm -rf "gpurun_out/pmc/$name"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "gpurun_out/pmc/$name" -- python3 bench.py $ARGS > "gpurun_out/pmc/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit 1; fi
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU
run sq3 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 scripts/pmc_summary.py gpurun_out/pmc > gpurun_out/pmc/summary.txt 2>&1
cat gpurun_out/pmc/summary.txt
