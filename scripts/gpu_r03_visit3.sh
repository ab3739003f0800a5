#!/usr/bin/env bash
# Round-3 visit 3: (1) new GPU tests (640k-tap filters), (2) copy-ceiling kernel occupancy sweep, (3) PMC view of the frame
# assembly pass (interleave_tiled_kernel) at config 3 against config 5, serial launches. Output: gpurun_out/r03f/
set -u
out=gpurun_out/r03f
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "640k" > $out/pytest_640k.log 2>&1; echo "pytest 640k rc=$?"; tail -3 $out/pytest_640k.log
for b in 2 4 8 16 32; do
  MIUPS_EXP_COPY_BLOCKS_PER_CU=$b timeout -k 10 120 python -c "
import totton_rasp_gpu_dsp_amd as ups
print('copy blocks/CU $b:', round(ups.device_copy_rate(0, 1<<30, 5),1), 'GB/s (1 GiB)', round(ups.device_copy_rate(0, 1<<28, 5),1), '(256 MiB)')" 2>&1 | tail -1
done | tee $out/copy_sweep.txt
rocprofv3 -L > $out/counters_available.txt 2>&1
export MIUPS_EXP_PIPELINE=0
pmc() {  # pmc <config> <group name> <counters...>
  local c=$1 g=$2; shift 2
  rm -rf $out/pmc_c$c/$g
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/pmc_c$c/$g -- python3 bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $out/pmc_c${c}_$g.log 2>&1
  local rc=$?
  echo "pmc config $c $g rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit 1; fi
}
for c in 3 5; do
  pmc $c fetch FETCH_SIZE
  pmc $c write WRITE_SIZE
  pmc $c tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
  pmc $c ea_rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
  pmc $c ea_wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
  pmc $c ea_stall TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_sum
  pmc $c sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  pmc $c sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM
  pmc $c tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum
  pmc $c grbm GRBM_GUI_ACTIVE GRBM_COUNT
  python3 scripts/pmc_summary.py $out/pmc_c$c > $out/pmc_c$c.txt 2>&1
  find $out/pmc_c$c -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $out/ktrace_c$c.csv
  rm -rf $out/pmc_c$c
done
grep -A40 "interleave_tiled" $out/pmc_c3.txt | head -60
exit 0
