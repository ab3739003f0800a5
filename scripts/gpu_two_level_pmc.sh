#!/usr/bin/env bash
# issue / LDS / L2 counters of the two-level kernels at the 8x 2m geometry (separate --pmc passes)
set -u
out=gpurun_out/two_level_pmc
mkdir -p $out
export TMPDIR=/tmp
r=${1:-8}
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR"; do
  g=$(echo $grp | cut -d' ' -f1)
  rm -rf $out/raw/$g
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/raw/$g -- python3 scripts/staged_rate.py $r > $out/$g.log 2>&1
  rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed"; exit 1; fi
done
python3 scripts/pmc_summary.py $out/raw > $out/summary_$r.txt 2>&1
rm -rf $out/raw
grep -c mean $out/summary_$r.txt
