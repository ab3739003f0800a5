#!/usr/bin/env bash
# Round-3 measurement visit: the bench line as the driver runs it, one rocprofv3 --kernel-trace --stats CSV PER CONFIG of
# the same workloads, and the memory-traffic PMC passes (FETCH_SIZE / WRITE_SIZE / TCC hit-miss, one counter group per run)
# for every config the line reports. Output: gpurun_out/r03m/ (copied into profiles/ by scripts/traffic_from_pmc.py and by hand).
set -u
out=${OUT:-gpurun_out/r03m}
mkdir -p $out
export TMPDIR=/tmp
step() {
  local name=$1 secs=$2; shift 2
  timeout -k 10 "$secs" "$@" > "$out/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed: stopping"; exit 1; fi
  return 0
}
step bench 600 python bench.py --gpus 1 --steps 20 --warmup 5
tail -c 600 $out/bench.log; echo
for c in 2 3 4 5; do
  rm -rf $out/ks_$c
  step ks_$c 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_$c -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --config $c
  f=$(find $out/ks_$c -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" $out/kernel_stats_config$c.csv; head -4 "$f"; fi
  rm -rf $out/ks_$c
done
rm -rf $out/ks_2l
step ks_2l 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_2l -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --config 2 --blocks 2048
f=$(find $out/ks_2l -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/kernel_stats_config2_2048blocks.csv; rm -rf $out/ks_2l
MEM_GROUPS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum")
# issue side of the dominant kernel (what bounds it when HBM does not): instruction counts, busy/active cycles, LDS conflicts
ISSUE_GROUPS=("SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR")
PMC_GROUPS=("${MEM_GROUPS[@]}")
pmc() {  # pmc <tag> <bench args...>   counter groups: PMC_GROUPS, one rocprofv3 run each
  local tag=$1; shift
  for grp in "${PMC_GROUPS[@]}"; do
    local g=$(echo $grp | cut -d' ' -f1)
    rm -rf $out/pmc_$tag/$g
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc_$tag/$g -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras "$@" > $out/pmc_$tag.$g.log 2>&1
    local rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pmc $tag $g killed: stopping"; exit 1; fi
  done
  python3 scripts/pmc_summary.py $out/pmc_$tag > $out/pmc_$tag.txt 2>&1
  rm -rf $out/pmc_$tag
  echo "pmc $tag: $(grep -c mean $out/pmc_$tag.txt) counters"
}
pmc c2_256 --config 2
pmc c2_2048 --config 2 --blocks 2048
pmc c3 --config 3
pmc c4 --config 4
pmc c5 --config 5
PMC_GROUPS=("${ISSUE_GROUPS[@]}")
pmc issue_c2_256 --config 2
exit 0
