#!/usr/bin/env bash
# Round-4 measurement visit: the bench line as the driver runs it, one rocprofv3 --kernel-trace --stats CSV PER CONFIG (and per
# extra row), and the memory-traffic PMC passes (FETCH_SIZE / WRITE_SIZE / TCC hit-miss, one counter group per run) for every
# config and row the line reports. Output: gpurun_out/r04m/ (-> profiles/ by scripts/traffic_from_pmc.py r04_m gpurun_out/r04m).
# PART=1: bench + kernel stats; PART=2: PMC passes of the configs; PART=3: PMC passes of the extra rows + issue counters.
set -u
out=${OUT:-gpurun_out/r04m}
mkdir -p $out
export TMPDIR=/tmp
part=${PART:-1}
step() {
  local name=$1 secs=$2; shift 2
  timeout -k 10 "$secs" "$@" > "$out/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed: stopping"; exit 1; fi
  return 0
}
ks() {  # ks <tag> <bench args>: rocprofv3 --kernel-trace --stats of one bench command
  local tag=$1; shift
  rm -rf $out/ks_$tag
  step ks_$tag 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_$tag -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@"
  f=$(find $out/ks_$tag -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" $out/kernel_stats_$tag.csv; head -3 "$f" | cut -c1-200; fi
  rm -rf $out/ks_$tag
}
MEM_GROUPS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum")
ISSUE_GROUPS=("SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR")
PMC_GROUPS=("${MEM_GROUPS[@]}")
pmc() {  # pmc <tag> <bench args...>   counter groups: PMC_GROUPS, one rocprofv3 run each
  local tag=$1; shift
  for grp in "${PMC_GROUPS[@]}"; do
    local g=$(echo $grp | cut -d' ' -f1)
    rm -rf $out/pmc_$tag/$g
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc_$tag/$g -- python3 bench.py --steps 5 --warmup 2 --prime-seconds 0.02 --no-cpu-baseline "$@" > $out/pmc_$tag.$g.log 2>&1
    local rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pmc $tag $g killed: stopping"; exit 1; fi
  done
  python3 scripts/pmc_summary.py $out/pmc_$tag > $out/pmc_$tag.txt 2>&1
  rm -rf $out/pmc_$tag
  echo "pmc $tag: $(grep -c mean $out/pmc_$tag.txt) counters"
}
if [ $part = 1 ]; then
  step bench 600 python bench.py --gpus 1 --steps 20 --warmup 5
  tail -c 400 $out/bench.log; echo
  for c in 2 3 4 5; do ks config$c --no-extras --config $c; done
  ks config2_2048blocks --no-extras --config 2 --blocks 2048
  for r in 8x80k_2ch 8x80k_32ch 2m_8x 2m_2x; do ks row_$r --row $r; done
fi
if [ $part = 2 ]; then
  pmc c2_256 --no-extras --config 2
  pmc c2_2048 --no-extras --config 2 --blocks 2048
  pmc c3 --no-extras --config 3
  pmc c4 --no-extras --config 4
  pmc c5 --no-extras --config 5
fi
if [ $part = 3 ]; then
  for r in 8x80k_2ch 8x80k_32ch 2m_8x 2m_2x; do pmc $r --row $r; done
  PMC_GROUPS=("${ISSUE_GROUPS[@]}")
  pmc issue_c2_256 --no-extras --config 2
fi
if [ $part = 4 ]; then  # issue-side SQ counters of the transform kernels of configs 3-5 (what bounds them: roofline.compute)
  PMC_GROUPS=("${ISSUE_GROUPS[@]}")
  pmc issue_c3 --no-extras --config 3
  pmc issue_c4 --no-extras --config 4
  pmc issue_c5 --no-extras --config 5
fi
exit 0
