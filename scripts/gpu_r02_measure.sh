#!/usr/bin/env bash
# Round-2 measurement visit: the bench line as the driver runs it, its rocprofv3 kernel stats, and the memory-traffic
# PMC passes (FETCH_SIZE / WRITE_SIZE / TCC hit-miss, one counter group per run) for every config the line reports.
# Output: gpurun_out/r02/ (copied into profiles/ by hand).
set -u
out=gpurun_out/r02
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
tail -c 1500 $out/bench.log; echo
rm -rf $out/kstats; mkdir -p $out/kstats
step bench_kstats 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kstats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
find $out/kstats -name "*kernel_stats.csv" | head -2
pmc() {  # pmc <tag> <bench args...>
  local tag=$1; shift
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    local g=$(echo $grp | cut -d' ' -f1)
    rm -rf $out/pmc_$tag/$g
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc_$tag/$g -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras "$@" > $out/pmc_$tag.$g.log 2>&1
    local rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pmc $tag $g killed: stopping"; exit 1; fi
  done
  python3 scripts/pmc_summary.py $out/pmc_$tag > $out/pmc_$tag.txt 2>&1
  echo "pmc $tag: $(grep -c mean $out/pmc_$tag.txt) counters"
}
pmc c2_256 --config 2
pmc c2_2048 --config 2 --blocks 2048
pmc c3 --config 3
pmc c4 --config 4
pmc c5 --config 5
exit 0
