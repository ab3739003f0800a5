#!/usr/bin/env bash
# L2 hit/miss + memory-side traffic of one bench config: gpu_pmc_tcc.sh <config> -> gpurun_out/pmc_tcc_<config>/summary.txt
set -u
c=$1; shift
out=gpurun_out/pmc_tcc_$c
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
run() {
  local name=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline > "$out/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit 1; fi
}
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 scripts/pmc_summary.py "$out" > "$out/summary.txt" 2>&1
grep -A6 "fused" "$out/summary.txt"
