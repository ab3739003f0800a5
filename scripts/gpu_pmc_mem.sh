#!/usr/bin/env bash
# Memory-traffic PMC passes only (FETCH_SIZE, WRITE_SIZE, L2 hit/miss), one group per
# rocprofv3 run. Usage: gpu_pmc_mem.sh <tag> [bench args...] -> gpurun_out/pmc_<tag>/summary.txt
set -u
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p "$out"
export TMPDIR=/tmp
ARGS=${*:-"--steps 5 --warmup 2 --no-cpu-baseline"}
run() {
  local name=$1; shift
  rm -rf "$out/$name"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py $ARGS > "$out/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit 1; fi
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
python3 scripts/pmc_summary.py "$out" > "$out/summary.txt" 2>&1
cat "$out/summary.txt"
