#!/usr/bin/env bash
# Round-2 visit 1: new parity tests, VALU/LDS micro-benchmark, K=16384-as-two-8192 experiment, baselines of configs 2-5.
set -u
mkdir -p gpurun_out/v1
export TMPDIR=/tmp
step() {  # step <name> <seconds> <cmd...>
  local name=$1 secs=$2; shift 2
  echo "=== $name"
  timeout -k 10 "$secs" "$@" > "gpurun_out/v1/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc"; tail -n 4 "gpurun_out/v1/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed: stopping"; exit 1; fi
  return 0
}
step pytest_gpu 900 python -m pytest tests -x -q -m gpu
step ubench 120 scripts/ubench/valu_rates
step c2_2048 200 python bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline
step c2_256 200 python bench.py --config 2 --blocks 256 --steps 20 --warmup 3 --no-cpu-baseline
MIUPS_EXP_FORCE_SPLIT=1 step c2_2048_split13 200 python bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline
MIUPS_EXP_FORCE_SPLIT=1 step c2_256_split13 200 python bench.py --config 2 --blocks 256 --steps 20 --warmup 3 --no-cpu-baseline
MIUPS_EXP_FORCE_SPLIT=1 step c5_split13 200 python bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline
step c3 200 python bench.py --config 3 --steps 10 --warmup 2 --no-cpu-baseline
step c4 200 python bench.py --config 4 --steps 10 --warmup 2 --no-cpu-baseline
step c5 200 python bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline
exit 0
