#!/usr/bin/env bash
# Round-3 visit 1: the whole GPU suite (incl. the bench-shape parity tests), the bench line with output probes,
# and one rocprofv3 --kernel-trace --stats CSV PER CONFIG (the round-2 file blended all configs).
# Output: gpurun_out/r03a/
set -u
out=gpurun_out/r03a
mkdir -p $out
export TMPDIR=/tmp
step() {
  local name=$1 secs=$2; shift 2
  timeout -k 10 "$secs" "$@" > "$out/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed: stopping"; tail -5 "$out/$name.log"; exit 1; fi
  return 0
}
step pytest 1000 python -m pytest tests -q -m gpu --durations=15
tail -25 $out/pytest.log
step bench 400 python bench.py --gpus 1 --steps 20 --warmup 5
tail -c 600 $out/bench.log; echo
for c in 2 3 4 5; do
  rm -rf $out/ks_$c
  step ks_$c 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_$c -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --config $c
  f=$(find $out/ks_$c -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" $out/kernel_stats_config$c.csv; head -6 "$f"; fi
  rm -rf $out/ks_$c
done
exit 0
