#!/usr/bin/env bash
# Round-3 visit 2: GPU suite on the current library (trickled stores, pruned last pass, channel split, socket control
# plane), the full bench line. Output: gpurun_out/r03e/
set -u
out=gpurun_out/r03e
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
step pytest 1000 python -m pytest tests -q -m gpu --durations=10
tail -30 $out/pytest.log
step bench 500 python bench.py --gpus 1 --steps 20 --warmup 5
tail -c 900 $out/bench.log; echo
exit 0
