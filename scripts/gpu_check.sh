#!/usr/bin/env bash
# One GPU-box visit: parity tests, smoke, bench, kernel-trace profile.
# Every step runs under its own timeout; a step that is killed (124/137) ends
# the visit (no further GPU work after a hang). Logs go to gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
step() {  # step <name> <seconds> <cmd...>
  local name=$1 secs=$2; shift 2
  echo "=== $name"
  timeout -k 10 "$secs" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc"; tail -n 5 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed: stopping"; exit 1; fi
  return 0
}
step pytest_gpu 900 python -m pytest tests -x -q -m gpu
step smoke 300 python -c "import __graft_entry__ as g; g.smoke()"
step bench 420 python bench.py --steps ${BENCH_STEPS:-20} --warmup 3
if [ "${PROFILE:-1}" = "1" ]; then
  rm -rf gpurun_out/prof
  step rocprof 420 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline
  find gpurun_out/prof -name "*stats*" | head
fi
exit 0
