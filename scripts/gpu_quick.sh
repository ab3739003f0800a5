#!/usr/bin/env bash
# Parity + the four bench configs (headline only), one line each. Usage: gpu_quick.sh [tag]
set -u
tag=${1:-quick}
mkdir -p gpurun_out/$tag
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/$tag/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 gpurun_out/$tag/pytest.log)"
run() {
  local name=$1; shift
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras "$@" > gpurun_out/$tag/$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc $(tail -n 1 gpurun_out/$tag/$name.log | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(d['value'],'Msamples/s ms/step',d['ms_per_step'],'frac',d['roofline']['frac'])
except Exception as e: print('')
")"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
}
for round in 1 2; do
run c2_256 --config 2
run c2_2048 --config 2 --blocks 2048
run c3 --config 3
run c4 --config 4
run c5 --config 5
done
