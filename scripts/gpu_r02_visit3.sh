#!/usr/bin/env bash
# Round-2 visit 3: parity with packed-math butterflies; bench 2-5; epilogue-vs-occupied-CUs stamps experiment.
set -u
mkdir -p gpurun_out/v3
export TMPDIR=/tmp
step() {
  local name=$1 secs=$2; shift 2
  timeout -k 10 "$secs" "$@" > "gpurun_out/v3/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc $(tail -n 1 gpurun_out/v3/$name.log | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(d['value'],'Msamples/s ms/step',d['ms_per_step'],'frac',d['roofline']['frac'])
except Exception as e: print('')
")"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed: stopping"; exit 1; fi
  return 0
}
step pytest_gpu 900 python -m pytest tests -x -q -m gpu
tail -3 gpurun_out/v3/pytest_gpu.log
for c in 2 3 4 5; do
  step c${c} 200 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline
done
step c2_256 200 python bench.py --config 2 --blocks 256 --steps 20 --warmup 3 --no-cpu-baseline
for b in 64 128 256; do
  STAMPS_BLOCKS=$b STAMPS_BRIEF=1 timeout -k 10 200 python scripts/stamps_report.py > gpurun_out/v3/stamps_c2_$b.txt 2>&1
  echo "stamps blocks=$b: $(grep -E 'epilogue|total|phase_in|fwd_first' gpurun_out/v3/stamps_c2_$b.txt | tr -s ' ' | tr '\n' ';')"
done
exit 0
