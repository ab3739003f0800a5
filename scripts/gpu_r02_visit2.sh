#!/usr/bin/env bash
# Round-2 visit 2: parity after the engine rework; pipelined (two-stream) launches on/off per config.
set -u
mkdir -p gpurun_out/v2
export TMPDIR=/tmp
step() {
  local name=$1 secs=$2; shift 2
  timeout -k 10 "$secs" "$@" > "gpurun_out/v2/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc $(tail -n 1 gpurun_out/v2/$name.log | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(d['value'],'Msamples/s ms/step',d['ms_per_step'],'frac',d['roofline']['frac'])
except Exception as e: print('')
")"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed: stopping"; exit 1; fi
  return 0
}
step pytest_gpu 900 python -m pytest tests -x -q -m gpu
tail -3 gpurun_out/v2/pytest_gpu.log
for c in 3 4 5; do
  MIUPS_EXP_PIPELINE=0 step c${c}_serial 200 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline
  MIUPS_EXP_PIPELINE=1 step c${c}_pipe 200 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline
  MIUPS_EXP_PIPELINE=1 MIUPS_EXP_CHUNK_ROUNDS=2 step c${c}_pipe_r2 200 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline
  MIUPS_EXP_PIPELINE=0 step c${c}_serial_b 200 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline
done
MIUPS_EXP_STEREO_EXT=1 MIUPS_EXP_PIPELINE=1 step c2_ext_pipe 200 python bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline
MIUPS_EXP_STEREO_EXT=1 MIUPS_EXP_PIPELINE=0 step c2_ext_serial 200 python bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline
step c2 200 python bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline
exit 0
