#!/usr/bin/env bash
# Bench every BASELINE config that runs on one GPU (short runs, no CPU baseline).
set -u
mkdir -p gpurun_out
for c in ${CONFIGS:-2 3 4 5}; do
  timeout -k 10 280 python bench.py --config $c --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline > gpurun_out/config_$c.log 2>&1
  rc=$?
  echo "config $c rc=$rc $(python3 -c "import json; d=json.loads(open('gpurun_out/config_$c.log').read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms/step', d['config']['kernel_path'], 'frac', d['roofline']['frac'])" 2>&1 | tail -1)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
