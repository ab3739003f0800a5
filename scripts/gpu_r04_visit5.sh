#!/usr/bin/env bash
# Round-4 visit 5: cooperative frames at K = 4096 -- cap x launch length, same box, interleaved with the frame-pass-only arm.
set -u
out=gpurun_out/r04e
mkdir -p $out
export TMPDIR=/tmp
for blocks in 256 512 1024 2048; do
  for round in 1 2; do
    for arm in nocoop cap8 cap12 cap16; do
      unset MIUPS_EXP_NO_COOP_FRAMES MIUPS_EXP_COOP_CAP
      case $arm in nocoop) export MIUPS_EXP_NO_COOP_FRAMES=1;; cap*) export MIUPS_EXP_COOP_CAP=${arm#cap};; esac
      timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --config 3 --blocks $blocks > $out/c3_${blocks}_$arm.$round.log 2>&1
      rc=$?
      echo "c3 $blocks $arm round $round rc=$rc $(python3 -c "import json; d=json.loads(open('$out/c3_${blocks}_$arm.$round.log').read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s  ms/step', d['ms_per_step'])" 2>&1 | tail -1)"
      if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
    done
  done
done
exit 0
