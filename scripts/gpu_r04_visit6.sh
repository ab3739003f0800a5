#!/usr/bin/env bash
# Round-4 visit 6: short launches -- does "a pair's last workgroup assembles ALL its tiles at once" (a cap nobody reaches)
# overlap round 1's frames with round 2's transforms? configs[2] at 256 / 384 blocks, and K = 8192 (four rounds).
set -u
out=gpurun_out/r04f
mkdir -p $out
export TMPDIR=/tmp
run() {  # run <tag> <arm> <bench args>
  local tag=$1 arm=$2; shift 2
  unset MIUPS_EXP_NO_COOP_FRAMES MIUPS_EXP_COOP_CAP MIUPS_EXP_COOP_FRAMES
  case $arm in nocoop) export MIUPS_EXP_NO_COOP_FRAMES=1;; cap*) export MIUPS_EXP_COOP_FRAMES=1 MIUPS_EXP_COOP_CAP=${arm#cap};; esac
  for round in 1 2; do
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" > $out/${tag}_$arm.$round.log 2>&1
    rc=$?
    echo "$tag $arm round $round rc=$rc $(python3 -c "import json; d=json.loads(open('$out/${tag}_$arm.$round.log').read().strip().splitlines()[-1]); d=d['rows'][0] if 'rows' in d else d; print(d['value'], 'Msamples/s  ms/step', d['ms_per_step'], 'coop', d['config'].get('coop_frames'))" 2>&1 | tail -1)"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  done
}
for arm in nocoop cap16 cap32 cap64 cap128; do run c3_256 $arm --config 3; done
for arm in nocoop cap16 cap128; do run c3_384 $arm --config 3 --blocks 384; done
for arm in nocoop cap8 cap16 cap64; do run r8x32 $arm --row 8x80k_32ch; done
exit 0
