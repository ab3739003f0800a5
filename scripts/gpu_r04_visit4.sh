#!/usr/bin/env bash
# Round-4 visit 4: where do cooperative frames pay? Same-box A/B over launch length (rounds of workgroups) at K = 4096, and
# at K = 8192 (two workgroups per CU).
set -u
out=gpurun_out/r04d
mkdir -p $out
export TMPDIR=/tmp
ab() {  # ab <tag> <bench args>
  local tag=$1; shift
  for round in 1 2; do
    for arm in coop nocoop; do
      if [ $arm = nocoop ]; then export MIUPS_EXP_NO_COOP_FRAMES=1; else unset MIUPS_EXP_NO_COOP_FRAMES; fi
      timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" > $out/ab_${tag}_$arm.$round.log 2>&1
      rc=$?
      echo "$tag $arm round $round rc=$rc $(python3 -c "import json; d=json.loads(open('$out/ab_${tag}_$arm.$round.log').read().strip().splitlines()[-1]); d=d['rows'][0] if 'rows' in d else d; print(d['value'], 'Msamples/s  ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'coop', d['config'].get('coop_frames'))" 2>&1 | tail -1)"
      if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
    done
  done
  unset MIUPS_EXP_NO_COOP_FRAMES
}
ab c3_512 --config 3 --blocks 512
ab c3_2048 --config 3 --blocks 2048
ab r8x32 --row 8x80k_32ch
for cap in 8 16 52; do
  MIUPS_EXP_COOP_CAP=$cap timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --config 3 --blocks 1024 > $out/cap_$cap.log 2>&1
  echo "c3_1024 cap $cap: $(python3 -c "import json; d=json.loads(open('$out/cap_$cap.log').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
exit 0
