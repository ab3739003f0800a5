#!/usr/bin/env bash
# Round-4 visit 3: cooperative frames with batched (high memory-level-parallelism) tiles: parity, then same-box A/B.
set -u
out=gpurun_out/r04c
mkdir -p $out
export TMPDIR=/tmp
step() {
  local name=$1 secs=$2; shift 2
  timeout -k 10 "$secs" "$@" > "$out/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc"; tail -n 3 "$out/$name.log" | cut -c1-300
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed: stopping"; exit 1; fi
  return $rc
}
step coop_tests 400 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_stream_host.py -x -q -m gpu -k "cooperative or page_lock or cli_multi" || exit 1
ab() {  # ab <tag> <bench args>
  local tag=$1; shift
  for round in 1 2; do
    for arm in coop nocoop; do
      if [ $arm = nocoop ]; then export MIUPS_EXP_NO_COOP_FRAMES=1; else unset MIUPS_EXP_NO_COOP_FRAMES; fi
      timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" > $out/ab_${tag}_$arm.$round.log 2>&1
      rc=$?
      echo "$tag $arm round $round rc=$rc $(python3 -c "import json; d=json.loads(open('$out/ab_${tag}_$arm.$round.log').read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s  ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'coop', d['config'].get('coop_frames'))" 2>&1 | tail -1)"
      if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
    done
  done
  unset MIUPS_EXP_NO_COOP_FRAMES
}
ab c3 --config 3
ab c5 --config 5
ab c3_1024 --config 3 --blocks 1024
ab c5_256 --config 5 --blocks 256
step pitched 120 python scripts/pitched_abort_repro.py
exit 0
