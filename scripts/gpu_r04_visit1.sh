#!/usr/bin/env bash
# Round-4 visit 1: the GPU suite (new: EQ fold against the true-stream golden, host-copy rule tests), smoke, the bench line
# with the new rows (8x 80k-tap at K = 8192, roofline.compute). Output: gpurun_out/r04a/.
set -u
out=gpurun_out/r04a
mkdir -p $out
export TMPDIR=/tmp
step() {
  local name=$1 secs=$2; shift 2
  timeout -k 10 "$secs" "$@" > "$out/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc"; tail -n 3 "$out/$name.log" | cut -c1-400
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed: stopping"; exit 1; fi
  return 0
}
step tests 900 python -m pytest tests -x -q -m gpu
step smoke 200 python -c "import __graft_entry__ as g; g.smoke()"
step bench 600 python bench.py --gpus 1 --steps 20 --warmup 5
exit 0
