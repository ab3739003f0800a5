#!/usr/bin/env bash
# Round-4 visit 7: the whole GPU suite and smoke on the final code, then measurement part 1 (bench line + kernel stats).
set -u
out=gpurun_out/r04g
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -n 4 $out/tests.log | cut -c1-300
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $out/smoke.log
PART=1 OUT=gpurun_out/r04m bash scripts/gpu_r04_measure.sh
exit 0
