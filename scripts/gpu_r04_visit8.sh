#!/usr/bin/env bash
# Round-4 visit 8: the whole GPU suite + smoke on the code with the widened pair rule, then PMC part 2 (configs).
set -u
out=gpurun_out/r04h
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -n 4 $out/tests.log | cut -c1-300
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $out/smoke.log
PART=2 OUT=gpurun_out/r04m bash scripts/gpu_r04_measure.sh
exit 0
