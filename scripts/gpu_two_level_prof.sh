#!/usr/bin/env bash
# per-kernel time of the two-level path at the 2m geometries (rocprofv3 --kernel-trace --stats of scripts/staged_rate.py)
set -u
out=gpurun_out/two_level
mkdir -p $out
export TMPDIR=/tmp
for r in 8 4 2; do
  rm -rf $out/ks_$r
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_$r -- python3 scripts/staged_rate.py $r > $out/rate_$r.log 2>&1
  rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed"; exit 1; fi
  f=$(find $out/ks_$r -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/kernel_stats_$r.csv
  rm -rf $out/ks_$r
  tail -1 $out/rate_$r.log
  head -8 $out/kernel_stats_$r.csv | cut -c1-200
done
