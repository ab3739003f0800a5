#!/usr/bin/env bash
# Per-kernel time of one bench config: gpu_kstats.sh <config> [extra bench args] -> gpurun_out/kstats_<config>/
set -u
c=$1; shift
out=gpurun_out/kstats_$c
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline "$@" > "$out/bench.log" 2>&1
rc=$?
echo "config $c rc=$rc"
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -d, -f1-8 "$f" | sed -E 's/\(miups::Geometry[^"]*//' | cut -c1-150
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
