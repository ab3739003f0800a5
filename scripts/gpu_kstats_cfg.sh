#!/usr/bin/env bash
# Per-kernel times of one bench config: gpu_kstats_cfg.sh <config> [extra bench args]
set -u
c=$1; shift
export TMPDIR=/tmp
rm -rf gpurun_out/ks_$c
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$c -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --config $c "$@" > gpurun_out/ks_$c.log 2>&1 || { tail -5 gpurun_out/ks_$c.log; exit 1; }
echo "== config $c $*"
python3 scripts/kstats_print.py gpurun_out/ks_$c
