#!/usr/bin/env bash
# Config 4 (split form): stage timeline of a MIUPS_STAMPS variant, then the per-kernel times of the product.
# usage: gpu_c4_probe.sh [stamps variant]
set -u
v=${1:-STAMPS15}
mkdir -p gpurun_out
MIUPS_LIB=$PWD/totton-rasp-gpu-dsp_amd/lib_ablate/libmi_upsampler_$v.so STAMPS_CONFIG=4 STAMPS_BRIEF=1 \
  timeout -k 10 200 python scripts/stamps_report.py > gpurun_out/stamps_c4_$v.txt 2>&1 || exit 1
cat gpurun_out/stamps_c4_$v.txt
export TMPDIR=/tmp
rm -rf gpurun_out/c4k
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4k -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --config 4 > gpurun_out/c4k.log 2>&1 || exit 1
python3 scripts/kstats_print.py gpurun_out/c4k
