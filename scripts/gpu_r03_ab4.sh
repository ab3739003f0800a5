#!/usr/bin/env bash
# Round-3 A/B 4: config 4 (2x filters, split kernel): trickled stores + pruned last pass of every half transform
set -u
mkdir -p gpurun_out/r03d
bash scripts/gpu_ab_arms.sh "--config 4" k15b k15c || exit 1
MIUPS_LIB=$PWD/totton-rasp-gpu-dsp_amd/lib_ablate/libmi_upsampler_ST15c.so STAMPS_CONFIG=4 STAMPS_BRIEF=1 timeout -k 10 200 python scripts/stamps_report.py > gpurun_out/r03d/stamps_ST15c.txt 2>&1 || exit 1
cat gpurun_out/r03d/stamps_ST15c.txt
