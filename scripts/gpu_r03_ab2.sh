#!/usr/bin/env bash
# Round-3 A/B 2: pruned last inverse pass and trickled LDS stores (K = 16384: configs 2, 5; K = 4096: config 3), then the
# stage timelines (MIUPS_STAMPS builds) of: round-2 code (STb), current (STc), radix-32 plan (STr32 + MIUPS_EXP_R32).
set -u
for a in "--config 2 --blocks 2048" "--config 2" "--config 5"; do
  echo "=== $a"
  bash scripts/gpu_ab_arms.sh "$a" k14d k14c || exit 1
done
echo "=== --config 3"
bash scripts/gpu_ab_arms.sh "--config 3" k12b k12d k12c || exit 1
mkdir -p gpurun_out/r03b
for v in STb STc; do
  MIUPS_LIB=$PWD/totton-rasp-gpu-dsp_amd/lib_ablate/libmi_upsampler_$v.so STAMPS_BRIEF=1 timeout -k 10 200 python scripts/stamps_report.py > gpurun_out/r03b/stamps_$v.txt 2>&1 || exit 1
  echo "--- $v"; grep -E "%\)|epilogue|total" gpurun_out/r03b/stamps_$v.txt
done
MIUPS_EXP_R32=1 MIUPS_LIB=$PWD/totton-rasp-gpu-dsp_amd/lib_ablate/libmi_upsampler_STr32.so STAMPS_BRIEF=1 timeout -k 10 200 python scripts/stamps_report.py > gpurun_out/r03b/stamps_STr32.txt 2>&1 || exit 1
echo "--- STr32 (radix-32 plan: the 'mid16' rows are its radix-32 pass, there is no mid256)"; grep -E "%\)|epilogue|total" gpurun_out/r03b/stamps_STr32.txt
