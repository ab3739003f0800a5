#!/usr/bin/env bash
# Round-3 A/B 3: what bounds the whole-frame epilogue (config 2)? Streaming (nt) frame stores; timing-only builds with
# the frame stores / the plane loads removed; stage timelines at 256 and 2048 blocks.
set -u
mkdir -p gpurun_out/r03c
for a in "--config 2 --blocks 2048" "--config 2"; do
  echo "=== $a"
  bash scripts/gpu_ab_arms.sh "$a" k14c k14e || exit 1
done
for v in STc STnt STnost STnold; do
  for b in 256 2048; do
    MIUPS_LIB=$PWD/totton-rasp-gpu-dsp_amd/lib_ablate/libmi_upsampler_$v.so STAMPS_BLOCKS=$b STAMPS_BRIEF=1 timeout -k 10 200 python scripts/stamps_report.py > gpurun_out/r03c/stamps_${v}_$b.txt 2>&1 || exit 1
    echo "--- $v blocks $b: $(grep -E 'fwd_first|epilogue|total' gpurun_out/r03c/stamps_${v}_$b.txt | tr -s ' ' | tr '\n' ';')"
  done
done
