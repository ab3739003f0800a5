#!/usr/bin/env bash
# Stage timeline (MIUPS_STAMPS builds) of several experiment variants, one process each.
# usage: gpu_variants.sh <variant>...   (library lib_ablate/libmi_upsampler_<variant>.so)
set -u
mkdir -p gpurun_out
for v in "$@"; do
  export MIUPS_LIB=$PWD/totton-rasp-gpu-dsp_amd/lib_ablate/libmi_upsampler_$v.so
  STAMPS_BRIEF=1 timeout -k 10 200 python scripts/stamps_report.py > "gpurun_out/stamps_$v.txt" 2>&1
  rc=$?
  echo "=== $v rc=$rc"
  grep -E "%\)|epilogue|total" "gpurun_out/stamps_$v.txt"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
