#!/usr/bin/env bash
# A/B of library variants on one box: gpu_ab.sh "<bench args>" <variant|base>...   (variant = lib_ablate/libmi_upsampler_<v>.so)
# Two interleaved rounds so that box warm-up does not favour one arm.
set -u
mkdir -p gpurun_out/ab
args=$1; shift
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset MIUPS_LIB; else export MIUPS_LIB=$PWD/totton-rasp-gpu-dsp_amd/lib_ablate/libmi_upsampler_$v.so; fi
    timeout -k 10 200 python bench.py $args --no-cpu-baseline > gpurun_out/ab/$v.$round.log 2>&1
    rc=$?
    echo "$v round $round rc=$rc $(python3 -c "import json,sys; d=json.loads(open('gpurun_out/ab/$v.$round.log').read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s  ms/step', d['ms_per_step'], ' kernel_ms', d['roofline']['kernel_ms_avg'], ' frac', d['roofline']['frac'])" 2>&1 | tail -1)"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  done
done
