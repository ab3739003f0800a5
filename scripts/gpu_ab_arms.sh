#!/usr/bin/env bash
# Same-box A/B of (library variant, environment) arms: gpu_ab_arms.sh "<bench args>" "<variant>[:ENV=1[,ENV2=1]]"...
# variant "base" = the product library. Two interleaved rounds. bench.py's output probes (fp64 truth) run in every arm.
set -u
mkdir -p gpurun_out/ab
args=$1; shift
for round in 1 2; do
  for arm in "$@"; do
    v=${arm%%:*}; envs=""
    if [ "$arm" != "$v" ]; then envs=$(echo "${arm#*:}" | tr ',' ' '); fi
    if [ "$v" = base ]; then unset MIUPS_LIB; else export MIUPS_LIB=$PWD/totton-rasp-gpu-dsp_amd/lib_ablate/libmi_upsampler_$v.so; fi
    tag=$(echo "$arm" | tr ':,=' '___')
    env $envs timeout -k 10 200 python bench.py $args --no-cpu-baseline --no-extras > gpurun_out/ab/$tag.$round.log 2>&1
    rc=$?
    echo "$arm round $round rc=$rc $(python3 -c "import json,sys; d=json.loads(open('gpurun_out/ab/$tag.$round.log').read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s  ms/step', d['ms_per_step'], ' kernel_ms', d['roofline']['kernel_ms_avg'], ' frac', d['roofline']['frac'], ' check', d['output_check']['worst_err_over_tol'])" 2>&1 | tail -1)"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  done
done
