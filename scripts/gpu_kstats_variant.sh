#!/usr/bin/env bash
# Per-kernel times of one bench config with a variant library: gpu_kstats_variant.sh <variant> <config> [bench args]
set -u
v=$1; c=$2; shift 2
export MIUPS_LIB=$PWD/totton-rasp-gpu-dsp_amd/lib_ablate/libmi_upsampler_$v.so
bash scripts/gpu_kstats_cfg.sh $c "$@"
