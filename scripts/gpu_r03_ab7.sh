#!/usr/bin/env bash
# Round-3 A/B 7: frame pass CO-RESIDENT with the transform kernel at K = 16384: transform capped at 232 registers (two waves
# per SIMD leave 48 registers per lane), frame pass at <= 48 registers and 17 KB of LDS, pipelined two-stream launches.
set -u
echo "=== --config 5"
bash scripts/gpu_ab_arms.sh "--config 5" base base:MIUPS_EXP_PIPELINE=1 k14cap k14cap:MIUPS_EXP_PIPELINE=1,MIUPS_EXP_TILE_TI=16 k14cap:MIUPS_EXP_PIPELINE=1,MIUPS_EXP_TILE_TI=16,MIUPS_EXP_CHUNK_ROUNDS=2 base:MIUPS_EXP_TILE_TI=16 || exit 1
echo "=== --config 2 --blocks 2048 (stereo through the external frame pass)"
bash scripts/gpu_ab_arms.sh "--config 2 --blocks 2048" base base:MIUPS_EXP_STEREO_EXT=1 k14cap:MIUPS_EXP_STEREO_EXT=1,MIUPS_EXP_PIPELINE=1 || exit 1
