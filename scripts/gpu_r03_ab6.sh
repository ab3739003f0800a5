#!/usr/bin/env bash
# Round-3 A/B 6: the narrow form (one butterfly per thread, four waves per SIMD) with trickled LDS stores against the wide form
set -u
for a in "--config 2 --blocks 2048" "--config 2" "--config 5" "--config 3"; do
  echo "=== $a"
  bash scripts/gpu_ab_arms.sh "$a" base base:MIUPS_EXP_NARROW=1 || exit 1
done
