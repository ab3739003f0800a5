set -u
for a in "--config 2 --blocks 2048" "--config 2" "--config 5"; do
  echo "=== $a"
  bash scripts/gpu_ab_arms.sh "$a" k14b:MIUPS_EXP_NO_R32=1 k14a:MIUPS_EXP_NO_R32=1 k14b k14a || exit 1
done
