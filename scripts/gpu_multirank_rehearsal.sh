#!/usr/bin/env bash
# Rehearsal of the multi-rank bench path on a ONE-GPU box (every rank stacked on device 0: numbers mean nothing).
set -u
mkdir -p gpurun_out
echo "== launcher, 1 rank"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/mr_l1.log 2>&1; echo "rc=$? $(tail -c 200 gpurun_out/mr_l1.log | tr '\n' ' ')"
echo "== launcher, 2 ranks stacked"
BENCH_STACK_RANKS_FOR_TEST=1 BENCH_VISIBLE_DEVICES_FOR_TEST=2 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/mr_l2.log 2>&1; echo "rc=$? $(tail -c 700 gpurun_out/mr_l2.log | tr '\n' ' ')"
echo "== plain --gpus 2, stacked"
BENCH_STACK_RANKS_FOR_TEST=1 BENCH_VISIBLE_DEVICES_FOR_TEST=2 timeout -k 10 400 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/mr_p2.log 2>&1; echo "rc=$? $(tail -c 700 gpurun_out/mr_p2.log | tr '\n' ' ')"
