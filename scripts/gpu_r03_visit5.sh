#!/usr/bin/env bash
# Round-3 visit 5: multi-GPU host tests (time split), its end-to-end rate on one GPU. Output: gpurun_out/r03i/
set -u
out=gpurun_out/r03i
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_stream_host.py -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
timeout -k 10 300 python scripts/multi_split_rate.py > $out/multi_split_rate.txt 2>&1; cat $out/multi_split_rate.txt
