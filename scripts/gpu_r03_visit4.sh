#!/usr/bin/env bash
# Round-3 visit 4: GPU suite after the pitched-copy fallback, copy-ceiling forms, MultiEngine end-to-end rates of the
# channel split on one GPU (2 / 4 / 8 slots = 64 / 32 / 16-byte rows). Output: gpurun_out/r03h/
set -u
out=gpurun_out/r03h
mkdir -p $out
timeout -k 10 900 python -m pytest tests -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
MIUPS_EXP_COPY_VERBOSE=1 python -c "
import totton_rasp_gpu_dsp_amd as ups
print('copy ceiling', round(ups.device_copy_rate(0, 1<<30, 3),1), 'GB/s')" > $out/copy_forms.txt 2>&1; cat $out/copy_forms.txt
timeout -k 10 300 python scripts/multi_split_rate.py > $out/multi_split_rate.txt 2>&1; cat $out/multi_split_rate.txt
