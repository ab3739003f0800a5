#!/usr/bin/env bash
# Round-3 A/B 5: split form (config 4), second half transform's first-pass inputs parked in global memory instead of
# recomputed; then the GPU suite on the result and the copy-ceiling kernel.
set -u
mkdir -p gpurun_out/r03g
bash scripts/gpu_ab_arms.sh "--config 4" base:MIUPS_EXP_NO_PARK=1 base || exit 1
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r03g/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03g/pytest.log
python -c "
import totton_rasp_gpu_dsp_amd as ups
print('copy ceiling', round(ups.device_copy_rate(0, 1<<30, 5),1), 'GB/s')"
