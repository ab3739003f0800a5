#!/usr/bin/env bash
# Bench the same workload with ablated builds of the library (timing only).
set -u
mkdir -p gpurun_out
for v in "" NOSTORE PLANARSTORE; do
  if [ -z "$v" ]; then unset MIUPS_LIB; name=base; else export MIUPS_LIB=$PWD/totton-rasp-gpu-dsp_amd/lib_ablate/libmi_upsampler_$v.so; name=$v; fi
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/ablate_$name.log 2>&1
  rc=$?
  echo "$name rc=$rc $(python3 -c "import json,sys; d=json.loads(open('gpurun_out/ablate_$name.log').read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s kernel_ms', d['roofline']['kernel_ms_avg'])" 2>&1 | tail -1)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
