#!/usr/bin/env bash
# Build an experiment variant of the library: build_variant.sh NAME [-DFLAG ...]
# -> totton-rasp-gpu-dsp_amd/lib_ablate/libmi_upsampler_NAME.so (git-ignored; travels with gpurun)
# e.g. build_variant.sh k14 -DMIUPS_ONLY_LOG2K=14   (one transform length: builds in seconds)
# ISA=1 also keeps the device assembly in /tmp/variant_NAME/ (for instruction counts).
set -eu
name=$1; shift
pkg="$(cd "$(dirname "$0")/.." && pwd)/totton-rasp-gpu-dsp_amd"
mkdir -p "$pkg/lib_ablate" "/tmp/variant_$name"
extra=""
if [ "${ISA:-0}" = "1" ]; then extra="-save-temps=obj -Rpass-analysis=kernel-resource-usage"; fi
( cd "/tmp/variant_$name" && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -ffp-contract=on -DMIUPS_SINGLE_TU $extra "$@" -c "$pkg/csrc/engine.hip" -o "/tmp/variant_$name/engine.o" 2> "/tmp/variant_$name/remarks.txt" ) || { tail -30 "/tmp/variant_$name/remarks.txt"; exit 1; }
hipcc --offload-arch=gfx950 -shared -o "$pkg/lib_ablate/libmi_upsampler_$name.so" "/tmp/variant_$name/engine.o" "$pkg"/build/host/*.o "$pkg/build/capi.o" "$pkg/build/capi_host.o" "$pkg/build/multi_engine.o" "$pkg/build/filter_bank.o" -lpthread
echo "built $name"
