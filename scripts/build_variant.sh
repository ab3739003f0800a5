#!/usr/bin/env bash
# Build an experiment variant of the library: build_variant.sh NAME [-DFLAG ...]
# -> totton-rasp-gpu-dsp_amd/lib_ablate/libmi_upsampler_NAME.so (git-ignored; travels with gpurun)
set -eu
name=$1; shift
pkg="$(cd "$(dirname "$0")/.." && pwd)/totton-rasp-gpu-dsp_amd"
make -C "$pkg" >/dev/null
mkdir -p "$pkg/lib_ablate"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" -c "$pkg/csrc/engine.hip" -o "/tmp/engine_$name.o"
hipcc --offload-arch=gfx950 -shared -o "$pkg/lib_ablate/libmi_upsampler_$name.so" "/tmp/engine_$name.o" "$pkg"/build/host/*.o "$pkg/build/capi.o" "$pkg/build/capi_host.o"
echo "built $name"
