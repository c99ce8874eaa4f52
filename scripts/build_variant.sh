#!/bin/bash
# A/B builds of the library next to the default one:  scripts/build_variant.sh <name> [extra hipcc flags...]
#   -> nirrt_star_amd/libnirrt_hip_<name>.so   (load with NIRRT_HIP_SO=...; *.so files travel to the GPU box, not into git)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fno-optimize-sibling-calls \
  -mllvm -amdgpu-lower-module-lds-strategy=module -mllvm -sink-insts-to-avoid-spills=true "$@" -o nirrt_star_amd/libnirrt_hip_$name.so \
  nirrt_star_amd/csrc/nirrt_hip.hip nirrt_star_amd/csrc/pointops.hip
echo nirrt_star_amd/libnirrt_hip_$name.so
