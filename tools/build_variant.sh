#!/bin/bash
# Cross-compile a variant of the library in the dev container (no GPU needed):
#   tools/build_variant.sh <name> "<extra hipcc flags>"  ->  c2-ray3dm1d_helium_amd/libc2ray_hip_<name>.so
# Variants travel to the GPU box with the snapshot (*.so is git-ignored, not gpurun-ignored) and are compared there
# on ONE box with tools/abn.sh (C2R_LIB_PATH selects the build).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -pthread -ldl $1 \
  -o "c2-ray3dm1d_helium_amd/libc2ray_hip_${name}.so" c2-ray3dm1d_helium_amd/csrc/c2ray_hip.hip
python3 tools/kernel_resources.py "${2:-k_rates|k_chemistry}" --lib "c2-ray3dm1d_helium_amd/libc2ray_hip_${name}.so"
