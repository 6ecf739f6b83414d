#!/bin/bash
# Builds libd2d_hip.so (HIP kernels + C ABI) for gfx950, in-tree next to the sources.
#   -ffp-contract=off : no fused multiply-adds the reference does not perform (bit-exact parity)
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared \
  -Wall -Wno-unused-function ${D2D_EXTRA_FLAGS:-} \
  -o ${D2D_OUT:-libd2d_hip.so} d2d_hip.hip
echo "built $(pwd)/${D2D_OUT:-libd2d_hip.so}"
