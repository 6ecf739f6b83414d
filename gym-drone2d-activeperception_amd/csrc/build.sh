#!/bin/bash
# Builds libd2d_hip.so (HIP kernels + C ABI) for gfx950, in-tree next to the sources.
#   -ffp-contract=off : no fused multiply-adds the reference does not perform (bit-exact parity)
#   -mllvm -disable-machine-licm : the pass hoists every 64-bit constant of the persistent kernel's inlined phases to the kernel's
#                       entry, where the register allocator spills them (d2d_hip.hip, D2D_PH_INLINE); closed loops +3-6 %, the
#                       step kernels within 0.5 %
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared \
  -Wall -Wno-unused-function -mllvm -disable-machine-licm ${D2D_EXTRA_FLAGS:-} \
  -o ${D2D_OUT:-libd2d_hip.so} d2d_hip.hip
echo "built $(pwd)/${D2D_OUT:-libd2d_hip.so}"
