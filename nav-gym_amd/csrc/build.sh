#!/bin/bash
# Builds the gfx950 shared library in-tree (nav-gym_amd/nav_gym_amd/libnavsim_hip.so).
# hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED for parity with the oracle:
# no v_mul/v_add pair of the specified float32/float64 sequences may be fused.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${NAVSIM_OUT:-${HERE}/../nav_gym_amd/libnavsim_hip.so}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"${HIPCC}" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
    -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical -Wno-pass-failed ${NAVSIM_EXTRA_FLAGS:-} \
    -o "${OUT}" "${HERE}/navsim_kernels.hip"
echo "built ${OUT}"
