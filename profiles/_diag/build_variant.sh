#!/bin/bash
# Diagnostic / A-B builds of the library, reproducibly (round-3 verdict: the stamp builds were only described in prose).
#   profiles/_diag/build_variant.sh <name> "<extra hipcc flags>"    ->  build/libnavsim_<name>.so
# They travel to the GPU box with the snapshot (build/ is git-ignored, not gpurun-ignored) and are selected with
# NAVSIM_LIB=build/libnavsim_<name>.so (nav_gym_amd/lib.py).  Flags in use:
#   -DNAVSIM_ONLY_RULE=1            compile the step / pedestrian-scan kernels for ONE march rule only (1 = NAVSIM_MARCH_F32,
#                                   the default rule): a quarter of the build time; other rules return NAVSIM_E_UNSUPPORTED
#   -DNAVSIM_STAMPS                 s_memtime at the phase boundaries of every arena's workgroup (navsim_debug_set_stamps)
#   -DNAVSIM_STAMPS_REALTIME        ... from the chip-wide 100 MHz clock instead (comparable across XCDs)
#   -DNAVSIM_DIAG_NO_MERGE, -DNAVSIM_DIAG_CHEAP_DIR, -DNAVSIM_DIAG_NO_RESCAN   cost probes (WRONG results): pedestrians invisible /
#                                   cheap beam directions / no second scan after a crash
# Presets:  stamps = "-DNAVSIM_STAMPS -DNAVSIM_STAMPS_REALTIME -DNAVSIM_ONLY_RULE=1"
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$HERE/../.."
NAME="$1"; FLAGS="${2:-}"
if [ "$NAME" = "stamps" ] && [ -z "$FLAGS" ]; then FLAGS="-DNAVSIM_STAMPS -DNAVSIM_STAMPS_REALTIME -DNAVSIM_ONLY_RULE=1"; fi
mkdir -p "$ROOT/build"
NAVSIM_OUT="$ROOT/build/libnavsim_${NAME}.so" NAVSIM_EXTRA_FLAGS="$FLAGS" bash "$ROOT/nav-gym_amd/csrc/build.sh"
