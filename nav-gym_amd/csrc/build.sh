#!/bin/bash
# Builds the gfx950 shared library in-tree (nav-gym_amd/nav_gym_amd/libnavsim_hip.so): `make` over csrc/Makefile, one job
# per core (nine objects; 2 min 53 s from scratch on 8 cores, the pedestrian families being the long ones -- as one translation
# unit the library of round 4 took 4 min 12 s -- and a minute or less after an edit that leaves some objects alone).  hipcc cross-compiles without a GPU.
# NAVSIM_OUT / NAVSIM_OBJ redirect the library / the objects (diagnostic A/B builds keep their own object directory),
# NAVSIM_EXTRA_FLAGS adds compiler flags.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
JOBS="${NAVSIM_JOBS:-$(nproc)}"
ARGS=()
[ -n "${NAVSIM_OUT:-}" ] && ARGS+=("OUT=${NAVSIM_OUT}")
[ -n "${NAVSIM_OBJ:-}" ] && ARGS+=("OBJ=${NAVSIM_OBJ}")
[ -n "${HIPCC:-}" ] && ARGS+=("HIPCC=${HIPCC}")
# objects built with other flags must not be reused
if [ -n "${NAVSIM_EXTRA_FLAGS:-}" ] && [ -z "${NAVSIM_OBJ:-}" ]; then
    ARGS+=("OBJ=${HERE}/../../build/obj_$(echo "${NAVSIM_EXTRA_FLAGS}" | md5sum | cut -c1-12)")
fi
make -C "${HERE}" -j "${JOBS}" "${ARGS[@]}"
echo "built ${NAVSIM_OUT:-${HERE}/../nav_gym_amd/libnavsim_hip.so}"
