// preamble.hpp -- what every translation unit of libnavsim_hip.so starts with: headers, the gfx950-only check, the
// stamp macro of diagnostic builds.  Include it, open the anonymous namespace, then include the kernels_*.hpp
// sections the unit needs (they are not standalone headers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <type_traits>

#include "../../include/navsim.h"
#include "navmath.hpp"
#include "navsim_device.hpp"

#pragma clang fp contract(off)


// gfx950 only: bit-identity of the scans rests on properties of THIS ISA that are proven by exhaustive
// device tests -- v_rsq_f32's rounding inside sqrt_small_int, the float32-only march step (navmath.hpp) -- and on
// v_dot2_i32_i16 / v_pk_* forms that other targets lack.  Another offload arch must not compile silently.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "navsim_kernels.hip is written for gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif

namespace {
constexpr int kMaxWaves = 16;
constexpr int kRegenMaxPackedSide = 520;    // see navsim_regen
}  // namespace

// Diagnostic build only (-DNAVSIM_STAMPS, profiles/stamp_phases.py): s_memtime at the phase
// boundaries of each arena's workgroup, written to a buffer nothing else reads.  The shipped library
// is built without it (no stamp executes in the measured kernel).
#ifdef NAVSIM_STAMPS
namespace { __device__ unsigned long long* g_stamps = nullptr; }
#ifdef NAVSIM_STAMPS_REALTIME      // chip-wide 100 MHz clock (comparable across XCDs) instead of the per-XCD shader clock
#define NAVSIM_STAMP(i) do { if (threadIdx.x == 0 && g_stamps) g_stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define NAVSIM_STAMP(i) do { if (threadIdx.x == 0 && g_stamps) g_stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#else
#define NAVSIM_STAMP(i) do { } while (0)
#endif
