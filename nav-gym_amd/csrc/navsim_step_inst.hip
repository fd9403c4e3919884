// navsim_step_inst.hip -- ONE (threads per arena, pedestrians or not) family of the fused step kernel.
//
// The step kernel is a template over <BLOCK, PEDS, Field, RULE, RECT, PINL>; compiled from one translation unit its 200-odd
// instantiations made the library a four-minute build (round-4 verdict: "build time is unreported").  This file is compiled
// once per (NAVSIM_INST_BLOCK, NAVSIM_INST_PEDS) pair -- eight objects, in parallel (csrc/Makefile) -- and exports one
// launcher per pair, navsim_step_launch_<BLOCK>_<PEDS>, which navsim_kernels.hip's dispatch_step calls.  Same kernels, same
// code objects as the single unit produced; the launchers are internal to the library (hidden visibility).
#include "preamble.hpp"

#if !defined(NAVSIM_INST_BLOCK) || !defined(NAVSIM_INST_PEDS)
#error "compile with -DNAVSIM_INST_BLOCK=64|256|512|1024 -DNAVSIM_INST_PEDS=0|1 (csrc/Makefile)"
#endif

namespace {
#include "kernels_field.hpp"
#include "kernels_rect.hpp"
#include "kernels_plan.hpp"
#include "kernels_regen_dev.hpp"
#include "kernels_step.hpp"
#include "step_plan.hpp"

// navsim_prepare: walk the dispatch chain of a launch down to its kernel, set what has to be set once per kernel
// (hipFuncSetAttribute for dynamic LDS above 64 KB) and launch nothing -- so that nothing of the kind happens inside a
// hipGraph capture (round-3 advisor: navsim_regen's lone first-observation launch instantiates its own variant)
thread_local bool g_prepare_only = false;      // set per call from the entry point's argument
thread_local int g_aux = 0;                    // navsim_step_replan: max_queries
thread_local const StepInstall* g_install = nullptr;   // navsim_step_install (reset_only bit 4)
template <int BLOCK, bool PEDS, typename Field, int RECT, int RULE, bool PINL>
int launch_step_pinl(const navsim_config* c, const navsim_state* st, const navsim_step_io* io, int reset_only,
                     const uint8_t* mask, const StepPlan& p, hipStream_t s, int grid) {
    if constexpr (PEDS && PINL) {
        if (((reset_only >> 2) & 3) == 3 && (reset_only & 16)) {   // ... with the install of the staged worlds (packed fields)
            if constexpr (!std::is_same<Field, FieldF32>::value) {
                const size_t pl = plan_lds(c->map_h / 5, c->map_w / 5), lds = p.lds > pl ? p.lds : pl;
                if (allow_lds((const void*)navsim_step_replan_install_kernel<BLOCK, Field, RULE, RECT>, lds) != NAVSIM_OK) return NAVSIM_E_UNSUPPORTED;
                if (g_prepare_only) return NAVSIM_OK;
                navsim_step_replan_install_kernel<BLOCK, Field, RULE, RECT><<<grid + c->n_envs, BLOCK, lds, s>>>(
                    *c, *st, *io, *g_install, (unsigned)step_lds_scan_bytes(c, p.park), p.park, p.rect_off, grid, g_aux);
                return NAVSIM_OK;
            }
            return NAVSIM_E_UNSUPPORTED;
        }
        if (((reset_only >> 2) & 3) == 3) {                        // navsim_step_replan: the re-plan inside the step's launch
            const size_t pl = plan_lds(c->map_h / 5, c->map_w / 5), lds = p.lds > pl ? p.lds : pl;
            if (allow_lds((const void*)navsim_step_replan_kernel<BLOCK, Field, RULE, RECT>, lds) != NAVSIM_OK) return NAVSIM_E_UNSUPPORTED;
            if (g_prepare_only) return NAVSIM_OK;
            navsim_step_replan_kernel<BLOCK, Field, RULE, RECT><<<grid + c->n_envs, BLOCK, lds, s>>>(
                *c, *st, *io, (unsigned)step_lds_scan_bytes(c, p.park), p.park, p.rect_off, grid, g_aux);
            return NAVSIM_OK;
        }
        if (((reset_only >> 2) & 3) == NAVSIM_STEP_DUE) {          // navsim_step_part's compact launch (kernels_step.hpp)
            if (allow_lds((const void*)navsim_step_due_kernel<BLOCK, Field, RULE, RECT>, p.lds) != NAVSIM_OK) return NAVSIM_E_UNSUPPORTED;
            if (g_prepare_only) return NAVSIM_OK;
            navsim_step_due_kernel<BLOCK, Field, RULE, RECT><<<grid > 0 ? grid : c->n_envs, BLOCK, p.lds, s>>>(
                *c, *st, *io, (unsigned)step_lds_scan_bytes(c, p.park), p.park, p.rect_off);
            return NAVSIM_OK;
        }
    }
    if constexpr (PINL == PEDS && !std::is_same<Field, FieldF32>::value) {
        if (reset_only & 16) {                                     // navsim_step_install: packed fields, pedestrians inside the step
            if (allow_lds((const void*)navsim_step_install_kernel<BLOCK, PEDS, Field, RULE, RECT>, p.lds) != NAVSIM_OK) return NAVSIM_E_UNSUPPORTED;
            if (g_prepare_only) return NAVSIM_OK;
            navsim_step_install_kernel<BLOCK, PEDS, Field, RULE, RECT><<<grid > 0 ? grid : c->n_envs, BLOCK, p.lds, s>>>(
                *c, *st, *io, *g_install, reset_only & 1, mask, (unsigned)step_lds_scan_bytes(c, p.park), p.park, p.rect_off);
            return NAVSIM_OK;
        }
    }
    if (reset_only & 16) return NAVSIM_E_UNSUPPORTED;
    if constexpr (!PEDS) {
        // the plain form (kernels_step.hpp step_arena FEAT = false: no terminal observation, no next-step reset compiled in) for
        // the calls that use neither -- round 5's code, and what the c2 / c4 bench lines run
        const bool feat = io->final_obs || io->reset_mask || c->auto_reset == NAVSIM_AUTORESET_NEXT_STEP;
        if (!feat) {
            if (allow_lds((const void*)navsim_step_kernel<BLOCK, PEDS, Field, RULE, RECT, PINL, false>, p.lds) != NAVSIM_OK) return NAVSIM_E_UNSUPPORTED;
            if (g_prepare_only) return NAVSIM_OK;
            navsim_step_kernel<BLOCK, PEDS, Field, RULE, RECT, PINL, false><<<grid > 0 ? grid : c->n_envs, BLOCK, p.lds, s>>>(
                *c, *st, *io, reset_only, mask, (unsigned)step_lds_scan_bytes(c, p.park), p.park, p.rect_off);
            return NAVSIM_OK;
        }
    }
    if (allow_lds((const void*)navsim_step_kernel<BLOCK, PEDS, Field, RULE, RECT, PINL>, p.lds) != NAVSIM_OK) return NAVSIM_E_UNSUPPORTED;
    if (g_prepare_only) return NAVSIM_OK;
    navsim_step_kernel<BLOCK, PEDS, Field, RULE, RECT, PINL><<<grid > 0 ? grid : c->n_envs, BLOCK, p.lds, s>>>(
        *c, *st, *io, reset_only, mask, (unsigned)step_lds_scan_bytes(c, p.park), p.park, p.rect_off);
    return NAVSIM_OK;
}
// pedestrian variants: the form without the pedestrian phase when ped_update_kernel has run (reset_only bit 1) or
// nothing is integrated at all (a reset-only launch), else the form that carries it
template <int BLOCK, bool PEDS, typename Field, int RECT, int RULE>
int launch_step_kernel(const navsim_config* c, const navsim_state* st, const navsim_step_io* io, int reset_only,
                       const uint8_t* mask, const StepPlan& p, hipStream_t s, int grid) {
    if constexpr (PEDS) {
        if ((reset_only & 3) == 0 || (reset_only & 16)) return launch_step_pinl<BLOCK, PEDS, Field, RECT, RULE, true>(c, st, io, reset_only, mask, p, s, grid);
    }
    return launch_step_pinl<BLOCK, PEDS, Field, RECT, RULE, false>(c, st, io, reset_only, mask, p, s, grid);
}

template <int BLOCK, bool PEDS, typename Field, int RECT>
int launch_step_rule(const navsim_config* c, const navsim_state* st, const navsim_step_io* io, int reset_only,
                     const uint8_t* mask, const StepPlan& p, hipStream_t s, int grid) {
#ifdef NAVSIM_ONLY_RULE     // experiment builds (profiles/_diag/build_variant.sh): one march rule compiled, a quarter of the build time
    if (march_rule_variant(c) != NAVSIM_ONLY_RULE) return NAVSIM_E_UNSUPPORTED;
    return launch_step_kernel<BLOCK, PEDS, Field, RECT, NAVSIM_ONLY_RULE>(c, st, io, reset_only, mask, p, s, grid);
#else
    switch (march_rule_variant(c)) {
        case NAVSIM_MARCH_F32: return launch_step_kernel<BLOCK, PEDS, Field, RECT, NAVSIM_MARCH_F32>(c, st, io, reset_only, mask, p, s, grid);
        case NAVSIM_MARCH_F32_FMA: return launch_step_kernel<BLOCK, PEDS, Field, RECT, NAVSIM_MARCH_F32_FMA>(c, st, io, reset_only, mask, p, s, grid);
        case kMarchF64Exact32:
            if constexpr (!std::is_same<Field, FieldF32>::value)
                return launch_step_kernel<BLOCK, PEDS, Field, RECT, kMarchF64Exact32>(c, st, io, reset_only, mask, p, s, grid);
            [[fallthrough]];
        default: return launch_step_kernel<BLOCK, PEDS, Field, RECT, NAVSIM_MARCH_F64>(c, st, io, reset_only, mask, p, s, grid);
    }
#endif
}

template <int BLOCK, bool PEDS, typename Field>
int launch_step_field(const navsim_config* c, const navsim_state* st, const navsim_step_io* io, int reset_only,
                      const uint8_t* mask, const StepPlan& p, hipStream_t s, int grid) {
    if (p.rect == 2) return launch_step_rule<BLOCK, PEDS, Field, 2>(c, st, io, reset_only, mask, p, s, grid);
    return p.rect ? launch_step_rule<BLOCK, PEDS, Field, 1>(c, st, io, reset_only, mask, p, s, grid)
                  : launch_step_rule<BLOCK, PEDS, Field, 0>(c, st, io, reset_only, mask, p, s, grid);
}

template <int BLOCK, bool PEDS>
int launch_step_family(const navsim_config* c, const navsim_state* st, const navsim_step_io* io, int reset_only,
                       const uint8_t* mask, const StepPlan& p, hipStream_t s, int grid) {
    if (c->field_format == NAVSIM_FIELD_U16T && !st->field_overflow)           // no saturated cell anywhere
        return launch_step_field<BLOCK, PEDS, FieldU16TN>(c, st, io, reset_only, mask, p, s, grid);
    if (c->field_format == NAVSIM_FIELD_U16T)
        return launch_step_field<BLOCK, PEDS, FieldU16T>(c, st, io, reset_only, mask, p, s, grid);
    return launch_step_rule<BLOCK, PEDS, FieldF32, 0>(c, st, io, reset_only, mask, p, s, grid);
}

}  // namespace

#define NAVSIM_CAT_(a, b, c) a##b##_##c
#define NAVSIM_CAT(a, b, c) NAVSIM_CAT_(a, b, c)

// reset_only: bit 0 = a reset-only launch, bit 1 = ped_update_kernel has already advanced the pedestrians, bits 2-3 = the
// NAVSIM_STEP_* part of navsim_step_part (3: navsim_step_replan, aux = its max_queries), bit 4 = navsim_step_install
// (install = its StepInstall); grid > 0: that
// many workgroups (st->launch_order names their arenas); prepare_only: set the kernel's attributes, launch nothing
extern "C" __attribute__((visibility("hidden")))
int NAVSIM_CAT(navsim_step_launch_, NAVSIM_INST_BLOCK, NAVSIM_INST_PEDS)(const navsim_config* c, const navsim_state* st,
                                                                        const navsim_step_io* io, int reset_only,
                                                                        const uint8_t* mask, void* stream, int grid,
                                                                        int prepare_only, int aux, const void* install) {
    // (a NAVSIM_STEP_DUE launch is the ordinary step kernel on fewer workgroups: its grid does not change the plan)
    const bool due_part = ((reset_only >> 2) & 3) >= NAVSIM_STEP_DUE;       // (also navsim_step_replan: grid = its front workgroups)
    g_aux = aux;
    g_install = (const StepInstall*)install;
    const StepPlan p = plan_step(c, st, due_part ? 0 : grid);
    if (p.block != NAVSIM_INST_BLOCK) return NAVSIM_E_UNSUPPORTED;
    g_prepare_only = prepare_only != 0;
    const int rc = launch_step_family<NAVSIM_INST_BLOCK, (NAVSIM_INST_PEDS != 0)>(c, st, io, reset_only, mask, p, (hipStream_t)stream, grid);
    g_prepare_only = false;
    return rc;
}

// diagnostic builds (-DNAVSIM_STAMPS): every unit has its own copy of the stamp pointer
extern "C" __attribute__((visibility("hidden")))
int NAVSIM_CAT(navsim_step_set_stamps_, NAVSIM_INST_BLOCK, NAVSIM_INST_PEDS)(unsigned long long* buf) {
#ifdef NAVSIM_STAMPS
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &buf, sizeof(buf)) == hipSuccess ? NAVSIM_OK : NAVSIM_E_LAUNCH;
#else
    (void)buf;
    return NAVSIM_E_UNSUPPORTED;
#endif
}
