// tuning.hpp -- where the launchers read their schedule knobs (PBR_TUNE_*, include/pbr_hip.h).
//
// A knob has three sources, in this order:
//   1. the CALL: pbr_render_desc.tuning (a caller-owned pbr_tuning, NULL = none); entries equal to PBR_TUNE_UNSET are skipped.  This is
//      how product code tunes a launch: nothing outlives the call, two threads with different settings do not see each other;
//   2. the PROCESS: pbr_set_tuning(knob, value) -- a test / bench / profiling hook, documented as process-global in the header.  Atomic
//      words, so a concurrent launch reads the old or the new value of a knob, never a torn one;
//   3. the library's rule (the initial value of 2).
// The launchers keep the names the knobs had as plain globals: `g_block_log2` is now an expression that resolves 1 -> 2 -> 3.
// The call's pbr_tuning reaches the helper functions of a launch (fill_args, pick_vec, ...) through a thread-local pointer that every
// extern "C" entry point taking a descriptor sets for its own duration (TuningScope): per call and per thread, no shared state.
#pragma once
#include <atomic>
#include <cstdint>

#include "../../include/pbr_hip.h"

namespace pbr {

extern std::atomic<int> g_knobs[PBR_TUNE_COUNT];            // cook_torrance.hip: the process-wide values, initialised to the rules
extern thread_local const pbr_tuning *t_tuning;             // the current call's overrides (NULL outside a call / when the caller gave none)

inline int knob(int k) {
    const pbr_tuning *t = t_tuning;
    if (t && t->knob[k] != PBR_TUNE_UNSET) return t->knob[k];
    return g_knobs[k].load(std::memory_order_relaxed);
}

struct TuningScope {
    const pbr_tuning *previous;
    explicit TuningScope(const pbr_render_desc *d) : previous(t_tuning) { t_tuning = d ? d->tuning : nullptr; }
    ~TuningScope() { t_tuning = previous; }
    TuningScope(const TuningScope &) = delete;
    TuningScope &operator=(const TuningScope &) = delete;
};

}  // namespace pbr

#define g_block_log2       (::pbr::knob(PBR_TUNE_BLOCK_LOG2))       // workgroup size: 64 (6), 128 (7) or 256 (8) lanes; 0 = rule (64; 256 for the one-pixel kernels)
#define g_f16_vec          (::pbr::knob(PBR_TUNE_F16_VEC))          // pixels per lane for fp16 maps with one light: 8 (16-byte loads) or 4
#define g_lds_bytes        (::pbr::knob(PBR_TUNE_LDS_BYTES))        // unused dynamic LDS per one-wave workgroup: an occupancy governor (-1 = rule, ct_launch.hpp)
#define g_batch_inner      (::pbr::knob(PBR_TUNE_BATCH_INNER))      // materials per lane of the several-lights kernels: -1 = rule (4 | 2 | off), 0 = off, 2 | 4 = forced
#define g_scalar_base      (::pbr::knob(PBR_TUNE_SCALAR_BASE))      // scalar plane addresses: 0 never, 1 = rule (single materials), 2 = whenever the launch allows them
#define g_max_vec          (::pbr::knob(PBR_TUNE_MAX_VEC))          // at most this many pixels per lane (1 = the one-pixel kernels everywhere)
#define g_bwd_run          (::pbr::knob(PBR_TUNE_BWD_RUN))          // tiles per wave of the streamed backward kernel (fp16 maps, one light): -1 = rule, 0 = off
#define g_mse_stream       (::pbr::knob(PBR_TUNE_MSE_STREAM))       // ct_loss.hip: streamed loss step for fp16 maps with one light (1) or the one-tile kernels (0)
#define g_tile_repeat      (::pbr::knob(PBR_TUNE_TILE_REPEAT))      // tiled maps: the repeat-inner kernels, forward and backward (-1 = rule: on, 0 = the wrap-around form)
#define g_resize_up2       (::pbr::knob(PBR_TUNE_RESIZE_UP2))       // resize.hip: the register-only kernels (two-tap up-scales, whole-factor down-scales) and the row walk from 7x down-scales up (1), the strip kernels (0), the row walk at every down-scale it can take (2)
