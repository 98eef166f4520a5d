// gvk_tuning.h — internal: the A/B knobs of gvk_set_tuning (include/gvk.h GVK_TUNE_*), shared by the kernel translation units
// (defined in gvk_tuning.cpp), and the small host helpers every launcher uses.
#pragma once
#include <hip/hip_runtime.h>

#include "gvk.h"
#include "gvk_internal.h"

#define GVK_HIDDEN __attribute__((visibility("hidden")))
extern GVK_HIDDEN int g_variant;         // GVK_TUNE_VARIANT
extern GVK_HIDDEN int g_run_cap;         // GVK_TUNE_RUN_CAP (0 = the default of 20, run_cap_for)
extern GVK_HIDDEN int g_split_hits;      // GVK_TUNE_SPLIT_HITS (samples per table row one launch may hold; 0 = never split a batch)
extern GVK_HIDDEN int g_hot_order;       // GVK_TUNE_HOT_ORDER (measurement: which blocks of a train_hot_kernel launch come first; 1 = long chains, pairs, the other chains)
extern GVK_HIDDEN int g_hot_serialized;  // GVK_TUNE_HOT_SERIALIZED (measurement: gvk_train_episode_hot launches the chains and the pairs of a unit one after the other)
extern GVK_HIDDEN int g_round_steps;    // GVK_TUNE_ROUND_STEPS (-1 = as the caller's form says; 0 = never rounds; 1 .. 8 = rounds of so many entries per task)
extern GVK_HIDDEN int g_chain_cap;       // GVK_TUNE_CHAIN_CAP (entries one chain task trains in sequence; 0 = the default of 7)
#if defined(GVK_AB_BUILDS)  // knobs of the A/B library only (make ab -> build/ab/libgvk_ab.so)
extern GVK_HIDDEN int g_lanes_per_pair;  // GVK_TUNE_LANES_PER_PAIR
extern GVK_HIDDEN int g_generation;      // GVK_TUNE_GENERATION (0 = one launch per batch)
extern GVK_HIDDEN int g_segment_steps;   // GVK_TUNE_SEGMENT_STEPS (0 = off)
extern GVK_HIDDEN int g_skip_loss;       // GVK_TUNE_SKIP_LOSS (gvk_train_episode leaves out the loss of batches nobody can read)
#else
constexpr int g_lanes_per_pair = 0, g_generation = 0, g_segment_steps = 0, g_skip_loss = 1;
#endif

constexpr int kMaxRunCap = 4096;  // GVK_TUNE_RUN_CAP

namespace {

inline int fail(int code, const char *what) { return gvk_fail(code, "%s", what); }

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return gvk_fail(GVK_EHIP, "%s: %s", what, hipGetErrorString(e));
    return GVK_OK;
}

inline int default_lanes(int dim) {
    switch (dim) {
        case 32: return 8;
        case 64: return 16;
        case 96: return 8;
        case 128: return 16;
        case 256: return 16;
        case 512: return 32;
    }
    return 0;
}

}  // namespace
