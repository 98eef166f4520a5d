// gvk_tuning.cpp — the A/B knobs (gvk_set_tuning, include/gvk.h GVK_TUNE_*): one definition for all kernel translation units.
#include "gvk_tuning.h"

int g_variant = 0;         // GVK_TUNE_VARIANT
int g_run_cap = 0;         // GVK_TUNE_RUN_CAP (0 = the default of 20, run_cap_for)
int g_split_hits = 2;      // GVK_TUNE_SPLIT_HITS (samples per table row one launch may hold; 0 = never split a batch)
int g_hot_order = 1;       // GVK_TUNE_HOT_ORDER (measurement: which blocks of a train_hot_kernel launch come first; 1 = long chains, pairs, the other chains)
int g_hot_serialized = 0;  // GVK_TUNE_HOT_SERIALIZED (measurement: gvk_train_episode_hot launches the chains and the pairs of a unit one after the other)
int g_round_steps = -1;   // GVK_TUNE_ROUND_STEPS (-1 = as the caller's form says; 0 = never rounds; 1 .. 8 = rounds of so many entries per task)
int g_chain_cap = 0;       // GVK_TUNE_CHAIN_CAP (entries one chain task trains in sequence; 0 = the default of 7)
#if defined(GVK_AB_BUILDS)  // knobs of the A/B library only (make ab -> build/ab/libgvk_ab.so)
int g_lanes_per_pair = 0;  // GVK_TUNE_LANES_PER_PAIR
int g_generation = 0;      // GVK_TUNE_GENERATION (0 = one launch per batch)
int g_segment_steps = 0;   // GVK_TUNE_SEGMENT_STEPS (0 = off)
int g_skip_loss = 1;       // GVK_TUNE_SKIP_LOSS (gvk_train_episode leaves out the loss of batches nobody can read)
#endif

extern "C" {

int gvk_set_tuning(int key, int value) {
    if (key == GVK_TUNE_VARIANT) {
#if defined(GVK_AB_BUILDS)
        if (value < 0 || value > 4) return fail(GVK_EINVAL, "gvk_set_tuning: variant must be 0 ... 4");
#else
        if (value != 0 && value != 2 && value != 4)
            return fail(GVK_EINVAL, "gvk_set_tuning: variant must be 0, 2 or 4 (1 and 3 exist in the A/B library only: make ab)");
#endif
        g_variant = value;
        return GVK_OK;
    }
    if (key == GVK_TUNE_RUN_CAP) {
        if (value < 0 || value > kMaxRunCap) return fail(GVK_EINVAL, "gvk_set_tuning: run cap must be in [0, 4096]");
        g_run_cap = value;
        return GVK_OK;
    }
    if (key == GVK_TUNE_CHAIN_CAP) {
        if (value < 0 || value > (1 << 20)) return fail(GVK_EINVAL, "gvk_set_tuning: chain cap must be in [0, 2^20]");
        g_chain_cap = value;
        return GVK_OK;
    }
    if (key == GVK_TUNE_ROUND_STEPS) {
        if (value < -1 || value > 8) return fail(GVK_EINVAL, "gvk_set_tuning: round steps must be -1 (default), 0 (one round) or 1 .. 8");
        g_round_steps = value;
        return GVK_OK;
    }
    if (key == GVK_TUNE_HOT_ORDER) {
        if (value < 0 || value > 2) return fail(GVK_EINVAL, "gvk_set_tuning: block order must be 0 .. 2");
        g_hot_order = value;
        return GVK_OK;
    }
    if (key == GVK_TUNE_HOT_SERIALIZED) {
        g_hot_serialized = value != 0;
        return GVK_OK;
    }
    if (key == GVK_TUNE_SPLIT_HITS) {
        if (value < 0) return fail(GVK_EINVAL, "gvk_set_tuning: samples per row and launch must be >= 0");
        g_split_hits = value;
        return GVK_OK;
    }
#if defined(GVK_AB_BUILDS)
    if (key == GVK_TUNE_LANES_PER_PAIR) {
        if (value != 0 && value != 8 && value != 16 && value != 32 && value != 64)
            return fail(GVK_EINVAL, "gvk_set_tuning: lanes per pair must be 0, 8, 16, 32 or 64");
        g_lanes_per_pair = value;
        return GVK_OK;
    }
    if (key == GVK_TUNE_SEGMENT_STEPS) {
        if (value != 0 && value != 1 && value != 2 && value != 4)
            return fail(GVK_EINVAL, "gvk_set_tuning: segment steps must be 0, 1, 2 or 4");
        g_segment_steps = value;
        return GVK_OK;
    }
    if (key == GVK_TUNE_SKIP_LOSS) {
        if (value != 0 && value != 1) return fail(GVK_EINVAL, "gvk_set_tuning: flag must be 0 or 1");
        g_skip_loss = value;
        return GVK_OK;
    }
    if (key == GVK_TUNE_GENERATION) {
        if (value < 0) return fail(GVK_EINVAL, "gvk_set_tuning: generation size must be >= 0");
        g_generation = value;
        return GVK_OK;
    }
#else
    if (key == GVK_TUNE_LANES_PER_PAIR || key == GVK_TUNE_SEGMENT_STEPS || key == GVK_TUNE_SKIP_LOSS ||
        key == GVK_TUNE_GENERATION) {
        if (value == (key == GVK_TUNE_SKIP_LOSS ? 1 : 0)) return GVK_OK;  // the default is all the product library has
        return fail(GVK_EINVAL, "gvk_set_tuning: this knob exists in the A/B library only (make -C graphvite_amd/csrc ab)");
    }
#endif
    return fail(GVK_EINVAL, "gvk_set_tuning: unknown key");
}

int gvk_has_ab_builds(void) {
#if defined(GVK_AB_BUILDS)
    return 1;
#else
    return 0;
#endif
}

}  // extern "C"
