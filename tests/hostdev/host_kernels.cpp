// TEST INFRASTRUCTURE ONLY — never part of the product, never loaded by it.
//
// The device entry points of include/gvk.h implemented ON THE HOST by calling the CPU oracle (oracle/gv_oracle.c), plus the
// allocator behind tests/hostdev/hip/hip_runtime.h.  Linked with the engine's own sources into
// tests/hostdev/build/libgvk_host.so (tests/hostdev/Makefile), which the `-m "not gpu"` tests load through GVK_LIBRARY: the
// solver engine's host logic then runs without a GPU, its "kernels" being the SEQUENTIAL oracle.  The GPU parity tests use
// the same library (in a process of its own) as the sequential pipeline the HIP path is compared with.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <vector>

#include "gvk.h"
#include "gvk_internal.h"

extern "C" {
// oracle/gv_oracle.c
float gvo_lr(float init_lr, int linear, int batch_id, int num_batch);
size_t gvo_hot_lists(const uint32_t *batch, const uint32_t *negatives, int batch_size, int k, uint32_t kv, uint32_t kc,
                     uint32_t *chain_start, uint32_t *entries);
int gvo_train_hot(int dim, float *vertex, float *context, const uint32_t *batch, const uint32_t *negatives, float *loss,
                  int batch_size, int k, float lr, float wd, float negative_weight, uint32_t kv, uint32_t kc,
                  const uint32_t *chain_start, const uint32_t *entries, uint32_t cap, uint32_t max_tasks, int lerp);
int gvo_hot_unit_chains(int dim, float *vertex, float *context, float lr, float wd, float negative_weight, uint32_t kv, uint32_t kc,
                        const uint32_t *chain_start, const uint32_t *entries, uint32_t cap, uint32_t max_tasks, int k);
void gvo_set_pairs_concurrent(int on);
void gvo_set_long_task(uint32_t entries);
void gvo_set_round_steps(uint32_t steps);
void gvo_set_hub_snapshot(const float *vertex_hub_rows, uint32_t kv, const float *context_hub_rows, uint32_t kc);
void gvo_set_pairs_read_before(int on);
void gvo_set_pairs_at(float at);
int gvo_train_hot_moments(int dim, int type, float *vertex, float *context, float *vm1, float *cm1, float *vm2, float *cm2, const uint32_t *batch,
                          const uint32_t *negatives, float *loss, int batch_size, int k, float lr, float wd, float negative_weight, const float *hp,
                          uint32_t kv, uint32_t kc, const uint32_t *chain_start, const uint32_t *entries);
int gvo_train_pairs_hot(int dim, float *vertex, float *context, const uint32_t *batch, const uint32_t *negatives, float *loss,
                        int batch_size, int k, float lr, float wd, float negative_weight, uint32_t kv, uint32_t kc,
                        const float *before_vertex, const float *before_context);
int gvo_train(int dim, int type, float *vertex, float *context, float *vm1, float *cm1, float *vm2, float *cm2,
              const uint32_t *batch, const uint32_t *negatives, float *loss, int batch_size, int k, float lr, float wd,
              float negative_weight, const float *hp);
void gvo_predict(int dim, const float *vertex, const float *context, const uint32_t *batch, float *logits, int batch_size);
void gvo_negative_draw_batch(const float *prob, const uint32_t *alias, uint32_t count, uint64_t seed, uint32_t batch_id,
                             int batch_size, int k, uint32_t *out);
void gvo_negative_draw_class_batch(const uint32_t *first, const uint32_t *count, const float *prob, const uint32_t *alias,
                                   uint32_t num_class, uint64_t seed, uint32_t batch_id, int batch_size, int k, uint32_t *out);
void gvo_sample_pairs(const float *prob, const uint32_t *alias, const uint32_t *block_pairs, uint32_t count, uint64_t seed,
                      uint64_t first, size_t n, uint32_t *out);
int gvo_sample_walks_device(const uint64_t *flat, const uint32_t *edges_uv, const float *edge_prob, const uint32_t *edge_alias,
                            uint32_t D, const float *nb_prob, const uint32_t *nb_alias, const uint32_t *sorted_nb,
                            const uint32_t *local, int biased, float p, float q, uint64_t seed, uint64_t first_walk,
                            uint32_t *pool, size_t pool_pairs, int L, int aug, int shuffle_base);
}

// ---- "device" memory ---------------------------------------------------------------------------------------------------

extern "C" {
size_t gvh_memory_limit = (size_t)64 << 30;
size_t gvh_memory_used = 0;
}

namespace {
std::mutex g_mutex;
std::map<void *, size_t> g_blocks;
}  // namespace

hipError_t gvh_malloc(void **p, size_t bytes) {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (gvh_memory_used + bytes > gvh_memory_limit) return hipErrorOutOfMemory;
    *p = malloc(bytes ? bytes : 1);
    if (!*p) return hipErrorOutOfMemory;
    g_blocks[*p] = bytes;
    gvh_memory_used += bytes;
    return hipSuccess;
}

hipError_t gvh_free(void *p) {
    if (!p) return hipSuccess;
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_blocks.find(p);
    if (it != g_blocks.end()) {
        gvh_memory_used -= it->second;
        g_blocks.erase(it);
    }
    free(p);
    return hipSuccess;
}

// ---- what the tests can observe ---------------------------------------------------------------------------------------

namespace {
struct Launch {
    uint32_t batch_id;
    float lr;
    int batch_size;
    uint32_t n_vertex;
};
std::vector<Launch> g_launches;
int g_split_hits = 2;
typedef void (*BatchObserver)(const uint32_t *pairs, int batch_size, uint32_t batch_id, const float *vertex, const float *context);
BatchObserver g_observer = nullptr;

void unpack(const gvk_alias_entry *table, size_t n, std::vector<float> &prob, std::vector<uint32_t> &alias) {
    prob.resize(n), alias.resize(n);
    for (size_t i = 0; i < n; i++) prob[i] = table[i].prob, alias[i] = table[i].alias;
}
}  // namespace

extern "C" {

int gvh_is_host_build(void) { return 1; }
// called with every batch right before it is trained (tests look at the pairs a block is trained on)
void gvh_set_batch_observer(BatchObserver observer) { g_observer = observer; }
void gvh_set_memory_limit(size_t bytes) { gvh_memory_limit = bytes; }

// every gvk_train / batch of gvk_train_episode since the last clear: (batch id, lr, batch size, rows of the head table)
size_t gvh_launch_log(uint32_t *batch_ids, float *lrs, int32_t *batch_sizes, uint32_t *rows, size_t capacity) {
    std::lock_guard<std::mutex> lock(g_mutex);
    for (size_t i = 0; i < g_launches.size() && i < capacity; i++) {
        if (batch_ids) batch_ids[i] = g_launches[i].batch_id;
        if (lrs) lrs[i] = g_launches[i].lr;
        if (batch_sizes) batch_sizes[i] = g_launches[i].batch_size;
        if (rows) rows[i] = g_launches[i].n_vertex;
    }
    return g_launches.size();
}
void gvh_launch_log_clear(void) {
    std::lock_guard<std::mutex> lock(g_mutex);
    g_launches.clear();
}

// ---- include/gvk.h ---------------------------------------------------------------------------------------------------------

int gvk_negative_draw(void *, const gvk_alias_entry *table, uint32_t count, uint64_t seed, uint32_t batch_id, uint32_t *negatives,
                      int batch_size, int num_negative) {
    std::vector<float> prob;
    std::vector<uint32_t> alias;
    unpack(table, count, prob, alias);
    gvo_negative_draw_batch(prob.data(), alias.data(), count, seed, batch_id, batch_size, num_negative, negatives);
    return GVK_OK;
}

int gvk_negative_draw_classes(void *, const gvk_class_entry *classes, uint32_t class_count, uint64_t seed, uint32_t batch_id,
                              uint32_t *negatives, int batch_size, int num_negative) {
    std::vector<float> prob(class_count);
    std::vector<uint32_t> alias(class_count), first(class_count), count(class_count);
    for (uint32_t i = 0; i < class_count; i++)
        prob[i] = classes[i].prob, alias[i] = classes[i].alias, first[i] = classes[i].first, count[i] = classes[i].count;
    gvo_negative_draw_class_batch(first.data(), count.data(), prob.data(), alias.data(), class_count, seed, batch_id, batch_size,
                                  num_negative, negatives);
    return GVK_OK;
}

static int train_batch(int dim, const gvk_optimizer *o, float lr, const gvk_tables *t, const uint32_t *pairs,
                       const gvk_negative_source *neg, uint32_t batch_id, float *loss, int batch_size, int k, float negative_weight) {
    std::vector<uint32_t> drawn;
    const uint32_t *negatives = neg->negatives;
    if (k > 0 && !negatives) {
        drawn.resize((size_t)batch_size * k);
        if (neg->classes)
            gvk_negative_draw_classes(nullptr, neg->classes, neg->class_count, neg->seed, batch_id, drawn.data(), batch_size, k);
        else if (neg->table)
            gvk_negative_draw(nullptr, neg->table, neg->count, neg->seed, batch_id, drawn.data(), batch_size, k);
        else
            return gvk_fail(GVK_EINVAL, "gvk_train: num_negative > 0 but neither negatives nor an alias table given");
        negatives = drawn.data();
    }
    const float hp[3] = {o->hp0, o->hp1, o->epsilon};
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        g_launches.push_back({batch_id, lr, batch_size, t->n_vertex});
    }
    if (g_observer) g_observer(pairs, batch_size, batch_id, t->vertex, t->context);
    return gvo_train(dim, o->type, t->vertex, t->context, t->vertex_moment1, t->context_moment1, t->vertex_moment2,
                     t->context_moment2, pairs, negatives, loss, batch_size, k, lr, o->weight_decay, negative_weight, hp) == 0
               ? GVK_OK
               : gvk_fail(GVK_ENOMEM, "gvk_train: out of memory");
}

int gvk_train(void *, int dim, const gvk_optimizer *optimizer, const gvk_tables *tables, const uint32_t *pairs,
              const gvk_negative_source *negative, uint32_t batch_id, float *loss, int batch_size, int num_negative,
              float negative_weight) {
    if (batch_size == 0) return GVK_OK;
    return train_batch(dim, optimizer, optimizer->lr, tables, pairs, negative, batch_id, loss, batch_size, num_negative, negative_weight);
}

// GVH_SHUFFLE_POOL=1 (experiment): the samples of a call — a block visit's batches — in a random order before they are trained: what
// an executor learns from a well-mixed stream (the order thousands of concurrent device walks produce) instead of the samplers' walk order
static const uint32_t *shuffled_pool(const uint32_t *pairs, int num_batches, int batch_size, uint32_t first_batch_id, std::vector<uint32_t> &shuffled) {
    if (!(getenv("GVH_SHUFFLE_POOL") && atoi(getenv("GVH_SHUFFLE_POOL")) && num_batches > 0)) return pairs;
    const size_t n = (size_t)num_batches * batch_size;
    shuffled.assign(pairs, pairs + 2 * n);
    uint64_t state = 0x9E3779B97F4A7C15ull * (first_batch_id + 1);
    for (size_t i = n - 1; i > 0; i--) {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        const size_t j = (size_t)((state >> 33) % (i + 1));
        std::swap(shuffled[2 * i], shuffled[2 * j]), std::swap(shuffled[2 * i + 1], shuffled[2 * j + 1]);
    }
    return shuffled.data();
}

int gvk_train_episode(void *, int dim, const gvk_optimizer *optimizer, int linear_schedule, const gvk_tables *tables,
                      const uint32_t *pairs, const gvk_negative_source *negative, uint32_t first_batch_id, uint32_t batch_id_stride,
                      uint32_t total_batches, int num_batches, float *loss, int batch_size, int num_negative, float negative_weight) {
    std::vector<uint32_t> shuffled;
    pairs = shuffled_pool(pairs, num_batches, batch_size, first_batch_id, shuffled);
    for (int i = 0; i < num_batches; i++) {
        const uint32_t id = first_batch_id + (uint32_t)i * batch_id_stride;
        const float lr = gvo_lr(optimizer->lr, linear_schedule, (int)id, (int)total_batches);
        const int rc = train_batch(dim, optimizer, lr, tables, pairs + (size_t)i * batch_size * 2, negative, id, loss, batch_size,
                                   num_negative, negative_weight);
        if (rc != GVK_OK) return rc;
    }
    return GVK_OK;
}

// Hub rows by chains (gvk.h): a device-execution concern — what the chains restore is the sequential result, and the host
// build's kernels ARE sequential.  The work lists are not needed; the batches are trained as gvk_train_episode trains them.
int gvk_hot_plan(int dim, int batch_size, int num_negative, uint32_t hot_vertex, uint32_t hot_context, int num_batch, int parts, int,
                 size_t *bytes) {
    if (!bytes || dim <= 0 || parts < 1 || batch_size % parts || batch_size <= 0 || num_negative < 0 || num_batch < 0 || (uint64_t)hot_vertex + hot_context == 0)
        return gvk_fail(GVK_EINVAL, "gvk_hot_plan: bad argument");
    *bytes = 256;
    return GVK_OK;
}

int gvk_hot_build(void *, int, void *workspace, size_t, const uint32_t *pool, int, int, int, const gvk_negative_source *negative, uint32_t,
                  uint32_t, uint32_t, uint32_t, int, int) {
    return workspace && pool && negative ? GVK_OK : gvk_fail(GVK_EINVAL, "gvk_hot_build: null argument");
}

int gvk_hot_build_sliced(void *, int, void *workspace, size_t, const uint32_t *pool, int, int, int, const gvk_negative_source *negative, uint32_t,
                         uint32_t, uint32_t, uint32_t, int, int, int units_per_launch) {
    return workspace && pool && negative && units_per_launch >= 0 ? GVK_OK : gvk_fail(GVK_EINVAL, "gvk_hot_build_sliced: bad argument");
}

int gvk_train_episode_hot(void *stream, int dim, const gvk_optimizer *optimizer, int linear_schedule, const gvk_tables *tables,
                          const uint32_t *pairs, const gvk_negative_source *negative, uint32_t first_batch_id,
                          uint32_t batch_id_stride, uint32_t total_batches, int num_batches, float *loss, int batch_size,
                          int num_negative, float negative_weight, void *workspace, size_t, uint32_t hot_vertex,
                          uint32_t hot_context, int workspace_batches, int parts, int chain_cap, int form) {
    if (!workspace || num_batches > workspace_batches || hot_vertex > tables->n_vertex || hot_context > tables->n_context)
        return gvk_fail(GVK_EINVAL, "gvk_train_episode_hot: bad argument");
    // GVH_EXECUTOR: instead of the sequential batch, the device path's execution model unit by unit (oracle/gv_oracle.c
    // gvo_train_hot: chains of both families from the unit's start state, then its pairs; `form` & GVK_HOT_LERP or GVH_LERP=1:
    // the pairs read hub rows along the chains' way) — an executor simulator for what a change of the device path does to
    // learning, no GPU needed.  "units": unit after unit; "pipelined": the chains of unit u + 1 are computed before the pairs of
    // unit u have written anything, as one launch of the product does.  (The Hogwild losses of rows that are not hub rows are
    // not simulated: the pairs run in sample order.)
    std::vector<uint32_t> shuffled_pairs;
    const char *executor = getenv("GVH_EXECUTOR");
    if (executor && *executor && strcmp(executor, "sequential")) pairs = shuffled_pool(pairs, num_batches, batch_size, first_batch_id, shuffled_pairs);
    if (!executor || !*executor || !strcmp(executor, "sequential"))
        return gvk_train_episode(stream, dim, optimizer, linear_schedule, tables, pairs, negative, first_batch_id, batch_id_stride,
                                 total_batches, num_batches, loss, batch_size, num_negative, negative_weight);
    gvo_set_pairs_read_before(getenv("GVH_PAIRS_READ") ? (!strcmp(getenv("GVH_PAIRS_READ"), "before") ? 1 : (!strcmp(getenv("GVH_PAIRS_READ"), "lerp") ? 2 : 0)) : 0);  // experiment: hub rows as the unit found them
    gvo_set_pairs_at(getenv("GVH_PAIRS_AT") ? (float)atof(getenv("GVH_PAIRS_AT")) : -1.0f);  // ... SGD: with lerp, at this fixed place of the chains' way
    if (optimizer->type != GVK_SGD) {
        // a moment optimizer (round 6): every unit in the serialized form of its chains (gvo_train_hot_moments: one sequential task per hub
        // row from the unit's start state, then the unit's pairs in sample order) — what the unit scheme itself costs, without Hogwild
        const int n = batch_size / parts, k = num_negative;
        std::vector<uint32_t> start(hot_vertex + hot_context + 1), entries(2 * (size_t)(k + 1) * n + 1), negatives((size_t)batch_size * std::max(k, 1));
        const float hp[3] = {optimizer->hp0, optimizer->hp1, optimizer->epsilon};
        for (int i = 0; i < num_batches; i++) {
            const uint32_t id = first_batch_id + (uint32_t)i * batch_id_stride;
            if (negative->classes) gvk_negative_draw_classes(nullptr, negative->classes, negative->class_count, negative->seed, id, negatives.data(), batch_size, k);
            else gvk_negative_draw(nullptr, negative->table, negative->count, negative->seed, id, negatives.data(), batch_size, k);
            const float lr = gvo_lr(optimizer->lr, linear_schedule, (int)id, (int)total_batches);
            for (int q = 0; q < parts; q++) {
                const uint32_t *part = pairs + ((size_t)i * batch_size + (size_t)q * n) * 2, *neg = negatives.data() + (size_t)q * n * k;
                gvo_hot_lists(part, neg, n, k, hot_vertex, hot_context, start.data(), entries.data());
                if (gvo_train_hot_moments(dim, optimizer->type, tables->vertex, tables->context, tables->vertex_moment1, tables->context_moment1,
                                          tables->vertex_moment2, tables->context_moment2, part, neg, loss + (size_t)q * n, n, k, lr, optimizer->weight_decay,
                                          negative_weight, hp, hot_vertex, hot_context, start.data(), entries.data()))
                    return gvk_fail(GVK_ENOMEM, "gvk_train_episode_hot: out of memory");
            }
        }
        return GVK_OK;
    }
    const int lerp = (form & GVK_HOT_LERP) || (getenv("GVH_LERP") && atoi(getenv("GVH_LERP")));
    // GVH_PAIRS=concurrent: the pairs of a unit as one launch runs them (reads as the unit found the rows, the later of two writers stays)
    gvo_set_pairs_concurrent(getenv("GVH_PAIRS") && !strcmp(getenv("GVH_PAIRS"), "concurrent"));
    // GVH_LONG_TASK=t (experiments): a chain of more than cap entries as tasks of t entries, as many as it takes (1: every entry on
    // its own); default 0: the device path's tasks of cap entries side by side, at most GVH_MAX_TASKS
    gvo_set_long_task(getenv("GVH_LONG_TASK") ? (uint32_t)atoi(getenv("GVH_LONG_TASK")) : 0u);
    // form & GVK_HOT_ROUNDS: the tasks of a long chain work in rounds of GVK_HOT_ROUND_STEPS entries, the row re-composed between rounds
    // (train_long_chains_in_rounds); GVH_ROUND_STEPS=s overrides (0: one round)
    gvo_set_round_steps(getenv("GVH_ROUND_STEPS") ? (uint32_t)atoi(getenv("GVH_ROUND_STEPS")) : ((form & GVK_HOT_ROUNDS) ? (uint32_t)GVK_HOT_ROUND_STEPS : 0u));
    const int pipelined = strstr(executor, "pipelined") != nullptr, n = batch_size / parts, k = num_negative;
    if (getenv("GVH_CHAIN_CAP")) chain_cap = atoi(getenv("GVH_CHAIN_CAP"));
    const uint32_t cap = (uint32_t)std::min(chain_cap > 0 ? chain_cap : 7, 7);  // chain_cap_for, gvk_chains.hip
    const uint32_t max_tasks = getenv("GVH_MAX_TASKS") ? (uint32_t)atoi(getenv("GVH_MAX_TASKS")) : (dim == 512 ? 8u : (dim == 32 || dim == 96 ? 32u : 16u));
    std::vector<uint32_t> start(hot_vertex + hot_context + 1), entries(2 * (size_t)(k + 1) * n + 1);
    std::vector<uint32_t> all((size_t)num_batches * batch_size * std::max(k, 1));
    for (int i = 0; i < num_batches; i++) {
        const uint32_t id = first_batch_id + (uint32_t)i * batch_id_stride;
        uint32_t *out = all.data() + (size_t)i * batch_size * k;
        if (negative->classes)
            gvk_negative_draw_classes(nullptr, negative->classes, negative->class_count, negative->seed, id, out, batch_size, k);
        else
            gvk_negative_draw(nullptr, negative->table, negative->count, negative->seed, id, out, batch_size, k);
    }
    const int units = num_batches * parts;
    const size_t hv = (size_t)hot_vertex * dim, hc = (size_t)hot_context * dim;
    std::vector<uint32_t> none(hot_vertex + hot_context + 1, 0);
    // with_chains / with_pairs of unit u on the tables as they are
    auto unit = [&](int u, bool with_chains, bool with_pairs, const float *before_v, const float *before_c) {
        const int i = u / parts, q = u % parts;
        const uint32_t *part = pairs + ((size_t)i * batch_size + (size_t)q * n) * 2, *negatives = all.data() + ((size_t)i * batch_size + (size_t)q * n) * k;
        const float lr = gvo_lr(optimizer->lr, linear_schedule, (int)(first_batch_id + (uint32_t)i * batch_id_stride), (int)total_batches);
        if (with_chains) gvo_hot_lists(part, negatives, n, k, hot_vertex, hot_context, start.data(), entries.data());
        if (with_chains && with_pairs)
            return gvo_train_hot(dim, tables->vertex, tables->context, part, negatives, loss + (size_t)q * n, n, k, lr, optimizer->weight_decay,
                                 negative_weight, hot_vertex, hot_context, start.data(), entries.data(), cap, max_tasks, lerp);
        if (with_chains)
            return gvo_hot_unit_chains(dim, tables->vertex, tables->context, lr, optimizer->weight_decay, negative_weight, hot_vertex,
                                       hot_context, start.data(), entries.data(), cap, max_tasks, k);
        // pairs only: the hub rows in the tables are the rows after the unit's chains; with lerp they are read on the way from
        // (before_v, before_c)
        return gvo_train_pairs_hot(dim, tables->vertex, tables->context, part, negatives, loss + (size_t)q * n, n, k, lr, optimizer->weight_decay,
                                   negative_weight, hot_vertex, hot_context, lerp ? before_v : nullptr, lerp ? before_c : nullptr);
    };
    if (strstr(executor, "batch")) {
        // "batchahead": the chains of a whole batch (unit after unit, the hub rows after every unit kept) run BEFORE the pairs of
        // the batch before it — the chain front one to two batches ahead of the pairs, as a chain stream beside one pairs launch
        // per batch would run —, and the pairs of unit u then read the hub rows as the chains of unit u left them.
        // "batchlerp": the same chains; the pairs of a batch read a hub row on the straight line between GVH_KNOTS + 1 kept
        // states (1: from the batch's start to its end).
        const bool whole = strstr(executor, "batchlerp") != nullptr;
        const int knots = getenv("GVH_KNOTS") ? atoi(getenv("GVH_KNOTS")) : 1;
        if (whole && (knots < 1 || parts % knots)) return gvk_fail(GVK_EINVAL, "GVH_KNOTS must divide the parts");
        std::vector<std::vector<float>> sv(2 * (parts + 1)), sc(2 * (parts + 1));
        auto save = [&](int b, int j) {
            auto &v = sv[(b & 1) * (parts + 1) + j], &c = sc[(b & 1) * (parts + 1) + j];
            v.assign(tables->vertex, tables->vertex + hv), c.assign(tables->context, tables->context + hc);
        };
        auto put = [&](const std::vector<float> &v, const std::vector<float> &c) {
            memcpy(tables->vertex, v.data(), hv * 4), memcpy(tables->context, c.data(), hc * 4);
        };
        auto chains_of_batch = [&](int b) {
            save(b, 0);
            for (int q = 0; q < parts; q++) {
                if (unit(b * parts + q, true, false, nullptr, nullptr)) return 1;
                save(b, q + 1);
            }
            return 0;
        };
        auto pairs_range = [&](int i, int q0, int count, const float *bv, const float *bc) {
            const uint32_t *part = pairs + ((size_t)i * batch_size + (size_t)q0 * n) * 2, *negatives = all.data() + ((size_t)i * batch_size + (size_t)q0 * n) * k;
            const float lr = gvo_lr(optimizer->lr, linear_schedule, (int)(first_batch_id + (uint32_t)i * batch_id_stride), (int)total_batches);
            return gvo_train_pairs_hot(dim, tables->vertex, tables->context, part, negatives, loss + (size_t)q0 * n, n * count, k, lr,
                                       optimizer->weight_decay, negative_weight, hot_vertex, hot_context, bv, bc);
        };
        if (chains_of_batch(0)) return gvk_fail(GVK_ENOMEM, "gvk_train_episode_hot: out of memory");
        std::vector<float> front_v(hv), front_c(hc);
        for (int i = 0; i < num_batches; i++) {
            if (i + 1 < num_batches && chains_of_batch(i + 1)) return gvk_fail(GVK_ENOMEM, "gvk_train_episode_hot: out of memory");
            front_v.assign(tables->vertex, tables->vertex + hv), front_c.assign(tables->context, tables->context + hc);
            const int base = (i & 1) * (parts + 1);
            if (whole) {
                const int seg = parts / knots;
                for (int j = 0; j < knots; j++) {
                    put(sv[base + (j + 1) * seg], sc[base + (j + 1) * seg]);
                    if (pairs_range(i, j * seg, seg, sv[base + j * seg].data(), sc[base + j * seg].data())) return gvk_fail(GVK_ENOMEM, "out of memory");
                }
            } else {
                for (int q = 0; q < parts; q++) {
                    put(sv[base + q + 1], sc[base + q + 1]);
                    if (pairs_range(i, q, 1, lerp ? sv[base + q].data() : nullptr, lerp ? sc[base + q].data() : nullptr)) return gvk_fail(GVK_ENOMEM, "out of memory");
                }
            }
            put(front_v, front_c);
        }
        return GVK_OK;
    }
    if (!pipelined) {
        for (int u = 0; u < units; u++)
            if (unit(u, true, true, nullptr, nullptr)) return gvk_fail(GVK_ENOMEM, "gvk_train_episode_hot: out of memory");
        return GVK_OK;
    }
    // GVH_GROUP=g (experiment, round 6): the chains of g consecutive units read their hub PARTNERS as the group found them (their own
    // rows go on from unit to unit) — what one launch for the chains of g units would compute
    const int group = getenv("GVH_GROUP") ? std::max(atoi(getenv("GVH_GROUP")), 1) : 1;
    std::vector<float> snap_v, snap_c;
    auto group_start = [&](int u) {
        if (group <= 1 || u % group) return;
        snap_v.assign(tables->vertex, tables->vertex + hv), snap_c.assign(tables->context, tables->context + hc);
        gvo_set_hub_snapshot(snap_v.data(), hot_vertex, snap_c.data(), hot_context);
    };
    struct ClearSnapshot { ~ClearSnapshot() { gvo_set_hub_snapshot(nullptr, 0, nullptr, 0); } } clear_snapshot;
    // the product form: launch u = the pairs of unit u + the chains of unit u + 1, which start before those pairs have written
    // anything and leave their rows in another mirror — the chains of unit u + 1 are computed from the tables as the pairs of
    // unit u - 1 left them, the pairs of unit u read the hub rows the chains of unit u left
    std::vector<float> before_v(hv), before_c(hc), now_v(hv), now_c(hc), next_v(hv), next_c(hc);
    memcpy(before_v.data(), tables->vertex, hv * 4), memcpy(before_c.data(), tables->context, hc * 4);
    group_start(0);
    if (unit(0, true, false, nullptr, nullptr)) return gvk_fail(GVK_ENOMEM, "gvk_train_episode_hot: out of memory");
    for (int u = 0; u < units; u++) {
        memcpy(now_v.data(), tables->vertex, hv * 4), memcpy(now_c.data(), tables->context, hc * 4);
        if (u + 1 < units) {  // the chains of the next unit, from the tables as they are now; their rows land after this unit's pairs
            group_start(u + 1);
            if (unit(u + 1, true, false, nullptr, nullptr)) return gvk_fail(GVK_ENOMEM, "gvk_train_episode_hot: out of memory");
            memcpy(next_v.data(), tables->vertex, hv * 4), memcpy(next_c.data(), tables->context, hc * 4);
            memcpy(tables->vertex, now_v.data(), hv * 4), memcpy(tables->context, now_c.data(), hc * 4);
        }
        if (unit(u, false, true, before_v.data(), before_c.data())) return gvk_fail(GVK_ENOMEM, "gvk_train_episode_hot: out of memory");
        before_v.swap(now_v), before_c.swap(now_c);
        if (u + 1 < units) memcpy(tables->vertex, next_v.data(), hv * 4), memcpy(tables->context, next_c.data(), hc * 4);
    }
    return GVK_OK;
}

// The chain-stream executor (gvk.h gvk_ahead_*): the same stand-in — sequential batches, or with GVH_EXECUTOR the device path's
// execution model ("batchahead" is the model of this executor: the chains of a batch before the pairs of the batch before it).
int gvk_ahead_plan(int dim, int batch_size, int num_negative, uint32_t hot_vertex, uint32_t hot_context, int num_batch, int parts, int chain_cap,
                   size_t *bytes) {
    return gvk_hot_plan(dim, batch_size, num_negative, hot_vertex, hot_context, num_batch, parts, chain_cap, bytes);
}

int gvk_ahead_build(void *stream, int dim, void *workspace, size_t workspace_bytes, const uint32_t *pool, int batch_size, int num_batch,
                    int num_negative, const gvk_negative_source *negative, uint32_t first_batch_id, uint32_t batch_id_stride,
                    uint32_t hot_vertex, uint32_t hot_context, int parts, int chain_cap, int) {
    return gvk_hot_build(stream, dim, workspace, workspace_bytes, pool, batch_size, num_batch, num_negative, negative, first_batch_id,
                         batch_id_stride, hot_vertex, hot_context, parts, chain_cap);
}

void gvk_ahead_release(void *) {}

int gvk_train_episode_ahead(void *stream, void *, int dim, const gvk_optimizer *optimizer, int linear_schedule, const gvk_tables *tables,
                            const uint32_t *pairs, const gvk_negative_source *negative, uint32_t first_batch_id, uint32_t batch_id_stride,
                            uint32_t total_batches, int num_batches, float *loss, int batch_size, int num_negative, float negative_weight,
                            void *workspace, size_t workspace_bytes, uint32_t hot_vertex, uint32_t hot_context, int workspace_batches,
                            int parts, int chain_cap, int, int, int form) {
    return gvk_train_episode_hot(stream, dim, optimizer, linear_schedule, tables, pairs, negative, first_batch_id, batch_id_stride,
                                 total_batches, num_batches, loss, batch_size, num_negative, negative_weight, workspace, workspace_bytes,
                                 hot_vertex, hot_context, workspace_batches, parts, chain_cap, form);
}

int gvk_predict(void *, int dim, const float *vertex, const float *context, const uint32_t *pairs, float *logits, int batch_size) {
    gvo_predict(dim, vertex, context, pairs, logits, batch_size);
    return GVK_OK;
}

int gvk_alias_sample(void *, const gvk_alias_entry *, uint32_t, const double *, uint32_t *, int) {
    return gvk_fail(GVK_EHIP, "gvk_alias_sample: not part of the host build");
}

int gvk_sample_pairs(void *, const gvk_alias_entry *table, const uint32_t *block_pairs, uint32_t count, uint64_t seed,
                     uint64_t first_index, uint32_t *pool, size_t n) {
    std::vector<float> prob;
    std::vector<uint32_t> alias;
    unpack(table, count, prob, alias);
    gvo_sample_pairs(prob.data(), alias.data(), block_pairs, count, seed, first_index, n, pool);
    return GVK_OK;
}

int gvk_sample_edges(void *, const gvk_edge_entry *table, uint32_t count, uint64_t seed, uint64_t first_index, uint32_t *pool, size_t n) {
    std::vector<float> prob(count);
    std::vector<uint32_t> alias(count), pairs((size_t)count * 2);
    for (uint32_t i = 0; i < count; i++)
        prob[i] = table[i].prob, alias[i] = table[i].alias, pairs[2 * (size_t)i] = table[i].tail, pairs[2 * (size_t)i + 1] = table[i].head;
    gvo_sample_pairs(prob.data(), alias.data(), pairs.data(), count, seed, first_index, n, pool);
    return GVK_OK;
}

int gvk_sample_walks(void *, const gvk_walk_graph *g, uint64_t seed, uint64_t first_walk, uint32_t *pool, size_t pool_pairs,
                     int walk_length, int augmentation_step, int shuffle_base) {
    std::vector<float> ep, np;
    std::vector<uint32_t> ea, na;
    unpack(g->edge_table, g->num_edge_entries, ep, ea);
    unpack(g->neighbor_table, g->num_edge_entries, np, na);
    return gvo_sample_walks_device(g->flat_offsets, g->edges_uv, ep.data(), ea.data(), g->num_edge_entries, np.data(), na.data(),
                                   g->sorted_neighbors, g->local, g->biased, g->p, g->q, seed, first_walk, pool, pool_pairs,
                                   walk_length, augmentation_step, shuffle_base) == 0
               ? GVK_OK
               : gvk_fail(GVK_EINVAL, "gvk_sample_walks: bad arguments");
}

// gvk_sample_walks_blocks restated: the walks of gvk_sample_walks (vertex ids instead of rows), every pair binned into the
// pool of its (head partition, tail partition) block; pair i of a walk to stripe ((walk / 64) + (i % sb) * (stripes / sb)) % stripes, in walk order.
static float thinning_uniform(uint64_t walk, uint64_t i, uint64_t seed) {  // gvk.h gvk_sample_walks_blocks_thinned
    uint32_t h = (uint32_t)walk ^ (uint32_t)(walk >> 32) * 0x85ebca6bu ^ (uint32_t)i * 0x9e3779b9u ^ (uint32_t)seed * 0xc2b2ae35u;
    h ^= h >> 16, h *= 0x85ebca6bu, h ^= h >> 13, h *= 0xc2b2ae35u, h ^= h >> 16;
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

int gvk_sample_walks_blocks(void *stream, const gvk_walk_graph *g, const int32_t *part, int P, uint64_t seed, uint64_t first_walk,
                            uint64_t num_walks, uint32_t *pools, const uint64_t *offsets, uint32_t *counters, uint32_t capacity,
                            int num_stripe, int walk_length, int aug, int shuffle_base) {
    return gvk_sample_walks_blocks_thinned(stream, g, part, P, seed, first_walk, num_walks, pools, offsets, counters, capacity, num_stripe, walk_length, aug,
                                           shuffle_base, nullptr);
}

int gvk_sample_walks_blocks_thinned(void *, const gvk_walk_graph *g, const int32_t *part, int P, uint64_t seed, uint64_t first_walk,
                                    uint64_t num_walks, uint32_t *pools, const uint64_t *offsets, uint32_t *counters, uint32_t capacity,
                                    int num_stripe, int walk_length, int aug, int shuffle_base, const float *accept) {
    if (capacity % (uint32_t)num_stripe || capacity % (uint32_t)shuffle_base)
        return gvk_fail(GVK_EINVAL, "gvk_sample_walks_blocks: stripes / shuffle base must divide the pool size");
    std::vector<float> ep, np;
    std::vector<uint32_t> ea, na, identity(g->num_vertex);
    unpack(g->edge_table, g->num_edge_entries, ep, ea);
    unpack(g->neighbor_table, g->num_edge_entries, np, na);
    for (uint32_t v = 0; v < g->num_vertex; v++) identity[v] = v;
    const uint64_t per_walk = (uint64_t)aug * walk_length - (uint64_t)aug * (aug - 1) / 2;
    const uint32_t stripe_capacity = capacity / (uint32_t)num_stripe, sb = (uint32_t)shuffle_base;
    const uint32_t apart = (uint32_t)num_stripe / sb > 0 ? (uint32_t)num_stripe / sb : 1;
    std::vector<uint32_t> pairs(per_walk * 2);
    for (uint64_t t = 0; t < num_walks; t++) {
        if (gvo_sample_walks_device(g->flat_offsets, g->edges_uv, ep.data(), ea.data(), g->num_edge_entries, np.data(), na.data(),
                                    g->sorted_neighbors, identity.data(), g->biased, g->p, g->q, seed, first_walk + t, pairs.data(),
                                    per_walk, walk_length, aug, 1) != 0)
            return gvk_fail(GVK_EINVAL, "gvk_sample_walks_blocks: bad arguments");
        const uint32_t wave = (uint32_t)((t / 64) % (uint64_t)num_stripe);
        for (uint64_t i = 0; i < per_walk; i++) {
            const uint32_t stripe = (wave + (uint32_t)(i % sb) * apart) % (uint32_t)num_stripe;
            const uint32_t tail = pairs[2 * i], head = pairs[2 * i + 1];
            const int block = part[head] * P + part[tail];
            const uint64_t first = offsets[block];
            if (first == ~(uint64_t)0) continue;
            if (accept && thinning_uniform(first_walk + t, i, seed) >= accept[block]) continue;
            const uint32_t slot = counters[(size_t)block * num_stripe + stripe]++;
            if (slot >= stripe_capacity) continue;
            uint32_t *record = pools + 2 * (first + ((size_t)stripe * stripe_capacity + slot));
            record[0] = g->local[tail], record[1] = g->local[head];
        }
    }
    return GVK_OK;
}

int gvk_spread_pairs(void *, const uint32_t *pool_in, uint32_t *pool_out, size_t num_pair, int units) {
    if (units < 1 || num_pair % (size_t)units || !pool_in || !pool_out || pool_in == pool_out)
        return gvk_fail(GVK_EINVAL, "gvk_spread_pairs: bad argument");
    const size_t per = num_pair / (size_t)units;
    for (size_t i = 0; i < num_pair; i++) {
        const size_t to = (i % (size_t)units) * per + i / (size_t)units;
        pool_out[2 * to] = pool_in[2 * i], pool_out[2 * to + 1] = pool_in[2 * i + 1];
    }
    return GVK_OK;
}

// per batch: a stable sort of the records by the low row bits of the head (what gvk_group_pairs promises)
int gvk_group_pairs(void *, const uint32_t *pool_in, uint32_t *pool_out, void *workspace, size_t *workspace_bytes, int batch_size,
                    int num_batch, int row_bits) {
    if (!workspace_bytes) return gvk_fail(GVK_EINVAL, "gvk_group_pairs: workspace_bytes is null");
    if (!workspace) {
        *workspace_bytes = 16;
        return GVK_OK;
    }
    const uint32_t mask = row_bits >= 32 ? 0xffffffffu : ((1u << row_bits) - 1);
    std::vector<uint32_t> order((size_t)batch_size);
    for (int b = 0; b < num_batch; b++) {
        const uint32_t *in = pool_in + (size_t)b * batch_size * 2;
        uint32_t *out = pool_out + (size_t)b * batch_size * 2;
        for (int i = 0; i < batch_size; i++) order[i] = (uint32_t)i;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return (in[2 * x + 1] & mask) < (in[2 * y + 1] & mask); });
        std::vector<uint32_t> sorted((size_t)batch_size * 2);
        for (int i = 0; i < batch_size; i++) sorted[2 * i] = in[2 * order[i]], sorted[2 * i + 1] = in[2 * order[i] + 1];
        memcpy(out, sorted.data(), sorted.size() * 4);
    }
    return GVK_OK;
}

int gvk_probe_row_traffic(void *, int, float *, float *, const uint32_t *, const uint32_t *, float, int) { return GVK_OK; }

int gvk_set_tuning(int key, int value) {
    if (key == GVK_TUNE_SPLIT_HITS) g_split_hits = value;
    return GVK_OK;
}

int gvk_train_launches(int batch_size, uint32_t rows) {  // the product's rule (gvk_pairs.hip launches_for)
    if (g_split_hits <= 0 || rows == 0 || batch_size <= 0) return 1;
    const int64_t per_launch = (int64_t)rows * g_split_hits, want = ((int64_t)batch_size + per_launch - 1) / per_launch;
    if (want <= 1) return 1;
    for (int64_t q = want; q <= batch_size && q <= 8 * want; q++)
        if (batch_size % q == 0) return (int)q;
    for (int64_t q = want; q > 1; q--)
        if (batch_size % q == 0) return (int)q;
    return 1;
}

int gvk_has_ab_builds(void) { return 0; }

int gvk_describe_train(int dim, int, int num_negative, int, int, uint32_t, char *name, size_t capacity) {
    snprintf(name, capacity, "host build: the sequential CPU oracle (dim %d, %d negative(s))", dim, num_negative);
    return GVK_OK;
}

}  // extern "C"
