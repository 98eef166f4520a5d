// gvk_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the node-embedding hot path and
// their C-ABI launchers (include/gvk.h).  Written for MI355X only: no CUDA dual path.
//
// Kernel shape.  The path is a skinny gather-dot-scatter at ~1 flop/byte, so the only
// roofline is HBM and the design goal is "as many independent 512-byte row requests in flight as
// possible, every request a full-line coalesced burst, nothing read twice":
//   * a group of G lanes (G = 16 at dim 128) owns one {tail, head} pair; a wavefront carries 64/G
//     pairs.  Each lane holds dim/G consecutive-by-chunk floats of every row in VGPRs (float4 chunks:
//     16 B per lane per request, G*16 B contiguous per row request).  Rows never touch LDS — there is
//     no cross-lane reuse to stage for; the "vertex buffer" of the reference's kernel
//     (include/instance/gpu/graph.cuh:51,59,93) is simply the lane's registers.
//   * the negative draw is fused: Philox4x32-10 keyed by (seed; sample, batch, j) + one 8-byte alias
//     entry load (include/gvk.h "RNG contract"), issued concurrently with the pair load.
//   * all rows of a pair (vertex, negatives, positive) are requested before the first is used
//     (one-ahead prefetch of the next target row), so a pair costs two dependent HBM round trips.
//   * the dot product is reduced with DPP butterflies inside a 16-lane row (quad_perm, row_half_mirror,
//     row_mirror): 4 VALU adds, no LDS, every lane ends with the same sum (no broadcast step).
//   * updates are Hogwild exactly like the reference: plain stores, no atomics.
//
// Arithmetic follows include/instance/model/graph.h:40-85 and include/core/optimizer.h:161-210 term by
// term; only the summation order of the dot product differs (lane partials + butterfly).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <cmath>

#include "gvk.h"
#include "gvk_internal.h"

namespace {

constexpr float kEpsilon = 1e-15f;  // include/util/common.h:28
constexpr int kBlock = 256;         // 4 wavefronts; no LDS, no barrier -> block size only sets dispatch granularity

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

int g_variant = 0;         // GVK_TUNE_VARIANT
int g_run_cap = 0;         // GVK_TUNE_RUN_CAP (0 = the default of 20, run_cap_for)
int g_split_hits = 2;      // GVK_TUNE_SPLIT_HITS (samples per table row one launch may hold; 0 = never split a batch)
int g_hot_order = 1;       // GVK_TUNE_HOT_ORDER (measurement: which blocks of a train_hot_kernel launch come first; 1 = long chains, pairs, the other chains)
int g_hot_serialized = 0;  // GVK_TUNE_HOT_SERIALIZED (measurement: gvk_train_episode_hot launches the chains and the pairs of a unit one after the other)
int g_chain_cap = 0;       // GVK_TUNE_CHAIN_CAP (entries one chain task trains in sequence; 0 = the default of 7)
#if defined(GVK_AB_BUILDS)  // knobs of the A/B library only (make ab -> build/ab/libgvk_ab.so)
int g_lanes_per_pair = 0;  // GVK_TUNE_LANES_PER_PAIR
int g_generation = 0;      // GVK_TUNE_GENERATION (0 = one launch per batch)
int g_segment_steps = 0;   // GVK_TUNE_SEGMENT_STEPS (0 = off)
int g_skip_loss = 1;       // GVK_TUNE_SKIP_LOSS (gvk_train_episode leaves out the loss of batches nobody can read)
#else
constexpr int g_lanes_per_pair = 0, g_generation = 0, g_segment_steps = 0, g_skip_loss = 1;
#endif

struct TrainArgs {
    float *vertex, *context, *vm1, *cm1, *vm2, *cm2;
    const uint32_t *pairs;
    const uint32_t *negatives;
    const gvk_alias_entry *table;
    const gvk_class_entry *classes;  // non-null: negatives are drawn by weight class (count = number of classes)
    float *loss;
    uint64_t seed;
    uint32_t count, batch_id;
    int batch_size, k;
    int run_cap;  // train_runs_kernel: longest run of adjacent same-head pairs one lane group trains in sequence
    int first_sample;  // this launch trains samples [first_sample, batch_size) of the batch (a batch split over several launches)
    uint32_t hot_vertex, hot_context;  // HOT builds: head / context rows below these local ids belong to chains (train_hot_kernel) and are not stored here
    const float *hub_now;     // HOT builds: the hub rows [hot_vertex + hot_context][dim] (head rows first) as the chains of the pairs' unit left them
    const float *hub_before;  // HOT == 2: ... and as those chains found them (a sample reads a hub row on the straight line between the two)
    float hub_step;           // HOT == 2: 1 / samples of the unit
    float lr, wd, neg_weight, hp0, hp1, eps;
};

// ---- cross-lane ------------------------------------------------------------------------------

template <int CTRL>
__device__ __forceinline__ float dpp(float x) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}

// Sum over the G lanes of a group; every lane of the group returns the same value.
template <int G>
__device__ __forceinline__ float group_sum(float x) {
    if (G >= 2) x += dpp<0xB1>(x);   // quad_perm [1,0,3,2]
    if (G >= 4) x += dpp<0x4E>(x);   // quad_perm [2,3,0,1]
    if (G >= 8) x += dpp<0x141>(x);  // row_half_mirror: quad <-> other quad of the 8
    if (G >= 16) x += dpp<0x140>(x); // row_mirror: 8 <-> other 8 of the 16-lane DPP row
    if (G >= 32) x += __shfl_xor(x, 16);
    if (G >= 64) x += __shfl_xor(x, 32);
    return x;
}

// ---- Philox4x32-10 ----------------------------------------------------------------------------

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

constexpr uint32_t kTagNegative = 0x6e656721u;

constexpr uint32_t kTagNegativeClass = 0x6e656743u;

struct Draw {
    uint32_t index;
    float u;
    uint32_t v;  // class draws: the word that picks the row inside the class
};

__device__ __forceinline__ Draw negative_slot(uint64_t seed, uint32_t batch_id, uint32_t sample, uint32_t j,
                                              uint32_t count) {
    uint32_t w[4];
    philox4x32_10(sample, batch_id, j >> 1, kTagNegative, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    uint32_t wa = (j & 1) ? w[2] : w[0], wb = (j & 1) ? w[3] : w[1];
    Draw d;
    d.index = __umulhi(wa, count);
    d.u = (float)(wb >> 8) * (1.0f / 16777216.0f);
    return d;
}

__device__ __forceinline__ uint32_t resolve(const Draw &d, const gvk_alias_entry &e) {
    return d.u < e.prob ? d.index : e.alias;
}

// The negative of (sample, j) inside the training kernels, from whichever structure the caller gave (the branch is
// uniform over the launch).  Row table: one random 8-byte slot of a table as long as the partition (8 MB at 1M rows: a
// memory request per draw).  Class table: rows of equal weight form a class — a few thousand classes, 16 bytes each,
// resident in the caches — and the draw is class (alias method over the classes) then a uniform row of the class.
struct NegEntry {
    uint32_t prob_bits, alias, first, count;
};

__device__ __forceinline__ Draw negative_slot(const TrainArgs &a, uint32_t sample, uint32_t j) {
    if (!a.classes) return negative_slot(a.seed, a.batch_id, sample, j, a.count);
    uint32_t w[4];
    philox4x32_10(sample, a.batch_id, j, kTagNegativeClass, (uint32_t)a.seed, (uint32_t)(a.seed >> 32), w);
    Draw d;
    d.index = __umulhi(w[0], a.count);
    d.u = (float)(w[1] >> 8) * (1.0f / 16777216.0f);
    d.v = w[2];
    return d;
}

__device__ __forceinline__ NegEntry load_entry(const TrainArgs &a, const Draw &d) {
    NegEntry e = {0, 0, 0, 0};
    if (a.classes) {
        const u32x4 x = *reinterpret_cast<const u32x4 *>(a.classes + d.index);
        e.prob_bits = x.x, e.alias = x.y, e.first = x.z, e.count = x.w;
    } else {
        const gvk_alias_entry t = a.table[d.index];
        e.prob_bits = __float_as_uint(t.prob), e.alias = t.alias;
    }
    return e;
}

__device__ __forceinline__ uint32_t resolve(const TrainArgs &a, const Draw &d, const NegEntry &e) {
    const bool self = d.u < __uint_as_float(e.prob_bits);
    if (!a.classes) return self ? d.index : e.alias;
    uint32_t first = e.first, count = e.count;
    if (!self) {
        const u32x2 other = *reinterpret_cast<const u32x2 *>(&a.classes[e.alias].first);
        first = other.x, count = other.y;
    }
    return first + __umulhi(d.v, count);
}

// ---- rows in registers ---------------------------------------------------------------------------

template <int DIM, int G>
struct Layout {
    static constexpr int V = DIM / G;                       // floats per lane
    static constexpr int CW = (V % 4 == 0) ? 4 : (V % 2 == 0 ? 2 : 1);  // floats per request
    static constexpr int NC = V / CW;                       // requests per row per lane
    static_assert(DIM % G == 0, "dim must split over the lane group");
};

// the row that starts at `base`
template <int DIM, int G>
__device__ __forceinline__ void load_row_at(const float *base, int lane, float (&r)[DIM / G]) {
    typedef Layout<DIM, G> L;
    const float *row = base + lane * L::CW;
#pragma unroll
    for (int c = 0; c < L::NC; c++) {
        const float *p = row + c * G * L::CW;
        if (L::CW == 4) {
#if defined(GVK_AB_BUILDS) && defined(GVK_EXPERIMENT_NT_ROWS)  // A/B build only (scripts/experiments/gpu_r2_nt.sh): rows marked streaming in the caches
            f32x4 x = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
#else
            f32x4 x = *reinterpret_cast<const f32x4 *>(p);
#endif
            r[c * 4 + 0] = x.x; r[c * 4 + 1] = x.y; r[c * 4 + 2] = x.z; r[c * 4 + 3] = x.w;
        } else if (L::CW == 2) {
            f32x2 x = *reinterpret_cast<const f32x2 *>(p);
            r[c * 2 + 0] = x.x; r[c * 2 + 1] = x.y;
        } else {
            r[c] = *p;
        }
    }
}

template <int DIM, int G>
__device__ __forceinline__ void load_row(const float *table, uint32_t id, int lane, float (&r)[DIM / G]) {
    load_row_at<DIM, G>(table + (size_t)id * DIM, lane, r);
}

template <int DIM, int G>
__device__ __forceinline__ void store_row_at(float *base, int lane, const float (&r)[DIM / G]) {
    typedef Layout<DIM, G> L;
    float *row = base + lane * L::CW;
#pragma unroll
    for (int c = 0; c < L::NC; c++) {
        float *p = row + c * G * L::CW;
        if (L::CW == 4) {
            f32x4 x = {r[c * 4 + 0], r[c * 4 + 1], r[c * 4 + 2], r[c * 4 + 3]};
#if defined(GVK_AB_BUILDS) && defined(GVK_EXPERIMENT_NT_ROWS) && GVK_EXPERIMENT_NT_ROWS >= 2
            __builtin_nontemporal_store(x, reinterpret_cast<f32x4 *>(p));
#else
            *reinterpret_cast<f32x4 *>(p) = x;
#endif
        } else if (L::CW == 2) {
            f32x2 x = {r[c * 2 + 0], r[c * 2 + 1]};
            *reinterpret_cast<f32x2 *>(p) = x;
        } else {
            *p = r[c];
        }
    }
}

template <int DIM, int G>
__device__ __forceinline__ void store_row(float *table, uint32_t id, int lane, const float (&r)[DIM / G]) {
    store_row_at<DIM, G>(table + (size_t)id * DIM, lane, r);
}

template <int N>
__device__ __forceinline__ void copy_row(float (&dst)[N], const float (&src)[N]) {
#pragma unroll
    for (int i = 0; i < N; i++) dst[i] = src[i];
}

// ---- arithmetic (include/util/math.h:30-33, include/core/optimizer.h:161-210) -------------------

// x > 0 ? 1 / (1 + expf(-x)) : expf(x) / (expf(x) + 1), the same bits with one exponential and one division: both branches
// evaluate expf(-|x|)
__device__ __forceinline__ float sigmoidf(float x) {
    const float t = expf(-fabsf(x));
    return (x > 0 ? 1.0f : t) / (1 + t);
}

template <int OPT>
__device__ __forceinline__ float update(const TrainArgs &a, float parameter, float gradient, float weight,
                                        float &m1, float &m2) {
    if (OPT == GVK_SGD) return a.lr * weight * (gradient + a.wd * parameter);
    float regularized = weight * (gradient + a.wd * parameter);
    if (OPT == GVK_MOMENTUM) {
        m1 = a.hp0 * m1 + (1 - a.hp0) * regularized;
        return a.lr * m1;
    }
    if (OPT == GVK_ADAGRAD) {
        m1 += regularized * regularized;
        return a.lr * regularized / (sqrtf(m1) + a.eps);
    }
    if (OPT == GVK_RMSPROP) {
        m1 = a.hp0 * m1 + (1 - a.hp0) * regularized * regularized;
        return a.lr * regularized / sqrtf(m1 + a.eps);
    }
    m1 = a.hp0 * m1 + (1 - a.hp0) * regularized;
    m2 = a.hp1 * m2 + (1 - a.hp1) * regularized * regularized;
    return a.lr * m1 / (sqrtf(m2) + a.eps);
}

// ---- training kernel -------------------------------------------------------------------------------

// KT > 0 fixes num_negative at compile time (the loop unrolls and every row request of the pair is issued
// up front); DRAW fixes the negative source (-1: decided at run time).  WAVES is the occupancy the register
// allocator is asked for (waves per SIMD).
// HOT: the hub rows of both tables — local ids below a.hot_vertex / a.hot_context; partitions are ordered by falling
// degree — belong to the chains of train_hot_kernel: this body reads them and never stores them.
template <int DIM, int G, int OPT, int KT, int DRAW, int HOT>
__device__ __forceinline__ void train_pair(const TrainArgs &a, const int tid) {
    constexpr int V = DIM / G;
    constexpr int NM = OPT == GVK_SGD ? 0 : (OPT == GVK_ADAM ? 2 : 1);  // moments per row
    constexpr int M1 = NM >= 1 ? V : 1, M2 = NM >= 2 ? V : 1;

    const int s = a.first_sample + tid / G, lane = tid % G;
    if (s >= a.batch_size) return;  // whole groups leave together: G divides 64

    const int k = KT > 0 ? KT : a.k;
    const bool draw = DRAW < 0 ? a.negatives == nullptr : DRAW != 0;

    // round trip 1: the pair, requested before anything is computed, and the first negative's alias slot (Philox runs
    // while the pair is on its way)
    const u32x2 pr = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(a.pairs) + s);
    Draw d0 = {0, 0, 0};
    NegEntry e0 = {0, 0, 0, 0};
    uint32_t neg0 = 0;
    if (k > 0) {
        if (draw) {
            d0 = negative_slot(a, (uint32_t)s, 0);
            e0 = load_entry(a, d0);
        } else {
            neg0 = __builtin_nontemporal_load(a.negatives + (size_t)s * k);
        }
    }
    const uint32_t tail = pr.x, head = pr.y;  // records are {tail, head}

    // HOT: a hub row is read from the mirror the chains of the unit stored it to (the table's copy is written once, when
    // the call ends), every other row from its table
    auto vertex_row = [&](const uint32_t id) __attribute__((always_inline)) -> const float * {
        if (HOT != 0 && id < a.hot_vertex) return a.hub_now + (size_t)id * DIM;
        return a.vertex + (size_t)id * DIM;
    };
    auto context_row = [&](const uint32_t id) __attribute__((always_inline)) -> const float * {
        if (HOT != 0 && id < a.hot_context) return a.hub_now + ((size_t)a.hot_vertex + id) * DIM;
        return a.context + (size_t)id * DIM;
    };

    // round trip 2: vertex row (+ moments) and the first target row.  With one negative (KT == 1) the positive's row
    // is requested here as well, ahead of the negative's: both ids of the pair are known, the negative's still waits
    // for its alias entry.
    // HOT == 2 (lerp): a hub row is read where its chain was when it met the sample — on the straight line from the row as
    // the unit's chains found it (hub_before) to the row as they left it (hub_now), at the sample's place in the unit.  The
    // second row is requested together with the first and the two are combined when the row is first used.
    constexpr int VB = HOT == 2 ? V : 1;
    const float at = HOT == 2 ? ((float)(s - a.first_sample) + 0.5f) * a.hub_step : 0.0f;
    auto before_row = [&](const bool is_hub, const size_t slot, float (&b)[VB]) __attribute__((always_inline)) {
        if constexpr (HOT == 2) {
            if (is_hub) load_row_at<DIM, G>(a.hub_before + slot * DIM, lane, reinterpret_cast<float(&)[V]>(b));
        }
    };
    auto on_the_way = [&](const bool is_hub, float (&r)[V], const float (&b)[VB]) __attribute__((always_inline)) {
        if constexpr (HOT == 2) {
            if (is_hub) {
#pragma unroll
                for (int i = 0; i < V; i++) r[i] = b[i] + at * (r[i] - b[i]);
            }
        }
    };
    float v[V], vm1[M1], vm2[M2], v_before[VB];
    const bool v_hub = HOT == 2 && head < a.hot_vertex;
    load_row_at<DIM, G>(vertex_row(head), lane, v);
    before_row(v_hub, head, v_before);
    if constexpr (NM >= 1) load_row<DIM, G>(a.vm1, head, lane, reinterpret_cast<float(&)[V]>(vm1));
    if constexpr (NM >= 2) load_row<DIM, G>(a.vm2, head, lane, reinterpret_cast<float(&)[V]>(vm2));
    constexpr bool kTailEarly = KT == 1 && NM == 0;
    float early[kTailEarly ? V : 1], early_before[kTailEarly ? VB : 1];
    const bool early_hub = HOT == 2 && kTailEarly && tail < a.hot_context;
    if constexpr (kTailEarly) {
        load_row_at<DIM, G>(context_row(tail), lane, reinterpret_cast<float(&)[V]>(early));
        before_row(early_hub, (size_t)a.hot_vertex + tail, reinterpret_cast<float(&)[VB]>(early_before));
    }

    uint32_t id_cur = k > 0 ? (draw ? resolve(a, d0, e0) : neg0) : tail;
    float cur[V], cur1[M1], cur2[M2], cur_before[VB];
    bool cur_hub = HOT == 2 && id_cur < a.hot_context;
    load_row_at<DIM, G>(context_row(id_cur), lane, cur);
    before_row(cur_hub, (size_t)a.hot_vertex + id_cur, cur_before);
    if constexpr (NM >= 1) load_row<DIM, G>(a.cm1, id_cur, lane, reinterpret_cast<float(&)[V]>(cur1));
    if constexpr (NM >= 2) load_row<DIM, G>(a.cm2, id_cur, lane, reinterpret_cast<float(&)[V]>(cur2));

    float sample_loss = 0;
    auto target_step = [&](const int j) __attribute__((always_inline)) {
        // request the next target row before touching the current one
        uint32_t id_nxt = 0;
        float nxt[V], nxt1[M1], nxt2[M2], nxt_before[VB];
        bool nxt_hub = false;
        if (j < k) {
            if (j + 1 < k) {
                if (draw) {
                    Draw d = negative_slot(a, (uint32_t)s, (uint32_t)(j + 1));
                    id_nxt = resolve(a, d, load_entry(a, d));
                } else {
                    id_nxt = __builtin_nontemporal_load(a.negatives + (size_t)s * k + j + 1);
                }
            } else {
                id_nxt = tail;
            }
            if constexpr (kTailEarly) {
                copy_row(nxt, reinterpret_cast<float(&)[V]>(early));  // requested before the negative's row
                copy_row(nxt_before, reinterpret_cast<float(&)[VB]>(early_before));
                nxt_hub = early_hub;
            } else {
                nxt_hub = HOT == 2 && id_nxt < a.hot_context;
                load_row_at<DIM, G>(context_row(id_nxt), lane, nxt);
                before_row(nxt_hub, (size_t)a.hot_vertex + id_nxt, nxt_before);
                if constexpr (NM >= 1) load_row<DIM, G>(a.cm1, id_nxt, lane, reinterpret_cast<float(&)[V]>(nxt1));
                if constexpr (NM >= 2) load_row<DIM, G>(a.cm2, id_nxt, lane, reinterpret_cast<float(&)[V]>(nxt2));
            }
        }
        if (j == 0) on_the_way(v_hub, v, v_before);
        on_the_way(cur_hub, cur, cur_before);

        // forward: model/graph.h:40-45
        float partial = 0;
#pragma unroll
        for (int i = 0; i < V; i++) partial += v[i] * cur[i];
        const float logit = group_sum<G>(partial);
        const float prob = sigmoidf(logit);
        // gpu/graph.cuh:77-87
        float gradient, weight;
        if (j == k) {
            gradient = prob - 1;
            weight = 1;
            sample_loss += weight * -logf(prob + kEpsilon);
        } else {
            gradient = prob;
            weight = a.neg_weight;
            sample_loss += weight * -logf(1 - prob + kEpsilon);
        }
        // backward: model/graph.h:47-58 — both updates use the pre-update v and c
#pragma unroll
        for (int i = 0; i < V; i++) {
            const float vi = v[i], ci = cur[i];
            v[i] -= update<OPT>(a, vi, gradient * ci, weight, vm1[NM >= 1 ? i : 0], vm2[NM >= 2 ? i : 0]);
            cur[i] -= update<OPT>(a, ci, gradient * vi, weight, cur1[NM >= 1 ? i : 0], cur2[NM >= 2 ? i : 0]);
        }
        if (HOT == 0 || id_cur >= a.hot_context) store_row<DIM, G>(a.context, id_cur, lane, cur);
        if constexpr (NM >= 1) store_row<DIM, G>(a.cm1, id_cur, lane, reinterpret_cast<float(&)[V]>(cur1));
        if constexpr (NM >= 2) store_row<DIM, G>(a.cm2, id_cur, lane, reinterpret_cast<float(&)[V]>(cur2));

        if (j < k) {
            // The next row was requested before this one was updated. If it is the same row (a negative
            // equal to the next negative / to the positive tail), carry the updated registers forward so
            // the pair sees its own update, as the reference's sequential warp does.
            const bool same = id_nxt == id_cur;
#pragma unroll
            for (int i = 0; i < V; i++) cur[i] = same ? cur[i] : nxt[i];
            if constexpr (HOT == 2) {
                copy_row(cur_before, nxt_before);
                cur_hub = !same && nxt_hub;  // the same row again: the registers already hold the row on its way, updated
            }
            if constexpr (NM >= 1) {
#pragma unroll
                for (int i = 0; i < V; i++) cur1[i] = same ? cur1[i] : nxt1[i];
            }
            if constexpr (NM >= 2) {
#pragma unroll
                for (int i = 0; i < V; i++) cur2[i] = same ? cur2[i] : nxt2[i];
            }
            id_cur = id_nxt;
        }
    };
    if constexpr (KT > 0) {
#pragma unroll
        for (int j = 0; j <= KT; j++) target_step(j);
    } else {
        for (int j = 0; j <= k; j++) target_step(j);
    }

    if (lane == 0) __builtin_nontemporal_store(sample_loss / (1 + k * a.neg_weight), a.loss + s);
    if (HOT == 0 || head >= a.hot_vertex) store_row<DIM, G>(a.vertex, head, lane, v);
    if constexpr (NM >= 1) store_row<DIM, G>(a.vm1, head, lane, reinterpret_cast<float(&)[V]>(vm1));
    if constexpr (NM >= 2) store_row<DIM, G>(a.vm2, head, lane, reinterpret_cast<float(&)[V]>(vm2));
}

// Waves per SIMD the training kernels are built for: four (128 registers) wherever the rows a lane group holds at once — the
// head row, the current and the next target row, each with its moment rows — fit; the moment optimizers at 12 or 16 floats per
// lane do not (76 .. 456 bytes of scratch per lane at four): three waves (168 registers), Adam at 16 floats per lane two (256).
constexpr int train_waves(int v, int opt, bool runs) {
    const int m = opt == GVK_SGD ? 0 : (opt == GVK_ADAM ? 2 : 1);
    if (m == 0) return 4;
    if (v >= 16) return m == 2 ? 2 : 3;
    if (v >= 12) return m == 2 ? (runs ? 2 : 3) : (runs ? 3 : 4);
    if (v >= 8) return m == 2 ? 3 : 4;
    return 4;
}

template <int DIM, int G, int OPT, int KT = 0, int DRAW = -1, int WAVES = train_waves(DIM / G, OPT, false)>
__global__ void __launch_bounds__(kBlock, WAVES) train_kernel(const TrainArgs a) {
    train_pair<DIM, G, OPT, KT, DRAW, 0>(a, blockIdx.x * kBlock + threadIdx.x);
}

// ---- training kernel, runs of same-head pairs ------------------------------------------------------------------
//
// The shipped kernel.  Same lane layout, loads, arithmetic and negative draw as train_kernel above (which stays as
// the per-pair A/B build, GVK_TUNE_VARIANT 2); the unit of work of a lane group is a RUN instead of a pair: the pairs
// j = s, s + 1, ... that sit next to each other in the batch, share the head row of pair s and lie in the same
// run_cap-aligned segment of the batch.  The lane group of the run's first pair keeps the head row in registers over
// the whole run — one load, one store, every pair of the run sees the updates of the pairs before it, exactly like
// consecutive iterations of one warp's grid-stride loop in the reference (gpu/graph.cuh:54-94) — and the lane groups
// of the other pairs of the run retire at once.  Sample j keeps its own identity: negatives are drawn for (batch, j),
// loss goes to loss[j].
//
// Batches in sampler order have almost no adjacent same-head pairs and behave as before.  After gvk_group_pairs
// every head row of a batch is one or more runs: the row crosses HBM once per run instead of once per pair, and of
// the m pairs of a batch that share a hub row, min(m, run_cap) consecutive updates survive instead of one (the
// remaining ceil(m / run_cap) - 1 lane groups train the same row concurrently from the same start; the last store wins, as it
// does between any two concurrent warps of the reference).  run_cap = 20 is how many times the reference's
// <<<8192, 512>>> launch refills a V100 (5120 resident warps) within one default batch, i.e. how many generations of
// updates to one row that launch can chain (run_cap_for).
//
// Pipelining inside a run: the header of pair j + 1 is loaded one pair ahead, its first alias slot as soon as the
// header says the run continues, and its first target row while the positive target of pair j is computed — the
// one-ahead row prefetch of the per-pair kernel carried across pairs.
template <int DIM, int G, int OPT, int KT = 0, int DRAW = -1, int WAVES = train_waves(DIM / G, OPT, true)>
__global__ void __launch_bounds__(kBlock, WAVES) train_runs_kernel(const TrainArgs a) {
    constexpr int V = DIM / G;
    constexpr int NM = OPT == GVK_SGD ? 0 : (OPT == GVK_ADAM ? 2 : 1);  // moments per row
    constexpr int M1 = NM >= 1 ? V : 1, M2 = NM >= 2 ? V : 1;

    const int tid = blockIdx.x * kBlock + threadIdx.x;
    const int s = a.first_sample + tid / G, lane = tid % G;
    if (s >= a.batch_size) return;  // whole groups leave together: G divides 64

    const int k = KT > 0 ? KT : a.k;
    const bool draw = DRAW < 0 ? a.negatives == nullptr : DRAW != 0;
    const int R = a.run_cap;
    const u32x2 *records = reinterpret_cast<const u32x2 *>(a.pairs);

    // round trip 1: the pair, its neighbours in the segment, and the first negative's alias slot
    Draw d0 = {0, 0, 0};
    NegEntry e0 = {0, 0, 0, 0};
    uint32_t neg0 = 0;
    if (k > 0) {
        if (draw) {
            d0 = negative_slot(a, (uint32_t)s, 0);
            e0 = load_entry(a, d0);
        } else {
            neg0 = __builtin_nontemporal_load(a.negatives + (size_t)s * k);
        }
    }
    const u32x2 pr = records[s];
    uint32_t tail = pr.x;
    const uint32_t head = pr.y;  // records are {tail, head}
    // runs stay inside their segment: run_cap samples, counted from the first sample of this launch
    const int first_of_segment = s - (s - a.first_sample) % R;
    const int limit = first_of_segment + R < a.batch_size ? first_of_segment + R : a.batch_size;
    u32x2 next_pr = {0, 0};
    if (s + 1 < limit) next_pr = records[s + 1];
    if (s > first_of_segment && records[s - 1].y == head) return;  // this pair belongs to the run of a pair before it

    // round trip 2: vertex row (+ moments) and the first target row
    float v[V], vm1[M1], vm2[M2];
    load_row<DIM, G>(a.vertex, head, lane, v);
    if constexpr (NM >= 1) load_row<DIM, G>(a.vm1, head, lane, reinterpret_cast<float(&)[V]>(vm1));
    if constexpr (NM >= 2) load_row<DIM, G>(a.vm2, head, lane, reinterpret_cast<float(&)[V]>(vm2));

    uint32_t id_cur = k > 0 ? (draw ? resolve(a, d0, e0) : neg0) : tail;
    float cur[V], cur1[M1], cur2[M2];
    load_row<DIM, G>(a.context, id_cur, lane, cur);
    if constexpr (NM >= 1) load_row<DIM, G>(a.cm1, id_cur, lane, reinterpret_cast<float(&)[V]>(cur1));
    if constexpr (NM >= 2) load_row<DIM, G>(a.cm2, id_cur, lane, reinterpret_cast<float(&)[V]>(cur2));

    int j = s;  // the pair being trained
    while (true) {
        // does the run go on with pair j + 1?  If so its first alias slot and the header of pair j + 2 are requested
        // now; both are back long before the positive step below needs them.
        const bool more = j + 1 < limit && next_pr.y == head;
        Draw dn = {0, 0, 0};
        NegEntry en = {0, 0, 0, 0};
        uint32_t negn = 0;
        u32x2 after_pr = {0, 0};
        if (more) {
            if (k > 0) {
                if (draw) {
                    dn = negative_slot(a, (uint32_t)(j + 1), 0);
                    en = load_entry(a, dn);
                } else {
                    negn = __builtin_nontemporal_load(a.negatives + (size_t)(j + 1) * k);
                }
            }
            if (j + 2 < limit) after_pr = records[j + 2];
        }

        float sample_loss = 0;
        auto target_step = [&](const int t) __attribute__((always_inline)) {
            // request the next target row before touching the current one: the next negative, the positive, or —
            // at the positive — the first target of the next pair of the run
            const bool has_next = t < k || more;
            uint32_t id_nxt = 0;
            float nxt[V], nxt1[M1], nxt2[M2];
            if (has_next) {
                if (t + 1 < k) {
                    if (draw) {
                        Draw d = negative_slot(a, (uint32_t)j, (uint32_t)(t + 1));
                        id_nxt = resolve(a, d, load_entry(a, d));
                    } else {
                        id_nxt = __builtin_nontemporal_load(a.negatives + (size_t)j * k + t + 1);
                    }
                } else if (t < k) {
                    id_nxt = tail;
                } else {
                    id_nxt = k > 0 ? (draw ? resolve(a, dn, en) : negn) : next_pr.x;
                }
                load_row<DIM, G>(a.context, id_nxt, lane, nxt);
                if constexpr (NM >= 1) load_row<DIM, G>(a.cm1, id_nxt, lane, reinterpret_cast<float(&)[V]>(nxt1));
                if constexpr (NM >= 2) load_row<DIM, G>(a.cm2, id_nxt, lane, reinterpret_cast<float(&)[V]>(nxt2));
            }

            // forward: model/graph.h:40-45
            float partial = 0;
#pragma unroll
            for (int i = 0; i < V; i++) partial += v[i] * cur[i];
            const float logit = group_sum<G>(partial);
            const float prob = sigmoidf(logit);
            // gpu/graph.cuh:77-87
            float gradient, weight;
            if (t == k) {
                gradient = prob - 1;
                weight = 1;
                sample_loss += weight * -logf(prob + kEpsilon);
            } else {
                gradient = prob;
                weight = a.neg_weight;
                sample_loss += weight * -logf(1 - prob + kEpsilon);
            }
            // backward: model/graph.h:47-58 — both updates use the pre-update v and c
#pragma unroll
            for (int i = 0; i < V; i++) {
                const float vi = v[i], ci = cur[i];
                v[i] -= update<OPT>(a, vi, gradient * ci, weight, vm1[NM >= 1 ? i : 0], vm2[NM >= 2 ? i : 0]);
                cur[i] -= update<OPT>(a, ci, gradient * vi, weight, cur1[NM >= 1 ? i : 0], cur2[NM >= 2 ? i : 0]);
            }
            store_row<DIM, G>(a.context, id_cur, lane, cur);
            if constexpr (NM >= 1) store_row<DIM, G>(a.cm1, id_cur, lane, reinterpret_cast<float(&)[V]>(cur1));
            if constexpr (NM >= 2) store_row<DIM, G>(a.cm2, id_cur, lane, reinterpret_cast<float(&)[V]>(cur2));

            if (has_next) {
                // The next row was requested before this one was updated.  If it is the same row, carry the updated
                // registers forward so that the run sees its own update, as the reference's sequential warp does.
                const bool same = id_nxt == id_cur;
#pragma unroll
                for (int i = 0; i < V; i++) cur[i] = same ? cur[i] : nxt[i];
                if constexpr (NM >= 1) {
#pragma unroll
                    for (int i = 0; i < V; i++) cur1[i] = same ? cur1[i] : nxt1[i];
                }
                if constexpr (NM >= 2) {
#pragma unroll
                    for (int i = 0; i < V; i++) cur2[i] = same ? cur2[i] : nxt2[i];
                }
                id_cur = id_nxt;
            }
        };
        if constexpr (KT > 0) {
#pragma unroll
            for (int t = 0; t <= KT; t++) target_step(t);
        } else {
            for (int t = 0; t <= k; t++) target_step(t);
        }
        if (lane == 0) __builtin_nontemporal_store(sample_loss / (1 + k * a.neg_weight), a.loss + j);
        if (!more) break;
        j++;
        tail = next_pr.x;
        next_pr = after_pr;
    }

    store_row<DIM, G>(a.vertex, head, lane, v);
    if constexpr (NM >= 1) store_row<DIM, G>(a.vm1, head, lane, reinterpret_cast<float(&)[V]>(vm1));
    if constexpr (NM >= 2) store_row<DIM, G>(a.vm2, head, lane, reinterpret_cast<float(&)[V]>(vm2));
}

// ---- hub rows: chains --------------------------------------------------------------------------------------------------
//
// The samples of a launch run concurrently, and of the updates that hold a row at the same time one survives (Hogwild, as
// between two warps of the reference).  For most rows of a large table that never happens; a HUB row — the top hub of the
// benchmark graph is the head of 1 in 100 samples and the tail of as many — is in flight hundreds of times per launch
// and keeps a handful of its updates, where the reference's CPU solver (and its GPU kernel on the card it was written
// for, far less concurrent) keeps them all: link-prediction AUC 0.650 against 0.668 on the headline shape.
//
// With hub rows, a batch is trained as UNITS (its `parts`), and a unit is two kinds of work.  The hub rows of both tables —
// the first hot_vertex / hot_context local ids; partitions are ordered by falling degree — are each owned by a CHAIN: a
// lane group holds the row in registers and applies every update the unit has for it one after the other (for a head row
// the targets of its samples, negatives first; for a context row the heads it is the tail or the negative of), reading
// the partner rows (D of them in flight) and writing nothing but its own row, once, at the end.  Everything else is the
// per-pair body (train_pair<HOT>): every sample, all arithmetic, but hub rows are only read.  So a hub row has ONE writer
// per unit and loses nothing.
//
// Where hub rows live.  During a call they live in three MIRRORS M[0..2] ([hot_vertex + hot_context][dim], head rows
// first) in the workspace, not in the tables: the chains of unit u read M[(u - 1) % 3] — their own row AND every partner
// that is a hub row itself, so that a sample between two hub rows updates both from the values the unit started with, as
// the reference does (model/graph.h:47-58); two chains that read each other's fresh stores would compound the step they
// share, DESIGN.md §3.1.2 — and store to M[u % 3]; the pairs of unit u read M[u % 3] (and, lerp, M[(u - 1) % 3]).  One
// launch runs the pairs of unit u and, in its first blocks, the chains of unit u + 1: nothing it reads is written by it.
// The tables' hub rows are written once, when the call ends (hub_rows_kernel).
//
// A chain longer than `cap` entries is a LONG chain: a whole workgroup trains it, up to kBlock / G tasks of consecutive
// entries side by side, composed through LDS in task order — deterministic given the work lists, no atomics.  The chains'
// work lists are built by hot_list_kernel.
struct HotArgs {
    const uint32_t *chain_start;  // [chains + 1] offsets of this unit into entries
    const uint32_t *before_start[2];  // the same of the one or two units before it (null: none): which rows the mirror `to` has missed
    const uint32_t *entries;      // partner row | label << 31 (label 1 = positive)
    const uint32_t *long_list;    // [0] = number of long chains (more than cap entries), then from [4] on a record {chain, first entry, entries, -} each
    const uint32_t *short_list;   // [0] = number of chains of 1 .. cap entries, then from [16] on a record of 16 words each: {chain, entries, -, -, the entries themselves}
    const float *from;            // mirror the chains read: own rows and hub partners as the unit finds them
    float *to;                    // mirror the chains store to
    uint32_t chains;              // hot_vertex + hot_context
    uint32_t long_capacity;
    uint32_t cap;                 // entries of one task (at most kShortEntries)
    float lr;                     // learning rate of the chains' batch (the pairs of the same launch may belong to another batch)
    float log2_decay_positive, log2_decay_negative;  // log2(1 - lr wd), log2(1 - lr negative_weight wd): decay of an entry by label
    int order, pair_blocks;       // grid order (0: chains first, 1: long chains, pairs, the other chains, 2: pairs first)
    int long_blocks, short_blocks, copy_blocks;  // grid: [long chains | chains of 1 .. cap entries, kBlock / G per block | rows without entries | pairs]
#if defined(GVK_TIMESTAMPS)  // measurement build (make ts): where the time of a launch goes, eight 100 MHz stamps per workgroup
    unsigned long long *stamps;
#endif
};

// Measurement build only (make -C graphvite_amd/csrc ts -> build/ts/libgvk_ts.so, scripts/experiments/stamps.py): thread 0 of
// every workgroup of a train_hot_kernel launch leaves eight words — its role and the 100 MHz clock at the points of its path.
#if defined(GVK_TIMESTAMPS)
#define GVK_STAMP(h, slot) do { if ((h).stamps && threadIdx.x == 0) (h).stamps[(size_t)blockIdx.x * 8 + (slot)] = (unsigned long long)wall_clock64(); } while (0)
#define GVK_STAMP_VALUE(h, slot, value) do { if ((h).stamps && threadIdx.x == 0) (h).stamps[(size_t)blockIdx.x * 8 + (slot)] = (unsigned long long)(value); } while (0)
#else
#define GVK_STAMP(h, slot) do { } while (0)
#define GVK_STAMP_VALUE(h, slot, value) do { } while (0)
#endif

template <int DIM, int G>
struct ChainShape {
    static constexpr int V = DIM / G;
    static constexpr int D = V <= 4 ? 8 : (V <= 8 ? 4 : 2);  // partner rows in flight per lane group of a long chain's task
    static constexpr int NG = kBlock / G;     // lane groups of a block = most tasks of a long chain
    static_assert(D <= G, "the entry window is two fetches of G entries");
};

// Entries [begin, end) of one chain applied one after the other to `own` (the row of `chain`, in the registers of a lane
// group).  Every lane group of the wavefront comes here together, each with its own chain and range (an empty range: a
// group without work) — the loop runs as long as any group has entries left; a group past its end keeps requesting its own
// mirror row and trains with weight 0.  Every step issues exactly one row request and consumes the one issued D steps
// earlier, with no branch around either, so the wait before a step is "all but the D - 1 youngest" and not "all".
template <int DIM, int G>
__device__ __forceinline__ void chain_steps(const TrainArgs &a, const HotArgs &h, const uint32_t chain, const uint32_t begin,
                                            const uint32_t end, const int lane, float (&own)[DIM / G]) {
    typedef ChainShape<DIM, G> S;
    constexpr int V = S::V, D = S::D;
    const bool is_vertex = chain < a.hot_vertex;
    const float *partner_table = is_vertex ? a.context : a.vertex;
    const uint32_t partner_hot = is_vertex ? a.hot_context : a.hot_vertex;  // partners below this id are hub rows: read from the mirror
    const float *partner_mirror = h.from + (is_vertex ? (size_t)a.hot_vertex * DIM : (size_t)0);
    const float *idle = h.from + (size_t)chain * DIM;
    // the work list, G entries per fetch, two fetches resident: entries [blk, blk + 2 G)
    uint32_t blk = begin;
    uint32_t e_cur = blk + lane < end ? h.entries[blk + lane] : 0;
    uint32_t e_nxt = blk + G + lane < end ? h.entries[blk + G + lane] : 0;
    // entry p of the list (through the window): the row it names (past the end: the group's own mirror row) and its label
    auto row_of = [&](const uint32_t p, uint32_t &label) __attribute__((always_inline)) -> const float * {
        const uint32_t o = p - blk;
        const uint32_t e = (uint32_t)__shfl((int)(o < (uint32_t)G ? e_cur : e_nxt), (int)(o & (G - 1)), G);
        label = e >> 31;
        const uint32_t id = e & 0x7fffffffu;
        const float *row = id < partner_hot ? partner_mirror + (size_t)id * DIM : partner_table + (size_t)id * DIM;
        return p < end ? row : idle;
    };
    float ring[D][V];
    uint32_t labels = 0;  // bit i: the label of the entry whose row sits in ring[i]
#pragma unroll
    for (int i = 0; i < D; i++) {
        uint32_t label;
        load_row_at<DIM, G>(row_of(begin + i, label), lane, ring[i]);
        labels |= label << i;
    }
    for (uint32_t base = begin; __builtin_amdgcn_ballot_w64(base < end) != 0; base += D) {
        const uint32_t f = blk + 2 * G + lane;
        const uint32_t e_fut = h.entries[f < end ? f : (begin < end ? end - 1 : 0)];  // the window after e_nxt, asked for ahead of its use
#pragma unroll
        for (int i = 0; i < D; i++) {
            const uint32_t p = base + i;
            const bool positive = (labels >> i & 1u) != 0;
            const float(&c)[V] = ring[i];
            // forward / backward of one target: model/graph.h:40-58, gpu/graph.cuh:77-87 — on the own row only
            float partial = 0;
#pragma unroll
            for (int x = 0; x < V; x++) partial += own[x] * c[x];
            const float prob = sigmoidf(group_sum<G>(partial));
            const float gradient = positive ? prob - 1 : prob;
            const float weight = p < end ? (positive ? 1.0f : a.neg_weight) : 0.0f;
#pragma unroll
            for (int x = 0; x < V; x++) own[x] -= h.lr * weight * (gradient * c[x] + a.wd * own[x]);  // optimizer.h:161-164
            // the slot is free: the row of entry p + D takes it (D - 1 requests stay in flight while a step computes)
            uint32_t label;
            load_row_at<DIM, G>(row_of(p + D, label), lane, ring[i]);
            labels = (labels & ~(1u << i)) | label << i;
        }
        if (base + D >= blk + G) {  // the next steps look beyond e_nxt: move the window
            blk += G;
            e_cur = e_nxt;
            e_nxt = e_fut;
        }
    }
}

// Sum over the lanes of a group of a small count (exact in fp32).
template <int G>
__device__ __forceinline__ float group_count(const uint32_t x) {
    return group_sum<G>((float)x);
}

// Chains of 1 .. cap entries (cap <= kShortEntries = 7), one lane group each: block b trains records [b NG, (b + 1) NG) of the
// unit's short list.  A record carries the chain's entries, so a chain costs two dependent round trips: its record (asked for
// together with the list's length), then its own row and every partner row at once; then at most seven steps.
constexpr int kShortEntries = 7;

// The n <= kShortEntries entries entry_of(0 .. n - 1) of one chain applied one after the other to `own`: every partner row is
// requested before the first step (where the registers hold them: dims up to 128), so the chain waits for memory once.
template <int DIM, int G, class EntryOf>
__device__ __forceinline__ void short_steps(const TrainArgs &a, const HotArgs &h, const uint32_t chain, const uint32_t n, const int lane,
                                            float (&own)[DIM / G], EntryOf entry_of) {
    constexpr int V = DIM / G, N = kShortEntries;
    constexpr int D = V <= 8 ? N : (V <= 12 ? 3 : 2);  // partner rows in flight: all of them where the registers hold them
    const bool is_vertex = chain < a.hot_vertex;
    const float *partner_table = is_vertex ? a.context : a.vertex;
    const uint32_t partner_hot = is_vertex ? a.hot_context : a.hot_vertex;
    const float *partner_mirror = h.from + (is_vertex ? (size_t)a.hot_vertex * DIM : (size_t)0);
    const float *idle = h.from + (size_t)chain * DIM;
    float ring[D][V];
    uint32_t labels = 0;
    auto request = [&](const int i) __attribute__((always_inline)) {  // the row of entry i into its slot of the ring
        const uint32_t e = entry_of(i);
        const uint32_t id = e & 0x7fffffffu;
        const float *row = id < partner_hot ? partner_mirror + (size_t)id * DIM : partner_table + (size_t)id * DIM;
        load_row_at<DIM, G>((uint32_t)i < n ? row : idle, lane, ring[i % D]);
        labels |= (e >> 31) << i;
    };
#pragma unroll
    for (int i = 0; i < D; i++) request(i);
#pragma unroll
    for (int i = 0; i < N; i++) {
        const bool positive = (labels >> i & 1u) != 0;
        const float(&c)[V] = ring[i % D];
        float partial = 0;
#pragma unroll
        for (int x = 0; x < V; x++) partial += own[x] * c[x];
        const float prob = sigmoidf(group_sum<G>(partial));
        const float gradient = positive ? prob - 1 : prob;
        const float weight = (uint32_t)i < n ? (positive ? 1.0f : a.neg_weight) : 0.0f;
#pragma unroll
        for (int x = 0; x < V; x++) own[x] -= h.lr * weight * (gradient * c[x] + a.wd * own[x]);  // optimizer.h:161-164
        if (i + D < N) request(i + D);
    }
}

template <int DIM, int G>
__device__ __forceinline__ void train_short_chains(const TrainArgs &a, const HotArgs &h, const uint32_t block) {
    typedef ChainShape<DIM, G> S;
    constexpr int LW = G < 16 ? G : 16;  // lanes that hold the record's sixteen words
    const int lane = threadIdx.x % G, group = threadIdx.x / G;
    const uint32_t at = block * S::NG + group;
    // the record's sixteen words across the lanes of the group; the list's length arrives with them
    const uint32_t *record = h.short_list + 16 + 16 * (size_t)(at < h.chains ? at : h.chains - 1);
    const uint32_t word0 = record[lane % LW], word1 = LW < 16 ? record[8 + lane % LW] : 0;
    const uint32_t count = h.short_list[0] < h.chains ? h.short_list[0] : h.chains;
    if (block * S::NG >= count) return;  // the whole block at once
    GVK_STAMP_VALUE(h, 0, 2);
    GVK_STAMP(h, 2);  // the record is here
    auto word = [&](const int i) __attribute__((always_inline)) -> uint32_t {
        return (uint32_t)(i < LW ? __shfl((int)word0, i, G) : __shfl((int)word1, i - LW, G));
    };
    const bool mine = at < count;
    const uint32_t chain = mine ? word(0) : 0, n = mine ? word(1) : 0;
    float own[S::V];
    load_row_at<DIM, G>(h.from + (size_t)chain * DIM, lane, own);
    short_steps<DIM, G>(a, h, chain, n, lane, own, [&](const int i) __attribute__((always_inline)) { return word(4 + i); });
    GVK_STAMP(h, 5);  // steps done
    if (mine) store_row_at<DIM, G>(h.to + (size_t)chain * DIM, lane, own);
}

// Hub rows the unit has no entry for pass from mirror to mirror unchanged — those that need it: the mirror `to` was last
// written R units ago (R mirrors in rotation), so a row is behind there only if a chain stored it since, i.e. if it had
// entries in one of the R - 1 units before this one (all mirrors start a call equal).  Block b looks at chains [64 b, 64 b +
// 64), each lane group at four of them (all four rows requested before the first is stored).
template <int DIM, int G>
__device__ __forceinline__ void copy_idle_rows(const HotArgs &h, const uint32_t block) {
    typedef ChainShape<DIM, G> S;
    constexpr int R = 4;
    const int lane = threadIdx.x % G, group = threadIdx.x / G;
    float row[R][S::V];
    bool behind[R];
#pragma unroll
    for (int i = 0; i < R; i++) {
        const uint32_t chain = (block * R + i) * S::NG + group;
        behind[i] = false;
        if (chain < h.chains && h.chain_start[chain] == h.chain_start[chain + 1]) {
#pragma unroll
            for (int v = 0; v < 2; v++)
                if (h.before_start[v]) behind[i] = behind[i] || h.before_start[v][chain] != h.before_start[v][chain + 1];
        }
    }
#pragma unroll
    for (int i = 0; i < R; i++) {
        const uint32_t chain = (block * R + i) * S::NG + group;
        if (behind[i]) load_row_at<DIM, G>(h.from + (size_t)chain * DIM, lane, row[i]);
    }
#pragma unroll
    for (int i = 0; i < R; i++) {
        const uint32_t chain = (block * R + i) * S::NG + group;
        if (behind[i]) store_row_at<DIM, G>(h.to + (size_t)chain * DIM, lane, row[i]);
    }
}

// Long chains, one workgroup each (block b takes long chains b, b + long_blocks, ...): T <= NG tasks of consecutive
// entries (whole samples for a head chain) trained side by side by the block's lane groups and composed.  An update is
// own <- d own - lr w g c with d = 1 - lr w wd: weight decay is a factor that depends on the entry's label only, so the
// decay of the entries BEFORE a task (before_), of the task itself and of the entries AFTER it (after_) are known in closed
// form from label counts (every task counts its own positives; the counts meet in LDS).  A task starts from the row as the
// earlier tasks' decay leaves it, and what it adds to the row is its end state carried through the later tasks' decay:
//     row <- total row + sum over tasks (after_t end_t - total row),         total = before_ x task x after_
// which composes the tasks' decay exactly (a hub row of the benchmark graph decays to 0.48 of itself within ONE batch —
// summing plain deltas of 8 tasks would take it to 0.30) and leaves only the gradients' dependence on the other tasks'
// steps to first order.  The sum runs in task order in one lane group: the same bits on every run.
template <int DIM, int G>
__device__ __forceinline__ void train_long_chains(const TrainArgs &a, const HotArgs &h, const uint32_t block) {
    typedef ChainShape<DIM, G> S;
    constexpr int V = S::V, NG = S::NG;
    __shared__ float ends[NG][DIM];
    __shared__ float positives[NG];
    const int lane = threadIdx.x % G, group = threadIdx.x / G;
    // the block's first record is asked for together with the list's length (one round trip)
    u32x4 record = *reinterpret_cast<const u32x4 *>(h.long_list + 4 + 4 * (size_t)(block < h.long_capacity ? block : 0));
    const uint32_t count = h.long_list[0] < h.long_capacity ? h.long_list[0] : h.long_capacity;
    for (uint32_t j = block; j < count; j += (uint32_t)h.long_blocks) {
        if (j != block) record = *reinterpret_cast<const u32x4 *>(h.long_list + 4 + 4 * (size_t)j);
        const uint32_t chain = record.x, first = record.y, n = record.z, last = first + n;
        if (j == block) {
            GVK_STAMP_VALUE(h, 0, 1);
            GVK_STAMP(h, 2);  // the record is here
            GVK_STAMP_VALUE(h, 7, n);
        }
        // NG tasks at most: a longer chain gets longer tasks
        uint32_t per = h.cap;
        if ((uint64_t)per * NG < n) per = (n + NG - 1) / NG;
        const uint32_t tasks = (n + per - 1) / per;
        const bool mine = (uint32_t)group < tasks;
        const uint32_t begin = mine ? first + (uint32_t)group * per : last;
        const uint32_t end = last - begin > per ? begin + per : last;
        float own[V];
        load_row_at<DIM, G>(h.from + (size_t)chain * DIM, lane, own);
        const uint32_t mine_entry = begin + lane < end ? h.entries[begin + lane] : 0;  // the task's first G entries, one per lane
        uint32_t inside = mine_entry >> 31;
        for (uint32_t p = begin + G + lane; p < end; p += G) inside += h.entries[p] >> 31;
        const float pi = group_count<G>(inside);
        if (lane == 0) positives[group] = pi;
        __syncthreads();
        if (j == block) GVK_STAMP(h, 3);  // own row and the task's entries are here
        float pb = 0, pa = 0;
        for (uint32_t t = 0; t < tasks; t++) {
            const float x = positives[t];
            pb += t < (uint32_t)group ? x : 0.0f;
            pa += t > (uint32_t)group ? x : 0.0f;
        }
        const float before_ = exp2f(pb * h.log2_decay_positive + ((float)(begin - first) - pb) * h.log2_decay_negative);
        const float after_ = exp2f(pa * h.log2_decay_positive + ((float)(last - end) - pa) * h.log2_decay_negative);
        const float total = exp2f((pb + pi + pa) * h.log2_decay_positive + ((float)n - (pb + pi + pa)) * h.log2_decay_negative);
#pragma unroll
        for (int x = 0; x < V; x++) own[x] *= before_;
        if (per <= (uint32_t)kShortEntries)  // the usual task: all its partner rows at once
            short_steps<DIM, G>(a, h, chain, end - begin, lane, own,
                                [&](const int i) __attribute__((always_inline)) { return (uint32_t)__shfl((int)mine_entry, i, G); });
        else  // a chain of more than NG tasks of seven entries (the largest hubs): longer tasks, rows D at a time
            chain_steps<DIM, G>(a, h, chain, begin, end, lane, own);
        if (mine) {
#pragma unroll
            for (int x = 0; x < V; x++) own[x] *= after_;
            store_row_at<DIM, G>(&ends[group][0], lane, own);
        }
        if (j == block) GVK_STAMP(h, 4);  // this task's steps are done
        __syncthreads();
        if (j == block) GVK_STAMP(h, 5);  // every task's steps are done
        if (group == 0) {
            float sum[V], row0[V];
            load_row_at<DIM, G>(h.from + (size_t)chain * DIM, lane, row0);
#pragma unroll
            for (int x = 0; x < V; x++) sum[x] = (1.0f - (float)tasks) * total * row0[x];
            for (uint32_t t = 0; t < tasks; t++) {
                float part[V];
                load_row_at<DIM, G>(&ends[t][0], lane, part);
#pragma unroll
                for (int x = 0; x < V; x++) sum[x] += part[x];
            }
            store_row_at<DIM, G>(h.to + (size_t)chain * DIM, lane, sum);
        }
        __syncthreads();
        if (j == block) GVK_STAMP(h, 6);  // composed and stored
    }
}

// HOT: 1 = the pairs read a hub row as the chains of their unit left it, 2 = on the straight line from where those chains
// found it to where they left it, at the sample's place in the unit (lerp)
// Built for four wavefronts per SIMD (128 registers; the short chains keep seven partner rows per lane group in flight; three
// at dims 256 and 512, sixteen floats of a row per lane): the chains and the pairs of a unit of the sizes this kernel trains (a
// part of a batch) are then resident side by side.
template <int DIM, int G, int KT, int HOT>
__global__ void __launch_bounds__(kBlock, DIM / G > 12 ? 3 : 4) train_hot_kernel(const TrainArgs a, const HotArgs h) {
    // the grid: [long chains | pairs | short chains | idle rows] — the long chains, whose tasks wait for memory three times in
    // a row, are dispatched first, the bulk (the pairs) next; the short chains and the copies fill in behind
    const int b = blockIdx.x;
    GVK_STAMP_VALUE(h, 0, 0);
    GVK_STAMP(h, 1);  // the workgroup starts
    const int pairs_first = h.order == 2 ? 0 : (h.order == 1 ? h.long_blocks : h.long_blocks + h.short_blocks + h.copy_blocks);
    const int long_first = h.order == 2 ? h.pair_blocks : 0;
    const int short_first = h.order == 0 ? h.long_blocks : h.long_blocks + h.pair_blocks;
    if (b >= long_first && b < long_first + h.long_blocks) {
        train_long_chains<DIM, G>(a, h, b - long_first);
    } else if (b >= pairs_first && b < pairs_first + h.pair_blocks) {
        GVK_STAMP_VALUE(h, 0, 3);
        train_pair<DIM, G, GVK_SGD, KT, 1, HOT>(a, (b - pairs_first) * kBlock + threadIdx.x);
        GVK_STAMP(h, 5);  // thread 0's sample is trained (its stores are on their way)
    } else if (b >= short_first && b < short_first + h.short_blocks) {
        train_short_chains<DIM, G>(a, h, b - short_first);
    } else {
        copy_idle_rows<DIM, G>(h, b - short_first - h.short_blocks);
    }
}

// Hub rows between the tables and a mirror: to_mirror != 0 copies the first hot_vertex rows of the head table and the first
// hot_context rows of the tail table into the mirror (a call's first step), else the mirror into the tables (its last).
__global__ void __launch_bounds__(kBlock) hub_rows_kernel(float *vertex, float *context, float *mirror, const uint32_t hot_vertex,
                                                          const uint32_t hot_context, const int dim, const int to_mirror) {
    const size_t quads = (size_t)dim / 4, head_quads = (size_t)hot_vertex * quads, all = head_quads + (size_t)hot_context * quads;
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= all) return;
    f32x4 *in_table = i < head_quads ? reinterpret_cast<f32x4 *>(vertex) + i : reinterpret_cast<f32x4 *>(context) + (i - head_quads);
    f32x4 *in_mirror = reinterpret_cast<f32x4 *>(mirror) + i;
    if (to_mirror) *in_mirror = *in_table;
    else *in_table = *in_mirror;
}

// The chains' work lists, one workgroup per unit: counting sort of the unit's updates to hub rows by row.  Chain c <
// hot_vertex is head row c: per sample with that head, the sample's k negatives (label 0) then its tail (label 1), in
// that order.  Chain hot_vertex + r is context row r: the head of every sample r is the tail (label 1) or a negative
// (label 0) of.  Negatives are drawn exactly as the training kernel draws them (same counters, same tables).  The order
// of the samples inside a chain is the order the atomics retire in — any order is a valid sequential order.  Chains of
// more than cap entries are listed in long_list (train_long_chains: records {chain, first entry, entries, -} from word 4 on),
// those of 1 .. cap entries in short_list (train_short_chains: records of 16 words from word 16 on — {chain, entries, first
// entry, -} and, from word 4, the entries themselves).
constexpr int kListThreads = 1024;

__global__ void __launch_bounds__(kListThreads) hot_list_kernel(TrainArgs a, const uint32_t first_batch_id, const uint32_t stride,
                                                                uint32_t *chain_start_all, uint32_t *entries_all, uint32_t *long_all,
                                                                uint32_t *short_all, const uint32_t entry_capacity, const uint32_t long_capacity,
                                                                const uint32_t cap, const int parts) {
    extern __shared__ uint32_t bins[];  // [chains]
    __shared__ uint32_t wave_total[kListThreads / 64];
    __shared__ uint32_t long_count, short_count;
    const uint32_t chains = a.hot_vertex + a.hot_context;
    // list blockIdx.x = part (blockIdx.x % parts) of batch (blockIdx.x / parts): samples [lo, hi) of the batch
    const int B = a.batch_size, k = a.k;
    const int batch = blockIdx.x / parts, lo = (int)(blockIdx.x % parts) * (B / parts), hi = lo + B / parts;
    const u32x2 *records = reinterpret_cast<const u32x2 *>(a.pairs) + (size_t)batch * B;
    uint32_t *chain_start = chain_start_all + (size_t)blockIdx.x * (chains + 1);
    uint32_t *entries = entries_all + (size_t)blockIdx.x * entry_capacity;
    uint32_t *long_list = long_all + (size_t)blockIdx.x * 4 * (1 + (size_t)long_capacity);
    uint32_t *short_list = short_all + (size_t)blockIdx.x * 16 * (1 + (size_t)chains);
    a.batch_id = first_batch_id + (uint32_t)batch * stride;

    for (uint32_t i = threadIdx.x; i < chains; i += kListThreads) bins[i] = 0;
    if (threadIdx.x == 0) long_count = 0, short_count = 0;
    __syncthreads();
    // A: how many entries every chain gets
    for (int s = lo + threadIdx.x; s < hi; s += kListThreads) {
        const u32x2 pr = records[s];
        if (pr.y < a.hot_vertex) atomicAdd(&bins[pr.y], (uint32_t)(k + 1));
        if (pr.x < a.hot_context) atomicAdd(&bins[a.hot_vertex + pr.x], 1u);
        for (int j = 0; j < k; j++) {
            const Draw d = negative_slot(a, (uint32_t)s, (uint32_t)j);
            const uint32_t n = resolve(a, d, load_entry(a, d));
            if (n < a.hot_context) atomicAdd(&bins[a.hot_vertex + n], 1u);
        }
    }
    __syncthreads();
    // exclusive scan over the chains: thread t owns the bins [t * per, (t + 1) * per)
    {
        const uint32_t per = (chains + kListThreads - 1) / kListThreads;
        const uint32_t lo = threadIdx.x * per < chains ? threadIdx.x * per : chains;
        const uint32_t hi = lo + per < chains ? lo + per : chains;
        uint32_t sum = 0;
        for (uint32_t i = lo; i < hi; i++) sum += bins[i];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        uint32_t inclusive = sum;
        for (int step = 1; step < 64; step <<= 1) {
            const uint32_t up = __shfl_up(inclusive, step);
            if (lane >= step) inclusive += up;
        }
        if (lane == 63) wave_total[wave] = inclusive;
        __syncthreads();
        uint32_t running = inclusive - sum;
        for (int w = 0; w < wave; w++) running += wave_total[w];
        for (uint32_t i = lo; i < hi; i++) {
            const uint32_t count = bins[i];
            chain_start[i] = running;
            bins[i] = running;  // the chain's cursor
            if (count > cap) {
                const uint32_t slot = atomicAdd(&long_count, 1u);
                if (slot < long_capacity) *reinterpret_cast<u32x4 *>(long_list + 4 + 4 * (size_t)slot) = u32x4{i, running, count, 0u};
            } else if (count > 0) {
                const uint32_t slot = atomicAdd(&short_count, 1u);
                *reinterpret_cast<u32x4 *>(short_list + 16 + 16 * (size_t)slot) = u32x4{i, count, running, 0u};
            }
            running += count;
        }
        if (threadIdx.x == kListThreads - 1) chain_start[chains] = running;
    }
    __syncthreads();
    if (threadIdx.x == 0) long_list[0] = long_count, short_list[0] = short_count;
    // B: scatter
    for (int s = lo + threadIdx.x; s < hi; s += kListThreads) {
        const u32x2 pr = records[s];
        const bool hot_head = pr.y < a.hot_vertex;
        uint32_t at = hot_head ? atomicAdd(&bins[pr.y], (uint32_t)(k + 1)) : 0;
        for (int j = 0; j < k; j++) {
            const Draw d = negative_slot(a, (uint32_t)s, (uint32_t)j);
            const uint32_t n = resolve(a, d, load_entry(a, d));
            if (hot_head) entries[at + j] = n;
            if (n < a.hot_context) entries[atomicAdd(&bins[a.hot_vertex + n], 1u)] = pr.y;
        }
        if (hot_head) entries[at + k] = pr.x | 0x80000000u;
        if (pr.x < a.hot_context) entries[atomicAdd(&bins[a.hot_vertex + pr.x], 1u)] = pr.y | 0x80000000u;
    }
    __syncthreads();
    // C: the short chains' entries into their records (written by this workgroup above: its own stores are visible to it
    // after the barrier)
    __threadfence_block();
    for (uint32_t r = threadIdx.x / 8; r < short_count; r += kListThreads / 8) {
        uint32_t *record = short_list + 16 + 16 * (size_t)r;
        const uint32_t n = record[1], first = record[2], i = threadIdx.x % 8;
        if (i < n) record[4 + i] = entries[first + i];
    }
}

#if defined(GVK_AB_BUILDS)  // A/B baselines: only in build/ab/libgvk_ab.so (make ab), never in the product library

// ---- A/B: SGD with one negative, a wavefront owns a segment of the batch ----------------------------------------------
//
// Measured alternative to train_runs_kernel (no faster on large tables, slower on cache-resident shards).  Lane
// layout and arithmetic are those of train_kernel; the unit of work is a SEGMENT of S = (64 / G) * D consecutive pairs
// per wavefront, trained in D steps of 64 / G pairs (one pair per lane group and step):
//
//   phase 1  all D pair headers of every lane group and the alias slots of their negatives           (1 round trip)
//   phase 2  every row the segment needs — both context rows of every pair, and the head row of every pair that
//            STARTS a run — requested at once: 2-3 x S x 512 B in flight per wavefront                (1 round trip)
//   phase 3  arithmetic only.  Pairs of the segment that sit next to each other and share a head row form a run; a
//            run is trained in sequence on ONE register copy of the row, which travels from lane group to lane group
//            (ds_bpermute), exactly as consecutive iterations of one warp update its shared-memory copy in the
//            reference (gpu/graph.cuh:54-94).  Pairs that start a run are independent of each other and run in the
//            same step side by side.  The row is stored once, by the last pair of the run.
//
// With batches in sampler order runs are rare and this is the per-pair kernel with D pairs per lane group in flight.
// After gvk_group_pairs every head row of a batch is a sequence of adjacent pairs: the row crosses HBM once per run
// instead of once per pair (a 100k batch of the benchmark graph has 70k distinct head rows: a tenth of all row traffic
// disappears), and of the pairs of a batch that share a hub row up to S consecutive updates survive instead of one.
// Because every row was requested in phase 2, a run costs no memory round trip per pair — the dependent chain is
// arithmetic only (about 0.1 us per pair), which is what train_runs_kernel above could not avoid.
// LOSS = 0 builds leave the per-sample loss out: gvk_train_episode only needs it for the batch whose loss can still be
// read afterwards (every batch overwrites the same loss buffer).
template <int DIM, int G, int D, int DRAW, int WAVES, int LOSS = 1>
__global__ void __launch_bounds__(kBlock, WAVES) train_segment_kernel(const TrainArgs a) {
    constexpr int V = DIM / G;
    constexpr int NG = 64 / G;  // lane groups of a wavefront = pairs per step
    constexpr int S = NG * D;   // pairs per wavefront
    constexpr uint32_t kNone = 0xffffffffu;  // row ids are below 2^32 - 1 (gvk_tables.n_vertex is a uint32 count)

    const int wave = (blockIdx.x * kBlock + threadIdx.x) / 64;
    const int lane64 = threadIdx.x % 64, g = lane64 / G, lane = lane64 % G;
    const int base = wave * S;
    if (base >= a.batch_size) return;  // whole wavefronts leave together
    const bool draw = DRAW != 0;
    const u32x2 *records = reinterpret_cast<const u32x2 *>(a.pairs);
    const int before = (lane64 + 64 - G) & 63, after = (lane64 + G) & 63;  // same lane of the neighbouring lane groups

    // phase 1: headers and alias slots
    uint32_t head[D], tail[D], neg[D];
    Draw dr[D];
    NegEntry en[D];
#pragma unroll
    for (int i = 0; i < D; i++) {
        const int s = base + i * NG + g;
        head[i] = kNone, tail[i] = 0, neg[i] = 0;
        dr[i] = {0, 0, 0};
        en[i] = {0, 0, 0, 0};
        if (s < a.batch_size) {
            if (draw) {
                dr[i] = negative_slot(a, (uint32_t)s, 0);
                en[i] = load_entry(a, dr[i]);
            } else {
                neg[i] = __builtin_nontemporal_load(a.negatives + s);
            }
            const u32x2 pr = __builtin_nontemporal_load(records + s);
            tail[i] = pr.x, head[i] = pr.y;
        }
    }
    // run structure: cont = this pair continues the run of the pair before it; last = the run ends with this pair
    bool cont[D], last[D];
#pragma unroll
    for (int i = 0; i < D; i++) {
        const uint32_t same_step = __shfl(head[i], before);
        const uint32_t step_before = i > 0 ? __shfl(head[i - 1], before) : kNone;
        const uint32_t pred = g > 0 ? same_step : step_before;
        cont[i] = head[i] != kNone && pred == head[i];
        const uint32_t next_same = __shfl(head[i], after);
        const uint32_t step_after = i + 1 < D ? __shfl(head[i + 1], after) : kNone;
        const uint32_t succ = g < NG - 1 ? next_same : step_after;
        last[i] = succ != head[i];
    }

    // phase 2: every row of the segment
    float vl_[D][V], cn_[D][V], cp_[D][V];
#pragma unroll
    for (int i = 0; i < D; i++) {
        if (head[i] != kNone) {
            if (draw) neg[i] = resolve(a, dr[i], en[i]);
            load_row<DIM, G>(a.context, neg[i], lane, cn_[i]);
            load_row<DIM, G>(a.context, tail[i], lane, cp_[i]);
            if (!cont[i]) load_row<DIM, G>(a.vertex, head[i], lane, vl_[i]);
        }
    }

    // phase 3
    float v[V];
#pragma unroll
    for (int i = 0; i < V; i++) v[i] = 0;
#pragma unroll
    for (int i = 0; i < D; i++) {
        const bool valid = head[i] != kNone;
        float(&vl)[V] = vl_[i];
        float(&cn)[V] = cn_[i];
        float(&cp)[V] = cp_[i];
        // position in the chain of this step: 0 = nothing to wait for in this step (a run start, or lane group 0, whose
        // predecessor finished in the step before), d = d lane groups of this step come first
        int depth = 0;
        {
            const uint64_t chain = __ballot(cont[i]);
            bool run = true;
#pragma unroll
            for (int r = 0; r < NG - 1; r++) {
                const int gg = g - r;
                run = run && gg >= 1 && ((chain >> (gg * G)) & 1);
                depth += run ? 1 : 0;
            }
        }
        auto train_pair = [&]() __attribute__((always_inline)) {
            float sample_loss = 0;
            // negative target, then the positive one (gpu/graph.cuh:63-88; model/graph.h:40-58)
            {
                float partial = 0;
#pragma unroll
                for (int x = 0; x < V; x++) partial += v[x] * cn[x];
                const float prob = sigmoidf(group_sum<G>(partial));
                if (LOSS) sample_loss += a.neg_weight * -logf(1 - prob + kEpsilon);
                float m1 = 0, m2 = 0;
#pragma unroll
                for (int x = 0; x < V; x++) {
                    const float vi = v[x], ci = cn[x];
                    v[x] -= update<GVK_SGD>(a, vi, prob * ci, a.neg_weight, m1, m2);
                    cn[x] -= update<GVK_SGD>(a, ci, prob * vi, a.neg_weight, m1, m2);
                }
                store_row<DIM, G>(a.context, neg[i], lane, cn);
                if (neg[i] == tail[i]) copy_row(cp, cn);  // the pair sees its own update
            }
            {
                float partial = 0;
#pragma unroll
                for (int x = 0; x < V; x++) partial += v[x] * cp[x];
                const float prob = sigmoidf(group_sum<G>(partial));
                if (LOSS) sample_loss += -logf(prob + kEpsilon);
                float m1 = 0, m2 = 0;
#pragma unroll
                for (int x = 0; x < V; x++) {
                    const float vi = v[x], ci = cp[x];
                    v[x] -= update<GVK_SGD>(a, vi, (prob - 1) * ci, 1.0f, m1, m2);
                    cp[x] -= update<GVK_SGD>(a, ci, (prob - 1) * vi, 1.0f, m1, m2);
                }
                store_row<DIM, G>(a.context, tail[i], lane, cp);
            }
            if (LOSS && lane == 0)
                __builtin_nontemporal_store(sample_loss / (1 + a.neg_weight), a.loss + base + i * NG + g);
        };
#pragma unroll 1
        for (int t = 0; t < NG; t++) {
            const bool mine = valid && depth == t;
            if (!__any(mine)) break;  // depths are contiguous: nobody is deeper either
            if (__any(mine && cont[i])) {  // the row of the run moves on to the next lane group
                float vin[V];
#pragma unroll
                for (int x = 0; x < V; x++) vin[x] = __shfl(v[x], before);
                if (mine && cont[i]) copy_row(v, vin);
            }
            if (mine) {
                if (!cont[i]) copy_row(v, vl);
                train_pair();
                if (last[i]) store_row<DIM, G>(a.vertex, head[i], lane, v);
            }
        }
    }
}

// ---- A/B baseline: the reference's kernel SHAPE on wave64 ---------------------------------------------------------
// One wavefront per pair in a grid-stride loop, the vertex row staged in LDS, context rows read-modify-written in
// global memory one element pair per lane, shuffle-down reduction + broadcast — i.e. include/instance/gpu/graph.cuh:
// 36-95 with kWarpSize = 64 and the draw fused.  It exists only so that bench.py --variant 3 can measure what a
// warp-shaped translation reaches on this chip next to the shipped layout; nothing else launches it.
template <int DIM>
__global__ void __launch_bounds__(512) train_kernel_reference_shape(const TrainArgs a) {
    __shared__ float buffer[512 / 64][DIM];
    const int lane = threadIdx.x % 64, wave = threadIdx.x / 64;
    const int waves = gridDim.x * (512 / 64);
    float *vertex_buffer = buffer[wave];
    const int k = a.k;
    for (int s = blockIdx.x * (512 / 64) + wave; s < a.batch_size; s += waves) {
        const uint32_t tail = a.pairs[2 * s], head = a.pairs[2 * s + 1];
        float *vertex = a.vertex + (size_t)head * DIM;
        for (int i = lane; i < DIM; i += 64) vertex_buffer[i] = vertex[i];
        float sample_loss = 0;
        for (int j = 0; j <= k; j++) {
            uint32_t id = tail;
            if (j < k) {
                if (a.negatives) {
                    id = a.negatives[(size_t)s * k + j];
                } else {
                    const Draw d = negative_slot(a, (uint32_t)s, (uint32_t)j);
                    id = resolve(a, d, load_entry(a, d));
                }
            }
            float *context = a.context + (size_t)id * DIM;
            float x = 0;
            for (int i = lane; i < DIM; i += 64) x += vertex_buffer[i] * context[i];
            for (int delta = 1; delta < 64; delta *= 2) x += __shfl_down(x, delta);
            const float logit = __shfl(x, 0);
            const float prob = sigmoidf(logit);
            float gradient, weight;
            if (j == k) {
                gradient = prob - 1;
                weight = 1;
                sample_loss += weight * -logf(prob + kEpsilon);
            } else {
                gradient = prob;
                weight = a.neg_weight;
                sample_loss += weight * -logf(1 - prob + kEpsilon);
            }
            for (int i = lane; i < DIM; i += 64) {
                const float v = vertex_buffer[i], c = context[i];
                vertex_buffer[i] -= a.lr * weight * (gradient * c + a.wd * v);
                context[i] -= a.lr * weight * (gradient * v + a.wd * c);
            }
        }
        if (lane == 0) a.loss[s] = sample_loss / (1 + k * a.neg_weight);
        for (int i = lane; i < DIM; i += 64) vertex[i] = vertex_buffer[i];
    }
}

#endif  // GVK_AB_BUILDS

// ---- predict / alias kernels ------------------------------------------------------------------------------

template <int DIM, int G>
__global__ void __launch_bounds__(kBlock) predict_kernel(const float *vertex, const float *context,
                                                         const uint32_t *pairs, float *logits, int batch_size) {
    constexpr int V = DIM / G;
    const int tid = blockIdx.x * kBlock + threadIdx.x;
    const int s = tid / G, lane = tid % G;
    if (s >= batch_size) return;
    const u32x2 pr = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(pairs) + s);
    float v[V], c[V];
    load_row<DIM, G>(vertex, pr.y, lane, v);
    load_row<DIM, G>(context, pr.x, lane, c);
    float partial = 0;
#pragma unroll
    for (int i = 0; i < V; i++) partial += v[i] * c[i];
    const float logit = group_sum<G>(partial);
    if (lane == 0) logits[s] = logit;
}

// The memory traffic of train_kernel<DIM, G, SGD, k = 1> and nothing else (gvk_probe_row_traffic): the same lane layout,
// the same rows read and written, no arithmetic to speak of and no dependent draw (the negative row is given).
template <int DIM, int G>
__global__ void __launch_bounds__(kBlock) probe_rows_kernel(float *vertex, float *context, const uint32_t *pairs,
                                                            const uint32_t *negatives, float bump, int batch_size) {
    constexpr int V = DIM / G;
    const int tid = blockIdx.x * kBlock + threadIdx.x;
    const int s = tid / G, lane = tid % G;
    if (s >= batch_size) return;
    const u32x2 pr = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(pairs) + s);
    const uint32_t negative = __builtin_nontemporal_load(negatives + s);
    float v[V], c[V], n[V];
    load_row<DIM, G>(vertex, pr.y, lane, v);
    load_row<DIM, G>(context, pr.x, lane, c);
    load_row<DIM, G>(context, negative, lane, n);
#pragma unroll
    for (int i = 0; i < V; i++) v[i] += bump, c[i] += bump, n[i] += bump;
    store_row<DIM, G>(context, negative, lane, n);
    store_row<DIM, G>(context, pr.x, lane, c);
    store_row<DIM, G>(vertex, pr.y, lane, v);
}

__global__ void __launch_bounds__(kBlock) alias_sample_kernel(const gvk_alias_entry *table, uint32_t count,
                                                              const double *rand, uint32_t *result, int n) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    // gpu::Sample narrows both uniforms to Float, then sample() takes them as double
    const float r1 = (float)rand[2 * (size_t)i], r2 = (float)rand[2 * (size_t)i + 1];
    uint32_t index = (uint32_t)((double)r1 * count);
    if (index >= count) index = count - 1;  // cuRAND's (0, 1] can yield index == count in the reference (latent OOB)
    const gvk_alias_entry e = table[index];
    result[i] = r2 < e.prob ? index : e.alias;
}

__global__ void __launch_bounds__(kBlock) negative_draw_kernel(const gvk_alias_entry *table, uint32_t count,
                                                               uint64_t seed, uint32_t batch_id, uint32_t *out,
                                                               int batch_size, int k) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= batch_size * k) return;
    const uint32_t s = i / k, j = i % k;
    const Draw d = negative_slot(seed, batch_id, s, j, count);
    out[i] = resolve(d, table[d.index]);
}

__global__ void __launch_bounds__(kBlock) negative_draw_classes_kernel(const gvk_class_entry *classes, uint32_t count,
                                                                       uint64_t seed, uint32_t batch_id, uint32_t *out,
                                                                       int batch_size, int k) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= batch_size * k) return;
    TrainArgs a;  // the draw of the training kernels, verbatim
    a.classes = classes, a.count = count, a.seed = seed, a.batch_id = batch_id;
    const Draw d = negative_slot(a, (uint32_t)(i / k), (uint32_t)(i % k));
    out[i] = resolve(a, d, load_entry(a, d));
}

constexpr uint32_t kTagPositive = 0x706f7321u;

__global__ void __launch_bounds__(kBlock) sample_pairs_kernel(const gvk_alias_entry *table, const u32x2 *block_pairs,
                                                              uint32_t count, uint64_t seed, uint64_t first_index,
                                                              u32x2 *pool, size_t n) {
    const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= n) return;
    const uint64_t i = first_index + t;
    uint32_t w[4];
    philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), 0, kTagPositive, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    Draw d;
    d.index = __umulhi(w[0], count);
    d.u = (float)(w[1] >> 8) * (1.0f / 16777216.0f);
    const uint32_t edge = resolve(d, table[d.index]);
    __builtin_nontemporal_store(block_pairs[edge], pool + t);
}

// the same draw from the packed form: the pair of a slot sits next to its probability, so a draw that keeps its slot
// (every draw on an unweighted graph) is ONE random 16-byte read instead of a slot and then a pair
__global__ void __launch_bounds__(kBlock) sample_edges_kernel(const gvk_edge_entry *table, uint32_t count, uint64_t seed,
                                                              uint64_t first_index, u32x2 *pool, size_t n) {
    const size_t t = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= n) return;
    const uint64_t i = first_index + t;
    uint32_t w[4];
    philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), 0, kTagPositive, (uint32_t)seed, (uint32_t)(seed >> 32), w);
    const uint32_t index = __umulhi(w[0], count);
    const float u = (float)(w[1] >> 8) * (1.0f / 16777216.0f);
    const u32x4 e = *reinterpret_cast<const u32x4 *>(table + index);
    u32x2 pair = {e.z, e.w};
    if (!(u < __uint_as_float(e.x))) pair = *reinterpret_cast<const u32x2 *>(&table[e.y].tail);
    __builtin_nontemporal_store(pair, pool + t);
}

constexpr uint32_t kTagWalk = 0x77616c6bu;

__device__ __forceinline__ bool has_neighbor(const gvk_walk_graph &g, uint32_t x, uint32_t u) {
    uint64_t lo = g.flat_offsets[x], hi = g.flat_offsets[x + 1];
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        const uint32_t v = g.sorted_neighbors[mid];
        if (v == u) return true;
        if (v < u)
            lo = mid + 1;
        else
            hi = mid;
    }
    return false;
}

constexpr int kMaxAugmentation = 16;

// One walk = one thread: chains of at most L steps, restarted from a fresh edge until the walk has produced `quota`
// pairs; emit(head vertex, tail vertex) receives every pair (chain[j - k], chain[j]), k = 1 .. min(aug, j), in order.
template <class Emit>
__device__ __forceinline__ void walk_pairs(const gvk_walk_graph &g, uint64_t seed, uint64_t walk, uint64_t quota, int L,
                                           int aug, Emit emit) {
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const float fmax = fmaxf(1.0f, fmaxf(1.0f / g.p, 1.0f / g.q));
    uint64_t emitted = 0;
    uint32_t draw = 0;
    uint32_t window[kMaxAugmentation];  // the last `aug` chain nodes, window[j % aug]
    while (emitted < quota) {
        // start (or restart) a chain from a weighted random edge
        uint32_t w[4];
        philox4x32_10((uint32_t)walk, (uint32_t)(walk >> 32), draw++, kTagWalk, k0, k1, w);
        Draw d;
        d.index = __umulhi(w[0], g.num_edge_entries);
        d.u = (float)(w[1] >> 8) * (1.0f / 16777216.0f);
        uint64_t edge = resolve(d, g.edge_table[d.index]);
        uint32_t previous = g.edges_uv[2 * edge], current = g.edges_uv[2 * edge + 1];
        window[0] = previous;
        int j = 1;  // index of `current` in the chain
        while (true) {
            // node j joined the chain: emit its pairs with the previous min(aug, j) nodes
            const int back = j < aug ? j : aug;
            for (int k = 1; k <= back && emitted < quota; k++) {
                emit(window[(j - k) % aug], current, emitted);
                emitted++;
            }
            window[j % aug] = current;
            if (j == L || emitted >= quota) break;
            const uint64_t base = g.flat_offsets[current], degree = g.flat_offsets[current + 1] - base;
            if (degree == 0) break;  // dead end: the chain stops here (graph.cuh:346-349,421-424)
            uint32_t next;
            while (true) {
                philox4x32_10((uint32_t)walk, (uint32_t)(walk >> 32), draw++, kTagWalk, k0, k1, w);
                d.index = __umulhi(w[0], (uint32_t)degree);
                d.u = (float)(w[1] >> 8) * (1.0f / 16777216.0f);
                const uint32_t neighbor = resolve(d, g.neighbor_table[base + d.index]);
                next = g.edges_uv[2 * (base + neighbor) + 1];
                if (!g.biased) break;
                const float f = next == previous ? 1.0f / g.p : (has_neighbor(g, next, previous) ? 1.0f : 1.0f / g.q);
                if ((float)(w[2] >> 8) * (1.0f / 16777216.0f) * fmax < f) break;
            }
            previous = current;
            current = next;
            j++;
        }
    }
}

__global__ void __launch_bounds__(kBlock) sample_walks_kernel(const gvk_walk_graph g, uint64_t seed, uint64_t first_walk,
                                                              u32x2 *pool, size_t pool_pairs, int L, int aug,
                                                              uint64_t pairs_per_walk, uint64_t sb, uint64_t num_walks) {
    const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= num_walks) return;
    const uint64_t begin = t * pairs_per_walk;
    const uint64_t end = begin + pairs_per_walk < pool_pairs ? begin + pairs_per_walk : pool_pairs;
    const uint64_t stride = pool_pairs / sb;
    walk_pairs(g, seed, first_walk + t, end - begin, L, aug, [&](uint32_t head, uint32_t tail, uint64_t i) {
        const uint64_t offset = begin + i, slot = offset % sb * stride + offset / sb;
        u32x2 record = {g.local[tail], g.local[head]};
        __builtin_nontemporal_store(record, pool + slot);
    });
}

// Random walks for SEVERAL partitions: a walk yields pairs for every (head partition, tail partition) block, so every
// pair is binned into the pool of its block, b = part[head] * P + part[tail] (GraphSampler::sample_random_walk's
// per-block pools, graph.cuh:357-373, filled by GPU threads instead of CPU threads).  Pairs for a block whose pool is
// full, or which this call does not collect, are dropped, as the reference drops them (solver.h:1045-1052).
// A pool is cut into `stripes` stripes with one slot counter each and a wavefront appends to stripe (wavefront id mod
// stripes): a single counter per block would take every atomic of the launch on P * P addresses (measured: 0.36 G
// pairs/s at 16 blocks), striped they spread over a few hundred times as many.  counters[b][stripe] keeps counting
// past the stripe's capacity, so the caller sees each block's share and which stripes are full.

struct BlockPools {
    u32x2 *pools;
    const uint64_t *offsets;  // [P * P] first pair of the block's pool, or ~0: not collected
    uint32_t *counters;       // [P * P][stripes]
    const int32_t *part;      // [num_vertex]
    uint32_t capacity, stripes, stripe_capacity, sb;
    int P;
};

__global__ void __launch_bounds__(kBlock) sample_walks_blocks_kernel(const gvk_walk_graph g, const BlockPools b, uint64_t seed,
                                                                     uint64_t first_walk, int L, int aug, uint64_t pairs_per_walk,
                                                                     uint64_t num_walks) {
    const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= num_walks) return;
    const uint32_t stripe = (uint32_t)((t / 64) % b.stripes), stride = b.capacity / b.sb;
    walk_pairs(g, seed, first_walk + t, pairs_per_walk, L, aug, [&](uint32_t head, uint32_t tail, uint64_t) {
        const int block = b.part[head] * b.P + b.part[tail];
        const uint64_t first = b.offsets[block];
        if (first == ~(uint64_t)0) return;
        const uint32_t slot = atomicAdd(b.counters + (size_t)block * b.stripes + stripe, 1u);
        if (slot >= b.stripe_capacity) return;
        const uint32_t position = stripe * b.stripe_capacity + slot;
        u32x2 record = {g.local[tail], g.local[head]};
        __builtin_nontemporal_store(record, b.pools + first + (position % b.sb * stride + position / b.sb));
    });
}

// ---- dispatch ----------------------------------------------------------------------------------------------

int fail(int code, const char *what) { return gvk_fail(code, "%s", what); }

int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return gvk_fail(GVK_EHIP, "%s: %s", what, hipGetErrorString(e));
    return GVK_OK;
}

int default_lanes(int dim) {
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

bool lanes_ok(int dim, int g) {
    switch (dim) {
        case 32: return g == 8 || g == 16;
        case 64: return g == 8 || g == 16;
        case 96: return g == 8 || g == 16;
        case 128: return g == 8 || g == 16 || g == 32 || g == 64;
        case 256: return g == 16 || g == 32 || g == 64;
        case 512: return g == 32 || g == 64;
    }
    return false;
}

typedef void (*TrainKernel)(const TrainArgs);

// RUNS picks train_runs_kernel or train_kernel.  Non-default lane groups exist in the A/B library only (SGD).
#if defined(GVK_AB_BUILDS)
template <int DIM, int G, bool RUNS>
TrainKernel pick_sgd(int opt) {
    if (opt != GVK_SGD) return nullptr;
    return RUNS ? train_runs_kernel<DIM, G, GVK_SGD> : train_kernel<DIM, G, GVK_SGD>;
}
#endif

template <int DIM, int G, bool RUNS>
TrainKernel pick_any(int opt) {
#define GVK_OPT(O) \
    case O: return RUNS ? train_runs_kernel<DIM, G, O> : train_kernel<DIM, G, O>;
    switch (opt) { GVK_OPT(GVK_SGD) GVK_OPT(GVK_MOMENTUM) GVK_OPT(GVK_ADAGRAD) GVK_OPT(GVK_RMSPROP) GVK_OPT(GVK_ADAM) }
#undef GVK_OPT
    return nullptr;
}

template <bool RUNS>
TrainKernel pick_train(int dim, int g, int opt) {
#define GVK_CASE(D, GG) \
    if (dim == D && g == GG) return pick_any<D, GG, RUNS>(opt);
    GVK_CASE(32, 8) GVK_CASE(64, 16) GVK_CASE(96, 8) GVK_CASE(128, 16) GVK_CASE(256, 16) GVK_CASE(512, 32)
#undef GVK_CASE
#if defined(GVK_AB_BUILDS)
#define GVK_CASE(D, GG) \
    if (dim == D && g == GG) return pick_sgd<D, GG, RUNS>(opt);
    GVK_CASE(32, 16) GVK_CASE(64, 8) GVK_CASE(96, 16) GVK_CASE(128, 8) GVK_CASE(128, 32) GVK_CASE(128, 64)
    GVK_CASE(256, 32) GVK_CASE(256, 64) GVK_CASE(512, 64)
#undef GVK_CASE
#endif
    return nullptr;
}

// Longest run one lane group trains in sequence (train_runs_kernel).  20 = how many times the reference's launch refills
// the card it was written for within one default batch — 8192 x 512 threads = one warp per sample (util/gpu.cuh:41-43), a
// V100 holds 80 SMs x 2048 threads = 5120 of those warps at a time, 100 000 / 5120 rounds up to 20 — i.e. how many
// generations of updates to one row that launch can chain; the same cap at every batch size (round 2 scaled it with the
// batch: a 500-sample batch then had runs of one, and walk-mode training fell 0.009 short of sequential).
constexpr int kRunCap = 20;
constexpr int kMaxRunCap = 4096;

int run_cap_for(int batch_size) {
    (void)batch_size;
    return g_run_cap > 0 ? g_run_cap : kRunCap;
}

int validate_train(int dim, const gvk_optimizer *o, const gvk_tables *t, const uint32_t *pairs,
                   const gvk_negative_source *neg, float *loss, int batch_size, int k) {
    if (!default_lanes(dim)) return fail(GVK_EDIM, "gvk_train: dim must be one of 32, 64, 96, 128, 256, 512");
    if (!o || !t || !neg) return fail(GVK_EINVAL, "gvk_train: null optimizer / tables / negative source");
    if (batch_size < 0 || k < 0) return fail(GVK_EINVAL, "gvk_train: negative batch_size or num_negative");
    if (batch_size == 0) return GVK_OK;
    if (!t->vertex || !t->context || !pairs || !loss) return fail(GVK_EINVAL, "gvk_train: null table / pairs / loss");
    if (o->type < GVK_SGD || o->type > GVK_ADAM) return fail(GVK_EINVAL, "gvk_train: unknown optimizer type");
    if (o->type != GVK_SGD && (!t->vertex_moment1 || !t->context_moment1))
        return fail(GVK_EINVAL, "gvk_train: optimizer needs first-moment tables");
    if (o->type == GVK_ADAM && (!t->vertex_moment2 || !t->context_moment2))
        return fail(GVK_EINVAL, "gvk_train: Adam needs second-moment tables");
    if (k > 0 && !neg->negatives && (!neg->table || neg->count == 0) && (!neg->classes || neg->class_count == 0))
        return fail(GVK_EINVAL, "gvk_train: num_negative > 0 but neither negatives nor an alias table given");
    if ((int64_t)batch_size * 64 > INT32_MAX) return fail(GVK_EINVAL, "gvk_train: batch_size too large");
    return 1;
}

// What launch_train would launch for this configuration under the current tuning (also what gvk_describe_train reports).
struct Choice {
    TrainKernel kernel = nullptr;
    int lanes = 0, run_cap = 1, steps = 0;  // steps > 0: train_segment_kernel, (64 / lanes) * steps pairs per wavefront
    int launches = 1;                       // the batch is trained as this many consecutive launches (launches_for)
    bool runs = false, fixed_k = false, reference_shape = false;
};

// Which kernel trains a batch, by the size of the head table:
//   * cache-resident tables (< 16 MiB: a BlogCatalog-sized graph).  Every batch touches every hub row hundreds of
//     times; the solver regroups the batches and train_runs_kernel trains each run of adjacent same-head samples in
//     sequence on one register copy of the row, up to run_cap_for(batch) = 20 of them at the default batch — that keeps
//     link-prediction AUC within 0.002 of sequential training there (0.8745 against 0.8747; the per-pair kernel: 0.8716
//     regrouped, 0.8734 in sampler order), for every optimizer and any number of negatives.  A run is a chain of
//     dependent row fetches, which costs a third of the rate on large tables and nothing that matters here: a
//     quick-start run (7000 batches) trains in well under a second either way;
//   * everything larger: the per-pair kernel.  Conflicts are rare enough there that runs change no AUC (§7).
constexpr size_t kResidentTableBytes = (size_t)16 << 20;

bool resident_table(int dim, uint32_t rows) { return (size_t)rows * dim * 4 < kResidentTableBytes; }

#if defined(GVK_AB_BUILDS)
// D pairs per lane group keep 3 * D rows of DIM / G floats in registers; past 128 VGPRs per lane the kernel is
// built for 2 wavefronts per SIMD (256 VGPRs) instead of spilling, and D = 4 exists only where that suffices.
template <int DIM, int G, int D>
constexpr int segment_waves() {
    return DIM / G * (3 * D + 2) + 40 <= 128 ? 4 : 2;
}

template <int DIM, int G, int D>
TrainKernel segment_build(bool draw, bool loss) {
    if constexpr (DIM / G * (3 * D + 2) + 40 > 256) {
        return nullptr;
    } else {
        constexpr int W = segment_waves<DIM, G, D>();
        if (draw) return loss ? train_segment_kernel<DIM, G, D, 1, W, 1> : train_segment_kernel<DIM, G, D, 1, W, 0>;
        return train_segment_kernel<DIM, G, D, 0, W, 1>;
    }
}

template <int DIM, int G>
TrainKernel pick_segment(int steps, bool draw, bool loss) {
    switch (steps) {
        case 1: return segment_build<DIM, G, 1>(draw, loss);
        case 2: return segment_build<DIM, G, 2>(draw, loss);
        case 4: return segment_build<DIM, G, 4>(draw, loss);
    }
    return nullptr;
}

#endif  // GVK_AB_BUILDS

// want_loss = false: the caller promises that nobody can read this batch's loss (a later batch overwrites it)
// A batch larger than a few samples per table row is trained as several launches.  Inside one launch
// every sample may run at the same time, and of the updates that hold a row at the same time one survives (Hogwild, as
// in the reference).  While a partition has about as many rows as a batch has samples that is rare; when a partition is
// small (a 100k-node graph cut into 16 partitions trains 100 000 samples on 6 250 rows) every row is in flight dozens of
// times per launch — a hub row thousands of times — and most of its updates are lost: link-prediction AUC 0.880 where the
// reference's loop reaches 0.903 at the same partition count.  So a batch is cut into Q equal parts of at most
// g_split_hits = 2 samples per row, each part regrouped on its own (gvk_group_pairs with batch_size / Q) and trained by
// its own launch: same samples, same negatives (a sample keeps its index in the batch), same lr; a later launch sees
// everything the earlier ones wrote.  Q = the smallest divisor of the batch size that is large enough; nothing changes
// for partitions of batch_size / 2 rows or more.
int launches_for(int batch_size, uint32_t rows) {
    if (g_split_hits <= 0 || rows == 0 || batch_size <= 0) return 1;
    const int64_t per_launch = (int64_t)rows * g_split_hits;
    const int64_t want = ((int64_t)batch_size + per_launch - 1) / per_launch;
    if (want <= 1) return 1;
    for (int64_t q = want; q <= batch_size && q <= 8 * want; q++)
        if (batch_size % q == 0) return (int)q;
    for (int64_t q = want; q > 1; q--)  // no divisor just above: the nearest one below
        if (batch_size % q == 0) return (int)q;
    return 1;
}

Choice choose_train(int dim, int opt, int k, bool explicit_negatives, int batch_size, uint32_t rows,
                    bool want_loss = true, uint32_t flags = 0) {
    Choice c;
    (void)want_loss;
#if defined(GVK_AB_BUILDS)
    if (g_variant == 3 && dim == 128 && opt == GVK_SGD) {  // the reference's launch shape, graph.cuh:487-490
        c.reference_shape = true;
        c.lanes = 64;
        return c;
    }
#endif
    c.lanes = g_lanes_per_pair && lanes_ok(dim, g_lanes_per_pair) && opt == GVK_SGD ? g_lanes_per_pair
                                                                                    : default_lanes(dim);
    const bool shipped_shape = opt == GVK_SGD && k == 1 && c.lanes == default_lanes(dim);
    const bool draw = !explicit_negatives;
    c.launches = g_generation > 0 ? 1 : launches_for(batch_size, rows);
#if defined(GVK_AB_BUILDS)
    // GVK_TUNE_SEGMENT_STEPS: train_segment_kernel (a wavefront owns a segment), the A/B alternative to runs for SGD with
    // one negative on the default lane layout
    c.steps = g_variant == 0 && g_generation == 0 && shipped_shape ? g_segment_steps : 0;
    if (c.steps > 0) {
#define GVK_SEGMENT(D, GG) \
    case D: c.kernel = pick_segment<D, GG>(c.steps, draw, want_loss || !g_skip_loss); break;
        switch (dim) {
            GVK_SEGMENT(32, 8) GVK_SEGMENT(64, 16) GVK_SEGMENT(96, 8) GVK_SEGMENT(128, 16) GVK_SEGMENT(256, 16)
            GVK_SEGMENT(512, 32)
        }
#undef GVK_SEGMENT
        if (c.kernel) {
            c.fixed_k = true;
            c.launches = 1;
            return c;
        }
        c.steps = 0;
    }
#endif
    // runs of same-head samples: cache-resident tables by default (resident_table), any table with GVK_TUNE_VARIANT 4
    // (never for the walk-ordered pools of DeepWalk / node2vec: GVK_PAIRS_OF_WALKS, gvk.h)
    c.runs = g_generation == 0 && (g_variant == 4 || (g_variant == 0 && resident_table(dim, rows) && !(flags & GVK_PAIRS_OF_WALKS)));
    c.run_cap = c.runs ? run_cap_for(batch_size) : 1;
    c.kernel = c.runs ? pick_train<true>(dim, c.lanes, opt) : pick_train<false>(dim, c.lanes, opt);
#if defined(GVK_AB_BUILDS)
    // A/B: compile-time-k builds for a few non-default lane groups (GVK_TUNE_LANES_PER_PAIR), so that the comparison
    // with the shipped layout is like for like
    if (opt == GVK_SGD && k == 1 && !c.runs && g_variant != 1 && c.lanes != default_lanes(dim)) {
#define GVK_ALT(D, GG) \
    if (dim == D && c.lanes == GG) c.kernel = draw ? train_kernel<D, GG, GVK_SGD, 1, 1> : train_kernel<D, GG, GVK_SGD, 1, 0>, c.fixed_k = true;
        GVK_ALT(64, 8) GVK_ALT(96, 16) GVK_ALT(128, 8)
#undef GVK_ALT
    }
#endif
    // compile-time k and negative source -> straight-line code (the generic build only with GVK_TUNE_VARIANT 1, A/B library)
    if (shipped_shape && g_variant != 1) {
        c.fixed_k = true;
#define GVK_K1(D, GG)                                                                                         \
    case D:                                                                                                   \
        c.kernel = c.runs ? (draw ? train_runs_kernel<D, GG, GVK_SGD, 1, 1> : train_runs_kernel<D, GG, GVK_SGD, 1, 0>) \
                          : (draw ? train_kernel<D, GG, GVK_SGD, 1, 1> : train_kernel<D, GG, GVK_SGD, 1, 0>);   \
        break;
        switch (dim) {
            GVK_K1(32, 8) GVK_K1(64, 16) GVK_K1(96, 8) GVK_K1(128, 16) GVK_K1(256, 16) GVK_K1(512, 32)
        }
#undef GVK_K1
    }
    return c;
}

int launch_train(hipStream_t stream, int dim, const gvk_optimizer *o, float lr, const gvk_tables *t,
                 const uint32_t *pairs, const gvk_negative_source *neg, uint32_t batch_id, float *loss,
                 int batch_size, int k, float negative_weight, bool want_loss = true) {
    const Choice c = choose_train(dim, o->type, k, neg->negatives != nullptr, batch_size, t->n_vertex, want_loss, t->flags);
    TrainArgs a;
    memset(&a, 0, sizeof(a));
    a.vertex = t->vertex; a.context = t->context;
    a.vm1 = t->vertex_moment1; a.cm1 = t->context_moment1;
    a.vm2 = t->vertex_moment2; a.cm2 = t->context_moment2;
    a.pairs = pairs; a.negatives = neg->negatives; a.table = neg->table; a.loss = loss;
    a.seed = neg->seed; a.count = neg->count; a.batch_id = batch_id;
    if (neg->classes) a.classes = neg->classes, a.count = neg->class_count;  // drawn by weight class
    a.batch_size = batch_size; a.k = k; a.run_cap = c.run_cap;
    a.lr = lr; a.wd = o->weight_decay; a.neg_weight = negative_weight;
    a.hp0 = o->hp0; a.hp1 = o->hp1; a.eps = o->epsilon;
#if defined(GVK_AB_BUILDS)
    if (c.reference_shape) {
        hipLaunchKernelGGL(train_kernel_reference_shape<128>, dim3(8192), dim3(512), 0, stream, a);
        return check_launch("gvk_train (reference-shape variant)");
    }
#endif
    if (!c.kernel) return fail(GVK_EINVAL, "gvk_train: no kernel for this (dim, lanes, optimizer)");
    if (c.steps > 0) {  // A/B: one wavefront per segment of (64 / lanes) * steps pairs
        const int per_wave = 64 / c.lanes * c.steps;
        const int64_t threads = ((int64_t)batch_size + per_wave - 1) / per_wave * 64;
        hipLaunchKernelGGL(c.kernel, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, a);
        return check_launch("gvk_train");
    }
    // consecutive launches of `chunk` samples each: one for the whole batch unless the partition is small
    // (launches_for: equal parts), or — A/B library, GVK_TUNE_GENERATION — launches of one generation of the reference's warps
    int chunk = batch_size / c.launches;
    if (g_generation > 0) chunk = g_generation;
    for (int first = 0; first < batch_size; first += chunk) {
        a.first_sample = first;
        a.batch_size = first + chunk < batch_size ? first + chunk : batch_size;
        const unsigned grid = (unsigned)(((int64_t)(a.batch_size - first) * c.lanes + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(c.kernel, dim3(grid), dim3(kBlock), 0, stream, a);
    }
    return check_launch("gvk_train");
}


// ---- hub rows: work lists + launch (train_hot_kernel) -----------------------------------------------------------------

struct HotLayout {
    size_t chain_start = 0, entries = 0, long_list = 0, short_list = 0, mirrors = 0, mirror_bytes = 0, bytes = 0;  // offsets into the workspace
    uint32_t chains = 0, entry_capacity = 0, long_capacity = 0, cap = 0;
};

constexpr uint32_t kMaxChains = 32768;  // one LDS counter per chain in hot_list_kernel (128 KB of the CU's 160 KB)
constexpr int kLongBlocks = 256;        // workgroups that walk the long chains of a unit (one per CU)

// entries one chain task trains in sequence: at most what a short record holds (train_short_chains)
constexpr uint32_t kDefaultChainCap = 7, kMaxChainCap = 7;

uint32_t chain_cap_for(int chain_cap) {
    const uint32_t want = chain_cap > 0 ? (uint32_t)chain_cap : (g_chain_cap > 0 ? (uint32_t)g_chain_cap : kDefaultChainCap);
    return std::min(want, kMaxChainCap);
}

// One work list per part of a batch (parts divides batch_size: gvk_train_launches): num_batch * parts lists; behind them
// the three mirrors of the hub rows (train_hot_kernel).
HotLayout hot_layout(int dim, int batch_size, int k, uint32_t hot_vertex, uint32_t hot_context, int num_batch, int parts, int chain_cap) {
    HotLayout l;
    l.chains = hot_vertex + hot_context;
    l.cap = chain_cap_for(chain_cap);
    num_batch *= parts;
    batch_size /= parts;
    // a sample adds at most k + 1 entries to its head's chain and one to the chain of each of its k + 1 targets
    l.entry_capacity = (uint32_t)(2 * (size_t)(k + 1) * (size_t)batch_size);
    l.long_capacity = std::min(l.chains, l.entry_capacity / (l.cap + 1) + 1);  // a long chain holds more than cap entries
    auto align = [](size_t x) { return (x + 255) / 256 * 256; };
    l.chain_start = 0;
    l.entries = align((size_t)num_batch * (l.chains + 1) * 4);
    l.long_list = l.entries + align((size_t)num_batch * l.entry_capacity * 4);
    l.short_list = l.long_list + align((size_t)num_batch * (1 + (size_t)l.long_capacity) * 16);
    l.mirrors = l.short_list + align((size_t)num_batch * (1 + (size_t)l.chains) * 64);
    l.mirror_bytes = align((size_t)l.chains * dim * 4);
    l.bytes = l.mirrors + 3 * l.mirror_bytes;
    return l;
}

int validate_hot(const char *what, int dim, int batch_size, int k, uint32_t hot_vertex, uint32_t hot_context, int num_batch, int parts) {
    if (!default_lanes(dim)) return gvk_fail(GVK_EDIM, "%s: dim must be one of 32, 64, 96, 128, 256, 512", what);
    if (batch_size <= 0 || k < 0 || num_batch < 0) return gvk_fail(GVK_EINVAL, "%s: bad sizes", what);
    if (parts < 1 || batch_size % parts) return gvk_fail(GVK_EINVAL, "%s: parts (%d) must divide the batch size", what, parts);
    if ((uint64_t)hot_vertex + hot_context == 0) return gvk_fail(GVK_EINVAL, "%s: no hub rows given", what);
    if ((uint64_t)hot_vertex + hot_context > kMaxChains)
        return gvk_fail(GVK_EINVAL, "%s: at most %u hub rows in all (%u + %u given)", what, kMaxChains, hot_vertex, hot_context);
    if (2 * (uint64_t)(k + 1) * (uint64_t)batch_size > 0x7fffffffull) return gvk_fail(GVK_EINVAL, "%s: batch too large", what);
    return GVK_OK;
}

void fill_negative(TrainArgs &a, const gvk_negative_source *neg) {
    a.negatives = nullptr; a.table = neg->table; a.seed = neg->seed; a.count = neg->count;
    if (neg->classes) a.classes = neg->classes, a.count = neg->class_count;
}

typedef void (*HotKernel)(const TrainArgs, const HotArgs);

HotKernel pick_hot(int dim, int k, int lerp) {
#define GVK_HOT(D, GG)                                                                                              \
    case D:                                                                                                         \
        return k == 1 ? (lerp ? train_hot_kernel<D, GG, 1, 2> : train_hot_kernel<D, GG, 1, 1>)                      \
                      : (lerp ? train_hot_kernel<D, GG, 0, 2> : train_hot_kernel<D, GG, 0, 1>);
    switch (dim) {
        GVK_HOT(32, 8) GVK_HOT(64, 16) GVK_HOT(96, 8) GVK_HOT(128, 16) GVK_HOT(256, 16) GVK_HOT(512, 32)
    }
#undef GVK_HOT
    return nullptr;
}

}  // namespace

extern "C" {

int gvk_hot_plan(int dim, int batch_size, int num_negative, uint32_t hot_vertex, uint32_t hot_context, int num_batch, int parts,
                 int chain_cap, size_t *bytes) {
    if (!bytes) return fail(GVK_EINVAL, "gvk_hot_plan: bytes is null");
    int rc = validate_hot("gvk_hot_plan", dim, batch_size, num_negative, hot_vertex, hot_context, num_batch, parts);
    if (rc != GVK_OK) return rc;
    if (chain_cap < 0) return fail(GVK_EINVAL, "gvk_hot_plan: negative chain_cap");
    *bytes = hot_layout(dim, batch_size, num_negative, hot_vertex, hot_context, num_batch, parts, chain_cap).bytes;
    return GVK_OK;
}

int gvk_hot_build(void *stream, int dim, void *workspace, size_t workspace_bytes, const uint32_t *pool, int batch_size, int num_batch,
                  int num_negative, const gvk_negative_source *negative, uint32_t first_batch_id, uint32_t batch_id_stride,
                  uint32_t hot_vertex, uint32_t hot_context, int parts, int chain_cap) {
    int rc = validate_hot("gvk_hot_build", dim, batch_size, num_negative, hot_vertex, hot_context, num_batch, parts);
    if (rc != GVK_OK) return rc;
    if (num_batch == 0) return GVK_OK;
    if (!workspace || !pool || !negative) return fail(GVK_EINVAL, "gvk_hot_build: null workspace / pool / negative source");
    if (negative->negatives) return fail(GVK_EINVAL, "gvk_hot_build: the chains need negatives drawn on the device");
    if (num_negative > 0 && (!negative->table || negative->count == 0) && (!negative->classes || negative->class_count == 0))
        return fail(GVK_EINVAL, "gvk_hot_build: no alias table given");
    if (chain_cap < 0) return fail(GVK_EINVAL, "gvk_hot_build: negative chain_cap");
    const HotLayout l = hot_layout(dim, batch_size, num_negative, hot_vertex, hot_context, num_batch, parts, chain_cap);
    if (workspace_bytes < l.bytes) return gvk_fail(GVK_EINVAL, "gvk_hot_build: workspace holds %zu bytes, %zu needed", workspace_bytes, l.bytes);
    TrainArgs a;
    memset(&a, 0, sizeof(a));
    a.pairs = pool;
    fill_negative(a, negative);
    a.batch_size = batch_size; a.k = num_negative;
    a.hot_vertex = hot_vertex; a.hot_context = hot_context;
    char *base = static_cast<char *>(workspace);
    const size_t lds = (size_t)l.chains * 4;
    if (lds > 48 * 1024) {  // beyond the default limit of dynamic LDS the kernel needs the attribute (per device)
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(hot_list_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kMaxChains * 4));
        if (e != hipSuccess) return gvk_fail(GVK_EHIP, "gvk_hot_build: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(hot_list_kernel, dim3((unsigned)(num_batch * parts)), dim3(kListThreads), lds, (hipStream_t)stream, a,
                       first_batch_id, batch_id_stride, reinterpret_cast<uint32_t *>(base + l.chain_start),
                       reinterpret_cast<uint32_t *>(base + l.entries), reinterpret_cast<uint32_t *>(base + l.long_list),
                       reinterpret_cast<uint32_t *>(base + l.short_list), l.entry_capacity, l.long_capacity, l.cap, parts);
    return check_launch("gvk_hot_build");
}

int gvk_train_episode_hot(void *stream, int dim, const gvk_optimizer *optimizer, int linear_schedule, const gvk_tables *tables,
                          const uint32_t *pairs, const gvk_negative_source *negative, uint32_t first_batch_id,
                          uint32_t batch_id_stride, uint32_t total_batches, int num_batches, float *loss, int batch_size,
                          int num_negative, float negative_weight, void *workspace, size_t workspace_bytes,
                          uint32_t hot_vertex, uint32_t hot_context, int workspace_batches, int parts, int chain_cap,
                          int form) {
    if (num_batches < 0 || num_batches > workspace_batches) return fail(GVK_EINVAL, "gvk_train_episode_hot: more batches than the work lists cover");
    int rc = validate_train(dim, optimizer, tables, pairs, negative, loss, batch_size, num_negative);
    if (rc <= 0) return rc;
    rc = validate_hot("gvk_train_episode_hot", dim, batch_size, num_negative, hot_vertex, hot_context, workspace_batches, parts);
    if (rc != GVK_OK) return rc;
    if (optimizer->type != GVK_SGD) return fail(GVK_EINVAL, "gvk_train_episode_hot: chains exist for SGD only");
    if (negative->negatives) return fail(GVK_EINVAL, "gvk_train_episode_hot draws negatives on device");
    if (hot_vertex > tables->n_vertex || hot_context > tables->n_context)
        return fail(GVK_EINVAL, "gvk_train_episode_hot: more hub rows than table rows");
    if (chain_cap < 0) return fail(GVK_EINVAL, "gvk_train_episode_hot: negative chain_cap");
    if (form & ~(GVK_HOT_SERIALIZED | GVK_HOT_LERP)) return fail(GVK_EINVAL, "gvk_train_episode_hot: unknown form bits");
    const HotLayout l = hot_layout(dim, batch_size, num_negative, hot_vertex, hot_context, workspace_batches, parts, chain_cap);
    if (!workspace || workspace_bytes < l.bytes) return fail(GVK_EINVAL, "gvk_train_episode_hot: workspace too small (gvk_hot_plan)");
    const bool lerp = (form & GVK_HOT_LERP) != 0, serialized = (form & GVK_HOT_SERIALIZED) != 0 || g_hot_serialized != 0;
    const HotKernel kernel = pick_hot(dim, num_negative, lerp);
    if (!kernel) return fail(GVK_EDIM, "gvk_train_episode_hot: no kernel for this dim");
    const int lanes = default_lanes(dim);
    char *base = static_cast<char *>(workspace);
    TrainArgs a;
    memset(&a, 0, sizeof(a));
    a.vertex = tables->vertex; a.context = tables->context;
    a.loss = loss;
    fill_negative(a, negative);
    a.batch_size = batch_size; a.k = num_negative; a.run_cap = 1;
    a.wd = optimizer->weight_decay; a.neg_weight = negative_weight;
    a.hot_vertex = hot_vertex; a.hot_context = hot_context;
    HotArgs h;
    memset(&h, 0, sizeof(h));
    h.chains = l.chains; h.long_capacity = l.long_capacity; h.cap = l.cap;
    const int groups = kBlock / lanes;
    const int short_blocks = (int)((l.chains + groups - 1) / groups);
    const int long_blocks = (int)std::min<uint32_t>(l.long_capacity, (uint32_t)kLongBlocks);
    const int copy_blocks = (int)((l.chains + 4 * groups - 1) / (4 * groups));
    // the unit of work is a PART of a batch (parts = 1: the batch): unit u = part u % parts of batch u / parts
    const int part_size = batch_size / parts, units = num_batches * parts;
    const unsigned pair_blocks = (unsigned)(((int64_t)part_size * lanes + kBlock - 1) / kBlock);
    if (num_batches == 0) return GVK_OK;
    // when every row of both tables is a hub row the pairs have nothing to store: they run for the last batch only, whose
    // per-sample loss a caller may read
    const bool chains_only = hot_vertex == tables->n_vertex && hot_context == tables->n_context;
    // mirrors in rotation: the chains of unit u read M[(u - 1) % R] and store to M[u % R], the pairs of unit u read M[u % R] — and,
    // lerp, M[(u - 1) % R], which the chains of unit u + 1 (same launch) must then not store to: R = 3; else R = 2
    const int R = lerp ? 3 : 2;
    auto mirror = [&](int u) { return reinterpret_cast<float *>(base + l.mirrors + (size_t)((u + R) % R) * l.mirror_bytes); };
    auto lr_of = [&](int i) {
        const uint32_t id = first_batch_id + (uint32_t)i * batch_id_stride;
        float scale = 1;
        if (linear_schedule) {  // optimizer.h:77-79
            scale = 1 - float(int(id)) / int(total_batches);
            if (scale < 1e-4f) scale = 1e-4f;
        }
        return optimizer->lr * scale;
    };
    auto chains_of = [&](int u) {  // the chain blocks of a launch work on unit u: from mirror u - 1 to mirror u
        h.chain_start = reinterpret_cast<const uint32_t *>(base + l.chain_start) + (size_t)u * (l.chains + 1);
        h.entries = reinterpret_cast<const uint32_t *>(base + l.entries) + (size_t)u * l.entry_capacity;
        h.long_list = reinterpret_cast<const uint32_t *>(base + l.long_list) + (size_t)u * 4 * (1 + (size_t)l.long_capacity);
        h.short_list = reinterpret_cast<const uint32_t *>(base + l.short_list) + (size_t)u * 16 * (1 + (size_t)l.chains);
        h.from = mirror(u - 1), h.to = mirror(u);
        for (int v = 0; v < 2; v++)  // the units since M[u % R] was last stored to: u - 1 .. u - R + 1
            h.before_start[v] = v < R - 1 && u - 1 - v >= 0 ? h.chain_start - (size_t)(v + 1) * (l.chains + 1) : nullptr;
        h.lr = lr_of(u / parts);
        h.log2_decay_positive = (float)std::log2(1.0 - (double)h.lr * a.wd);
        h.log2_decay_negative = (float)std::log2(1.0 - (double)h.lr * a.neg_weight * a.wd);
    };
    auto pairs_of = [&](int u) {  // the pair blocks of a launch work on unit u; false: nothing to do
        const int i = u / parts;
        a.lr = lr_of(i);
        a.batch_id = first_batch_id + (uint32_t)i * batch_id_stride;
        a.pairs = pairs + (size_t)i * batch_size * 2;
        a.first_sample = (u % parts) * part_size;
        a.batch_size = a.first_sample + part_size;
        a.hub_now = mirror(u), a.hub_before = mirror(u - 1);
        a.hub_step = 1.0f / (float)part_size;
        return !chains_only || i == num_batches - 1;
    };
#if defined(GVK_TIMESTAMPS)
    static unsigned long long *stamps = nullptr;
    constexpr size_t kStampBlocks = 8192;
    if (!stamps && hipMalloc(&stamps, kStampBlocks * 64) != hipSuccess) stamps = nullptr;
    h.stamps = stamps;
#endif
    auto launch = [&](bool with_chains, bool with_pairs) {
        h.long_blocks = with_chains ? long_blocks : 0;
        h.short_blocks = with_chains ? short_blocks : 0;
        h.copy_blocks = with_chains ? copy_blocks : 0;
        h.pair_blocks = with_pairs ? (int)pair_blocks : 0;
        h.order = g_hot_order;
        const unsigned grid = (unsigned)(h.long_blocks + h.short_blocks + h.copy_blocks + h.pair_blocks);
#if defined(GVK_TIMESTAMPS)
        h.stamps = grid > kStampBlocks ? nullptr : stamps;
        if (h.stamps && hipMemsetAsync(stamps, 0, kStampBlocks * 64, (hipStream_t)stream) != hipSuccess) h.stamps = nullptr;
#endif
        if (grid) hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, a, h);
#if defined(GVK_TIMESTAMPS)
        // GVK_STAMP_FILE=<prefix>: the stamps of the first 64 launches that carry chains and pairs, each run on its own (the stream
        // is drained after it), to <prefix>.<n>: eight ints {grid, long, pair, short, copy blocks, order, -, -}, then the records
        static int written = 0;
        if (h.stamps && getenv("GVK_STAMP_FILE") && with_chains && with_pairs && written < 64) {
            static std::vector<unsigned long long> host(kStampBlocks * 8);
            char name[512];
            snprintf(name, sizeof name, "%s.%d", getenv("GVK_STAMP_FILE"), written++);
            if (hipStreamSynchronize((hipStream_t)stream) == hipSuccess &&
                hipMemcpy(host.data(), stamps, kStampBlocks * 64, hipMemcpyDeviceToHost) == hipSuccess)
                if (FILE *f = fopen(name, "wb")) {
                    const int header[8] = {(int)grid, h.long_blocks, h.pair_blocks, h.short_blocks, h.copy_blocks, h.order, 0, 0};
                    fwrite(header, sizeof header, 1, f);
                    fwrite(host.data(), 64, grid, f);
                    fclose(f);
                }
        }
#endif
    };
    const unsigned mirror_blocks = (unsigned)(((size_t)l.chains * (size_t)(dim / 4) + kBlock - 1) / kBlock);
    // the hub rows enter the mirrors: every mirror = the tables' rows (a row without entries is only copied on while a mirror
    // is behind: copy_idle_rows)
    for (int m = 0; m < R; m++)
        hipLaunchKernelGGL(hub_rows_kernel, dim3(mirror_blocks), dim3(kBlock), 0, (hipStream_t)stream, a.vertex, a.context, mirror(m),
                           hot_vertex, hot_context, dim, 1);
    // A sample's updates to its rows are all computed from the rows as the sample found them (model/graph.h:47-58).  The
    // chains of a unit therefore run BEFORE its pairs: a chain reads the partner rows before the unit's pairs move them
    // towards the hub row (a chain that read them afterwards would compound the step it is about to take — every sample of a
    // hub row, thousands per epoch: the row's norm explodes), and the pairs train against the hub rows the chains left.
    // Pipelined: launch u trains the pairs of unit u and, in its first blocks, the chains of unit u + 1 — different samples,
    // different mirrors, so neither waits for the other — which hides the chains (few, sequential) behind the pairs (the bulk).
    if (serialized) {  // tests: per unit the chains, then the pairs, as two launches — a pure function of the work lists
        for (int u = 0; u < units; u++) {
            chains_of(u);
            const bool with_pairs = pairs_of(u);
            launch(true, false);
            launch(false, with_pairs);
        }
    } else {
        chains_of(0);
        launch(true, false);
        for (int u = 0; u < units; u++) {
            const bool with_pairs = pairs_of(u);
            if (u + 1 < units) chains_of(u + 1);
            launch(u + 1 < units, with_pairs);
        }
    }
    // ... and leave them: the tables' hub rows = M[last unit]
    hipLaunchKernelGGL(hub_rows_kernel, dim3(mirror_blocks), dim3(kBlock), 0, (hipStream_t)stream, a.vertex, a.context, mirror(units - 1),
                       hot_vertex, hot_context, dim, 0);
    return check_launch("gvk_train_episode_hot");
}

int gvk_train(void *stream, int dim, const gvk_optimizer *optimizer, const gvk_tables *tables,
              const uint32_t *pairs, const gvk_negative_source *negative, uint32_t batch_id, float *loss,
              int batch_size, int num_negative, float negative_weight) {
    int rc = validate_train(dim, optimizer, tables, pairs, negative, loss, batch_size, num_negative);
    if (rc <= 0) return rc;
    return launch_train((hipStream_t)stream, dim, optimizer, optimizer->lr, tables, pairs, negative, batch_id, loss,
                        batch_size, num_negative, negative_weight);
}

int gvk_train_episode(void *stream, int dim, const gvk_optimizer *optimizer, int linear_schedule,
                      const gvk_tables *tables, const uint32_t *pairs, const gvk_negative_source *negative,
                      uint32_t first_batch_id, uint32_t batch_id_stride, uint32_t total_batches, int num_batches,
                      float *loss, int batch_size, int num_negative, float negative_weight) {
    if (num_batches < 0) return fail(GVK_EINVAL, "gvk_train_episode: negative num_batches");
    int rc = validate_train(dim, optimizer, tables, pairs, negative, loss, batch_size, num_negative);
    if (rc <= 0) return rc;
    if (negative->negatives)
        return fail(GVK_EINVAL, "gvk_train_episode draws negatives on device; explicit negatives are per batch");
    for (int i = 0; i < num_batches; i++) {
        const uint32_t id = first_batch_id + (uint32_t)i * batch_id_stride;
        float scale = 1;
        if (linear_schedule) {  // optimizer.h:77-79
            scale = 1 - float(int(id)) / int(total_batches);
            if (scale < 1e-4f) scale = 1e-4f;
        }
        // every batch overwrites loss[]: only the last one's values can ever be read, the others skip computing them
        rc = launch_train((hipStream_t)stream, dim, optimizer, optimizer->lr * scale, tables,
                          pairs + (size_t)i * batch_size * 2, negative, id, loss, batch_size, num_negative,
                          negative_weight, i == num_batches - 1);
        if (rc != GVK_OK) return rc;
    }
    return GVK_OK;
}

int gvk_predict(void *stream, int dim, const float *vertex, const float *context, const uint32_t *pairs,
                float *logits, int batch_size) {
    if (!default_lanes(dim)) return fail(GVK_EDIM, "gvk_predict: dim must be one of 32, 64, 96, 128, 256, 512");
    if (batch_size < 0) return fail(GVK_EINVAL, "gvk_predict: negative batch_size");
    if (batch_size == 0) return GVK_OK;
    if (!vertex || !context || !pairs || !logits) return fail(GVK_EINVAL, "gvk_predict: null pointer");
    hipStream_t st = (hipStream_t)stream;
#define GVK_PREDICT(D, GG)                                                                                    \
    case D:                                                                                                   \
        hipLaunchKernelGGL((predict_kernel<D, GG>), dim3((unsigned)(((int64_t)batch_size * GG + kBlock - 1) / kBlock)), \
                           dim3(kBlock), 0, st, vertex, context, pairs, logits, batch_size);                 \
        break;
    switch (dim) {
        GVK_PREDICT(32, 8) GVK_PREDICT(64, 16) GVK_PREDICT(96, 8) GVK_PREDICT(128, 16) GVK_PREDICT(256, 16)
        GVK_PREDICT(512, 32)
    }
#undef GVK_PREDICT
    return check_launch("gvk_predict");
}

int gvk_probe_row_traffic(void *stream, int dim, float *vertex, float *context, const uint32_t *pairs,
                          const uint32_t *negatives, float bump, int batch_size) {
    if (!default_lanes(dim)) return fail(GVK_EDIM, "gvk_probe_row_traffic: dim must be one of 32, 64, 96, 128, 256, 512");
    if (batch_size <= 0) return GVK_OK;
    if (!vertex || !context || !pairs || !negatives) return fail(GVK_EINVAL, "gvk_probe_row_traffic: null pointer");
    if ((int64_t)batch_size * 64 > INT32_MAX) return fail(GVK_EINVAL, "gvk_probe_row_traffic: batch_size too large");
    hipStream_t st = (hipStream_t)stream;
#define GVK_PROBE(D, GG)                                                                                       \
    case D:                                                                                                    \
        hipLaunchKernelGGL((probe_rows_kernel<D, GG>), dim3((unsigned)(((int64_t)batch_size * GG + kBlock - 1) / kBlock)), \
                           dim3(kBlock), 0, st, vertex, context, pairs, negatives, bump, batch_size);          \
        break;
    switch (dim) {  // the lane groups of gvk_train
        GVK_PROBE(32, 8) GVK_PROBE(64, 16) GVK_PROBE(96, 8) GVK_PROBE(128, 16) GVK_PROBE(256, 16) GVK_PROBE(512, 32)
    }
#undef GVK_PROBE
    return check_launch("gvk_probe_row_traffic");
}

int gvk_alias_sample(void *stream, const gvk_alias_entry *table, uint32_t count, const double *rand,
                     uint32_t *result, int n) {
    if (n < 0) return fail(GVK_EINVAL, "gvk_alias_sample: negative n");
    if (n == 0) return GVK_OK;
    if (!table || !count || !rand || !result) return fail(GVK_EINVAL, "gvk_alias_sample: null pointer / empty table");
    hipLaunchKernelGGL(alias_sample_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream,
                       table, count, rand, result, n);
    return check_launch("gvk_alias_sample");
}

int gvk_negative_draw(void *stream, const gvk_alias_entry *table, uint32_t count, uint64_t seed,
                      uint32_t batch_id, uint32_t *negatives, int batch_size, int num_negative) {
    if (batch_size < 0 || num_negative < 0) return fail(GVK_EINVAL, "gvk_negative_draw: negative size");
    const int64_t n = (int64_t)batch_size * num_negative;
    if (n == 0) return GVK_OK;
    if (n > INT32_MAX) return fail(GVK_EINVAL, "gvk_negative_draw: too many draws for one call");
    if (!table || !count || !negatives) return fail(GVK_EINVAL, "gvk_negative_draw: null pointer / empty table");
    hipLaunchKernelGGL(negative_draw_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, table, count, seed, batch_id, negatives, batch_size, num_negative);
    return check_launch("gvk_negative_draw");
}

int gvk_negative_draw_classes(void *stream, const gvk_class_entry *classes, uint32_t class_count, uint64_t seed,
                              uint32_t batch_id, uint32_t *negatives, int batch_size, int num_negative) {
    if (batch_size < 0 || num_negative < 0) return fail(GVK_EINVAL, "gvk_negative_draw_classes: negative size");
    const int64_t n = (int64_t)batch_size * num_negative;
    if (n == 0) return GVK_OK;
    if (n > INT32_MAX) return fail(GVK_EINVAL, "gvk_negative_draw_classes: too many draws for one call");
    if (!classes || !class_count || !negatives) return fail(GVK_EINVAL, "gvk_negative_draw_classes: null pointer / empty table");
    hipLaunchKernelGGL(negative_draw_classes_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, classes, class_count, seed, batch_id, negatives, batch_size, num_negative);
    return check_launch("gvk_negative_draw_classes");
}

int gvk_sample_pairs(void *stream, const gvk_alias_entry *table, const uint32_t *block_pairs, uint32_t count,
                     uint64_t seed, uint64_t first_index, uint32_t *pool, size_t n) {
    if (n == 0) return GVK_OK;
    if (!table || !block_pairs || !count || !pool) return fail(GVK_EINVAL, "gvk_sample_pairs: null pointer / empty block");
    const size_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffu) return fail(GVK_EINVAL, "gvk_sample_pairs: pool too large for one call");
    hipLaunchKernelGGL(sample_pairs_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, table,
                       reinterpret_cast<const u32x2 *>(block_pairs), count, seed, first_index,
                       reinterpret_cast<u32x2 *>(pool), n);
    return check_launch("gvk_sample_pairs");
}

int gvk_sample_edges(void *stream, const gvk_edge_entry *table, uint32_t count, uint64_t seed, uint64_t first_index,
                     uint32_t *pool, size_t n) {
    if (n == 0) return GVK_OK;
    if (!table || !count || !pool) return fail(GVK_EINVAL, "gvk_sample_edges: null pointer / empty block");
    const size_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffu) return fail(GVK_EINVAL, "gvk_sample_edges: pool too large for one call");
    hipLaunchKernelGGL(sample_edges_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, table, count, seed,
                       first_index, reinterpret_cast<u32x2 *>(pool), n);
    return check_launch("gvk_sample_edges");
}

int gvk_sample_walks(void *stream, const gvk_walk_graph *graph, uint64_t seed, uint64_t first_walk, uint32_t *pool,
                     size_t pool_pairs, int walk_length, int augmentation_step, int shuffle_base) {
    if (pool_pairs == 0) return GVK_OK;
    if (!graph || !pool) return fail(GVK_EINVAL, "gvk_sample_walks: null pointer");
    if (!graph->flat_offsets || !graph->edges_uv || !graph->edge_table || !graph->neighbor_table || !graph->local ||
        !graph->num_edge_entries)
        return fail(GVK_EINVAL, "gvk_sample_walks: incomplete graph description");
    if (graph->biased && (!graph->sorted_neighbors || !(graph->p > 0) || !(graph->q > 0)))
        return fail(GVK_EINVAL, "gvk_sample_walks: node2vec needs sorted_neighbors and positive p, q");
    if (augmentation_step < 1 || augmentation_step > kMaxAugmentation)
        return fail(GVK_EINVAL, "gvk_sample_walks: augmentation_step must be in [1, 16]");
    if (augmentation_step > walk_length)
        return fail(GVK_EINVAL, "`random_walk_length` should be no less than `augmentation_step`");
    if (shuffle_base < 1 || pool_pairs % (size_t)shuffle_base)
        return fail(GVK_EINVAL, "gvk_sample_walks: pool size must be a multiple of the shuffle base");
    const uint64_t per_walk = (uint64_t)augmentation_step * walk_length -
                              (uint64_t)augmentation_step * (augmentation_step - 1) / 2;
    const uint64_t walks = (pool_pairs + per_walk - 1) / per_walk;
    const uint64_t blocks = (walks + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffu) return fail(GVK_EINVAL, "gvk_sample_walks: pool too large for one call");
    hipLaunchKernelGGL(sample_walks_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, *graph, seed,
                       first_walk, reinterpret_cast<u32x2 *>(pool), pool_pairs, walk_length, augmentation_step, per_walk,
                       (uint64_t)shuffle_base, walks);
    return check_launch("gvk_sample_walks");
}

int gvk_sample_walks_blocks(void *stream, const gvk_walk_graph *graph, const int32_t *part, int num_partition, uint64_t seed,
                            uint64_t first_walk, uint64_t num_walks, uint32_t *pools, const uint64_t *offsets,
                            uint32_t *counters, uint32_t capacity, int num_stripe, int walk_length, int augmentation_step,
                            int shuffle_base) {
    if (num_walks == 0) return GVK_OK;
    if (!graph || !part || !pools || !offsets || !counters) return fail(GVK_EINVAL, "gvk_sample_walks_blocks: null pointer");
    if (!graph->flat_offsets || !graph->edges_uv || !graph->edge_table || !graph->neighbor_table || !graph->local ||
        !graph->num_edge_entries)
        return fail(GVK_EINVAL, "gvk_sample_walks_blocks: incomplete graph description");
    if (graph->biased && (!graph->sorted_neighbors || !(graph->p > 0) || !(graph->q > 0)))
        return fail(GVK_EINVAL, "gvk_sample_walks_blocks: node2vec needs sorted_neighbors and positive p, q");
    if (num_partition < 1 || capacity == 0) return fail(GVK_EINVAL, "gvk_sample_walks_blocks: no partitions / empty pools");
    if (augmentation_step < 1 || augmentation_step > kMaxAugmentation)
        return fail(GVK_EINVAL, "gvk_sample_walks_blocks: augmentation_step must be in [1, 16]");
    if (augmentation_step > walk_length)
        return fail(GVK_EINVAL, "`random_walk_length` should be no less than `augmentation_step`");
    if (shuffle_base < 1 || capacity % (uint32_t)shuffle_base)
        return fail(GVK_EINVAL, "gvk_sample_walks_blocks: pool size must be a multiple of the shuffle base");
    if (num_stripe < 1 || capacity % (uint32_t)num_stripe)
        return fail(GVK_EINVAL, "gvk_sample_walks_blocks: the number of stripes must divide the pool size");
    const uint64_t per_walk = (uint64_t)augmentation_step * walk_length -
                              (uint64_t)augmentation_step * (augmentation_step - 1) / 2;
    const uint64_t blocks = (num_walks + kBlock - 1) / kBlock;
    if (blocks > 0x7fffffffu) return fail(GVK_EINVAL, "gvk_sample_walks_blocks: too many walks for one call");
    BlockPools b;
    b.pools = reinterpret_cast<u32x2 *>(pools), b.offsets = offsets, b.counters = counters, b.part = part;
    b.capacity = capacity, b.stripes = (uint32_t)num_stripe, b.stripe_capacity = capacity / (uint32_t)num_stripe, b.sb = (uint32_t)shuffle_base, b.P = num_partition;
    hipLaunchKernelGGL(sample_walks_blocks_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, *graph, b, seed,
                       first_walk, walk_length, augmentation_step, per_walk, num_walks);
    return check_launch("gvk_sample_walks_blocks");
}

int gvk_describe_train(int dim, int optimizer_type, int num_negative, int explicit_negatives, int batch_size,
                       uint32_t n_vertex, char *name, size_t capacity) {
    if (!default_lanes(dim)) return fail(GVK_EDIM, "gvk_describe_train: dim must be one of 32, 64, 96, 128, 256, 512");
    if (optimizer_type < GVK_SGD || optimizer_type > GVK_ADAM || !name || !capacity)
        return fail(GVK_EINVAL, "gvk_describe_train: unknown optimizer type or no buffer");
    static const char *const kOptimizers[] = {"SGD", "Momentum", "AdaGrad", "RMSprop", "Adam"};
    const Choice c = choose_train(dim, optimizer_type, num_negative, explicit_negatives != 0, batch_size, n_vertex);
    if (c.reference_shape)  // A/B library only
        snprintf(name, capacity, "train_kernel_reference_shape<%d> grid 8192x512", dim);
    else if (!c.kernel)
        return fail(GVK_EINVAL, "gvk_describe_train: no kernel for this (dim, lanes, optimizer)");
    else if (c.steps > 0)
        snprintf(name, capacity, "train_segment_kernel<%d,%d,SGD,k=1> %d pairs per wavefront", dim, c.lanes,
                 64 / c.lanes * c.steps);
    else
    {
        char split[48] = "";
        if (c.launches > 1) snprintf(split, sizeof(split), " in %d launches per batch", c.launches);
        snprintf(name, capacity, "%s<%d,%d,%s%s> run_cap %d%s%s", c.runs ? "train_runs_kernel" : "train_kernel", dim, c.lanes,
                 kOptimizers[optimizer_type], c.fixed_k ? ",k=1" : "", c.run_cap, split,
                 g_generation > 0 ? " in launches of one generation" : "");
    }
    return GVK_OK;
}

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

int gvk_train_launches(int batch_size, uint32_t n_vertex) { return launches_for(batch_size, n_vertex); }

int gvk_has_ab_builds(void) {
#if defined(GVK_AB_BUILDS)
    return 1;
#else
    return 0;
#endif
}

}  // extern "C"
