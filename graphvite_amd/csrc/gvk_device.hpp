// gvk_device.hpp — internal: what the hand-written gfx950 (CDNA4, wave64) kernels of the node-embedding hot path share —
// lane groups, the fused negative draw, rows in registers, the arithmetic, the per-pair body (train_pair).  The kernels and
// their C-ABI launchers (include/gvk.h): gvk_pairs.hip (per pair, runs, predict), gvk_chains.hip (hub rows by chains),
// gvk_samplers.hip (alias draws, positive sampling), gvk_group.hip (regrouping), gvk_tuning.cpp (A/B knobs).  Written for MI355X only: no CUDA dual path.
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

#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <cmath>

#include "gvk.h"
#include "gvk_internal.h"

#include "gvk_tuning.h"

namespace {

constexpr float kEpsilon = 1e-15f;  // include/util/common.h:28
constexpr int kBlock = 256;         // 4 wavefronts; no LDS, no barrier -> block size only sets dispatch granularity

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

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
    // HOT == 3 (gvk_train_episode_ahead): hub_now is a ring of versions per hub row, version at slot s of row i at
    // hub_now[(s * (hot_vertex + hot_context) + i) * dim]; slots: per sample of the batch slot_words words whose bytes name the slot its
    // head row (byte 0), its tail (byte 1) and its negatives (bytes 2 ..) are read at — written by hot_slots_kernel
    const uint32_t *slots;
    int slot_words;
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
// degree — belong to the chains of train_hot_kernel: this body reads them and never stores them (1: from the mirror hub_now, 2: on
// the line from hub_before to hub_now, 3: from the ring of versions hub_now at the slots a.slots names for the sample).
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
    uint32_t slot_word0 = 0;
    if constexpr (HOT == 3) slot_word0 = __builtin_nontemporal_load(a.slots + (size_t)s * a.slot_words);
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
    // the call ends), every other row from its table.  HOT == 3: from the ring, at the slot byte `which` of the sample's slot
    // words names (0 the head, 1 the tail, 2 + j negative j)
    auto slot_of = [&](const int which) __attribute__((always_inline)) -> size_t {
        if constexpr (HOT == 3) {
            const uint32_t word = which < 4 ? slot_word0 : a.slots[(size_t)s * a.slot_words + (which >> 2)];
            return (size_t)((word >> (8 * (which & 3))) & 0xffu) * (a.hot_vertex + a.hot_context);
        }
        return 0;
    };
    auto vertex_row = [&](const uint32_t id) __attribute__((always_inline)) -> const float * {
        if (HOT != 0 && id < a.hot_vertex) return a.hub_now + (slot_of(0) + id) * DIM;
        return a.vertex + (size_t)id * DIM;
    };
    auto context_row = [&](const uint32_t id, const int which) __attribute__((always_inline)) -> const float * {
        if (HOT != 0 && id < a.hot_context) return a.hub_now + (slot_of(which) + a.hot_vertex + id) * DIM;
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
        load_row_at<DIM, G>(context_row(tail, 1), lane, reinterpret_cast<float(&)[V]>(early));
        before_row(early_hub, (size_t)a.hot_vertex + tail, reinterpret_cast<float(&)[VB]>(early_before));
    }

    uint32_t id_cur = k > 0 ? (draw ? resolve(a, d0, e0) : neg0) : tail;
    float cur[V], cur1[M1], cur2[M2], cur_before[VB];
    bool cur_hub = HOT == 2 && id_cur < a.hot_context;
    load_row_at<DIM, G>(context_row(id_cur, k > 0 ? 2 : 1), lane, cur);
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
                load_row_at<DIM, G>(context_row(id_nxt, j + 1 < k ? 3 + j : 1), lane, nxt);
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
        if (HOT == 0 || id_cur >= a.hot_context) {  // a hub row and its moment rows belong to its chain (train_moment_chains): read here, never stored
            store_row<DIM, G>(a.context, id_cur, lane, cur);
            if constexpr (NM >= 1) store_row<DIM, G>(a.cm1, id_cur, lane, reinterpret_cast<float(&)[V]>(cur1));
            if constexpr (NM >= 2) store_row<DIM, G>(a.cm2, id_cur, lane, reinterpret_cast<float(&)[V]>(cur2));
        }

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
    if (HOT == 0 || head >= a.hot_vertex) {
        store_row<DIM, G>(a.vertex, head, lane, v);
        if constexpr (NM >= 1) store_row<DIM, G>(a.vm1, head, lane, reinterpret_cast<float(&)[V]>(vm1));
        if constexpr (NM >= 2) store_row<DIM, G>(a.vm2, head, lane, reinterpret_cast<float(&)[V]>(vm2));
    }
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

inline int validate_train(int dim, const gvk_optimizer *o, const gvk_tables *t, const uint32_t *pairs,
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

}  // namespace
