// gvk_chains.hip — hub rows trained by chains: the work lists (hot_list_kernel), the chains of 1 .. 7
// entries and the long chains, the launch that trains a unit's pairs beside the next unit's chains (train_hot_kernel), and their
// C-ABI (include/gvk.h: gvk_hot_plan, gvk_hot_build, gvk_train_episode_hot).
#include "gvk_device.hpp"

#include <chrono>
#include <map>
#include <mutex>

namespace {

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
// The logistic function of a chain step: v_exp_f32 + v_rcp_f32 — about 1e-7 relative, a dozen instructions where the expf and the
// division of sigmoidf (the pairs' form) take about forty.  A chain's step is a dependency chain of ~100 instructions that nothing
// else on its SIMD hides: at the shard size of an 8-GPU run, where a launch IS its longest chain, 13.0 -> 11.9 us per launch
// (profiles/r5/experiments/r5_chain_step_instructions.txt).  GVK_CHAIN_SIGMOID=0 builds the chains with sigmoidf.
#if !defined(GVK_CHAIN_SIGMOID)
#define GVK_CHAIN_SIGMOID 1
#endif
__device__ __forceinline__ float chain_sigmoid(float x) {
#if GVK_CHAIN_SIGMOID && !defined(GVK_SIMT_HOST)
    const float t = __builtin_amdgcn_exp2f(-fabsf(x) * 1.44269504088896340736f);
    return (x > 0 ? 1.0f : t) * __builtin_amdgcn_rcpf(1.0f + t);
#else
    return sigmoidf(x);
#endif
}

// own <- own - lr w (g c + wd own)  (optimizer.h:161-164).  GVK_CHAIN_UPDATE_FORM=1 (measurement variant): as (1 - lr w wd) own - (lr w g) c, two
// instructions per element instead of three — but the dim-128 build then spills 20 bytes
#if !defined(GVK_CHAIN_UPDATE_FORM)
#define GVK_CHAIN_UPDATE_FORM 0
#endif
#if GVK_CHAIN_UPDATE_FORM
#define GVK_CHAIN_UPDATE(own, c, weight, gradient)                                             \
    do {                                                                                       \
        const float keep_ = 1.0f - h.lr * (weight) * a.wd, push_ = h.lr * (weight) * (gradient); \
        _Pragma("unroll") for (int x = 0; x < V; x++) own[x] = keep_ * own[x] - push_ * c[x];  \
    } while (0)
#else
#define GVK_CHAIN_UPDATE(own, c, weight, gradient)                                             \
    do {                                                                                       \
        _Pragma("unroll") for (int x = 0; x < V; x++) own[x] -= h.lr * (weight) * ((gradient) * c[x] + a.wd * own[x]); \
    } while (0)
#endif

#if !defined(GVK_HOT_BLOCK)
#define GVK_HOT_BLOCK 256
#endif
constexpr int kHotBlock = GVK_HOT_BLOCK;  // threads of a train_hot_kernel workgroup: kHotBlock / lanes lane groups = the most tasks of a long chain

struct HotArgs {
    const uint32_t *chain_start;  // [chains + 1] offsets of this unit into entries
    const uint32_t *before_start[2];  // the same of the one or two units before it (null: none): which rows the mirror `to` has missed
    const uint32_t *entries;      // partner row | label << 31 (label 1 = positive)
    const uint32_t *long_list;    // [0] = number of long chains (more than cap entries), then from [4] on a record {chain, first entry, entries, -} each
    const uint32_t *short_list;   // [0] = number of chains of 1 .. cap entries, then from [16] on a record of 16 words each: {chain, entries, -, -, the entries themselves}
    const float *from;            // mirror the chains read: own rows and hub partners as the unit finds them (versioned: the ring)
    float *to;                    // mirror the chains store to (versioned: the ring)
    // versioned (gvk_train_episode_ahead): the hub rows live in a ring of `ring_slots` versions PER ROW — version v of row i at
    // ring[(v % ring_slots) * slot_stride + i] — instead of whole mirrors; a chain reads its row at the slot its record names and
    // stores it to the next one, a hub partner is read at the slot its entry names (written there by hot_slots_kernel)
    uint32_t slot_stride;         // rows between two slots of the ring (versioned: chains; mirrors: 0)
    uint32_t ring_slots;
    int versioned;
    // grouped (gvk_train_episode_ahead with group > 1): one launch carries the chains of `group` consecutive units; a chain whose row
    // had entries in an earlier unit of the same launch waits for that unit's chain to publish it (published[row] = that unit + 1)
    uint32_t *published;
    uint32_t unit;                // this unit's index in the call (a chain publishes unit + 1)
    int grouped;
    // train_group_kernel: the units of a launch lie `*_stride` words apart in the work lists
    uint32_t units_in_launch, start_stride, entries_stride, long_stride, short_stride;
    uint32_t chains;              // hot_vertex + hot_context
    uint32_t long_capacity;
    uint32_t cap;                 // entries of one task (at most kShortEntries)
    uint32_t round_steps;         // a task of more entries applies so many per round (gvk.h GVK_HOT_ROUND_STEPS; 0: all of them in one round)
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

// Where hub rows are read and stored.  Mirrors: row `index` of the mirror, whatever the slot.  Versioned: the row's version at
// `slot` of the ring; a chain stores its row one slot on.
template <int DIM>
__device__ __forceinline__ const float *hub_from(const HotArgs &h, const uint32_t index, const uint32_t slot) {
    return h.from + ((size_t)slot * h.slot_stride + index) * DIM;
}
template <int DIM>
__device__ __forceinline__ float *hub_to(const HotArgs &h, const uint32_t chain, const uint32_t slot_from) {
    const uint32_t slot = h.versioned ? (slot_from + 1 == h.ring_slots ? 0u : slot_from + 1) : 0u;
    return h.to + ((size_t)slot * h.slot_stride + chain) * DIM;
}
// The partner row an entry names.  Mirrors: entry = id | label << 31, a partner below partner_hot is a hub row of the mirror.
// Versioned (hot_slots_kernel): a hub partner is id (15 bits) | slot << 16 | 1 << 30 | label << 31, any other row id (30 bits) | label << 31.
constexpr uint32_t kEntryHub = 0x40000000u;
template <int DIM>
__device__ __forceinline__ const float *partner_of(const HotArgs &h, const uint32_t e, const float *partner_table, const uint32_t partner_hot,
                                                   const uint32_t partner_base) {
    if (h.versioned) {
        if (e & kEntryHub) return h.from + ((size_t)((e >> 16) & 0xffu) * h.slot_stride + partner_base + (e & 0x7fffu)) * DIM;
        return partner_table + (size_t)(e & 0x3fffffffu) * DIM;
    }
    const uint32_t id = e & 0x7fffffffu;
    return id < partner_hot ? h.from + (size_t)(partner_base + id) * DIM : partner_table + (size_t)id * DIM;
}

// Grouped launches: the own row of a chain may have been stored by another workgroup of the SAME launch (the chain of an earlier unit
// of the group), possibly on another XCD, whose L2 this one does not see: such a row is stored and loaded at agent scope (sc1: through
// to memory), the hand-off is published[row] — measured in scripts/experiments/r6_micro/chain_micro.hip (ii): coherent loads and stores
// with a relaxed flag pass a mirror between 256 workgroups without fences (a release / acquire fence pair per hand-off costs ten times more).
template <int DIM, int G>
__device__ __forceinline__ void load_row_coherent(const float *base, const int lane, float (&r)[DIM / G]) {
    typedef Layout<DIM, G> L;
#pragma unroll
    for (int c = 0; c < L::NC; c++)
#pragma unroll
        for (int x = 0; x < L::CW; x++)
            r[c * L::CW + x] = __hip_atomic_load(base + lane * L::CW + c * G * L::CW + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int DIM, int G>
__device__ __forceinline__ void store_row_coherent(float *base, const int lane, const float (&r)[DIM / G]) {
    typedef Layout<DIM, G> L;
#pragma unroll
    for (int c = 0; c < L::NC; c++)
#pragma unroll
        for (int x = 0; x < L::CW; x++)
            __hip_atomic_store(base + lane * L::CW + c * G * L::CW + x, r[c * L::CW + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a record's fourth word: the slot the own row is read at (8 bits) | the unit + 1 whose chain stored it (16 bits) << 8 | 1 << 31 when
// that unit belongs to this launch (hot_slots_kernel)
constexpr uint32_t kRecordWaits = 0x80000000u;
// the chain's row is there: published[chain] has reached the unit the record names (bounded: a bug must not hang the GPU — past the
// bound the chain goes on with what it finds and the launch's result is wrong, which the parity tests see)
__device__ __forceinline__ void await_row(const HotArgs &h, const uint32_t chain, const uint32_t word) {
    if (!h.grouped || !(word & kRecordWaits)) return;
    const uint32_t want = (word >> 8) & 0xffffu;
    for (uint32_t spins = 0; __hip_atomic_load(h.published + chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && spins < (1u << 20); spins++)
        __builtin_amdgcn_s_sleep(2);
}
template <int DIM, int G>
__device__ __forceinline__ void load_own_row(const HotArgs &h, const uint32_t chain, const uint32_t word, const int lane, float (&r)[DIM / G]) {
    const float *at = hub_from<DIM>(h, chain, h.versioned ? word & 0xffu : 0u);
    if (h.grouped && (word & kRecordWaits)) load_row_coherent<DIM, G>(at, lane, r);
    else load_row_at<DIM, G>(at, lane, r);
}
// the chain's row to its next slot; grouped: at agent scope, then published (the stores of a wavefront are complete before its flag is set)
template <int DIM, int G>
__device__ __forceinline__ void store_own_row(const HotArgs &h, const uint32_t chain, const uint32_t word, const int lane, const bool writer,
                                              const float (&r)[DIM / G]) {
    float *at = hub_to<DIM>(h, chain, h.versioned ? word & 0xffu : 0u);
    if (!h.grouped) {
        if (writer) store_row_at<DIM, G>(at, lane, r);
        return;
    }
    if (writer) store_row_coherent<DIM, G>(at, lane, r);
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0): this wavefront's stores have been acknowledged
    if (writer && lane == 0) __hip_atomic_store(h.published + chain, h.unit + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#if !defined(GVK_CHAIN_STEPS_INLINE)
#define GVK_CHAIN_STEPS_INLINE __forceinline__
#endif
template <int DIM, int G>
struct ChainShape {
    static constexpr int V = DIM / G;
#if !defined(GVK_CHAIN_RING)
#define GVK_CHAIN_RING 4  // measurement knob: partner rows in flight per lane group at 8 floats per lane (dims 128 and 96 x 8 lanes: 12)
#endif
    static constexpr int D = V <= 4 ? 8 : (V <= 8 ? GVK_CHAIN_RING : 2);  // partner rows in flight per lane group of a long chain's task
    static constexpr int NG = kHotBlock / G;  // lane groups of a block = most tasks of a long chain
    static_assert(D <= G, "the entry window is two fetches of G entries");
};

// Entries [begin, end) of one chain applied one after the other to `own` (the row of `chain`, in the registers of a lane
// group).  Every lane group of the wavefront comes here together, each with its own chain and range (an empty range: a
// group without work) — the loop runs as long as any group has entries left; a group past its end keeps requesting its own
// mirror row and trains with weight 0.  Every step issues exactly one row request and consumes the one issued D steps
// earlier, with no branch around either, so the wait before a step is "all but the D - 1 youngest" and not "all".
struct NoPrepare {
    __device__ __forceinline__ void operator()() const {}
};
// prepare(): called once, after the first partner rows have been requested and before the first step — where a grouped launch's chain
// waits for its own row (await_row): what it waits for is another workgroup's work, and its own requests are on their way meanwhile
template <int DIM, int G, class Prepare = NoPrepare>
__device__ GVK_CHAIN_STEPS_INLINE void chain_steps(const TrainArgs &a, const HotArgs &h, const uint32_t chain, const uint32_t slot_from,
                                            const uint32_t begin, const uint32_t end, const int lane, float (&own)[DIM / G], Prepare prepare = Prepare()) {
    typedef ChainShape<DIM, G> S;
    constexpr int V = S::V, D = S::D;
    const bool is_vertex = chain < a.hot_vertex;
    const float *partner_table = is_vertex ? a.context : a.vertex;
    const uint32_t partner_hot = is_vertex ? a.hot_context : a.hot_vertex;  // partners below this id are hub rows: read from the mirror
    const uint32_t partner_base = is_vertex ? a.hot_vertex : 0u;
    const float *idle = hub_from<DIM>(h, chain, slot_from);
    // the work list, G entries per fetch, two fetches resident: entries [blk, blk + 2 G)
    uint32_t blk = begin;
    uint32_t e_cur = blk + lane < end ? h.entries[blk + lane] : 0;
    uint32_t e_nxt = blk + G + lane < end ? h.entries[blk + G + lane] : 0;
    // entry p of the list (through the window): the row it names (past the end: the group's own mirror row) and its label
    auto row_of = [&](const uint32_t p, uint32_t &label) __attribute__((always_inline)) -> const float * {
        const uint32_t o = p - blk;
        const uint32_t e = (uint32_t)__shfl((int)(o < (uint32_t)G ? e_cur : e_nxt), (int)(o & (G - 1)), G);
        label = e >> 31;
        const float *row = partner_of<DIM>(h, e, partner_table, partner_hot, partner_base);
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
    prepare();
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
            const float prob = chain_sigmoid(group_sum<G>(partial));
            const float gradient = positive ? prob - 1 : prob;
            const float weight = p < end ? (positive ? 1.0f : a.neg_weight) : 0.0f;
            GVK_CHAIN_UPDATE(own, c, weight, gradient);  // optimizer.h:161-164
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
// GVK_TASK_ROWS (measurement knob, 16 in a build for fewer wavefronts per SIMD): a long chain's task of up to so many entries has all its
// partner rows in flight at once (N below) instead of a ring of four
#if !defined(GVK_TASK_ROWS)
#define GVK_TASK_ROWS 7
#endif
template <int DIM, int G, class EntryOf, class Prepare = NoPrepare, int N = kShortEntries>
__device__ __forceinline__ void short_steps(const TrainArgs &a, const HotArgs &h, const uint32_t chain, const uint32_t slot_from, const uint32_t n,
                                            const int lane, float (&own)[DIM / G], EntryOf entry_of, Prepare prepare = Prepare()) {
    constexpr int V = DIM / G;
    constexpr int D = V <= 8 ? N : (V <= 12 ? 3 : 2);  // partner rows in flight: all of them where the registers hold them
    const bool is_vertex = chain < a.hot_vertex;
    const float *partner_table = is_vertex ? a.context : a.vertex;
    const uint32_t partner_hot = is_vertex ? a.hot_context : a.hot_vertex;
    const uint32_t partner_base = is_vertex ? a.hot_vertex : 0u;
    const float *idle = hub_from<DIM>(h, chain, slot_from);
    float ring[D][V];
    uint32_t labels = 0;
    auto request = [&](const int i) __attribute__((always_inline)) {  // the row of entry i into its slot of the ring
        const uint32_t e = entry_of(i);
        const float *row = partner_of<DIM>(h, e, partner_table, partner_hot, partner_base);
        load_row_at<DIM, G>((uint32_t)i < n ? row : idle, lane, ring[i % D]);
        labels |= (e >> 31) << i;
    };
#pragma unroll
    for (int i = 0; i < D; i++) request(i);
    prepare();
#pragma unroll
    for (int i = 0; i < N; i++) {
        const bool positive = (labels >> i & 1u) != 0;
        const float(&c)[V] = ring[i % D];
        float partial = 0;
#pragma unroll
        for (int x = 0; x < V; x++) partial += own[x] * c[x];
        const float prob = chain_sigmoid(group_sum<G>(partial));
        const float gradient = positive ? prob - 1 : prob;
        const float weight = (uint32_t)i < n ? (positive ? 1.0f : a.neg_weight) : 0.0f;
        GVK_CHAIN_UPDATE(own, c, weight, gradient);  // optimizer.h:161-164
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
    const uint32_t chain = mine ? word(0) : 0, n = mine ? word(1) : 0, where = mine && h.versioned ? word(3) : 0, slot_from = where & 0xffu;
    float own[S::V];
    short_steps<DIM, G>(a, h, chain, slot_from, n, lane, own, [&](const int i) __attribute__((always_inline)) { return word(4 + i); },
                        [&]() __attribute__((always_inline)) {  // the partner rows are on their way: now the own row (a grouped launch: once it is there)
                            await_row(h, chain, where);
                            load_own_row<DIM, G>(h, chain, where, lane, own);
                        });
    GVK_STAMP(h, 5);  // steps done
    store_own_row<DIM, G>(h, chain, where, lane, mine, own);
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
__device__ __forceinline__ void train_long_chains_one_round(const TrainArgs &a, const HotArgs &h, const uint32_t block) {
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
        const uint32_t chain = record.x, first = record.y, n = record.z, last = first + n, where = h.versioned ? record.w : 0u, slot_from = where & 0xffu;
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
        auto own_row = [&]() __attribute__((always_inline)) {  // after the task's first partner rows have been requested (a grouped launch waits here)
            await_row(h, chain, where);
            load_own_row<DIM, G>(h, chain, where, lane, own);
#pragma unroll
            for (int x = 0; x < V; x++) own[x] *= before_;
        };
        auto entry_of = [&](const int i) __attribute__((always_inline)) { return (uint32_t)__shfl((int)mine_entry, i, G); };
        if (per <= (uint32_t)kShortEntries)  // the usual task: all its partner rows at once
            short_steps<DIM, G>(a, h, chain, slot_from, end - begin, lane, own, entry_of, own_row);
        else if (GVK_TASK_ROWS > kShortEntries && V <= 8 && per <= (uint32_t)(GVK_TASK_ROWS < G ? GVK_TASK_ROWS : G))
            short_steps<DIM, G, decltype(entry_of), decltype(own_row), (GVK_TASK_ROWS > kShortEntries ? (GVK_TASK_ROWS < G ? GVK_TASK_ROWS : G) : kShortEntries)>(
                a, h, chain, slot_from, end - begin, lane, own, entry_of, own_row);
        else  // a chain of more than NG tasks of seven entries (the largest hubs): longer tasks, rows D at a time
            chain_steps<DIM, G>(a, h, chain, slot_from, begin, end, lane, own, own_row);
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
            load_own_row<DIM, G>(h, chain, where, lane, row0);
#pragma unroll
            for (int x = 0; x < V; x++) sum[x] = (1.0f - (float)tasks) * total * row0[x];
            for (uint32_t t = 0; t < tasks; t++) {
                float part[V];
                load_row_at<DIM, G>(&ends[t][0], lane, part);
#pragma unroll
                for (int x = 0; x < V; x++) sum[x] += part[x];
            }
            store_own_row<DIM, G>(h, chain, where, lane, true, sum);
        }
        __syncthreads();
        if (j == block) GVK_STAMP(h, 6);  // composed and stored
    }
}

// The same with ROUNDS (gvk_train_episode_hot form GVK_HOT_ROUNDS; a kernel build of its own, so that each build has one stream loop
// and stays within its registers).
// Long chains in rounds, one workgroup each (block b takes long chains b, b + long_blocks, ...): T <= NG tasks of consecutive
// entries (whole samples for a head chain) trained side by side by the block's lane groups and composed.  An update is
// own <- d own - lr w g c with d = 1 - lr w wd: weight decay is a factor that depends on the entry's label only, so the
// decay of the entries BEFORE a task (before_), of the task itself and of the entries AFTER it (after_) are known in closed
// form from label counts (every task counts its own positives; the counts meet in LDS).  A task starts from the row as the
// earlier tasks' decay leaves it, and what it adds to the row is its end state carried through the later tasks' decay:
//     row <- total row + sum over tasks (after_t end_t - total row),         total = before_ x task x after_
// which composes the tasks' decay exactly (a hub row of the benchmark graph decays to 0.48 of itself within ONE batch —
// summing plain deltas of 8 tasks would take it to 0.30) and leaves only the gradients' dependence on the other tasks'
// steps to first order.  The sum runs in task order: the same bits on every run given the work lists.
//
// ROUNDS.  "To first order" has a price that grows with the entries that work side by side from one state: their gradient
// steps add up where the sequential loop's would have seen each other — on a graph whose largest hub heads 6 % of the
// samples a unit's 400 entries of that row, 16 tasks of 25, overshoot the reference's loop (AUC +0.008; sequential chains at
// the same parts: +0.0001).  Tasks of more than h.round_steps entries therefore work in rounds of that many:
// after every round the tasks' end states are composed — by every lane group on its own, from LDS — and the next round
// starts from the composed row: at most NG x round_steps entries side by side, however long the chain.  The entries of a
// task are still one stream through the ring of D rows in flight (a round's end does not drain it), the label counts of the
// next round's segments travel with the end states: ONE barrier per round.
template <int DIM, int G>
__device__ __forceinline__ void train_long_chains_in_rounds(const TrainArgs &a, const HotArgs &h, const uint32_t block) {
    typedef ChainShape<DIM, G> S;
#if !defined(GVK_ROUND_RING)
#define GVK_ROUND_RING 4
#endif
    // rows in flight: a round's end waits for the slowest lane group, so the rows of a round are asked for more than a round ahead where the registers allow
    constexpr int V = S::V, NG = S::NG, D = V <= 8 ? (GVK_ROUND_RING < G ? GVK_ROUND_RING : G) : S::D;
    __shared__ float ends[2][NG][DIM];
    __shared__ float counts[2][NG][2];  // per round (two in rotation) and task: positives, entries of the task's segment
    const int lane = threadIdx.x % G, group = threadIdx.x / G;
    // the block's first record is asked for together with the list's length (one round trip)
    u32x4 record = *reinterpret_cast<const u32x4 *>(h.long_list + 4 + 4 * (size_t)(block < h.long_capacity ? block : 0));
    const uint32_t count = h.long_list[0] < h.long_capacity ? h.long_list[0] : h.long_capacity;
    for (uint32_t j = block; j < count; j += (uint32_t)h.long_blocks) {
        if (j != block) record = *reinterpret_cast<const u32x4 *>(h.long_list + 4 + 4 * (size_t)j);
        const uint32_t chain = record.x, first = record.y, n = record.z, last = first + n, where = h.versioned ? record.w : 0u, slot_from = where & 0xffu;
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
        // entries a task applies per round (a round's segment fits one fetch of G entries)
        const uint32_t steps = h.round_steps && per > h.round_steps ? (h.round_steps < (uint32_t)G ? h.round_steps : (uint32_t)G) : per;
        float row[V], own[V];
        await_row(h, chain, where);
        load_own_row<DIM, G>(h, chain, where, lane, row);
        const uint32_t mine_entry = begin + lane < end ? h.entries[begin + lane] : 0;  // the task's first G entries, one per lane
        // what the tasks of a round need from each other: {positives, entries} of every task's segment, through LDS
        auto share_counts = [&](const int buffer, const uint32_t positives, const uint32_t length) __attribute__((always_inline)) {
            if (lane == 0) counts[buffer][group][0] = (float)positives, counts[buffer][group][1] = (float)length;
        };
        float after_ = 1, total = 1;
        uint32_t active = 0;  // tasks with entries in the round
        auto enter_round = [&](const int buffer) __attribute__((always_inline)) {  // after the barrier: own <- the row under the decay of the tasks before this one
            float pb = 0, nb = 0, pa = 0, na = 0, pi = 0, ni = 0;
            active = 0;
            for (uint32_t t = 0; t < tasks; t++) {
                const float p = counts[buffer][t][0], len = counts[buffer][t][1];
                pb += t < (uint32_t)group ? p : 0.0f, nb += t < (uint32_t)group ? len - p : 0.0f;
                pa += t > (uint32_t)group ? p : 0.0f, na += t > (uint32_t)group ? len - p : 0.0f;
                pi += t == (uint32_t)group ? p : 0.0f, ni += t == (uint32_t)group ? len - p : 0.0f;
                active += len > 0 ? 1u : 0u;
            }
            const float before_ = exp2f(pb * h.log2_decay_positive + nb * h.log2_decay_negative);
            after_ = exp2f(pa * h.log2_decay_positive + na * h.log2_decay_negative);
            total = exp2f((pb + pi + pa) * h.log2_decay_positive + (nb + ni + na) * h.log2_decay_negative);
#pragma unroll
            for (int x = 0; x < V; x++) own[x] = before_ * row[x];
        };
        auto compose = [&](const int buffer) __attribute__((always_inline)) {  // after the barrier: the round's end states into the row, by every group
#pragma unroll
            for (int x = 0; x < V; x++) row[x] *= (1.0f - (float)active) * total;
            for (uint32_t t = 0; t < active; t++) {
                float part[V];
                load_row_at<DIM, G>(&ends[buffer][t][0], lane, part);
#pragma unroll
                for (int x = 0; x < V; x++) row[x] += part[x];
            }
        };
        if (per <= (uint32_t)kShortEntries && steps == per) {
            // the usual long chain: tasks of up to seven entries, one round, all partner rows of a task at once
            share_counts(0, (uint32_t)group_count<G>(mine_entry >> 31), end - begin);
            __syncthreads();
            if (j == block) GVK_STAMP(h, 3);  // own row and the task's entries are here
            enter_round(0);
            short_steps<DIM, G>(a, h, chain, slot_from, end - begin, lane, own,
                                [&](const int i) __attribute__((always_inline)) { return (uint32_t)__shfl((int)mine_entry, i, G); });
            if (mine) {
#pragma unroll
                for (int x = 0; x < V; x++) own[x] *= after_;
                store_row_at<DIM, G>(&ends[0][group][0], lane, own);
            }
            if (j == block) GVK_STAMP(h, 4);  // this task's steps are done
            __syncthreads();
            if (j == block) GVK_STAMP(h, 5);  // every task's steps are done
            compose(0);
        } else {
            // longer tasks in rounds of `steps` entries (one round when steps == per): the task's entries as one stream, rows D at a time in flight
            const bool is_vertex = chain < a.hot_vertex;
            const float *partner_table = is_vertex ? a.context : a.vertex;
            const uint32_t partner_hot = is_vertex ? a.hot_context : a.hot_vertex;  // partners below this id are hub rows: read from the mirror
            const uint32_t partner_base = is_vertex ? a.hot_vertex : 0u;
            const float *idle = hub_from<DIM>(h, chain, slot_from);
            // the work list, G entries per fetch, two fetches resident: entries [blk, blk + 2 G)
            uint32_t blk = begin;
            uint32_t e_cur = mine_entry;
            uint32_t e_nxt = blk + G + lane < end ? h.entries[blk + G + lane] : 0;
            // entry p of the list (through the window): the row it names (past the end: the group's own mirror row) and its label
            auto row_of = [&](const uint32_t p, uint32_t &label) __attribute__((always_inline)) -> const float * {
                const uint32_t o = p - blk;
                const uint32_t e = (uint32_t)__shfl((int)(o < (uint32_t)G ? e_cur : e_nxt), (int)(o & (G - 1)), G);
                label = e >> 31;
                const float *at = partner_of<DIM>(h, e, partner_table, partner_hot, partner_base);
                return p < end ? at : idle;
            };
            float ring[D][V];
            uint32_t labels = 0;  // bit i: the label of the entry whose row sits in ring[i]
#pragma unroll
            for (int i = 0; i < D; i++) {
                uint32_t label;
                load_row_at<DIM, G>(row_of(begin + i, label), lane, ring[i]);
                labels |= label << i;
            }
            // the label count of the first round's segment — of a task's whole range when there is one round
            uint32_t segment = begin, length = end - segment < steps ? end - segment : steps;
            uint32_t inside = (uint32_t)lane < length ? mine_entry >> 31 : 0;
            for (uint32_t p = begin + G + lane; p < begin + length; p += G) inside += h.entries[p] >> 31;
            share_counts(0, (uint32_t)group_count<G>(inside), length);
            // the next round's segment, asked for a round ahead of its count
            auto segment_ahead = [&](const uint32_t from) __attribute__((always_inline)) -> uint32_t {
                return steps < per && from + lane < end && (uint32_t)lane < steps ? h.entries[from + lane] : 0;
            };
            uint32_t ahead = segment_ahead(begin + steps);
            __syncthreads();
            if (j == block) GVK_STAMP(h, 3);  // own row and the task's entries are here
            enter_round(0);
            uint32_t round = 0, in_round = 0;
            for (uint32_t base = 0; base < per; base += D) {  // `per` is the workgroup's: every lane group takes every step
                const uint32_t f = blk + 2 * G + lane;
                const uint32_t e_fut = h.entries[f < end ? f : (begin < end ? end - 1 : 0)];  // the window after e_nxt, asked for ahead of its use
#pragma unroll
                for (int i = 0; i < D; i++) {
                    const uint32_t p = begin + base + i;
                    const bool positive = (labels >> i & 1u) != 0;
                    const float(&c)[V] = ring[i];
                    // forward / backward of one target: model/graph.h:40-58, gpu/graph.cuh:77-87 — on the own row only
                    float partial = 0;
#pragma unroll
                    for (int x = 0; x < V; x++) partial += own[x] * c[x];
                    const float prob = chain_sigmoid(group_sum<G>(partial));
                    const float gradient = positive ? prob - 1 : prob;
                    const float weight = p < end && base + i < per ? (positive ? 1.0f : a.neg_weight) : 0.0f;
                    GVK_CHAIN_UPDATE(own, c, weight, gradient);  // optimizer.h:161-164
                    // the slot is free: the row of entry p + D takes it (D - 1 requests stay in flight while a step computes)
                    uint32_t label;
                    load_row_at<DIM, G>(row_of(p + D, label), lane, ring[i]);
                    labels = (labels & ~(1u << i)) | label << i;
                    if (++in_round == steps && base + i + 1 < per) {
                        // the round ends (the same step for every lane group): end states and the next segments' counts out, ...
                        const int buffer = (int)(round & 1u);
                        if (segment < end) {
#pragma unroll
                            for (int x = 0; x < V; x++) own[x] *= after_;
                            store_row_at<DIM, G>(&ends[buffer][group][0], lane, own);
                        }
                        segment += steps;
                        length = segment < end ? (end - segment < steps ? end - segment : steps) : 0;
                        share_counts(buffer ^ 1, (uint32_t)group_count<G>((uint32_t)lane < length ? ahead >> 31 : 0), length);
                        ahead = segment_ahead(segment + steps);
                        __syncthreads();
                        // ... composed by every group, and the next round starts from the composed row
                        compose(buffer);
                        enter_round(buffer ^ 1);
                        round++, in_round = 0;
                    }
                }
                if (base + D >= blk - begin + G) {  // the next steps look beyond e_nxt: move the window
                    blk += G;
                    e_cur = e_nxt;
                    e_nxt = e_fut;
                }
            }
            const int buffer = (int)(round & 1u);
            if (segment < end) {
#pragma unroll
                for (int x = 0; x < V; x++) own[x] *= after_;
                store_row_at<DIM, G>(&ends[buffer][group][0], lane, own);
            }
            if (j == block) GVK_STAMP(h, 4);  // this task's steps are done
            __syncthreads();
            if (j == block) GVK_STAMP(h, 5);  // every task's steps are done
            compose(buffer);
        }
        if (group == 0) store_own_row<DIM, G>(h, chain, where, lane, true, row);
        __syncthreads();
        if (j == block) GVK_STAMP(h, 6);  // composed and stored
    }
}

// ---- moment optimizers (Momentum, AdaGrad, RMSprop, Adam: gpu/graph.cuh:104-242) --------------------------------------------
//
// Their update does not compose — the step depends on the row's moment rows, which every step moves — so a hub row's chain is ONE
// task, however long: a lane group holds the row AND its moment rows in registers and applies the unit's entries one after the other
// (optimizer.h:170-210 on the own row only; the partner's side of a sample is its own chain's, or the pairs').  The row lives in the
// mirrors like an SGD chain's; its moment rows stay in the moment tables, which nothing but its chain writes (the pairs read them for
// the hub head's steps inside a sample and store neither).  Block b trains chains [b NG, (b + 1) NG); the launch ends with its
// longest chain — 250 dependent steps for the top hub of the headline shape: correct, not fast.
template <int DIM, int G, int OPT>
__device__ __forceinline__ void train_moment_chains(const TrainArgs &a, const HotArgs &h, const uint32_t block) {
    typedef ChainShape<DIM, G> S;
    constexpr int V = S::V, D = S::D, M2 = OPT == GVK_ADAM ? V : 1;
    const int lane = threadIdx.x % G, group = threadIdx.x / G;
    const uint32_t at = block * S::NG + group, chain = at < h.chains ? at : h.chains - 1;
    const uint32_t begin = h.chain_start[chain], end = at < h.chains ? h.chain_start[chain + 1] : begin;
    if (__builtin_amdgcn_ballot_w64(begin < end) == 0) return;  // no lane group of the wavefront has entries
    const bool is_vertex = chain < a.hot_vertex;
    const float *partner_table = is_vertex ? a.context : a.vertex;
    const uint32_t partner_hot = is_vertex ? a.hot_context : a.hot_vertex, partner_base = is_vertex ? a.hot_vertex : 0u;
    const uint32_t row_index = is_vertex ? chain : chain - a.hot_vertex;
    float *const moment1 = is_vertex ? a.vm1 : a.cm1, *const moment2 = is_vertex ? a.vm2 : a.cm2;
    const float *idle = hub_from<DIM>(h, chain, 0);
    float own[V], m1[V], m2[M2];
    load_row_at<DIM, G>(idle, lane, own);
    load_row<DIM, G>(moment1, row_index, lane, m1);
    if constexpr (OPT == GVK_ADAM) load_row<DIM, G>(moment2, row_index, lane, reinterpret_cast<float(&)[V]>(m2));
    TrainArgs o = a;  // the optimizer's constants with the learning rate of the chains' batch
    o.lr = h.lr;
    // the work list, G entries per fetch, two fetches resident: entries [blk, blk + 2 G) (chain_steps)
    uint32_t blk = begin;
    uint32_t e_cur = blk + lane < end ? h.entries[blk + lane] : 0;
    uint32_t e_nxt = blk + G + lane < end ? h.entries[blk + G + lane] : 0;
    auto row_of = [&](const uint32_t p, uint32_t &label) __attribute__((always_inline)) -> const float * {
        const uint32_t off = p - blk;
        const uint32_t e = (uint32_t)__shfl((int)(off < (uint32_t)G ? e_cur : e_nxt), (int)(off & (G - 1)), G);
        label = e >> 31;
        const float *row = partner_of<DIM>(h, e, partner_table, partner_hot, partner_base);
        return p < end ? row : idle;
    };
    float ring[D][V];
    uint32_t labels = 0;
#pragma unroll
    for (int i = 0; i < D; i++) {
        uint32_t label;
        load_row_at<DIM, G>(row_of(begin + i, label), lane, ring[i]);
        labels |= label << i;
    }
    for (uint32_t base = begin; __builtin_amdgcn_ballot_w64(base < end) != 0; base += D) {
        const uint32_t f = blk + 2 * G + lane;
        const uint32_t e_fut = h.entries[f < end ? f : (begin < end ? end - 1 : 0)];
#pragma unroll
        for (int i = 0; i < D; i++) {
            const uint32_t p = base + i;
            const bool positive = (labels >> i & 1u) != 0, active = p < end;
            const float(&c)[V] = ring[i];
            float partial = 0;
#pragma unroll
            for (int x = 0; x < V; x++) partial += own[x] * c[x];
            const float prob = sigmoidf(group_sum<G>(partial));
            const float gradient = positive ? prob - 1 : prob, weight = positive ? 1.0f : a.neg_weight;
#pragma unroll
            for (int x = 0; x < V; x++) {  // a step past the chain's end (another group of the wavefront is still working) leaves everything as it is
                float n1 = m1[x], n2 = m2[OPT == GVK_ADAM ? x : 0];
                const float step = update<OPT>(o, own[x], gradient * c[x], weight, n1, n2);
                own[x] = active ? own[x] - step : own[x];
                m1[x] = active ? n1 : m1[x];
                if constexpr (OPT == GVK_ADAM) m2[x] = active ? n2 : m2[x];
            }
            uint32_t label;
            load_row_at<DIM, G>(row_of(p + D, label), lane, ring[i]);
            labels = (labels & ~(1u << i)) | label << i;
        }
        if (base + D >= blk + G) {
            blk += G;
            e_cur = e_nxt;
            e_nxt = e_fut;
        }
    }
    if (begin < end) {
        store_row_at<DIM, G>(hub_to<DIM>(h, chain, 0), lane, own);
        store_row<DIM, G>(moment1, row_index, lane, m1);
        if constexpr (OPT == GVK_ADAM) store_row<DIM, G>(moment2, row_index, lane, reinterpret_cast<float(&)[V]>(m2));
    }
}

// grid: [chains, kHotBlock / G per block | pairs | rows without entries] — the chains, the launch's longest path, first
template <int DIM, int G, int OPT>
__global__ void __launch_bounds__(kHotBlock, train_waves(DIM / G, OPT, false)) train_hot_moment_kernel(const TrainArgs a, const HotArgs h) {
    const int b = blockIdx.x;
    if (b < h.long_blocks) train_moment_chains<DIM, G, OPT>(a, h, (uint32_t)b);
    else if (b < h.long_blocks + h.pair_blocks) train_pair<DIM, G, OPT, 0, 1, 1>(a, (b - h.long_blocks) * kHotBlock + threadIdx.x);
    else copy_idle_rows<DIM, G>(h, (uint32_t)(b - h.long_blocks - h.pair_blocks));
}

// HOT: 1 = the pairs read a hub row as the chains of their unit left it, 2 = on the straight line from where those chains
// found it to where they left it, at the sample's place in the unit (lerp)
// Built for four wavefronts per SIMD (128 registers; the short chains keep seven partner rows per lane group in flight; three
// at dims 256 and 512, sixteen floats of a row per lane): the chains and the pairs of a unit of the sizes this kernel trains (a
// part of a batch) are then resident side by side.
template <int DIM, int G, int KT, int HOT, int ROUNDS>
#if !defined(GVK_HOT_WAVES)
#define GVK_HOT_WAVES 4  // measurement knob: wavefronts per SIMD train_hot_kernel / chain_kernel are built for
#endif
__global__ void __launch_bounds__(kHotBlock, DIM / G > 12 ? 3 : GVK_HOT_WAVES) train_hot_kernel(const TrainArgs a, const HotArgs h) {
    // the grid: [long chains | pairs | short chains | idle rows] — the long chains, whose tasks wait for memory three times in
    // a row, are dispatched first, the bulk (the pairs) next; the short chains and the copies fill in behind
    const int b = blockIdx.x;
    GVK_STAMP_VALUE(h, 0, 0);
    GVK_STAMP(h, 1);  // the workgroup starts
    const int pairs_first = h.order == 2 ? 0 : (h.order == 1 ? h.long_blocks : h.long_blocks + h.short_blocks + h.copy_blocks);
    const int long_first = h.order == 2 ? h.pair_blocks : 0;
    const int short_first = h.order == 0 ? h.long_blocks : h.long_blocks + h.pair_blocks;
    if (b >= long_first && b < long_first + h.long_blocks) {
        if constexpr (ROUNDS != 0) train_long_chains_in_rounds<DIM, G>(a, h, b - long_first);
        else train_long_chains_one_round<DIM, G>(a, h, b - long_first);
    } else if (b >= pairs_first && b < pairs_first + h.pair_blocks) {
        GVK_STAMP_VALUE(h, 0, 3);
        train_pair<DIM, G, GVK_SGD, KT, 1, HOT>(a, (b - pairs_first) * kHotBlock + threadIdx.x);
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
                                                                const uint32_t cap, const int parts, const uint32_t first_unit) {
    extern __shared__ uint32_t bins[];  // [chains]
    __shared__ uint32_t wave_total[kListThreads / 64];
    __shared__ uint32_t long_count, short_count;
    const uint32_t chains = a.hot_vertex + a.hot_context;
    // list `unit` = part (unit % parts) of batch (unit / parts): samples [lo, hi) of the batch; a launch covers the units from first_unit on
    const int B = a.batch_size, k = a.k;
    const uint32_t unit = first_unit + blockIdx.x;
    const int batch = (int)(unit / (uint32_t)parts), lo = (int)(unit % (uint32_t)parts) * (B / parts), hi = lo + B / parts;
    const u32x2 *records = reinterpret_cast<const u32x2 *>(a.pairs) + (size_t)batch * B;
    uint32_t *chain_start = chain_start_all + (size_t)unit * (chains + 1);
    uint32_t *entries = entries_all + (size_t)unit * entry_capacity;
    uint32_t *long_list = long_all + (size_t)unit * 4 * (1 + (size_t)long_capacity);
    uint32_t *short_list = short_all + (size_t)unit * 16 * (1 + (size_t)chains);
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

// ---- the chain stream (gvk_train_episode_ahead) -----------------------------------------------------------------------------
//
// The same chains as a kernel of their own, on a stream of their own, a batch AHEAD of the pairs: a launch
// of train_hot_kernel lasts as long as its longest chain — record, row + entries, 16 dependent steps, compose, store: 10-12 us —
// while its pairs are done after 5-6; the chains are a dependency chain from unit to unit, the pairs are not.  So the chains of
// every unit run back to back on the chain stream (chain_kernel: the long and the short chains, nothing else) and the pairs of a
// batch follow on the caller's stream once that batch's chains are done (hot_pairs_kernel), while the chains of the next batch
// already run.  What makes that possible is where the hub rows live: not in whole mirrors (a mirror per unit in flight would have
// to be completed by copies of every row a unit did NOT touch), but in a ring of VERSIONS per row — a row's version advances by one
// in every unit that has entries for it; version v sits at slot v % ring_slots.  The work lists say which slot to read:
// hot_version_kernel counts the versions (ver[unit][row] = the row's slot after that unit), hot_slots_kernel writes them where they
// are needed — a chain's record (the slot its row is read at; it is stored one slot on), every entry whose partner is a hub row (the
// partner's slot BEFORE the unit: a sample between two hub rows updates both from the values the unit started with, as the mirrors
// did), and per sample the slots its pairs read its hub rows at (after the unit's chains).  Nothing is copied for rows without
// entries.  The ring holds 2 parts + 1 versions: the chains of batch b + 2 wait for the pairs of batch b, so between the oldest
// version a running pairs launch may read and the newest a chain writes lie at most 2 parts units.
template <int DIM, int G, int ROUNDS>
__global__ void __launch_bounds__(kHotBlock, DIM / G > 12 ? 3 : GVK_HOT_WAVES) chain_kernel(const TrainArgs a, const HotArgs h) {
    const int b = blockIdx.x;
    GVK_STAMP_VALUE(h, 0, 0);
    GVK_STAMP(h, 1);  // the workgroup starts
    if (b < h.long_blocks) {
        if constexpr (ROUNDS != 0) train_long_chains_in_rounds<DIM, G>(a, h, b);
        else train_long_chains_one_round<DIM, G>(a, h, b);
    } else {
        train_short_chains<DIM, G>(a, h, b - h.long_blocks);
    }
}

template <int DIM, int G, int KT>
__global__ void __launch_bounds__(kBlock, 4) hot_pairs_kernel(const TrainArgs a) {
    train_pair<DIM, G, GVK_SGD, KT, 1, 3>(a, blockIdx.x * kBlock + threadIdx.x);
}

// One launch for the chains of `units_in_launch` consecutive units AND the pairs of the units before them (gvk_train_episode_ahead with
// group > 1): grid [chains of the first unit | pairs | chains of the second unit | ...].  A launch of
// train_hot_kernel lasts as long as its unit's longest chain while its pairs are done in half the time; here the chains of the later
// units start as soon as their own row is there — published by the chain of the unit before, a workgroup of the same launch (await_row)
// — and their partners that are hub rows are read as the GROUP found them (hot_slots_kernel), so nothing else of the launch has to be
// waited for.  Workgroups are dispatched in grid order: whatever a chain waits for was dispatched before it.
template <int DIM, int G, int KT, int ROUNDS>
__global__ void __launch_bounds__(kHotBlock, DIM / G > 12 ? 3 : GVK_HOT_WAVES) train_group_kernel(const TrainArgs a, const HotArgs h) {
    const int per_unit = h.long_blocks + h.short_blocks;
    int b = blockIdx.x, sub = 0;
    if (b >= per_unit) {
        if (b < per_unit + h.pair_blocks) {
            train_pair<DIM, G, GVK_SGD, KT, 1, 3>(a, (b - per_unit) * kHotBlock + threadIdx.x);
            return;
        }
        b -= per_unit + h.pair_blocks;
        sub = 1 + b / per_unit, b %= per_unit;
    }
    if ((uint32_t)sub >= h.units_in_launch) return;
    HotArgs u = h;  // the unit's own work lists
    u.chain_start += (size_t)sub * h.start_stride, u.entries += (size_t)sub * h.entries_stride;
    u.long_list += (size_t)sub * h.long_stride, u.short_list += (size_t)sub * h.short_stride;
    u.unit = h.unit + (uint32_t)sub;
    if (b < h.long_blocks) {
        if constexpr (ROUNDS != 0) train_long_chains_in_rounds<DIM, G>(a, u, b);
        else train_long_chains_one_round<DIM, G>(a, u, b);
    } else {
        train_short_chains<DIM, G>(a, u, b - h.long_blocks);
    }
}

// ver[u][c] = slot of chain c's row after unit u: the number of units up to u that have entries for it, modulo the ring;
// last[u][c] = 1 + the last unit up to u that has entries for it (0: none in this call) — what a grouped launch's chains wait for
__global__ void __launch_bounds__(kBlock) hot_version_kernel(const uint32_t *chain_start_all, uint8_t *ver_all, uint16_t *last_all, const uint32_t chains,
                                                             const int units, const uint32_t ring_slots) {
    const uint32_t c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= chains) return;
    uint32_t v = 0, last = 0;
    for (int u = 0; u < units; u++) {
        const uint32_t *start = chain_start_all + (size_t)u * (chains + 1);
        if (start[c + 1] != start[c]) v = v + 1 == ring_slots ? 0u : v + 1, last = (uint32_t)u + 1;
        ver_all[(size_t)u * chains + c] = (uint8_t)v;
        last_all[(size_t)u * chains + c] = (uint16_t)last;
    }
}

// One workgroup per unit: the slots into the unit's work lists and the samples' slot words (see above)
__global__ void __launch_bounds__(kListThreads) hot_slots_kernel(TrainArgs a, const uint32_t first_batch_id, const uint32_t stride,
                                                                 const uint32_t *chain_start_all, uint32_t *entries_all, uint32_t *long_all,
                                                                 uint32_t *short_all, const uint8_t *ver_all, const uint16_t *last_all,
                                                                 uint32_t *slots_all, const uint32_t entry_capacity, const uint32_t long_capacity,
                                                                 const int parts, const int slot_words, const int group) {
    const uint32_t chains = a.hot_vertex + a.hot_context;
    const int B = a.batch_size, k = a.k, u = blockIdx.x;
    const int batch = u / parts, lo = (u % parts) * (B / parts), hi = lo + B / parts;
    const u32x2 *records = reinterpret_cast<const u32x2 *>(a.pairs) + (size_t)batch * B;
    const uint32_t *chain_start = chain_start_all + (size_t)u * (chains + 1);
    uint32_t *entries = entries_all + (size_t)u * entry_capacity;
    uint32_t *long_list = long_all + (size_t)u * 4 * (1 + (size_t)long_capacity);
    uint32_t *short_list = short_all + (size_t)u * 16 * (1 + (size_t)chains);
    // group > 1: the chains of `group` consecutive units run in one launch and read their hub PARTNERS as the group found them — at the
    // slots before the group's first unit —, their own row as the unit before left it (a chain of the same launch publishes it)
    const int first_of_group = u - u % group;
    const uint8_t *now = ver_all + (size_t)u * chains, *before = u ? now - chains : nullptr;
    const uint8_t *partners = first_of_group ? ver_all + (size_t)(first_of_group - 1) * chains : nullptr;
    const uint16_t *stored_by = u ? last_all + (size_t)(u - 1) * chains : nullptr;
    auto where_of = [&](const uint32_t chain) {  // a record's fourth word: slot | 1 + the unit that stored the row << 8 | it belongs to this launch << 31
        const uint32_t unit = stored_by ? stored_by[chain] : 0u;
        return (before ? (uint32_t)before[chain] : 0u) | unit << 8 | (unit > (uint32_t)first_of_group ? kRecordWaits : 0u);
    };
    a.batch_id = first_batch_id + (uint32_t)batch * stride;
    // the samples: where the unit's pairs read their hub rows — as the unit's chains left them
    for (int s = lo + threadIdx.x; s < hi; s += kListThreads) {
        const u32x2 pr = records[s];
        uint32_t *words = slots_all + ((size_t)batch * B + s) * slot_words;
        uint32_t word = (pr.y < a.hot_vertex ? (uint32_t)now[pr.y] : 0u) | (pr.x < a.hot_context ? (uint32_t)now[a.hot_vertex + pr.x] : 0u) << 8;
        for (int j = 0; j < k; j++) {
            const Draw d = negative_slot(a, (uint32_t)s, (uint32_t)j);
            const uint32_t n = resolve(a, d, load_entry(a, d));
            const int which = 2 + j;
            if ((which & 3) == 0) words[(which >> 2) - 1] = word, word = 0;
            word |= (n < a.hot_context ? (uint32_t)now[a.hot_vertex + n] : 0u) << (8 * (which & 3));
        }
        words[(k + 1) >> 2] = word;
    }
    // the entries: a hub partner is read at its slot BEFORE the unit
    const uint32_t total = chain_start[chains], head_entries = chain_start[a.hot_vertex];
    for (uint32_t e = threadIdx.x; e < total; e += kListThreads) {
        const uint32_t x = entries[e], label = x & 0x80000000u, id = x & 0x7fffffffu;
        const bool of_head = e < head_entries;  // the chains are listed by row, head rows first: this entry's partner is a context row
        const uint32_t hot = of_head ? a.hot_context : a.hot_vertex, base = of_head ? a.hot_vertex : 0u;
        entries[e] = id < hot ? (label | kEntryHub | (partners ? (uint32_t)partners[base + id] : 0u) << 16 | id) : x;
    }
    __syncthreads();
    __threadfence_block();
    // the records: the slot the chain's own row is read at, and (short chains) the entries as rewritten above
    const uint32_t long_count = long_list[0] < long_capacity ? long_list[0] : long_capacity, short_count = short_list[0];
    for (uint32_t j = threadIdx.x; j < long_count; j += kListThreads) {
        uint32_t *record = long_list + 4 + 4 * (size_t)j;
        record[3] = where_of(record[0]);
    }
    for (uint32_t r = threadIdx.x / 8; r < short_count; r += kListThreads / 8) {
        uint32_t *record = short_list + 16 + 16 * (size_t)r;
        const uint32_t n = record[1], first = record[2], i = threadIdx.x % 8;
        if (i == 7) record[3] = where_of(record[0]);
        if (i < n) record[4 + i] = entries[first + i];
    }
}

// The hub rows between the tables and the ring: into slot 0 (a call's first step; every version count starts at 0), and the
// tables' rows from each row's last version (its last step)
__global__ void __launch_bounds__(kBlock) hub_versions_kernel(float *vertex, float *context, float *ring, const uint8_t *last, const uint32_t hot_vertex,
                                                              const uint32_t hot_context, const int dim, const int to_ring) {
    const size_t quads = (size_t)dim / 4, head_quads = (size_t)hot_vertex * quads, all = head_quads + (size_t)hot_context * quads;
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= all) return;
    f32x4 *in_table = i < head_quads ? reinterpret_cast<f32x4 *>(vertex) + i : reinterpret_cast<f32x4 *>(context) + (i - head_quads);
    const size_t slot = to_ring || !last ? 0 : last[i / quads];
    f32x4 *in_ring = reinterpret_cast<f32x4 *>(ring) + slot * all + i;
    if (to_ring) *in_ring = *in_table;
    else *in_table = *in_ring;
}

// ---- hub rows: work lists + launch (train_hot_kernel) -----------------------------------------------------------------

struct HotLayout {
    size_t chain_start = 0, entries = 0, long_list = 0, short_list = 0, mirrors = 0, mirror_bytes = 0, bytes = 0;  // offsets into the workspace
    size_t versions = 0, slots = 0;  // versioned (gvk_ahead_*): ver[units][chains] bytes, the samples' slot words; `mirrors` is the ring
    size_t last = 0, published = 0;  // ... last[units][chains] (u16: 1 + the unit that stored a row last), published[chains]
    uint32_t chains = 0, entry_capacity = 0, long_capacity = 0, cap = 0, ring_slots = 0;
    int slot_words = 0;
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
HotLayout hot_layout(int dim, int batch_size, int k, uint32_t hot_vertex, uint32_t hot_context, int num_batch, int parts, int chain_cap,
                     bool versioned = false) {
    HotLayout l;
    l.chains = hot_vertex + hot_context;
    l.cap = chain_cap_for(chain_cap);
    const size_t samples = (size_t)num_batch * batch_size;
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
    if (versioned) {  // the ring of 2 parts + 1 versions per hub row in place of the mirrors (slot s at mirrors + s * chains * dim * 4: no padding between slots)
        l.ring_slots = 2u * (uint32_t)parts + 1;
        l.slot_words = (k + 2 + 3) / 4;
        l.mirror_bytes = (size_t)l.chains * dim * 4;
        l.versions = l.mirrors + align((size_t)l.ring_slots * l.mirror_bytes);
        l.slots = l.versions + align((size_t)num_batch * l.chains);
        l.last = l.slots + align(samples * l.slot_words * 4);
        l.published = l.last + align((size_t)num_batch * l.chains * 2);
        l.bytes = l.published + align((size_t)l.chains * 4);
    }
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

HotKernel pick_hot(int dim, int k, int lerp, int rounds) {
#define GVK_HOT_R(D, GG, R)                                                                                         \
        return k == 1 ? (lerp ? train_hot_kernel<D, GG, 1, 2, R> : train_hot_kernel<D, GG, 1, 1, R>)                \
                      : (lerp ? train_hot_kernel<D, GG, 0, 2, R> : train_hot_kernel<D, GG, 0, 1, R>);
#define GVK_HOT(D, GG)                                                                                              \
    case D:                                                                                                         \
        if (rounds) { GVK_HOT_R(D, GG, 1) }                                                                         \
        GVK_HOT_R(D, GG, 0)
    switch (dim) {
        GVK_HOT(32, 8) GVK_HOT(64, 16) GVK_HOT(96, 8) GVK_HOT(128, 16) GVK_HOT(256, 16) GVK_HOT(512, 32)
    }
#undef GVK_HOT_R
#undef GVK_HOT
    return nullptr;
}

HotKernel pick_hot_moment(int dim, int opt) {
#define GVK_HOT_M(D, GG)                                                                     \
    case D:                                                                                  \
        switch (opt) {                                                                       \
            case GVK_MOMENTUM: return train_hot_moment_kernel<D, GG, GVK_MOMENTUM>;          \
            case GVK_ADAGRAD: return train_hot_moment_kernel<D, GG, GVK_ADAGRAD>;            \
            case GVK_RMSPROP: return train_hot_moment_kernel<D, GG, GVK_RMSPROP>;            \
            case GVK_ADAM: return train_hot_moment_kernel<D, GG, GVK_ADAM>;                  \
        }                                                                                    \
        return nullptr;
    switch (dim) { GVK_HOT_M(32, 8) GVK_HOT_M(64, 16) GVK_HOT_M(96, 8) GVK_HOT_M(128, 16) GVK_HOT_M(256, 16) GVK_HOT_M(512, 32) }
#undef GVK_HOT_M
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
    return gvk_hot_build_sliced(stream, dim, workspace, workspace_bytes, pool, batch_size, num_batch, num_negative, negative, first_batch_id, batch_id_stride,
                                hot_vertex, hot_context, parts, chain_cap, 0);
}

int gvk_hot_build_sliced(void *stream, int dim, void *workspace, size_t workspace_bytes, const uint32_t *pool, int batch_size, int num_batch,
                         int num_negative, const gvk_negative_source *negative, uint32_t first_batch_id, uint32_t batch_id_stride,
                         uint32_t hot_vertex, uint32_t hot_context, int parts, int chain_cap, int units_per_launch) {
    if (units_per_launch < 0) return fail(GVK_EINVAL, "gvk_hot_build_sliced: negative units_per_launch");
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
    const int units = num_batch * parts, slice = units_per_launch > 0 && units_per_launch < units ? units_per_launch : units;
    for (int first = 0; first < units; first += slice)
        hipLaunchKernelGGL(hot_list_kernel, dim3((unsigned)std::min(slice, units - first)), dim3(kListThreads), lds, (hipStream_t)stream, a,
                           first_batch_id, batch_id_stride, reinterpret_cast<uint32_t *>(base + l.chain_start),
                           reinterpret_cast<uint32_t *>(base + l.entries), reinterpret_cast<uint32_t *>(base + l.long_list),
                           reinterpret_cast<uint32_t *>(base + l.short_list), l.entry_capacity, l.long_capacity, l.cap, parts, (uint32_t)first);
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
    const bool moments = optimizer->type != GVK_SGD;  // a moment optimizer: every chain one sequential task (train_moment_chains)
    if (moments && (form & (GVK_HOT_LERP | GVK_HOT_ROUNDS))) return fail(GVK_EINVAL, "gvk_train_episode_hot: lerp / rounds are forms of the SGD chains");
    if (negative->negatives) return fail(GVK_EINVAL, "gvk_train_episode_hot draws negatives on device");
    if (hot_vertex > tables->n_vertex || hot_context > tables->n_context)
        return fail(GVK_EINVAL, "gvk_train_episode_hot: more hub rows than table rows");
    if (chain_cap < 0) return fail(GVK_EINVAL, "gvk_train_episode_hot: negative chain_cap");
    if (form & ~(GVK_HOT_SERIALIZED | GVK_HOT_LERP | GVK_HOT_ROUNDS)) return fail(GVK_EINVAL, "gvk_train_episode_hot: unknown form bits");
    const HotLayout l = hot_layout(dim, batch_size, num_negative, hot_vertex, hot_context, workspace_batches, parts, chain_cap);
    if (!workspace || workspace_bytes < l.bytes) return fail(GVK_EINVAL, "gvk_train_episode_hot: workspace too small (gvk_hot_plan)");
    const bool lerp = (form & GVK_HOT_LERP) != 0, serialized = (form & GVK_HOT_SERIALIZED) != 0 || g_hot_serialized != 0;
    // rounds: asked for by the caller (GVK_HOT_ROUNDS) or forced either way by the measurement knob GVK_TUNE_ROUND_STEPS (0: never)
    const uint32_t round_steps = g_round_steps >= 0 ? (uint32_t)g_round_steps : ((form & GVK_HOT_ROUNDS) ? (uint32_t)GVK_HOT_ROUND_STEPS : 0u);
    const HotKernel kernel = moments ? pick_hot_moment(dim, optimizer->type) : pick_hot(dim, num_negative, lerp, round_steps != 0);
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
    a.vm1 = tables->vertex_moment1; a.cm1 = tables->context_moment1;
    a.vm2 = tables->vertex_moment2; a.cm2 = tables->context_moment2;
    a.hp0 = optimizer->hp0; a.hp1 = optimizer->hp1; a.eps = optimizer->epsilon;
    HotArgs h;
    memset(&h, 0, sizeof(h));
    h.chains = l.chains; h.long_capacity = l.long_capacity; h.cap = l.cap;
    h.round_steps = round_steps;
    const int groups = kHotBlock / lanes;
    // (a moment optimizer: every chain is one lane group's, kHotBlock / lanes chains per block — listed as the "long" blocks of the grid)
    const int short_blocks = moments ? 0 : (int)((l.chains + groups - 1) / groups);
    const int long_blocks = moments ? (int)((l.chains + groups - 1) / groups) : (int)std::min<uint32_t>(l.long_capacity, (uint32_t)kLongBlocks);
    const int copy_blocks = (int)((l.chains + 4 * groups - 1) / (4 * groups));
    // the unit of work is a PART of a batch (parts = 1: the batch): unit u = part u % parts of batch u / parts
    const int part_size = batch_size / parts, units = num_batches * parts;
    const unsigned pair_blocks = (unsigned)(((int64_t)part_size * lanes + kHotBlock - 1) / kHotBlock);
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
        if (grid) hipLaunchKernelGGL(kernel, dim3(grid), dim3(kHotBlock), 0, (hipStream_t)stream, a, h);
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


// ---- the chain stream: gvk_ahead_plan / gvk_ahead_build / gvk_train_episode_ahead (include/gvk.h) ---------------------------

int gvk_ahead_plan(int dim, int batch_size, int num_negative, uint32_t hot_vertex, uint32_t hot_context, int num_batch, int parts,
                   int chain_cap, size_t *bytes) {
    if (!bytes) return fail(GVK_EINVAL, "gvk_ahead_plan: bytes is null");
    int rc = validate_hot("gvk_ahead_plan", dim, batch_size, num_negative, hot_vertex, hot_context, num_batch, parts);
    if (rc != GVK_OK) return rc;
    if (chain_cap < 0) return fail(GVK_EINVAL, "gvk_ahead_plan: negative chain_cap");
    if (parts > 127) return fail(GVK_EINVAL, "gvk_ahead_plan: at most 127 parts (a slot is a byte, the ring holds 2 parts + 1 versions)");
    *bytes = hot_layout(dim, batch_size, num_negative, hot_vertex, hot_context, num_batch, parts, chain_cap, true).bytes;
    return GVK_OK;
}

int gvk_ahead_build(void *stream, int dim, void *workspace, size_t workspace_bytes, const uint32_t *pool, int batch_size, int num_batch,
                    int num_negative, const gvk_negative_source *negative, uint32_t first_batch_id, uint32_t batch_id_stride,
                    uint32_t hot_vertex, uint32_t hot_context, int parts, int chain_cap, int group) {
    if (parts > 127) return fail(GVK_EINVAL, "gvk_ahead_build: at most 127 parts");
    if (group < 1 || parts % group) return fail(GVK_EINVAL, "gvk_ahead_build: group must divide parts");
    if ((int64_t)num_batch * parts > 65534) return fail(GVK_EINVAL, "gvk_ahead_build: at most 65534 units per call (a unit's index is 16 bits of a record)");
    if (num_batch <= 0) return num_batch < 0 ? fail(GVK_EINVAL, "gvk_ahead_build: negative num_batch") : GVK_OK;
    if (hot_vertex > 0x7fffu || hot_context > 0x7fffu) return fail(GVK_EINVAL, "gvk_ahead_build: at most 32767 hub rows per table");
    const HotLayout l = hot_layout(dim, batch_size, num_negative, hot_vertex, hot_context, num_batch, parts, chain_cap, true);
    if (workspace_bytes < l.bytes) return gvk_fail(GVK_EINVAL, "gvk_ahead_build: workspace holds %zu bytes, %zu needed", workspace_bytes, l.bytes);
    // the work lists as gvk_hot_build writes them (the layouts share their first part) ...
    int rc = gvk_hot_build(stream, dim, workspace, workspace_bytes, pool, batch_size, num_batch, num_negative, negative, first_batch_id,
                           batch_id_stride, hot_vertex, hot_context, parts, chain_cap);
    if (rc != GVK_OK) return rc;
    // ... then the versions and the slots
    TrainArgs a;
    memset(&a, 0, sizeof(a));
    a.pairs = pool;
    fill_negative(a, negative);
    a.batch_size = batch_size; a.k = num_negative;
    a.hot_vertex = hot_vertex; a.hot_context = hot_context;
    char *base = static_cast<char *>(workspace);
    const int units = num_batch * parts;
    hipLaunchKernelGGL(hot_version_kernel, dim3((l.chains + kBlock - 1) / kBlock), dim3(kBlock), 0, (hipStream_t)stream,
                       reinterpret_cast<const uint32_t *>(base + l.chain_start), reinterpret_cast<uint8_t *>(base + l.versions),
                       reinterpret_cast<uint16_t *>(base + l.last), l.chains, units, l.ring_slots);
    hipLaunchKernelGGL(hot_slots_kernel, dim3((unsigned)units), dim3(kListThreads), 0, (hipStream_t)stream, a, first_batch_id, batch_id_stride,
                       reinterpret_cast<const uint32_t *>(base + l.chain_start), reinterpret_cast<uint32_t *>(base + l.entries),
                       reinterpret_cast<uint32_t *>(base + l.long_list), reinterpret_cast<uint32_t *>(base + l.short_list),
                       reinterpret_cast<const uint8_t *>(base + l.versions), reinterpret_cast<const uint16_t *>(base + l.last),
                       reinterpret_cast<uint32_t *>(base + l.slots), l.entry_capacity, l.long_capacity, parts, l.slot_words, group);
    return check_launch("gvk_ahead_build");
}

namespace {

typedef void (*ChainKernel)(const TrainArgs, const HotArgs);
typedef void (*HotPairsKernel)(const TrainArgs);

ChainKernel pick_chain(int dim, int rounds) {
#define GVK_CHAIN(D, GG) case D: return rounds ? chain_kernel<D, GG, 1> : chain_kernel<D, GG, 0>;
    switch (dim) { GVK_CHAIN(32, 8) GVK_CHAIN(64, 16) GVK_CHAIN(96, 8) GVK_CHAIN(128, 16) GVK_CHAIN(256, 16) GVK_CHAIN(512, 32) }
#undef GVK_CHAIN
    return nullptr;
}

ChainKernel pick_group(int dim, int k, int rounds) {
#define GVK_GROUP(D, GG)                                                                                   \
    case D:                                                                                                \
        return k == 1 ? (rounds ? train_group_kernel<D, GG, 1, 1> : train_group_kernel<D, GG, 1, 0>)       \
                      : (rounds ? train_group_kernel<D, GG, 0, 1> : train_group_kernel<D, GG, 0, 0>);
    switch (dim) { GVK_GROUP(32, 8) GVK_GROUP(64, 16) GVK_GROUP(96, 8) GVK_GROUP(128, 16) GVK_GROUP(256, 16) GVK_GROUP(512, 32) }
#undef GVK_GROUP
    return nullptr;
}

HotPairsKernel pick_hot_pairs(int dim, int k) {
#define GVK_HOT_PAIRS(D, GG) case D: return k == 1 ? hot_pairs_kernel<D, GG, 1> : hot_pairs_kernel<D, GG, 0>;
    switch (dim) { GVK_HOT_PAIRS(32, 8) GVK_HOT_PAIRS(64, 16) GVK_HOT_PAIRS(96, 8) GVK_HOT_PAIRS(128, 16) GVK_HOT_PAIRS(256, 16) GVK_HOT_PAIRS(512, 32) }
#undef GVK_HOT_PAIRS
    return nullptr;
}

// events of a chain stream, made once: "the chains of batch b are done" / "the pairs of batch b are done", four of each in rotation
struct AheadEvents {
    hipEvent_t chains_done[4], pairs_done[4], start;
};
std::mutex g_ahead_mutex;
std::map<void *, AheadEvents> g_ahead_events;

int ahead_events(void *chain_stream, AheadEvents **out) {
    std::lock_guard<std::mutex> lock(g_ahead_mutex);
    auto it = g_ahead_events.find(chain_stream);
    if (it == g_ahead_events.end()) {
        AheadEvents e;
        for (int i = 0; i < 4; i++)
            if (hipEventCreateWithFlags(&e.chains_done[i], hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess ||
                hipEventCreateWithFlags(&e.pairs_done[i], hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess)
                return fail(GVK_EHIP, "gvk_train_episode_ahead: hipEventCreate failed");
        if (hipEventCreateWithFlags(&e.start, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) return fail(GVK_EHIP, "gvk_train_episode_ahead: hipEventCreate failed");
        it = g_ahead_events.emplace(chain_stream, e).first;
    }
    *out = &it->second;
    return GVK_OK;
}

}  // namespace

void gvk_ahead_release(void *chain_stream) {  // before the stream is destroyed: the events made for it
    std::lock_guard<std::mutex> lock(g_ahead_mutex);
    auto it = g_ahead_events.find(chain_stream);
    if (it == g_ahead_events.end()) return;
    for (int i = 0; i < 4; i++) (void)hipEventDestroy(it->second.chains_done[i]), (void)hipEventDestroy(it->second.pairs_done[i]);
    (void)hipEventDestroy(it->second.start);
    g_ahead_events.erase(it);
}

int gvk_train_episode_ahead(void *stream, void *chain_stream, int dim, const gvk_optimizer *optimizer, int linear_schedule,
                            const gvk_tables *tables, const uint32_t *pairs, const gvk_negative_source *negative, uint32_t first_batch_id,
                            uint32_t batch_id_stride, uint32_t total_batches, int num_batches, float *loss, int batch_size,
                            int num_negative, float negative_weight, void *workspace, size_t workspace_bytes, uint32_t hot_vertex,
                            uint32_t hot_context, int workspace_batches, int parts, int chain_cap, int pair_launches, int group, int form) {
    if (num_batches < 0 || num_batches > workspace_batches) return fail(GVK_EINVAL, "gvk_train_episode_ahead: more batches than the work lists cover");
    if (group < 1 || parts % group) return fail(GVK_EINVAL, "gvk_train_episode_ahead: group must divide parts");
    int rc = validate_train(dim, optimizer, tables, pairs, negative, loss, batch_size, num_negative);
    if (rc <= 0) return rc;
    rc = validate_hot("gvk_train_episode_ahead", dim, batch_size, num_negative, hot_vertex, hot_context, workspace_batches, parts);
    if (rc != GVK_OK) return rc;
    if (optimizer->type != GVK_SGD) return fail(GVK_EINVAL, "gvk_train_episode_ahead: chains exist for SGD only");
    if (negative->negatives) return fail(GVK_EINVAL, "gvk_train_episode_ahead draws negatives on device");
    if (hot_vertex > tables->n_vertex || hot_context > tables->n_context)
        return fail(GVK_EINVAL, "gvk_train_episode_ahead: more hub rows than table rows");
    if (tables->n_vertex > 0x3fffffffu || tables->n_context > 0x3fffffffu)
        return fail(GVK_EINVAL, "gvk_train_episode_ahead: at most 2^30 rows per table (an entry's id field)");
    if (chain_cap < 0 || parts > 127) return fail(GVK_EINVAL, "gvk_train_episode_ahead: bad chain_cap / parts");
    if (form & ~(GVK_HOT_SERIALIZED | GVK_HOT_ROUNDS)) return fail(GVK_EINVAL, "gvk_train_episode_ahead: unknown form bits");
    if (pair_launches <= 0) pair_launches = parts;
    if (parts % pair_launches) return fail(GVK_EINVAL, "gvk_train_episode_ahead: pair_launches must divide parts");
    const bool serialized = (form & GVK_HOT_SERIALIZED) != 0 || g_hot_serialized != 0;
    const bool grouped = group > 1;  // one stream: a launch = the chains of `group` units + the pairs of the group before (train_group_kernel)
    if (!serialized && !grouped && (!chain_stream || chain_stream == stream))
        return fail(GVK_EINVAL, "gvk_train_episode_ahead: the chains need a stream of their own (chain_stream)");
    const HotLayout l = hot_layout(dim, batch_size, num_negative, hot_vertex, hot_context, workspace_batches, parts, chain_cap, true);
    if (!workspace || workspace_bytes < l.bytes) return fail(GVK_EINVAL, "gvk_train_episode_ahead: workspace too small (gvk_ahead_plan)");
    const uint32_t round_steps = g_round_steps >= 0 ? (uint32_t)g_round_steps : ((form & GVK_HOT_ROUNDS) ? (uint32_t)GVK_HOT_ROUND_STEPS : 0u);
    const ChainKernel chain = pick_chain(dim, round_steps != 0), fused = pick_group(dim, num_negative, round_steps != 0);
    const HotPairsKernel pair = pick_hot_pairs(dim, num_negative);
    if (!chain || !pair || !fused) return fail(GVK_EDIM, "gvk_train_episode_ahead: no kernel for this dim");
    if (num_batches == 0) return GVK_OK;
    const int lanes = default_lanes(dim);
    char *base = static_cast<char *>(workspace);
    float *ring = reinterpret_cast<float *>(base + l.mirrors);
    const uint8_t *versions = reinterpret_cast<const uint8_t *>(base + l.versions);
    TrainArgs a;
    memset(&a, 0, sizeof(a));
    a.vertex = tables->vertex; a.context = tables->context;
    a.loss = loss;
    fill_negative(a, negative);
    a.batch_size = batch_size; a.k = num_negative; a.run_cap = 1;
    a.wd = optimizer->weight_decay; a.neg_weight = negative_weight;
    a.hot_vertex = hot_vertex; a.hot_context = hot_context;
    a.hub_now = ring;
    a.slot_words = l.slot_words;
    HotArgs h;
    memset(&h, 0, sizeof(h));
    h.chains = l.chains; h.long_capacity = l.long_capacity; h.cap = l.cap;
    h.round_steps = round_steps;
    h.from = ring, h.to = ring;
    h.slot_stride = l.chains, h.ring_slots = l.ring_slots, h.versioned = 1;
    h.grouped = grouped, h.published = reinterpret_cast<uint32_t *>(base + l.published);
    h.start_stride = l.chains + 1, h.entries_stride = l.entry_capacity, h.long_stride = 4 * (1 + l.long_capacity), h.short_stride = 16 * (1 + l.chains);
    const int groups = kHotBlock / lanes;
    h.short_blocks = (int)((l.chains + groups - 1) / groups);
    h.long_blocks = (int)std::min<uint32_t>(l.long_capacity, (uint32_t)kLongBlocks);
    const int part_size = batch_size / parts;
    const bool chains_only = hot_vertex == tables->n_vertex && hot_context == tables->n_context;
    auto lr_of = [&](int i) {
        const uint32_t id = first_batch_id + (uint32_t)i * batch_id_stride;
        float scale = 1;
        if (linear_schedule) {  // optimizer.h:77-79
            scale = 1 - float(int(id)) / int(total_batches);
            if (scale < 1e-4f) scale = 1e-4f;
        }
        return optimizer->lr * scale;
    };
    auto chains_at = [&](int u) {  // the work lists, learning rate and index of unit u
        h.chain_start = reinterpret_cast<const uint32_t *>(base + l.chain_start) + (size_t)u * (l.chains + 1);
        h.entries = reinterpret_cast<const uint32_t *>(base + l.entries) + (size_t)u * l.entry_capacity;
        h.long_list = reinterpret_cast<const uint32_t *>(base + l.long_list) + (size_t)u * 4 * (1 + (size_t)l.long_capacity);
        h.short_list = reinterpret_cast<const uint32_t *>(base + l.short_list) + (size_t)u * 16 * (1 + (size_t)l.chains);
        h.lr = lr_of(u / parts);
        h.log2_decay_positive = (float)std::log2(1.0 - (double)h.lr * a.wd);
        h.log2_decay_negative = (float)std::log2(1.0 - (double)h.lr * a.neg_weight * a.wd);
        h.unit = (uint32_t)u;
    };
    auto chains_of = [&](int u, hipStream_t on) {  // the chains of unit u as a launch of their own
        chains_at(u);
        h.units_in_launch = 1;
        a.lr = h.lr;
        hipLaunchKernelGGL(chain, dim3((unsigned)(h.long_blocks + h.short_blocks)), dim3(kHotBlock), 0, on, a, h);
    };
    auto pairs_at = [&](int i, int first_part, int count) {  // the pairs of parts [first_part, first_part + count) of batch i: their arguments
        a.lr = lr_of(i);
        a.batch_id = first_batch_id + (uint32_t)i * batch_id_stride;
        a.pairs = pairs + (size_t)i * batch_size * 2;
        a.slots = reinterpret_cast<const uint32_t *>(base + l.slots) + (size_t)i * batch_size * l.slot_words;
        a.first_sample = first_part * part_size;
        a.batch_size = a.first_sample + count * part_size;
        return (unsigned)(((int64_t)count * part_size * lanes + kBlock - 1) / kBlock);
    };
    auto pairs_of = [&](int i, int first_part, int count, hipStream_t on) {  // the pairs of parts [first_part, first_part + count) of batch i
        if (chains_only && i != num_batches - 1) return;
        a.lr = lr_of(i);
        a.batch_id = first_batch_id + (uint32_t)i * batch_id_stride;
        a.pairs = pairs + (size_t)i * batch_size * 2;
        a.slots = reinterpret_cast<const uint32_t *>(base + l.slots) + (size_t)i * batch_size * l.slot_words;
        a.first_sample = first_part * part_size;
        a.batch_size = a.first_sample + count * part_size;
        const unsigned blocks = (unsigned)(((int64_t)count * part_size * lanes + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(pair, dim3(blocks), dim3(kBlock), 0, on, a);
    };
    const hipStream_t main = (hipStream_t)stream, side = (hipStream_t)chain_stream;
    const unsigned ring_blocks = (unsigned)(((size_t)l.chains * (size_t)(dim / 4) + kBlock - 1) / kBlock);
    // the hub rows enter the ring at slot 0
    hipLaunchKernelGGL(hub_versions_kernel, dim3(ring_blocks), dim3(kBlock), 0, main, a.vertex, a.context, ring, nullptr, hot_vertex, hot_context, dim, 1);
    if (grouped && hipMemsetAsync(base + l.published, 0, (size_t)l.chains * 4, main) != hipSuccess) return fail(GVK_EHIP, "gvk_train_episode_ahead: memset failed");
    auto launch_group = [&](int first_unit, bool with_chains, int pairs_first_unit) {  // the chains of units [first_unit, + group) and the pairs of the group before
        h.pair_blocks = 0, h.units_in_launch = 0;
        if (with_chains) chains_at(first_unit), h.units_in_launch = (uint32_t)group;
        if (pairs_first_unit >= 0 && (!chains_only || pairs_first_unit / parts == num_batches - 1))
            pairs_at(pairs_first_unit / parts, pairs_first_unit % parts, group), h.pair_blocks = (int)(((int64_t)group * part_size * lanes + kHotBlock - 1) / kHotBlock);
        const unsigned per_unit = (unsigned)(h.long_blocks + h.short_blocks);
        const unsigned grid = with_chains ? per_unit * (unsigned)group + (unsigned)h.pair_blocks : per_unit + (unsigned)h.pair_blocks;
        hipLaunchKernelGGL(fused, dim3(grid), dim3(kHotBlock), 0, main, a, h);
    };
    if (serialized && grouped) {  // tests: per group the chains (one launch: the later units wait for the rows of the earlier ones), then its pairs
        for (int u = 0; u < num_batches * parts; u += group) {
            launch_group(u, true, -1);
            launch_group(u, false, u);
        }
    } else if (grouped) {
        launch_group(0, true, -1);
        for (int u = 0; u < num_batches * parts; u += group) launch_group(u + group, u + group < num_batches * parts, u);
    } else if (serialized) {  // tests: per unit the chains, then the pairs, on one stream — a pure function of the work lists
        for (int u = 0; u < num_batches * parts; u++) {
            chains_of(u, main);
            pairs_of(u / parts, u % parts, 1, main);
        }
    } else {
        AheadEvents *events = nullptr;
        rc = ahead_events(chain_stream, &events);
        if (rc != GVK_OK) return rc;
        // what the caller put on `stream` before this call (the work lists, the tables) comes before the chains, too
        if (hipEventRecord(events->start, main) != hipSuccess || hipStreamWaitEvent(side, events->start, 0) != hipSuccess)
            return fail(GVK_EHIP, "gvk_train_episode_ahead: event record / wait failed");
        const int per_launch = parts / pair_launches;
        const auto host_t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < num_batches; i++) {
            // the chains of batch i wait for the pairs of batch i - 2: what those pairs read of the ring may be overwritten now.
            // GVK_AHEAD_THROTTLE=host (measurement): the HOST waits instead, so that the chain stream carries no barrier packet
            if (i >= 2) {
                static const bool on_host = getenv("GVK_AHEAD_THROTTLE") && !strcmp(getenv("GVK_AHEAD_THROTTLE"), "host");
                const hipError_t e = on_host ? hipEventSynchronize(events->pairs_done[(i - 2) & 3]) : hipStreamWaitEvent(side, events->pairs_done[(i - 2) & 3], 0);
                if (e != hipSuccess) return fail(GVK_EHIP, "gvk_train_episode_ahead: event wait failed");
            }
            for (int q = 0; q < parts; q++) chains_of(i * parts + q, side);
            if (hipEventRecord(events->chains_done[i & 3], side) != hipSuccess || hipStreamWaitEvent(main, events->chains_done[i & 3], 0) != hipSuccess)
                return fail(GVK_EHIP, "gvk_train_episode_ahead: event record / wait failed");
            for (int q = 0; q < parts; q += per_launch) pairs_of(i, q, per_launch, main);
            if (hipEventRecord(events->pairs_done[i & 3], main) != hipSuccess) return fail(GVK_EHIP, "gvk_train_episode_ahead: event record failed");
        }
        if (getenv("GVK_AHEAD_DEBUG"))  // measurement: what the host pays to enqueue a batch
            fprintf(stderr, "gvk_train_episode_ahead: %d batches enqueued in %.1f us per batch (%d + %d launches, 4 event calls each)\n", num_batches,
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - host_t0).count() / num_batches, parts, pair_launches);
    }
    // ... and leave it: the tables' hub rows = each row's last version (the pairs of the last batch have waited for its chains)
    hipLaunchKernelGGL(hub_versions_kernel, dim3(ring_blocks), dim3(kBlock), 0, main, a.vertex, a.context, ring,
                       versions + (size_t)(num_batches * parts - 1) * l.chains, hot_vertex, hot_context, dim, 0);
    return check_launch("gvk_train_episode_ahead");
}

}  // extern "C"
