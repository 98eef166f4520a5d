// gvk_pairs.hip — the per-pair training kernels (train_kernel: gvk_device.hpp; train_runs_kernel: runs of same-head pairs), the
// prediction and probe kernels, the choice of kernel by table size and their C-ABI launchers (include/gvk.h: gvk_train,
// gvk_train_episode, gvk_predict, gvk_probe_row_traffic, gvk_describe_train, gvk_train_launches).  The A/B kernels of the
// measurement library (make ab, -DGVK_AB_BUILDS) live here too, behind their guard.
#include "gvk_device.hpp"

namespace {

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

// ---- predict / probe kernels ------------------------------------------------------------------------------

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

// ---- dispatch ----------------------------------------------------------------------------------------------

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

int run_cap_for(int batch_size) {
    (void)batch_size;
    return g_run_cap > 0 ? g_run_cap : kRunCap;
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

}  // namespace

extern "C" {

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

int gvk_train_launches(int batch_size, uint32_t n_vertex) { return launches_for(batch_size, n_vertex); }

}  // extern "C"
