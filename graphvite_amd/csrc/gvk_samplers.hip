// gvk_samplers.hip — draws on the device: alias tables (gvk_alias_sample), the negatives of a batch as the training kernels draw
// them (gvk_negative_draw, gvk_negative_draw_classes), and the positive samples of section 8(f4): edge draws and random walks
// (gvk_sample_pairs, gvk_sample_edges, gvk_sample_walks, gvk_sample_walks_blocks; include/gvk.h).
#include "gvk_device.hpp"

namespace {

// ---- alias / sampling kernels ---------------------------------------------------------------------------------

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
// A pool is cut into `stripes` stripes with one slot counter each and a wavefront appends to stripes: a single counter
// per block would take every atomic of the launch on P * P addresses (measured: 0.36 G pairs/s at 16 blocks), striped
// they spread over a few hundred times as many.  counters[b][stripe] keeps counting past the stripe's capacity, so the
// caller sees each block's share and which stripes are full.
// The pseudo shuffle (graph.cuh:362-364,439-441) keeps the pairs of a walk that share a row — pair i and pairs i + 1 .. i + 2 aug
// - 1 — out of one another's batch: pair i goes to part i % sb of the pool.  A wavefront's 64 walks append in lock step, so
// here the part is chosen by the pair's INDEX IN ITS WALK, not by the slot it is handed: pair i of wavefront w goes to
// stripe (w + (i % sb) * (stripes / sb)) % stripes — capacity / sb records away from pair i + 1.  (Choosing the part by
// the slot, as a sequential sampler may, left a walk's pairs of a block a few slots apart in one launch; profiles/r5/experiments/
// r5_fs_shuffle_base.txt: +0.007 link-prediction AUC against the sequential loop on LINE with augmentation_step 2.)

struct BlockPools {
    u32x2 *pools;
    const uint64_t *offsets;  // [P * P] first pair of the block's pool, or ~0: not collected
    uint32_t *counters;       // [P * P][stripes]
    const int32_t *part;      // [num_vertex]
    const float *accept;      // [P * P] or null: a pair of block b is kept with this probability (gvk_sample_walks_blocks_thinned)
    uint32_t capacity, stripes, stripe_capacity, sb;
    int P;
};

// uniform in [0, 1) from (walk, pair index in the walk, seed): which pairs of a walk a thinned block keeps (include/gvk.h)
__device__ __forceinline__ float thinning_uniform(uint64_t walk, uint64_t i, uint64_t seed) {
    uint32_t h = (uint32_t)walk ^ (uint32_t)(walk >> 32) * 0x85ebca6bu ^ (uint32_t)i * 0x9e3779b9u ^ (uint32_t)seed * 0xc2b2ae35u;
    h ^= h >> 16, h *= 0x85ebca6bu, h ^= h >> 13, h *= 0xc2b2ae35u, h ^= h >> 16;
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

__global__ void __launch_bounds__(kBlock) sample_walks_blocks_kernel(const gvk_walk_graph g, const BlockPools b, uint64_t seed,
                                                                     uint64_t first_walk, int L, int aug, uint64_t pairs_per_walk,
                                                                     uint64_t num_walks) {
    const uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= num_walks) return;
    const uint32_t wave = (uint32_t)((t / 64) % b.stripes), apart = b.stripes / b.sb > 0 ? b.stripes / b.sb : 1;
    walk_pairs(g, seed, first_walk + t, pairs_per_walk, L, aug, [&](uint32_t head, uint32_t tail, uint64_t i) {
        const int block = b.part[head] * b.P + b.part[tail];
        const uint64_t first = b.offsets[block];
        if (first == ~(uint64_t)0) return;
        if (b.accept && thinning_uniform(first_walk + t, i, seed) >= b.accept[block]) return;
        const uint32_t stripe = (wave + (uint32_t)(i % b.sb) * apart) % b.stripes;
        const uint32_t slot = atomicAdd(b.counters + (size_t)block * b.stripes + stripe, 1u);
        if (slot >= b.stripe_capacity) return;
        u32x2 record = {g.local[tail], g.local[head]};
        __builtin_nontemporal_store(record, b.pools + first + ((size_t)stripe * b.stripe_capacity + slot));
    });
}

}  // namespace

extern "C" {

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
    return gvk_sample_walks_blocks_thinned(stream, graph, part, num_partition, seed, first_walk, num_walks, pools, offsets, counters, capacity,
                                           num_stripe, walk_length, augmentation_step, shuffle_base, nullptr);
}

int gvk_sample_walks_blocks_thinned(void *stream, const gvk_walk_graph *graph, const int32_t *part, int num_partition, uint64_t seed,
                                    uint64_t first_walk, uint64_t num_walks, uint32_t *pools, const uint64_t *offsets,
                                    uint32_t *counters, uint32_t capacity, int num_stripe, int walk_length, int augmentation_step,
                                    int shuffle_base, const float *accept) {
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
    b.pools = reinterpret_cast<u32x2 *>(pools), b.offsets = offsets, b.counters = counters, b.part = part, b.accept = accept;
    b.capacity = capacity, b.stripes = (uint32_t)num_stripe, b.stripe_capacity = capacity / (uint32_t)num_stripe, b.sb = (uint32_t)shuffle_base, b.P = num_partition;
    hipLaunchKernelGGL(sample_walks_blocks_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, *graph, b, seed,
                       first_walk, walk_length, augmentation_step, per_walk, num_walks);
    return check_launch("gvk_sample_walks_blocks");
}

}  // extern "C"
