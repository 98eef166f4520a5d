/* gvk.h — C ABI of the MI355X (gfx950) node-embedding training kernels ("gvk" = GraphVite kernels).
 *
 * This is the drop-in boundary of the hot path: plain pointers and sizes, no
 * torch / pybind / C++ types, caller owns every buffer, every call is stream-ordered and
 * re-entrant per device, and every call returns a status instead of aborting (the reference
 * aborts through glog CHECK, include/util/debug.h:27-38).
 *
 * Reference interfaces replaced (paths relative to the reference tree, graphvite v0.2.2):
 *   gvk_train            gpu::graph::train / train_1_moment / train_2_moment
 *                        (include/instance/gpu/graph.cuh:36-95, 104-167, 176-242) as launched by
 *                        GraphWorker::train_dispatch (include/instance/graph.cuh:467-554), together
 *                        with the negative draw that WorkerMixin::train_batch performs first
 *                        (curandGenerateUniformDouble + gpu::Sample, include/core/solver.h:1536-1539)
 *   gvk_train_episode    WorkerMixin::train's batch loop + train_batch's lr schedule
 *                        (include/core/solver.h:1511-1557, include/core/optimizer.h:77-79,132-134)
 *   gvk_predict          gpu::graph::predict (include/instance/gpu/graph.cuh:250-279) as launched by
 *                        GraphWorker::predict_dispatch (include/instance/graph.cuh:560-577)
 *   gvk_alias_build      AliasTable::build (include/base/alias_table.cuh:84-128)
 *   gvk_alias_sample     AliasTable::device_sample / gpu::Sample (include/base/alias_table.cuh:155-158,174-182)
 *   gvk_negative_draw    the same draw as gvk_train's fused on-device draw, as a standalone kernel
 *   gvk_class_table_build / gvk_negative_draw_classes
 *                        WorkerMixin::build_negative_sampler (include/core/solver.h:1264-1278) in the form the
 *                        solvers use by default: an alias table over the classes of equal-weight rows instead of
 *                        one slot per row — same distribution ("Negative sampling by weight classes" below)
 *   gvk_sample_pairs / gvk_sample_edges / gvk_sample_walks / gvk_sample_walks_blocks
 *                        SamplerMixin::sample, GraphSampler::sample_random_walk / sample_biased_random_walk
 *                        (include/core/solver.h:1012-1055, include/instance/graph.cuh:298-450) on the device (opt-in)
 *   gvk_group_pairs      no counterpart: a per-batch pre-pass (same samples, shared rows adjacent)
 *   gvk_spread_pairs     no counterpart: a per-pool pre-pass for walk-ordered pools (consecutive samples to consecutive launches)
 *   gvk_probe_row_traffic, gvk_describe_train, gvk_set_tuning, gvk_range_push / _pop
 *                        measurement aids (roofline.access_pattern, kernel label, A/B knobs, roctx ranges at the
 *                        reference's Timer scopes, include/util/time.h:28-60)
 *
 * Data layout in HBM
 *   embedding tables  row-major float32 [rows][dim] (Vector<dim,float>, include/base/vector.h:31-69);
 *                     512 B/row at dim 128.  Moment tables have the same shape.
 *   pairs             uint32 records {tail, head} (std::tuple<Index,Index> is laid out reversed,
 *                     include/instance/gpu/graph.cuh:55-57), ids local to the resident partitions.
 *   negatives         uint32 [batch_size][num_negative] (include/instance/gpu/graph.cuh:66)
 *   alias table       gvk_alias_entry[count]: prob and alias interleaved so that one 8-byte load
 *                     serves a draw (the reference keeps two arrays, alias_table.cuh:60-61)
 *
 * RNG contract (replaces cuRAND XORWOW, which nothing in the reference pins): Philox4x32-10.
 *   negative j of sample s in batch b:
 *       w = philox4x32_10(ctr = {s, b, j / 2, 0x6e656721}, key = {seed_lo, seed_hi})
 *       (w_a, w_b) = (w[0], w[1]) if j even else (w[2], w[3])
 *       index = (uint64(w_a) * count) >> 32            -- uniform slot, all of [0, count) reachable
 *       u     = float(w_b >> 8) * 2^-24                -- the 24-bit resolution gpu::Sample has
 *       negative = u < table[index].prob ? index : table[index].alias
 *   host uniform stream t (used by the CPU samplers, gvs.h): doubles 2i and 2i+1 come from
 *       w = philox4x32_10(ctr = {i_lo, i_hi, t, 0x686f7374}, key = seed);  d = (w_hi:w_lo >> 11) * 2^-53
 */
#ifndef GVK_H_
#define GVK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GVK_OK 0
#define GVK_EINVAL (-1) /* bad argument (null pointer, negative size, no negative source, ...) */
#define GVK_EDIM (-2)   /* dim is not one of 32, 64, 96, 128, 256, 512 (the reference's instantiations,
                           src/graphvite.cu:52-59) */
#define GVK_EHIP (-3)   /* the HIP runtime reported an error; gvk_last_error() has the text */
#define GVK_ENOMEM (-4)

/* OptimizerType, include/core/optimizer.h:27-34 */
enum { GVK_SGD = 0, GVK_MOMENTUM = 1, GVK_ADAGRAD = 2, GVK_RMSPROP = 3, GVK_ADAM = 4 };

typedef struct {
    float prob;
    uint32_t alias;
} gvk_alias_entry;

/* POD stand-in for the by-value `Optimizer` kernel argument (include/instance/gpu/graph.cuh:40). */
typedef struct {
    int32_t type;       /* GVK_SGD ... GVK_ADAM */
    float lr;           /* learning rate of THIS batch (schedule already applied) */
    float weight_decay;
    float hp0;          /* Momentum: momentum; RMSprop: alpha; Adam: beta1 (the union at optimizer.h:110-121) */
    float hp1;          /* Adam: beta2 */
    float epsilon;      /* AdaGrad / RMSprop / Adam */
} gvk_optimizer;

/* The embedding (and moment) tables a launch works on.  Moment pointers may be NULL for SGD. */
typedef struct {
    float *vertex;          /* [n_vertex][dim]   head partition */
    float *context;         /* [n_context][dim]  tail partition */
    float *vertex_moment1;  /* Momentum / AdaGrad / RMSprop / Adam */
    float *context_moment1;
    float *vertex_moment2;  /* Adam */
    float *context_moment2;
    uint32_t n_vertex, n_context;
    uint32_t flags;         /* GVK_PAIRS_* below; 0 = nothing known about the order of the pairs */
    uint32_t reserved;
} gvk_tables;

/* gvk_tables.flags: adjacent pairs of the batch that share a head row come from ONE random walk (DeepWalk / node2vec pools
 * in the sampler's order: a walk emits the pairs of a head node back to back, graph.cuh:357-373) — they are trained
 * pair by pair like any others, never chained into runs on one copy of the row: chaining the head side while the context
 * rows of the same walk are still updated concurrently shifts what is learned (link-prediction AUC +0.007 against the
 * reference's loop on the BlogCatalog-sized shape; pair by pair: -0.001). */
#define GVK_PAIRS_OF_WALKS 1u

/* Where the negatives of a batch come from: an explicit array (the reference's negative_batch), or —
 * when `negatives` is NULL — the alias table, drawn inside the training kernel per the RNG contract. */
typedef struct {
    const uint32_t *negatives;     /* device, [batch_size * num_negative], or NULL */
    const gvk_alias_entry *table;  /* device, [count] */
    uint32_t count;
    uint64_t seed;
    const struct gvk_class_entry *classes; /* device, [class_count], or NULL: the same distribution by weight classes */
    uint32_t class_count;
} gvk_negative_source;

/* Negative sampling by weight classes.  The reference draws a negative from an alias table with one slot per row of
 * the tail partition (solver.h:1264-1278, alias_table.cuh:148-152): 8 MB at 1M rows, one random memory request per draw —
 * 1 in 7 of everything a dim-32 sample asks of the memory system, 1 in 25 at dim 128.  The weights are
 * degree^0.75, and the rows of a partition are sorted by degree: rows of equal weight are contiguous and there are
 * only a few thousand distinct weights.  A class = a maximal run of consecutive rows of equal weight; the class table
 * is an alias table over the classes (class weight = rows * weight) whose entries also carry the run:
 *     w = philox4x32_10(ctr = {sample, batch_id, j, 0x6e656743}, key = seed)
 *     slot = (uint64(w[0]) * class_count) >> 32;  u = float(w[1] >> 8) * 2^-24
 *     c = u < classes[slot].prob ? slot : classes[slot].alias
 *     negative = classes[c].first + ((uint64(w[2]) * classes[c].count) >> 32)
 * Every row is drawn with probability weight / sum of weights, exactly as from the row table; the table is 16 bytes
 * per class and lives in the caches.  gvk_class_table_build (host) returns GVK_OK and *num_class; callers keep the row
 * table when the classes are not much fewer than the rows (real-valued edge weights: every row its own class). */
typedef struct gvk_class_entry {
    float prob;
    uint32_t alias;
    uint32_t first, count;
} gvk_class_entry;
int gvk_class_table_build(const float *weights, size_t n, gvk_class_entry *out, uint32_t *num_class);
/* negatives[s * num_negative + j] for a whole batch, as the training kernels draw them from a class table */
int gvk_negative_draw_classes(void *stream, const gvk_class_entry *classes, uint32_t class_count, uint64_t seed,
                              uint32_t batch_id, uint32_t *negatives, int batch_size, int num_negative);

/* One batch of negative-sampling SGD: for every {tail, head} pair, num_negative negative steps then the
 * positive step on a progressively updated copy of vertex[head]; context rows are updated in place,
 * Hogwild (no atomics), loss[s] = sample loss / (1 + num_negative * negative_weight).
 * On a head table smaller than 16 MiB, pairs that sit next to each other in the batch and share a head row (up to
 * batch_size / 5120 of them, rounded up) are trained as one run — one after the other on one
 * register copy of the row, as consecutive iterations of one warp in the reference (include/instance/gpu/graph.cuh:
 * 54-94); samples keep their own negatives and loss slots.  `stream` is a hipStream_t (NULL = default stream).  `batch_id` only feeds the negative draw. */
int gvk_train(void *stream, int dim, const gvk_optimizer *optimizer, const gvk_tables *tables,
              const uint32_t *pairs, const gvk_negative_source *negative, uint32_t batch_id, float *loss,
              int batch_size, int num_negative, float negative_weight);

/* `num_batches` consecutive batches from a device-resident pool: batch i uses pairs + i * batch_size * 2,
 * batch id = first_batch_id + i * batch_id_stride (the reference's workers draw ids from one shared atomic
 * counter, solver.h:142,1520, so W concurrent workers see ids W apart), and
 * lr = init_lr * schedule(id, total_batches) where schedule is max(1 - id / total, 1e-4) if linear_schedule
 * else 1 (`optimizer->lr` is init_lr here).  loss [batch_size] is overwritten by every batch, as in the
 * reference, so after the call it holds the losses of the LAST batch.  One kernel launch per batch. */
int gvk_train_episode(void *stream, int dim, const gvk_optimizer *optimizer, int linear_schedule,
                      const gvk_tables *tables, const uint32_t *pairs, const gvk_negative_source *negative,
                      uint32_t first_batch_id, uint32_t batch_id_stride, uint32_t total_batches, int num_batches,
                      float *loss, int batch_size, int num_negative, float negative_weight);

/* Hub rows trained by chains (SGD, negatives drawn on the device).  Inside one launch every sample runs at
 * the same time, and of the updates that hold a row at the same time one survives.  For the rows of a large table that is
 * rare; a hub row is in flight hundreds of times per launch and keeps a handful of its updates, where the reference's
 * sequential loop (gpu/graph.cuh:54-94 in sample order) keeps them all.  Here the first hot_vertex rows of the head table and
 * the first hot_context rows of the tail table (partitions are ordered by falling degree: the hub rows) are each owned by a
 * chain: one lane group holds the row in registers and applies every update the unit has for it one after the other, partner
 * rows read-only, and stores the row once; the per-pair work trains every sample as gvk_train does but only READS hub rows.
 * A unit is a batch or one of its `parts`.  The chains of a unit start from the hub rows as the unit found them — their own
 * row and every partner that is a hub row, so a sample between two hub rows updates both from their old values, as the
 * reference does (model/graph.h:47-58) — and the unit's pairs read the hub rows those chains left; during a call the hub
 * rows live in mirrors inside the workspace and are written to the tables when the call returns its last launch.
 *   gvk_hot_plan    bytes of device workspace the work lists of num_batch batches (and the mirrors of the hub rows) need
 *   gvk_hot_build   the work lists of num_batch consecutive batches of a device pool (ids first_batch_id + i * stride): per
 *                   unit and hub row the partner rows of its updates — for a head row the k negatives then the tail of every
 *                   sample it heads, for a context row the head of every sample it is the tail or a negative of; negatives
 *                   per the RNG contract, exactly as the training launch draws them
 *   gvk_train_episode_hot   gvk_train_episode over batches whose work lists sit in `workspace` (built for workspace_batches
 *                   batches starting with this call's first batch; same dim, pool, ids, negative source, hot_vertex /
 *                   hot_context).  One launch per unit: the pairs of unit u and, in its first blocks, the chains of unit
 *                   u + 1.  form: GVK_HOT_SERIALIZED = per unit two launches, chains then pairs — the result is then a pure
 *                   function of the work lists, what the oracle restates (parity tests of the kernels); GVK_HOT_LERP = a
 *                   sample reads a hub row not as its unit's chains left it but on the straight line from where they found it
 *                   to where they left it, at the sample's place in the unit (fewer parts for the same staleness).
 * parts (a divisor of batch_size, the same in all three calls; 1 = none): a batch is trained as `parts` equal units one after
 * the other, each with its own work lists, so that nothing reads a hub row that is more than a unit old.  When every row of
 * both tables is a hub row the pairs have nothing to store and only run for the last batch (its loss).
 * chain_cap (the same in all three calls; 0 = the default and the most, 7): entries one chain task trains in sequence — what a
 * chain's record carries, so that a chain of up to chain_cap entries costs two dependent round trips (its record; its own row
 * and all partner rows at once).  A longer chain is one workgroup's: up to 256 / lanes tasks of consecutive entries (chain_cap
 * each; longer tasks beyond that many) trained side by side and composed in task order (weight decay in closed form;
 * deterministic given the work lists).  form & GVK_HOT_ROUNDS: tasks of more than GVK_HOT_ROUND_STEPS entries work in ROUNDS of
 * that many — the tasks' end states are composed after every round and the next round starts from the composed row, so that
 * no more than (256 / lanes) x GVK_HOT_ROUND_STEPS entries ever work side by side from the same state, however many updates a
 * hub row meets in a unit (side by side, the entries' gradient steps add up where the sequential loop's see each other:
 * measured on a graph whose largest hub heads 6 % of the samples; a round's end costs a barrier). */
#define GVK_HOT_ROUND_STEPS 4 /* entries a task of a long chain applies per round when the caller asks for rounds (form GVK_HOT_ROUNDS) */
#define GVK_HOT_SERIALIZED 1
#define GVK_HOT_LERP 2
#define GVK_HOT_ROUNDS 4
int gvk_hot_plan(int dim, int batch_size, int num_negative, uint32_t hot_vertex, uint32_t hot_context, int num_batch, int parts,
                 int chain_cap, size_t *bytes);
int gvk_hot_build(void *stream, int dim, void *workspace, size_t workspace_bytes, const uint32_t *pool, int batch_size,
                  int num_batch, int num_negative, const gvk_negative_source *negative, uint32_t first_batch_id,
                  uint32_t batch_id_stride, uint32_t hot_vertex, uint32_t hot_context, int parts, int chain_cap);
/* gvk_hot_build as launches of at most units_per_launch units each (a unit = a part of a batch = one workgroup of the list kernel; 0 = all
 * in one launch = gvk_hot_build).  Same lists.  A measurement form: lists built ahead run beside the launches that train the chunk before
 * and slow some of them; thinner launches were measured slower still (profiles/r6/experiments/r6_list_slice_ab.txt), callers use gvk_hot_build. */
int gvk_hot_build_sliced(void *stream, int dim, void *workspace, size_t workspace_bytes, const uint32_t *pool, int batch_size,
                         int num_batch, int num_negative, const gvk_negative_source *negative, uint32_t first_batch_id,
                         uint32_t batch_id_stride, uint32_t hot_vertex, uint32_t hot_context, int parts, int chain_cap, int units_per_launch);
int gvk_train_episode_hot(void *stream, int dim, const gvk_optimizer *optimizer, int linear_schedule, const gvk_tables *tables,
                          const uint32_t *pairs, const gvk_negative_source *negative, uint32_t first_batch_id,
                          uint32_t batch_id_stride, uint32_t total_batches, int num_batches, float *loss, int batch_size,
                          int num_negative, float negative_weight, void *workspace, size_t workspace_bytes,
                          uint32_t hot_vertex, uint32_t hot_context, int workspace_batches, int parts, int chain_cap,
                          int form);

/* The same chains as a stream of their own, a batch AHEAD of the pairs (the default executor of hub-heavy tables).
 * A launch that carries a unit's pairs and the next unit's chains (gvk_train_episode_hot) lasts as long as its longest chain
 * (10-12 us: a dependency chain) while its pairs are done after 5-6.  Here the chains of every unit run back to back as launches
 * of their own on `chain_stream` and the pairs of a batch follow on `stream` once that batch's chains are done — while the
 * chains of the next batch already run.  Same chains, same pairs, same units as gvk_train_episode_hot (what a unit's chains
 * read of the hub rows, what its pairs read, is unchanged); what changes is when a chain sees the rows that are NOT hub rows:
 * up to two batches before the pairs have caught up (such a row is hit less than once per batch).  The hub rows live in a ring
 * of 2 parts + 1 versions per row inside the workspace — a row's version advances in every unit that has entries for it; the
 * work lists name the slot of every hub row a chain or a pair reads — so nothing is copied for rows a unit does not touch.
 *   gvk_ahead_plan   bytes of workspace: the work lists of num_batch batches, the versions, the samples' slot words, the ring
 *   gvk_ahead_build  gvk_hot_build + the versions and slots; on `stream` (a caller builds the lists of the next chunk on its
 *                    copy stream while this chunk trains)
 *   gvk_train_episode_ahead   the chains on chain_stream, the pairs on stream, ordered by events (the chains of batch b + 2
 *                    wait for the pairs of batch b); `stream` has everything of the call behind it when it returns, the
 *                    tables hold the hub rows again.  pair_launches (a divisor of parts; 0 = parts): launches the pairs of
 *                    a batch are trained as.  form: GVK_HOT_SERIALIZED (chain_stream unused: per unit the chains, then the
 *                    pairs, on `stream` — the function of the work lists the oracle restates), GVK_HOT_ROUNDS.
 * group (a divisor of parts, the same in build and train; 1 = the chain stream above): group > 1 is a third executor on ONE stream —
 * a launch carries the chains of `group` consecutive units and the pairs of the group before it (train_group_kernel): the chain of a
 * later unit of the group starts as soon as its own row has been published by the chain of the unit before (a workgroup of the same
 * launch; coherent loads and stores, a flag per row), and reads the partners that are hub rows as the GROUP found them — up to
 * group - 1 units older than gvk_train_episode_hot reads them.  chain_stream is unused then.
 * parts <= 127 (a slot is one byte); at most 32767 hub rows per table and 2^30 rows per table (fields of a work-list entry). */
int gvk_ahead_plan(int dim, int batch_size, int num_negative, uint32_t hot_vertex, uint32_t hot_context, int num_batch, int parts,
                   int chain_cap, size_t *bytes);
int gvk_ahead_build(void *stream, int dim, void *workspace, size_t workspace_bytes, const uint32_t *pool, int batch_size,
                    int num_batch, int num_negative, const gvk_negative_source *negative, uint32_t first_batch_id,
                    uint32_t batch_id_stride, uint32_t hot_vertex, uint32_t hot_context, int parts, int chain_cap, int group);
int gvk_train_episode_ahead(void *stream, void *chain_stream, int dim, const gvk_optimizer *optimizer, int linear_schedule,
                            const gvk_tables *tables, const uint32_t *pairs, const gvk_negative_source *negative,
                            uint32_t first_batch_id, uint32_t batch_id_stride, uint32_t total_batches, int num_batches, float *loss,
                            int batch_size, int num_negative, float negative_weight, void *workspace, size_t workspace_bytes,
                            uint32_t hot_vertex, uint32_t hot_context, int workspace_batches, int parts, int chain_cap,
                            int pair_launches, int group, int form);
/* the events gvk_train_episode_ahead keeps for a chain stream: to be released before the stream is destroyed */
void gvk_ahead_release(void *chain_stream);

/* logits[s] = dot(vertex[head_s], context[tail_s]) */
int gvk_predict(void *stream, int dim, const float *vertex, const float *context, const uint32_t *pairs,
                float *logits, int batch_size);

/* result[i] = table.sample((float)rand[2i], (float)rand[2i+1]) — the reference's double-driven draw. */
int gvk_alias_sample(void *stream, const gvk_alias_entry *table, uint32_t count, const double *rand,
                     uint32_t *result, int n);

/* negatives[s * num_negative + j] per the RNG contract (what gvk_train draws when negatives == NULL). */
int gvk_negative_draw(void *stream, const gvk_alias_entry *table, uint32_t count, uint64_t seed,
                      uint32_t batch_id, uint32_t *negatives, int batch_size, int num_negative);

/* Positive sampling on the device (SURVEY.md §8f rank 4 — the reference, and this repo's default, draw positives
 * on CPU threads, include/core/solver.h:1012-1055).  For LINE with augmentation_step 1 a positive sample of block
 * (head partition, tail partition) is an edge of that block drawn with probability proportional to its weight, so
 * with an alias table over the block's edges the whole block pool is one kernel:
 *     w = philox4x32_10(ctr = {i_lo, i_hi, 0, 0x706f7321}, key = seed),  i = first_index + t
 *     index = (uint64(w[0]) * count) >> 32;  u = float(w[1] >> 8) * 2^-24
 *     pool[t] = block_pairs[u < table[index].prob ? index : table[index].alias]          ({tail, head} records)
 * Same distribution as the CPU edge sampler restricted to the block (exact conditional, nothing dropped). */
int gvk_sample_pairs(void *stream, const gvk_alias_entry *table, const uint32_t *block_pairs, uint32_t count,
                     uint64_t seed, uint64_t first_index, uint32_t *pool, size_t n);

/* The same draw — same Philox words, same slot, same comparison, bit-identical pools — from the packed form of a block:
 * entry i = {table[i].prob, table[i].alias, block_pairs[2i], block_pairs[2i + 1]}.  A draw that keeps its slot is one
 * random 16-byte read (on an unweighted graph every probability is 1, so every draw); only a draw that takes the alias
 * reads a second entry.  The sampler competes with the training kernel for memory REQUESTS, not bytes: one request less
 * per sample is 4 % of the end-to-end rate at dim 128. */
typedef struct {
    float prob;
    uint32_t alias;
    uint32_t tail, head;
} gvk_edge_entry;
int gvk_sample_edges(void *stream, const gvk_edge_entry *table, uint32_t count, uint64_t seed, uint64_t first_index,
                     uint32_t *pool, size_t n);

/* Random-walk positive sampling on the device (same extension): what GraphSampler::sample_random_walk /
 * sample_biased_random_walk do on CPU threads (include/instance/graph.cuh:298-450), one walk per GPU thread.
 *   walk w: draw 0 picks a directed edge (c0 -> c1) from edge_table; every further draw picks the next node from the
 *   current node's out-edge alias table (neighbor_table, CSR-aligned with flat_offsets).  With `biased` (node2vec)
 *   a proposal x from node v reached from u is accepted with probability f(x) / max(1/p, 1, 1/q), f = 1/p if x == u,
 *   1 if u is an out-neighbour of x, 1/q otherwise — rejection sampling over the SAME per-vertex tables, which yields
 *   exactly the transition distribution of the reference's per-edge tables (graph.cuh:656-677) without their
 *   sum-of-deg^2 memory.  A node with no out-edge ends the chain; the thread restarts from a fresh edge.
 *   When node j >= 1 joins the chain the pairs (chain[j-k], chain[j]), k = 1..min(augmentation_step, j) are emitted
 *   as {tail, head} = {local[chain[j]], local[chain[j-k]]}; thread w owns pool offsets [w*M, (w+1)*M),
 *   M = aug*L - aug*(aug-1)/2, written through the pseudo shuffle slot = offset % sb * (pool_pairs / sb) + offset / sb.
 *   Draw d of walk w uses philox4x32_10(ctr = {w_lo, w_hi, d, 0x77616c6b}, key = seed): w[0] -> slot (multiply-shift),
 *   w[1] -> alias probability, w[2] -> acceptance (biased).  Single partition (local[] maps vertex -> row). */
typedef struct {
    const uint64_t *flat_offsets;           /* [num_vertex + 1] */
    const uint32_t *edges_uv;               /* [2 * num_edge_entries] {u, v}, CSR order */
    const gvk_alias_entry *edge_table;      /* [num_edge_entries] over the edge weights */
    const gvk_alias_entry *neighbor_table;  /* [num_edge_entries] per-vertex tables, CSR-aligned */
    const uint32_t *sorted_neighbors;       /* [num_edge_entries] out-neighbours of each vertex, ascending (biased only) */
    const uint32_t *local;                  /* [num_vertex] vertex -> row of the (single) partition */
    uint32_t num_vertex, num_edge_entries;
    int32_t biased;
    float p, q;
} gvk_walk_graph;

int gvk_sample_walks(void *stream, const gvk_walk_graph *graph, uint64_t seed, uint64_t first_walk, uint32_t *pool,
                     size_t pool_pairs, int walk_length, int augmentation_step, int shuffle_base);

/* The same walks for SEVERAL partitions (any number of GPUs): a walk yields pairs for every (head partition, tail
 * partition) block, so each pair is binned into the pool of block b = part[head] * P + part[tail] — the per-block pools
 * of GraphSampler::sample_random_walk (include/instance/graph.cuh:357-373) filled by GPU threads.  Walk w (= first_walk +
 * thread) draws exactly as in gvk_sample_walks and produces aug * L - aug * (aug - 1) / 2 pairs.  Block b's pool is
 * pools + 2 * offsets[b] (offsets in pairs; ~0 = this call does not collect b), capacity pairs long, cut into num_stripe
 * stripes (any divisor of capacity; a few hundred keeps the atomics off each other) of capacity / num_stripe slots with
 * one counter each — counters[b * num_stripe + stripe].  Pair i of a walk of wavefront w (= thread / 64) is appended to
 * stripe (w + (i % sb) * max(num_stripe / sb, 1)) % num_stripe: the pseudo shuffle of include/instance/graph.cuh:362-364,439-441 —
 * the pairs of a walk that share a row, capacity / sb records apart — with the part chosen by the pair's index in its
 * walk (a wavefront's 64 walks append in lock step: the slot says nothing about the walk).  Slot s of stripe k is record
 * k * (capacity / num_stripe) + s.  Pairs beyond a stripe's capacity are dropped (solver.h:1045-1052) but still counted, so the counters tell the
 * caller every block's share and which stripes are full.  The caller zeroes the counters once and repeats the call
 * (advancing first_walk) until the pools it needs are full.  Which pair lands in which slot depends on the order the
 * GPU retires the atomics; the MULTISET of pairs of a stripe that did not overflow is a pure function of (seed, walks). */
int gvk_sample_walks_blocks(void *stream, const gvk_walk_graph *graph, const int32_t *part, int num_partition, uint64_t seed,
                            uint64_t first_walk, uint64_t num_walks, uint32_t *pools, const uint64_t *offsets,
                            uint32_t *counters, uint32_t capacity, int num_stripe, int walk_length, int augmentation_step,
                            int shuffle_base);
/* The same with THINNING: accept (device memory, num_partition^2 floats, or NULL = gvk_sample_walks_blocks) gives every block the
 * probability with which a pair that belongs to it is kept — decided by a hash of (walk, the pair's index in its walk, seed), not by
 * when the pair arrives: u = fmix32(lo(walk) ^ hi(walk) * 0x85ebca6b ^ i * 0x9e3779b9 ^ lo(seed) * 0xc2b2ae35) >> 8 / 2^24 < accept[b].
 * Why: blocks receive unequal shares of the walks' pairs, a pool that is full drops what arrives LATER in the launch, and with
 * node2vec's rejection sampling the walks that arrive late are those that rejected most — a selection the AUC sees (+0.003 at
 * Youtube size in 4 partitions).  A caller that thins every block to the rate at which its pool fills together
 * with the others' (the engine: from the shares a first small launch shows) loses nothing to arrival order but the last per cent. */
int gvk_sample_walks_blocks_thinned(void *stream, const gvk_walk_graph *graph, const int32_t *part, int num_partition, uint64_t seed,
                                    uint64_t first_walk, uint64_t num_walks, uint32_t *pools, const uint64_t *offsets,
                                    uint32_t *counters, uint32_t capacity, int num_stripe, int walk_length, int augmentation_step,
                                    int shuffle_base, const float *accept);

/* Optional pool pre-pass: inside each of the num_batch batches (batch_size {tail, head} records each) of pool_in, make
 * the records that share a head row adjacent, writing the regrouped pool to pool_out (distinct from pool_in).  Each
 * batch keeps exactly its records; their order becomes ascending in the low row_bits bits of the head row, stable
 * otherwise (deterministic).  The order of the samples inside a batch has no meaning to gvk_train / the reference's
 * kernel (they run concurrently); adjacency makes a head row that several samples of a batch share one HBM fetch.
 * Two-call protocol, nothing is allocated here: workspace == NULL stores the scratch size this shape needs in
 * *workspace_bytes; otherwise workspace must hold *workspace_bytes bytes of device memory. */
int gvk_group_pairs(void *stream, const uint32_t *pool_in, uint32_t *pool_out, void *workspace,
                    size_t *workspace_bytes, int batch_size, int num_batch, int row_bits);

/* Optional pool pre-pass for the walk-ordered pools of DeepWalk / node2vec (a walk emits the pairs of a head node back to back
 * and meets a tail node in pairs a few records apart, graph.cuh:320-348): record i of pool_in goes to place
 * (i % units) * (num_pair / units) + i / units of pool_out (distinct from pool_in; units divides num_pair) — with units = the
 * number of launches that train the pool, consecutive records are trained by consecutive launches instead of side by side in
 * one, where all but one of the updates of the row they share would be lost.  Same records; the reference's sequential loop
 * does not care about their order. */
int gvk_spread_pairs(void *stream, const uint32_t *pool_in, uint32_t *pool_out, size_t num_pair, int units);

/* Host: Vose alias construction exactly as the reference orders it (FIFO queues, double mean).
 * index_bytes 4 -> uint32 alias[] (n <= 2^32 - 1), 8 -> uint64 alias[].  n must be > 0.  (The reference's loop
 * counters are int, alias_table.cuh:93-100, so it overflows past 2^31 entries; this builder does not.)
 * packed (optional, index_bytes 4 only) receives the interleaved device form. */
int gvk_alias_build(const float *weights, size_t n, float *prob, void *alias, int index_bytes,
                    gvk_alias_entry *packed);

/* Measurement aid (bench.py `roofline.access_pattern`): the memory traffic of gvk_train (SGD, one negative) with none
 * of its arithmetic and none of its dependent loads — per sample the head row, the tail row and the GIVEN negative row
 * are read, `bump` is added to every element and they are written back, in gvk_train's lane layout.  What this kernel
 * sustains is the ceiling of the memory system for the access pattern of the hot path (random rows of dim * 4 bytes,
 * half reads, half writes); gvk_train can at best match it. */
int gvk_probe_row_traffic(void *stream, int dim, float *vertex, float *context, const uint32_t *pairs,
                          const uint32_t *negatives, float bump, int batch_size);

/* Tuning knobs.  The product library (libgvk.so) has three; the A/B library (make -C graphvite_amd/csrc ab ->
 * build/ab/libgvk_ab.so, compiled with -DGVK_AB_BUILDS; bench.py / tests load it through GVK_LIBRARY) adds the measured
 * alternatives that do not ship.  Returns GVK_EINVAL for an unknown key, an unsupported value, or an A/B-only knob set
 * to anything but its default in the product library. */
#define GVK_TUNE_VARIANT 2        /* 0 = default: the per-pair kernel; on a head table smaller than 16 MiB train_runs_kernel
                                     (one lane group trains a run of adjacent same-head samples in sequence);
                                     2 = the per-pair kernel at any table size; 4 = train_runs_kernel at any table size.
                                     A/B library only: 1 = the per-pair kernel, generic build (run-time k); 3 = dim-128 SGD
                                     in the reference's kernel shape (one wavefront per pair, vertex row in LDS, 8192 x 512
                                     grid-stride launch) */
#define GVK_TUNE_RUN_CAP 3        /* train_runs_kernel: longest run a lane group trains in sequence: 0 = the default, 20 (the
                                     generations of the reference's launch of a default batch on the card it was written
                                     for), 1 = every pair on its own, up to 4096 */
#define GVK_TUNE_SPLIT_HITS 7     /* a batch is trained as gvk_train_launches() equal parts, one launch each, so that a launch
                                     holds at most `value` samples per row of the head table: default 2 (what keeps small
                                     partitions at the reference's learning quality); 0 = always one launch
                                     per batch */
#define GVK_TUNE_ROUND_STEPS 12    /* gvk_train_episode_hot: entries a task of a long chain applies per round: -1 (default) = what the caller's form says,
                                     0 = all of them in one round whatever the form says, 1 .. 8 = rounds of so many whatever it says */
#define GVK_TUNE_HOT_ORDER 10      /* measurement: which blocks of a train_hot_kernel launch are dispatched first — 0 the chains, 1 (default)
                                     the long chains, then the pairs, then the other chains, 2 the pairs */
#define GVK_TUNE_HOT_SERIALIZED 9 /* measurement: 1 = gvk_train_episode_hot always launches the chains and the pairs of a unit one after the
                                     other (GVK_HOT_SERIALIZED): their durations apart in a kernel trace */
#define GVK_TUNE_CHAIN_CAP 8      /* gvk_train_episode_hot: entries one chain task trains in sequence (a longer chain is cut into
                                     tasks trained side by side and composed): 0 = the default, 7 (also the most) */
/* A/B library only: */
#define GVK_TUNE_LANES_PER_PAIR 1 /* 0 = per-dim default; else 8, 16, 32 or 64 */
#define GVK_TUNE_GENERATION 4     /* parity experiment: C > 0 trains a batch as consecutive launches of at most C samples
                                     (per-pair kernel), the concurrency structure of the reference's launch on a card
                                     that keeps C warps resident; 0 = off (default) */
#define GVK_TUNE_SEGMENT_STEPS 5  /* 1, 2 or 4 = SGD with one negative runs train_segment_kernel (a wavefront owns 64 /
                                     lanes * steps consecutive pairs and chains the same-head runs inside it through
                                     registers, all rows requested up front); 0 = off (default) */
#define GVK_TUNE_SKIP_LOSS 6      /* train_segment_kernel only: 1 (default) = gvk_train_episode uses its loss-less build for
                                     batches whose loss[] a later batch of the same call overwrites; 0 = every batch
                                     computes it */
int gvk_set_tuning(int key, int value);
/* Q: gvk_train / gvk_train_episode train a batch of batch_size samples on a head table of n_vertex rows as Q consecutive
 * launches of batch_size / Q samples (Q divides batch_size; 1 unless the table has fewer than batch_size / 2 rows; see
 * GVK_TUNE_SPLIT_HITS).  A solver that regroups its pools (gvk_group_pairs) regroups them per part: batch_size / Q samples,
 * Q x the batches — a part is what runs concurrently, so a part is what is made of runs. */
int gvk_train_launches(int batch_size, uint32_t n_vertex);
/* 1 when this library was built with -DGVK_AB_BUILDS (the A/B baselines exist), 0 for the product library. */
int gvk_has_ab_builds(void);

/* The kernel gvk_train / gvk_train_episode launch for this configuration under the current tuning, as text
 * ("train_kernel<128,16,SGD,k=1> run_cap 1", "train_runs_kernel<128,16,SGD,k=1> run_cap 20") — what a benchmark should label
 * its measurement with.  n_vertex = rows of the head table (decides between the two). */
int gvk_describe_train(int dim, int optimizer_type, int num_negative, int explicit_negatives, int batch_size,
                       uint32_t n_vertex, char *name, size_t capacity);

/* Named host ranges for profilers (roctx; `rocprofv3 --marker-trace`): the scopes the reference times with its
 * USE_TIMER Timer (include/util/time.h:28-60, include/core/solver.h:622,645,1526-1552).  Nesting allowed, per thread. */
void gvk_range_push(const char *name);
void gvk_range_pop(void);

const char *gvk_last_error(void);
const char *gvk_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GVK_H_ */
