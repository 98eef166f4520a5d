/* gvs.h — C ABI of the host-side runtime that feeds the gvk kernels ("gvs" = GraphVite sampler/solver core).
 *
 * Everything the reference does on CPU threads around the training kernel, restated as a C-ABI library so
 * that any host (the Python package in this repo, or the reference's own C++ solver) can drive it:
 *
 *   gvs_graph_*        Graph<Index> / GraphMixin: edge-list loading, name<->id maps, undirected doubling,
 *                      normalisation, flatten() to CSR, save
 *                      (include/instance/graph.cuh:61-277, include/core/graph.h:45-125)
 *   gvs_partition      SolverMixin::partition + head/tail_locations (include/core/solver.h:873-887, 399-410)
 *   gvs_schedule       SolverMixin::get_schedule, non-tied branch (include/core/solver.h:519-575)
 *   gvs_sampler_*      SamplerMixin::sample (include/core/solver.h:1012-1055),
 *                      GraphSampler::sample_random_walk / sample_biased_random_walk
 *                      (include/instance/graph.cuh:298-450), GraphSolver::build_vertex_edge / build_edge_edge /
 *                      get_sample_function (include/instance/graph.cuh:645-721)
 *
 * Index type is uint32 (the reference's only bound instantiation for this path, src/graphvite.cu:52-59).
 * All functions return GVK_OK or a negative GVK_E* code (include/gvk.h) and set gvk_last_error(); none aborts.
 * Pool records are {tail, head} uint32 pairs with partition-local ids (the kernel's `pairs` layout).
 *
 * RNG: sampler thread t consumes the host uniform stream t of the RNG contract in include/gvk.h, two doubles
 * per alias draw.  The edge sampler consumes them in the reference's order (sample by sample); the walk samplers
 * advance the walks of one inner round in lockstep (all start edges, then step 2 of every live walk, ...) so that
 * each stage can be software-pipelined over the round.  Either way a fill is a pure function of (seed, number of
 * threads, stream positions), and the CPU oracle reproduces it bit for bit.
 */
#ifndef GVS_H_
#define GVS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gvs_graph gvs_graph;
typedef struct gvs_sampler gvs_sampler;

/* ---- graph ---------------------------------------------------------------------------------------- */

gvs_graph *gvs_graph_create(void);
void gvs_graph_destroy(gvs_graph *g);

/* Text edge list: "u v [w]" per line, `comment` starts a comment, `delimiters` as strtok takes them.
 * Ids are assigned in first-seen order; duplicate lines are kept; num_edge counts lines. */
int gvs_graph_load_file(gvs_graph *g, const char *file_name, int as_undirected, int normalization,
                        const char *delimiters, const char *comment);
/* Edge list of names (weights may be NULL = 1). */
int gvs_graph_load_names(gvs_graph *g, const char *const *u_names, const char *const *v_names, const float *weights,
                         size_t n, int as_undirected, int normalization);
/* Edge list of integer labels (the name of a vertex is the decimal form of its label). Same id assignment. */
int gvs_graph_load_labels(gvs_graph *g, const uint32_t *u_labels, const uint32_t *v_labels, const float *weights,
                          size_t n, int as_undirected, int normalization);
/* WordGraph::load_file_compact (include/instance/word_graph.cuh:73-181): the graph of word co-occurrences of a corpus,
 * one sentence per line; words seen fewer than min_count times are dropped, every pair of kept words at most `window`
 * tokens apart adds 1 to the weight of (u, v) and of (v, u).  num_edge counts both directions, as the reference does. */
int gvs_graph_load_corpus(gvs_graph *g, const char *file_name, int window, int min_count, int normalization,
                          const char *delimiters, const char *comment);
int gvs_graph_save(const gvs_graph *g, const char *file_name, int weighted, int anonymous);

uint32_t gvs_graph_num_vertex(const gvs_graph *g);
uint64_t gvs_graph_num_edge(const gvs_graph *g);          /* input lines, as the reference counts */
uint64_t gvs_graph_num_directed_edge(const gvs_graph *g); /* flattened entries (about 2x when undirected) */
int gvs_graph_as_undirected(const gvs_graph *g);
int gvs_graph_normalization(const gvs_graph *g);
int64_t gvs_graph_name2id(const gvs_graph *g, const char *name); /* -1 when absent */
/* Writes the NUL-terminated name into buf (truncated to buflen - 1); returns its full length, or -1. */
int64_t gvs_graph_id2name(const gvs_graph *g, uint32_t id, char *buf, size_t buflen);

/* Flattened (CSR) views, valid until the graph is reloaded or destroyed:
 *   edges_uv [2 * D] = {u, v} in vertex order then insertion order; edge_weights [D]; flat_offsets [N + 1];
 *   vertex_weights [N] (weighted out-degree; after normalisation the normalised sums). */
const uint32_t *gvs_graph_edges(const gvs_graph *g);
const float *gvs_graph_edge_weights(const gvs_graph *g);
const uint64_t *gvs_graph_flat_offsets(const gvs_graph *g);
const float *gvs_graph_vertex_weights(const gvs_graph *g);

/* CSR-aligned alias tables, one per vertex over the weights of its out-edges (the reference's build_vertex_edge,
 * include/instance/graph.cuh:645-653), in the interleaved form gvk_sample_walks reads: out[num_directed_edge]. */
int gvs_graph_neighbor_tables(const gvs_graph *g, int num_thread, gvk_alias_entry *out);

/* out[i] = powf(vertex_weights[ids[i]], exponent): the weights of a partition's negative sampler
 * (WorkerMixin::build_negative_sampler, include/core/solver.h:1263-1278), in the order of ids. */
int gvs_negative_weights(const float *vertex_weights, const uint32_t *ids, uint64_t n, float exponent, float *out);

/* ---- partition / schedule ------------------------------------------------------------------------------ */

/* Sort vertices by weight descending (ties: ascending id — the reference leaves ties to std::sort), deal them
 * zig-zag over P parts.  part[v], local[v] = position inside its part, part_sizes[P]. */
int gvs_partition(const float *weights, uint32_t n, int num_partition, int32_t *part, uint32_t *local,
                  uint32_t *part_sizes);
/* out[(step * W + worker) * 2 + {0, 1}] = {head partition, tail partition}; returns the number of steps
 * ((P / W)^2 * W, or 1 when P == 1), or a negative error.  P must be a multiple of W. */
int gvs_schedule(int num_partition, int num_worker, int32_t *out, size_t out_len);

/* ---- samplers ---------------------------------------------------------------------------------------------- */

#define GVS_MODE_EDGE 0        /* SamplerMixin::sample: every model when augmentation_step == 1 */
#define GVS_MODE_WALK 1        /* sample_random_walk: DeepWalk, and LINE when augmentation_step > 1 */
#define GVS_MODE_BIASED_WALK 2 /* sample_biased_random_walk: node2vec through per-edge alias tables (sum of deg^2) */
#define GVS_MODE_BIASED_REJECT 3 /* node2vec by rejection over the per-vertex tables: the same transition
                                    distribution with O(|E|) memory (propose x by edge weight, accept with probability
                                    f(x) / max(1/p, 1, 1/q); f as in graph.cuh:664-669).  Three uniforms per proposal. */

/* The graph must outlive the sampler (the reference borrows it the same way, solver.h:289).  part / local are
 * copied.  The alias table over all flattened edge weights is built at the first fill that needs it. */
gvs_sampler *gvs_sampler_create(const gvs_graph *g, const int32_t *part, const uint32_t *local, int num_partition,
                                uint64_t seed);
void gvs_sampler_destroy(gvs_sampler *s);

/* Builds what `mode` needs with num_thread threads: per-vertex alias tables (WALK, BIASED_REJECT) or node2vec
 * per-edge tables with return parameter p and in-out parameter q (BIASED_WALK; sum of deg^2 entries). */
int gvs_sampler_prepare(gvs_sampler *s, int mode, float p, float q, int num_thread);

/* EDGE mode, several partitions: builds the column table of one tail partition ahead of the first fill that uses
 * it (fills build it on demand otherwise). */
int gvs_sampler_prepare_column(gvs_sampler *s, int tail_partition, int num_thread);

typedef struct {
    int mode;
    int num_thread;         /* sampler threads; thread t fills slice [t*L, min((t+1)*L, pool_size)), L = ceil(pool_size / T) */
    int sample_batch_size;  /* EDGE: edges drawn per inner round (reference: walk_length * walk_batch) */
    int walk_length;        /* WALK modes */
    int walk_batch;         /* walks per inner round */
    int augmentation_step;
    int shuffle_base;       /* pseudo shuffle; pool_size % shuffle_base must be 0 */
    int tail_partition;     /* -1: fill all P*P block pools; r >= 0: only the blocks (*, r) — one GPU's column.
                               EDGE mode then draws from an alias table over just the edges whose tail lives in
                               partition r (the exact conditional distribution, nothing dropped); the walk modes
                               draw as usual and drop pairs that end elsewhere. */
    int os_threads;         /* OS threads that execute the num_thread slices (0 = one per slice, capped at the
                               hardware concurrency).  Fewer OS threads than slices balances stragglers; the
                               result depends on num_thread only. */
    int cpu_offset;         /* >= 0: pin OS thread k of the pool to the (cpu_offset + k)-th CPU this process may run on
                               (modulo their number); -1: leave placement to the scheduler.  A fill lasts a few
                               milliseconds — shorter than the scheduler takes to spread hundreds of freshly woken
                               threads over the cores — so unpinned fills are bimodal (10x slower when they stack up). */
} gvs_fill_config;

/* pools[hp * P + tp] -> pool_size {tail, head} records (entries of unfilled blocks may be NULL).
 * Blocks until every requested block pool is full.  Stream positions advance across calls. */
int gvs_sampler_fill(gvs_sampler *s, uint32_t *const *pools, uint64_t pool_size, const gvs_fill_config *config);

/* Introspection for tests: position (doubles consumed) of stream t; the built tables. */
uint64_t gvs_sampler_stream_position(const gvs_sampler *s, int thread);
int gvs_sampler_set_stream_position(gvs_sampler *s, int thread, uint64_t position);
const float *gvs_sampler_edge_prob(const gvs_sampler *s);
const uint64_t *gvs_sampler_edge_alias(const gvs_sampler *s);
const float *gvs_sampler_neighbor_prob(const gvs_sampler *s);      /* WALK: [D]; BIASED: [sum deg^2] */
const uint32_t *gvs_sampler_neighbor_alias(const gvs_sampler *s);
/* the same tables in the interleaved {prob, alias} form the samplers (and gvk_sample_walks) read; no copy */
const gvk_alias_entry *gvs_sampler_neighbor_slots(const gvs_sampler *s);
const uint64_t *gvs_sampler_edge_edge_offsets(const gvs_sampler *s); /* BIASED: [D + 1] */

/* EDGE-mode column table of tail partition r (built by the first filtered fill): the flattened edge ids it
 * covers and its alias table. */
int gvs_sampler_column(const gvs_sampler *s, int tail_partition, uint64_t *count, const uint64_t **edge_ids,
                       const float **prob, const uint64_t **alias);

/* The host uniform stream itself (RNG contract), for hosts that want to reproduce a fill. */
void gvs_host_uniforms(uint64_t seed, uint32_t stream, uint64_t first, size_t n, double *out);

#ifdef __cplusplus
}
#endif
#endif /* GVS_H_ */
