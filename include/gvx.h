/* gvx.h — C ABI of the native solver engine ("gvx" = GraphVite executor): the reference's GraphSolver /
 * SolverMixin / WorkerMixin orchestration (include/instance/graph.cuh:586-813, include/core/solver.h:87-888,
 * 1170-1623) as a C++ host runtime over the kernels (gvk.h) and the host samplers (gvs.h) — what the pybind11
 * module libgraphvite (graphvite_amd/csrc/bind/libgraphvite.cpp, the drop-in for src/graphvite.cu) is a thin
 * binding of, and what graphvite_amd.solver.GraphSolver (the Python drop-in) and bench.py drive.  ONE orchestrator, two
 * ways to place its W workers:
 *   gvx_solver_create              one process drives every GPU it is given, like the reference: one worker per entry of
 *                                  device_ids, each a set of HIP streams on its GPU;
 *   gvx_solver_create_distributed  one process per GPU (torchrun / mpirun): the process is worker `rank` of `world_size`.
 * Either way a worker holds the whole vertex table — a slab [P slots][1 + m][S][dim], the W slots of a head group
 * contiguous — and the context shards it owns for good; before a worker trains head partition hp it brings hp to ITS slot
 * of hp's group (at most one device-local copy), and after a schedule step ONE in-place all-gather of the group's slab
 * hands every worker what the others trained: RCCL over xGMI (ncclCommInitAll in one process, ncclCommInitRank across
 * processes; librccl is opened with dlopen at the first multi-GPU build).  Workers that share a GPU (device_ids = [0, 0],
 * tests) exchange by event-ordered device copies instead, and a program may supply its own transport (gvx_transport).
 * Random-walk pools drawn per worker are routed to the workers that train them by one all-to-all per episode.
 * When the model does not fit gpu_memory_limit, the one-process engine falls back to the reference's scheme: one head and
 * one tail partition per worker in HBM, travelling through host memory between blocks (solver.h:1435-1504).
 *
 * Reference interfaces replaced:
 *   gvx_solver_create   GraphSolver(device_ids, num_sampler_per_worker, gpu_memory_limit)   bind.h:438-441, solver.h:170-217
 *   gvx_solver_build    SolverMixin::build                                                  bind.h:443-457, solver.h:287-466
 *   gvx_solver_train    GraphSolver::train -> SolverMixin::train                            bind.h:459-487, graph.cuh:770-793,
 *                                                                                           solver.h:588-654
 *   gvx_solver_predict  SolverMixin::predict_numpy                                          bind.h:489-497, solver.h:660-735
 *   gvx_solver_clear    SolverMixin::clear                                                  bind.h:499-503, solver.h:741-751
 *   gvx_solver_embeddings  the numpy views of vertex_embeddings / context_embeddings        bind.h:90-106, 430-433
 *   gvx_solver_get      the read-only members                                               bind.h:408-429
 * Every call returns GVK_OK or a negative GVK_E* code and sets gvk_last_error(); nothing aborts.
 */
#ifndef GVX_H_
#define GVX_H_

#include <stddef.h>
#include <stdint.h>

#include "gvk.h"
#include "gvs.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GVX_AUTO 0 /* kAuto, include/util/common.h:29 */

typedef struct gvx_solver gvx_solver;

/* Optimizer + LRSchedule (include/core/optimizer.h:36-158, 216-300).  schedule: 0 constant, 1 linear, 2 custom —
 * custom calls schedule_function(batch_id, num_batch, user) on the host once per batch. */
typedef struct {
    int32_t type;         /* GVK_SGD ... GVK_ADAM, or -1 = "auto": the solver's default (SGD 0.025, weight decay 5e-3,
                             graph.cuh:634-636) with `lr` overriding its learning rate when > 0 */
    float lr, weight_decay;
    float hp0, hp1, epsilon; /* as gvk_optimizer */
    int32_t schedule;
    float (*schedule_function)(int batch_id, int num_batch, void *user);
    void *user;
} gvx_optimizer;

typedef struct {
    const char *model;           /* "DeepWalk", "LINE", "node2vec" */
    int num_epoch;
    int resume;
    int augmentation_step;       /* GVX_AUTO: log(1600) / log(|E| / |V|), graph.cuh:781-784 */
    int random_walk_length;
    int random_walk_batch_size;
    int shuffle_base;            /* GVX_AUTO: augmentation_step; forced to 1 for DeepWalk / node2vec */
    float p, q;
    int positive_reuse;
    float negative_sample_exponent;
    float negative_weight;
    int log_frequency;
} gvx_train_config;

/* The read-only members bind.h:408-429 exposes (strings are owned by the solver, valid until the next call). */
typedef struct {
    int dim, num_partition, num_negative, num_epoch, resume, episode_size, batch_size, augmentation_step;
    int random_walk_length, random_walk_batch_size, shuffle_base, positive_reuse, log_frequency, num_worker, num_sampler;
    float negative_sample_exponent, negative_weight, p, q;
    size_t gpu_memory_limit, gpu_memory_cost;
    const char *model;
    gvx_optimizer optimizer;
    uint64_t batch_id, num_batch;
    double train_seconds;        /* wall time of the episode loop of the last train() */
    int rank, num_local_worker;  /* rank of the first local worker; how many workers this process drives */
    int pair_order;              /* what the last train() did: 1 the sampler's order, 2 regrouped, 3 spread (walk-ordered pools, gvk_spread_pairs) */
    int sampler_mode;            /* GVS_MODE_* of the last train() */
    int device_sampling;
    uint32_t partition_rows;     /* S: rows of a partition's table */
    const char *transport;       /* "RCCL", "device copies", "caller-supplied transport" or "" (one worker) */
    uint32_t hub_rows;           /* rows per table the last train() trained by chains (the largest partition value; 0 = none) */
    int32_t hub_parts;           /* ... the parts it trained a batch as (the largest block value; 0 = no hub rows) */
    int32_t hub_lerp;            /* ... 1 = its pairs read hub rows along the chains' way (gvk.h GVK_HOT_LERP) */
    int32_t hub_rounds;          /* ... 1 = some block's long chains worked in rounds (gvk.h GVK_HOT_ROUNDS) */
    uint32_t lists_prefetched;   /* visits whose first chunk of work lists was built while the visit before trained, and trained as built (this solver, so far) */
} gvx_solver_members;

/* device_ids: num_device GPU ids (an id may repeat: its workers then share that GPU), or num_device == 0 for all
 * visible GPUs.  num_sampler_per_worker / gpu_memory_limit: GVX_AUTO = (usable CPUs / #worker) - 1 / free memory. */
gvx_solver *gvx_solver_create(int dim, const int *device_ids, int num_device, int num_sampler_per_worker,
                              size_t gpu_memory_limit);

/* A transport supplied by the embedding program instead of RCCL (the CPU tests run the engine over gloo this way).  Both
 * calls are collective over the world_size processes, operate on memory of the engine's device, may be asynchronous on
 * `stream` (the engine orders its own work behind that stream) and return GVK_OK or a negative code.
 *   all_gather  in place: `slab` holds world_size parts of `bytes` bytes, this rank's part already at slab + rank * bytes;
 *   all_to_all  part q of `send` arrives as part `rank` of rank q's `recv` (world_size parts of `bytes` bytes each).
 * Besides the exchanges of train(), build() calls all_gather twice with 4 bytes per rank (the ranks agree on the episode size every
 * one of them can allocate: gvx_engine.cpp allocate_pools) — on every rank, whatever happened to the rank locally.  A call that fails
 * must fail on EVERY rank (a collective that some ranks leave and others wait in cannot be recovered from here). */
typedef struct {
    int (*all_gather)(void *user, void *slab, size_t bytes, void *stream);
    int (*all_to_all)(void *user, const void *send, void *recv, size_t bytes, void *stream);
    void *user;
} gvx_transport;

#define GVX_UNIQUE_ID_BYTES 256 /* two RCCL unique ids: the exchange communicator and the one that routes walk pools */
/* Rank 0 calls this and broadcasts the bytes to every rank (torch.distributed, MPI, a file ...). */
int gvx_unique_id(void *out, size_t capacity);

/* One process per GPU: this process is worker `rank` of `world_size`, on GPU `device_id`.  unique_id: the bytes rank 0
 * obtained from gvx_unique_id (ignored when `transport` is given).  Every rank must make the same build / train calls
 * with the same arguments on the same graph; embeddings are complete on every rank after train(). */
gvx_solver *gvx_solver_create_distributed(int dim, int rank, int world_size, int device_id, const void *unique_id,
                                          size_t unique_id_bytes, const gvx_transport *transport,
                                          int num_sampler_per_worker, size_t gpu_memory_limit);
void gvx_solver_destroy(gvx_solver *s);

/* Beyond the reference's arguments (off by default; takes effect at the next train()).
 * GVX_DEVICE_SAMPLING 1: the positive samples are drawn by the GPUs themselves instead of the CPU sampler threads
 * (solver.h:1012-1055) — gvk_sample_edges over per-block edge alias tables for augmentation_step 1, gvk_sample_walks_blocks
 * for the random-walk modes: every worker walks the whole graph, keeps a 1/#worker slice of every block pool and copies
 * each slice to the worker that trains the block, GPU to GPU.  Resident (not streamed) mode only; graphs with fewer
 * than 2^32 directed edges. */
#define GVX_DEVICE_SAMPLING 1
/* GVX_PAIR_ORDER 0 (default): by table size — regrouped (gvk_group_pairs: the pairs of a part of a batch
 * that share a head row made adjacent) for cache-resident tables and for shard-sized tables of independent edge draws,
 * the sampler's order otherwise — except the walk-ordered pools of DeepWalk / node2vec, which are spread (gvk_spread_pairs:
 * consecutive records to consecutive launches) unless chains own every row; 1: always the sampler's order; 2: always
 * regrouped. */
#define GVX_PAIR_ORDER 2
/* GVX_SEED: seeds the embedding initialisation, the host samplers, the device samplers and the negative draws (default 0:
 * the values the engine has always used). */
#define GVX_SEED 3
/* GVX_NEGATIVE_TABLE 0 (default): an alias table over the weight classes when they are at least 8 x fewer than the rows;
 * 1: always one alias slot per row (the reference's); 2: always by class. */
#define GVX_NEGATIVE_TABLE 4
/* GVX_NODE2VEC_TABLE_LIMIT: entries of node2vec's per-edge alias tables (sum over edges of the head's degree, graph.cuh:
 * 656-677) beyond which the CPU samplers switch to rejection over the per-vertex tables (default 2^30). */
#define GVX_NODE2VEC_TABLE_LIMIT 5
/* GVX_HUB_ROWS (SGD): the hub rows of every partition are trained by chains (gvk_train_episode_hot): one
 * lane group per hub row applies all the updates a unit has for the row one after the other, so none of them is lost to a
 * concurrent one.  -2 (default): as GVX_FIDELITY says — with `auto`, -1 wherever chains exist (DeepWalk / node2vec on one
 * partition of at most 16384 rows: every row); -1: the rows a batch is expected to hit once or more (by its share of the
 * partition's degree, or of degree^exponent as a negative; at most 16384 per table); N > 0: the first N rows of every
 * partition; 0: off.  Batches keep the sampler's order (walk-ordered pools: spread over the units). */
#define GVX_HUB_ROWS 6
/* GVX_HUB_PARTS: with hub rows trained by chains, a batch is trained as this many equal parts (it must divide the batch size:
 * otherwise train() fails), each with its own chains and pairs; 0 (default): the rule — the largest hub row meets about 250 of its
 * updates per part, no row outside the chains is expected to be hit more than 0.125 times per part, walk-ordered pools get at
 * least augmentation_step^2 + 1 parts; at most 32 (gvk_train_launches() parts where every row is a hub row). */
#define GVX_HUB_PARTS 7
/* GVX_FIDELITY -1 (default, `auto`): the reference's learning quality wherever chains exist — on tables that do not live in
 * the caches, the rows a batch is expected to hit once or more are trained by chains (GVX_HUB_ROWS -1) and a batch as so many
 * parts that the largest hub row meets about 250 of its updates per part (eight on the headline shape): link-prediction AUC
 * within 0.002 of the reference's sequential loop there; the moment optimizers have no chains: they train every
 * row pair by pair and say so once.  1 (`reference`): the same, but a
 * configuration without chains is an error.  0 (`throughput`): every row is trained pair by pair (Hogwild, as the reference's
 * kernel); the hub rows of a hub-heavy graph then keep only some of their updates.  GVX_HUB_ROWS / GVX_HUB_PARTS given
 * explicitly take precedence. */
#define GVX_FIDELITY 8
/* GVX_HUB_LERP -1 (default): the rule; 0 / 1: with hub rows trained by chains, a sample reads a hub row as the chains of its
 * part left it / on the straight line from where they found it to where they left it, at the sample's place in the part
 * (gvk.h GVK_HOT_LERP).  GVX_HUB_CHAIN_CAP: entries one chain trains in sequence before it counts as a long chain (gvk.h chain_cap;
 * 0 = the default, 7; larger values are clamped to 7). */
#define GVX_HUB_LERP 9
#define GVX_HUB_CHAIN_CAP 10
/* GVX_HUB_ROUNDS -1 (default): the rule — long chains work in rounds (gvk.h GVK_HOT_ROUNDS) on graphs whose largest vertex takes
 * more than 2 % of the total degree; 0 / 1: never / always. */
#define GVX_HUB_ROUNDS 11
/* GVX_HUB_EXECUTOR -1 (default): the rule — the chains of the hub rows as a stream of their own, a batch ahead of the pairs
 * (gvk.h gvk_train_episode_ahead; the fused launches where GVX_HUB_LERP 1 is asked for or a callback computes the schedule);
 * 0: one launch per unit carries its pairs and the next unit's chains (gvk_train_episode_hot); 1: the chain stream.
 * GVX_HUB_PAIR_LAUNCHES 0 (default): under the chain stream the pairs of a batch are one launch per part; N (a divisor of the
 * parts, else ignored): N launches per batch.
 * GVX_HUB_EXECUTOR 2: one stream, a launch = the chains of GVX_HUB_GROUP consecutive parts + the pairs of the parts before them
 * (gvk.h `group`; GVX_HUB_GROUP: 0 = the rule's, N = a divisor of the parts, else the largest divisor below it). */
#define GVX_HUB_EXECUTOR 12
#define GVX_HUB_PAIR_LAUNCHES 13
#define GVX_HUB_GROUP 14
int gvx_solver_set(gvx_solver *s, int option, int64_t value);

/* The graph is borrowed until the next build / destroy (solver.h:289).  num_partition / episode_size: GVX_AUTO. */
int gvx_solver_build(gvx_solver *s, const gvs_graph *graph, const gvx_optimizer *optimizer, int num_partition,
                     int num_negative, int batch_size, int episode_size);
int gvx_solver_train(gvx_solver *s, const gvx_train_config *config);
/* samples: n pairs (v, c) of global vertex ids; logits[i] = <vertex[v_i], context[c_i]>. */
int gvx_solver_predict(gvx_solver *s, const int64_t *samples, size_t n, float *logits);
int gvx_solver_clear(gvx_solver *s);

/* which 0: vertex embeddings, 1: context embeddings — [num_vertex][dim] floats in global vertex order, owned by the
 * solver, stable from build() to the next build() / destroy (the numpy views alias this memory). */
float *gvx_solver_embeddings(gvx_solver *s, int which, uint64_t *num_vertex);
int gvx_solver_get(gvx_solver *s, gvx_solver_members *out);
/* GraphSolver::info() text; returns its length (truncated to capacity - 1 in buf). */
size_t gvx_solver_info(gvx_solver *s, char *buf, size_t capacity);
/* word2vec binary format, GraphSolver::save_embeddings (graph.cuh:796-805) */
int gvx_solver_save_embeddings(gvx_solver *s, const char *file_name);

/* ---- a training run step by step (bench.py; custom loops) -------------------------------------------------------------
 * gvx_solver_train is: open, then per episode { fill the next pool set while: for every schedule step { stage, train,
 * exchange } }, then close.  The same pieces one at a time, for every LOCAL worker at once (all W in one process, one per
 * process otherwise).  `set` is 0 / 1 (the two pool sets), `step` an index into the episode's step order
 * (gvx_session_steps of them), a block visit trains batches [first, first + count) of the step's pool.
 *   open      configure + device state + tables to HBM; resident_pools != 0: filled pool sets are kept in HBM, so that a
 *             stage never crosses PCIe (what a benchmark of the GPU path wants; device sampling always is)
 *   fill      the pools of every block the local workers train, by the CPU samplers (or on the device)
 *   stage     the pool of `step` from `set` into device buffer `buffer` (0 / 1): H2D copy and / or regrouping pass, on the
 *             copy stream; nothing when the pool is trained in place
 *   train     wait for the stage and for the exchange this block depends on, bring the head partition to the worker's slot,
 *             launch the batches; advances batch_id by count x W
 *   exchange  the all-gather of the head group trained at `step` (asynchronous; W == 1: nothing)
 *   wait      orders the compute streams behind every pending exchange (a fence for timed regions)
 *   close     device tables -> host arrays (gvx_solver_embeddings), device state released. */
int gvx_session_open(gvx_solver *s, const gvx_train_config *config, int resident_pools);
int gvx_session_steps(gvx_solver *s);
/* block (head partition, tail partition) local worker `worker` trains at `step` */
int gvx_session_block(gvx_solver *s, int step, int worker, int *head_partition, int *tail_partition);
int gvx_session_fill(gvx_solver *s, int set);
int gvx_session_stage(gvx_solver *s, int step, int set, int buffer);
int gvx_session_train(gvx_solver *s, int step, int set, int buffer, int first, int count);
int gvx_session_exchange(gvx_solver *s, int step);
int gvx_session_wait(gvx_solver *s);
int gvx_session_synchronize(gvx_solver *s);
int gvx_session_close(gvx_solver *s);
/* the compute stream (hipStream_t) of a local worker: where HIP events around train calls belong */
void *gvx_session_stream(gvx_solver *s, int worker);
/* mean per-sample loss of the most recent batch of a local worker (synchronises its compute stream) */
int gvx_session_loss(gvx_solver *s, int worker, float *mean_loss);
/* The ceiling of the memory system for what the batches of a block touch (gvk_probe_row_traffic on the tables and the
 * staged pool of local worker 0's block at `step`, negatives drawn as the training kernel draws them): `launches`
 * back-to-back launches timed with HIP events after as many untimed ones; *ms_per_launch receives the average. */
int gvx_session_probe(gvx_solver *s, int step, int set, int buffer, int launches, float *ms_per_launch);
/* bytes every worker sent in exchanges since open, and their number */
int gvx_session_exchange_stats(gvx_solver *s, uint64_t *bytes_sent_per_worker, uint64_t *exchanges);

/* Opens librccl, creates a one-rank communicator on `device` and runs the engine's two collectives through it (an
 * in-place all-gather and an all-to-all of one rank leave a buffer as it was): what a one-GPU box can check of the RCCL
 * carrier — the library loads, the entry points exist, the call sequence is accepted. */
int gvx_rccl_selftest(int device);

/* Logging of the native runtime (glog in the reference, src/graphvite.cu:81-88): messages below `threshold`
 * (0 INFO, 1 WARNING, 2 ERROR, 3 FATAL) are dropped; the rest go to stderr, or to `sink` when one is installed. */
void gvx_set_logging(int threshold, void (*sink)(int severity, const char *message, void *user), void *user);

#ifdef __cplusplus
}
#endif
#endif /* GVX_H_ */
