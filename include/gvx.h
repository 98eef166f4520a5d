/* gvx.h — C ABI of the native solver engine ("gvx" = GraphVite executor): the reference's GraphSolver /
 * SolverMixin / WorkerMixin orchestration (include/instance/graph.cuh:586-813, include/core/solver.h:87-888,
 * 1170-1623) as a C++ host runtime over the kernels (gvk.h) and the host samplers (gvs.h) — what the pybind11
 * module libgraphvite (graphvite_amd/csrc/bind/libgraphvite.cpp, the drop-in for src/graphvite.cu) is a thin
 * binding of.  One process drives every GPU it is given, like the reference: one worker per entry of device_ids,
 * each with its own HIP streams; a worker holds the whole vertex table and the context shards it owns for good,
 * and after a schedule step the workers copy the head shards they trained into each other's replicas directly,
 * GPU to GPU over xGMI (hipMemcpyPeerAsync) — no host staging, no collective library inside one process.
 * When that does not fit gpu_memory_limit, the engine falls back to the reference's scheme: one head and one tail
 * partition per worker in HBM, travelling through host memory between blocks (solver.h:1435-1504).
 * (Several processes, one GPU each, over RCCL: graphvite_amd.solver.GraphSolver.)
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
} gvx_solver_members;

/* device_ids: num_device GPU ids (an id may repeat: its workers then share that GPU), or num_device == 0 for all
 * visible GPUs.  num_sampler_per_worker / gpu_memory_limit: GVX_AUTO = (usable CPUs / #worker) - 1 / free memory. */
gvx_solver *gvx_solver_create(int dim, const int *device_ids, int num_device, int num_sampler_per_worker,
                              size_t gpu_memory_limit);
void gvx_solver_destroy(gvx_solver *s);

/* Beyond the reference's arguments (off by default; takes effect at the next train()).
 * GVX_DEVICE_SAMPLING 1: the positive samples are drawn by the GPUs themselves instead of the CPU sampler threads
 * (solver.h:1012-1055) — gvk_sample_edges over per-block edge alias tables for augmentation_step 1, gvk_sample_walks_blocks
 * for the random-walk modes: every worker walks the whole graph, keeps a 1/#worker slice of every block pool and copies
 * each slice to the worker that trains the block, GPU to GPU.  Resident (not streamed) mode only; graphs with fewer
 * than 2^32 directed edges. */
#define GVX_DEVICE_SAMPLING 1
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

/* Logging of the native runtime (glog in the reference, src/graphvite.cu:81-88): messages below `threshold`
 * (0 INFO, 1 WARNING, 2 ERROR, 3 FATAL) are dropped; the rest go to stderr, or to `sink` when one is installed. */
void gvx_set_logging(int threshold, void (*sink)(int severity, const char *message, void *user), void *user);

#ifdef __cplusplus
}
#endif
#endif /* GVX_H_ */
