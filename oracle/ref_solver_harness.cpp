// TEST INFRASTRUCTURE ONLY — never linked into or loaded by the product.
//
// Host build of the reference's OWN solver front end — Graph (include/instance/graph.cuh:62-276), SolverMixin::build
// with partition() and get_schedule() (include/core/solver.h:270-575,873-887), GraphSolver::get_sample_function with its
// vertex / edge alias tables (graph.cuh:645-721) and the three CPU samplers (solver.h:1012-1055, graph.cuh:298-450) —
// compiled from /root/reference where it lies, nothing copied, over the emulated CUDA runtime and cuRAND of
// ref_stubs/ (device memory = malloc; curandGenerateUniformDouble = whatever source the test installs, i.e. the same
// per-thread Philox stream the oracle and the product's samplers consume).  The GPU workers are constructed (their
// buffers are plain memory here) but never run.
//
//   hipcc -x hip --cuda-host-only ... (oracle/Makefile) -> oracle/_ref/libgvref_solver.so
#include <cstdint>
#include <string>
#include <tuple>
#include <vector>

#include <cuda_runtime.h>
#include <curand.h>

// From here on every `__device__` function of the reference (the optimizer updates, core/optimizer.h:162-213) is
// host-callable as well: the GPU workers' kernel is emulated below by a sequential host loop over the reference's own
// model and optimizer code.
#undef __device__
#define __device__ __attribute__((device)) __attribute__((host))

#ifndef GVREF_DIM
#define GVREF_DIM 128  // the embedding dimension this build of the reference's solver is instantiated for (oracle/Makefile: 128 and 96)
#endif

int gvref_device_count = 1;
size_t gvref_device_memory = (size_t)16 << 30;  // a P100's 16 GB, the card the reference's defaults were tuned on
gvref_uniform_source_t gvref_uniform_source = nullptr;
int gvref_generator_count = 0;

#include "instance/graph.cuh"
#include "instance/word_graph.cuh"

// The GPU workers never run here.  Their two dispatch functions are the only code that instantiates the reference's CUDA
// kernels (instance/gpu/graph.cuh, written for nvcc: `model.backward<...>` without the `template` disambiguator does
// not parse under clang), so they are specialised away before anything instantiates the worker class.
namespace graphvite {
typedef SolverMixin<GVREF_DIM, float, uint32_t, Graph, GraphSampler, GraphWorker> HarnessSolverBase;
// gpu::Sample (alias_table.cuh:176-185) run on the host: one draw per pair of uniforms, narrowed to Float as the kernel
// narrows them, through the table's own sample().  One guard the kernel does not have: a double uniform within 2^-25
// of 1 narrows to 1.0f, sample() then computes index == count and reads one entry past both tables (AddressSanitizer
// caught exactly that here, one draw in 3 * 10^7) — on the GPU a silent garbage negative, on the host a wild row index
// that corrupts the heap.  Such a uniform is moved to the last slot instead.
template <>
void AliasTable<float, uint32_t>::device_sample(const Memory<double, int> &rand, Memory<uint32_t, int> *result) {
    for (int i = 0; i < result->count; i++) {
        float rand1 = rand.device_ptr[i * 2], rand2 = rand.device_ptr[i * 2 + 1];
        if (rand1 >= 1.0f) rand1 = 0.99999994f;  // the largest float below 1
        result->device_ptr[i] = sample(rand1, rand2);
    }
}

// gpu::graph::train<Vector, Index, Model, kSGD> (instance/gpu/graph.cuh:36-95) on the worker's buffers, over the
// reference's own model code, so that the reference's WHOLE training loop (sampler threads, schedule, partition loads
// and write-backs, negative sampler, lr schedule) runs on the CPU.  Two execution models of the launch:
//
//   gvref_kernel_chunk == 0: SEQUENTIAL.  Samples of a batch are applied one after the other (no lost updates).
//
//   gvref_kernel_chunk == C > 0: CHUNK-SYNCHRONOUS, a model of the concurrency the launch has on the card the
//   reference was written for.  The kernel is launched <<<8192, 512>>> (util/gpu.cuh:41-43, graph.cuh:487-490): one
//   32-lane warp per sample, 131 072 warps, so with the default batch of 100 000 every sample has its own warp and
//   sample i runs in block i / 16.  A V100 keeps 80 SMs x 2048 threads = 320 such blocks = C = 5120 warps resident
//   and dispatches blocks in index order, so samples [0, C) run together, then [C, 2C), ...  Inside a chunk the warps
//   execute the same straight-line code at the same pace, which is modelled as lock step over the kernel's own
//   phases (graph.cuh:54-94): (1) every warp copies its vertex row into its shared-memory buffer; (2) for s = 0 ..
//   num_negative: every warp reads its context row, computes forward / backward on (buffer, row) with the reference's
//   own Model code, then every warp writes its context row back; (3) every warp writes its buffer back to the vertex
//   row.  Two warps of a chunk that write the same row: the higher sample index wins (the hardware leaves the winner
//   open; any fixed rule is one of its outcomes).  A later chunk sees everything earlier chunks wrote.  C >=
//   batch size is the fully concurrent launch, C == 1 the sequential loop.
//
//   gvref_kernel_reads == 1 (with C > 0): the harsher variant "all reads at chunk start": every warp of the chunk
//   fetches its vertex row AND all its num_negative + 1 context rows before any warp of the chunk writes anything (a
//   warp still sees its own writes when two of its targets are the same row); all writes land at the end of the chunk,
//   higher sample index / later target last.  This is what a launch of C samples does on hardware that requests every
//   row of a sample up front (the product's kernel does), and a lower bracket for the reference's kernel, whose
//   context rows are read at the point of use.
int gvref_kernel_chunk = 0;
int gvref_kernel_reads = 0;
int gvref_kernel_threads = 1;  // host threads that share the (independent) per-warp arithmetic of a phase

template <>
bool GraphWorker<HarnessSolverBase>::train_dispatch() {
    auto *solver = reinterpret_cast<graphvite::GraphSolver<GVREF_DIM, float, uint32_t> *>(this->solver);
    typedef graphvite::Vector<GVREF_DIM, float> Vec;
    typedef LINE<Vec> Model;  // DeepWalk and Node2Vec are the same arithmetic (model/graph.h:60-85)
    Vec *vertex_embeddings = embeddings[0]->device_ptr, *context_embeddings = embeddings[1]->device_ptr;
    const uint32_t *samples = batch.device_ptr, *negatives = negative_batch.device_ptr;
    const int num_sample = batch.count / 2, k = negative_batch.count / num_sample;
    const float negative_weight = solver->negative_weight;
    const Optimizer opt = optimizer;
    if (num_moment != 0) {
        // train_1_moment / train_2_moment (instance/gpu/graph.cuh:104-242) under the SEQUENTIAL model only: the vertex row is
        // buffered, its moment rows and the context's row and moment rows are updated in place, sample after sample; the
        // optimizer's update is bound exactly as graph.cuh:494-551 binds it.
        if (gvref_kernel_chunk > 0) return false;
        Vec *vm1 = (*moments[0])[0].device_ptr, *cm1 = (*moments[1])[0].device_ptr;
        Vec *vm2 = num_moment > 1 ? (*moments[0])[1].device_ptr : nullptr, *cm2 = num_moment > 1 ? (*moments[1])[1].device_ptr : nullptr;
        const int type = optimizer.type == "Momentum" ? 1 : optimizer.type == "AdaGrad" ? 2 : optimizer.type == "RMSprop" ? 3
                         : optimizer.type == "Adam" ? 4 : 0;
        if (!type || (type == 4) != (num_moment == 2)) return false;
        Vec vertex_buffer;
        for (int sample_id = 0; sample_id < num_sample; sample_id++) {
            const uint32_t head_id = samples[sample_id * 2 + 1];
            Vec &vertex = vertex_embeddings[head_id];
            vertex_buffer = vertex;
            float sample_loss = 0;
            for (int s = 0; s <= k; s++) {
                const bool label = s == k;
                const uint32_t tail_id = label ? samples[sample_id * 2] : negatives[sample_id * k + s];
                Vec &context = context_embeddings[tail_id];
                float logit;
                Model::forward(vertex_buffer, context, logit);
                const float prob = sigmoid(logit);
                float gradient, weight;
                if (label) {
                    gradient = prob - 1;
                    weight = 1;
                    sample_loss += weight * -log(prob + kEpsilon);
                } else {
                    gradient = prob;
                    weight = negative_weight;
                    sample_loss += weight * -log(1 - prob + kEpsilon);
                }
                switch (type) {
                    case 1: Model::template backward<kMomentum>(vertex_buffer, context, vm1[head_id], cm1[tail_id], gradient, opt, weight); break;
                    case 2: Model::template backward<kAdaGrad>(vertex_buffer, context, vm1[head_id], cm1[tail_id], gradient, opt, weight); break;
                    case 3: Model::template backward<kRMSprop>(vertex_buffer, context, vm1[head_id], cm1[tail_id], gradient, opt, weight); break;
                    default: Model::template backward<kAdam>(vertex_buffer, context, vm1[head_id], cm1[tail_id], vm2[head_id], cm2[tail_id], gradient, opt, weight);
                }
            }
            loss.device_ptr[sample_id] = sample_loss / (1 + k * negative_weight);
            vertex = vertex_buffer;
        }
        return true;
    }
    if (optimizer.type != "SGD") return false;
    // one target of one sample: graph.cuh:63-88
    auto target = [&](Vec &vertex_buffer, Vec &context, bool label, float &sample_loss) {
        float logit;
        Model::forward(vertex_buffer, context, logit);
        const float prob = sigmoid(logit);
        float gradient, weight;
        if (label) {
            gradient = prob - 1;
            weight = 1;
            sample_loss += weight * -log(prob + kEpsilon);
        } else {
            gradient = prob;
            weight = negative_weight;
            sample_loss += weight * -log(1 - prob + kEpsilon);
        }
        Model::template backward<kSGD>(vertex_buffer, context, gradient, opt, weight);
    };
    if (gvref_kernel_chunk <= 0) {
        Vec vertex_buffer;
        for (int sample_id = 0; sample_id < num_sample; sample_id++) {
            const uint32_t head_id = samples[sample_id * 2 + 1];  // each positive sample is {tail, head}
            Vec &vertex = vertex_embeddings[head_id];
            vertex_buffer = vertex;
            float sample_loss = 0;
            for (int s = 0; s <= k; s++) {
                const bool label = s == k;
                const uint32_t tail_id = label ? samples[sample_id * 2] : negatives[sample_id * k + s];
                target(vertex_buffer, context_embeddings[tail_id], label, sample_loss);
            }
            loss.device_ptr[sample_id] = sample_loss / (1 + k * negative_weight);
            vertex = vertex_buffer;
        }
        return true;
    }
    const int C = gvref_kernel_chunk, T = gvref_kernel_threads > 0 ? gvref_kernel_threads : 1;
    if (gvref_kernel_reads == 1) {
        std::vector<Vec> vertex_buffers(C), context_rows((size_t)C * (k + 1));
        auto target_id = [&](int sample_id, int s) {
            return s == k ? samples[sample_id * 2] : negatives[sample_id * k + s];
        };
        for (int first = 0; first < num_sample; first += C) {
            const int n = std::min(C, num_sample - first);
#pragma omp parallel for num_threads(T) schedule(static)
            for (int i = 0; i < n; i++) {
                const int sample_id = first + i;
                Vec *rows = &context_rows[(size_t)i * (k + 1)];
                vertex_buffers[i] = vertex_embeddings[samples[sample_id * 2 + 1]];
                for (int s = 0; s <= k; s++) rows[s] = context_embeddings[target_id(sample_id, s)];
                float sample_loss = 0;
                for (int s = 0; s <= k; s++) {
                    for (int e = 0; e < s; e++)  // the warp's own earlier write to the same row
                        if (target_id(sample_id, e) == target_id(sample_id, s)) rows[s] = rows[e];
                    target(vertex_buffers[i], rows[s], s == k, sample_loss);
                }
                loss.device_ptr[sample_id] = sample_loss / (1 + k * negative_weight);
            }
            for (int i = 0; i < n; i++) {
                const int sample_id = first + i;
                for (int s = 0; s <= k; s++) context_embeddings[target_id(sample_id, s)] = context_rows[(size_t)i * (k + 1) + s];
                vertex_embeddings[samples[sample_id * 2 + 1]] = vertex_buffers[i];
            }
        }
        return true;
    }
    std::vector<Vec> vertex_buffers(C), context_rows(C);
    std::vector<float> sample_losses(C);
    for (int first = 0; first < num_sample; first += C) {
        const int n = std::min(C, num_sample - first);
#pragma omp parallel for num_threads(T) schedule(static)
        for (int i = 0; i < n; i++) {
            vertex_buffers[i] = vertex_embeddings[samples[(first + i) * 2 + 1]];
            sample_losses[i] = 0;
        }
        for (int s = 0; s <= k; s++) {
            const bool label = s == k;
#pragma omp parallel for num_threads(T) schedule(static)
            for (int i = 0; i < n; i++) {
                const int sample_id = first + i;
                const uint32_t tail_id = label ? samples[sample_id * 2] : negatives[sample_id * k + s];
                context_rows[i] = context_embeddings[tail_id];
                target(vertex_buffers[i], context_rows[i], label, sample_losses[i]);
            }
            for (int i = 0; i < n; i++) {
                const int sample_id = first + i;
                context_embeddings[label ? samples[sample_id * 2] : negatives[sample_id * k + s]] = context_rows[i];
            }
        }
        for (int i = 0; i < n; i++) {
            loss.device_ptr[first + i] = sample_losses[i] / (1 + k * negative_weight);
            vertex_embeddings[samples[(first + i) * 2 + 1]] = vertex_buffers[i];
        }
    }
    return true;
}
template <>
bool GraphWorker<HarnessSolverBase>::predict_dispatch() { return false; }
}  // namespace graphvite

namespace {
typedef graphvite::Graph<uint32_t> GraphT;
typedef graphvite::GraphSolver<GVREF_DIM, float, uint32_t> SolverT;
struct Handle {
    GraphT graph;
    SolverT *solver = nullptr;
    ~Handle() { delete solver; }
};
}  // namespace

extern "C" {

void gvref_set_uniform_source(gvref_uniform_source_t source) { gvref_uniform_source = source; }

// Execution model of the emulated kernel launch (see train_dispatch above): chunk 0 = sequential, C > 0 =
// chunk-synchronous with C resident warps (reads_at_start 1: its harsher variant); threads = host threads for the per-warp arithmetic (results do not depend on it).
void gvref_set_kernel_model(int chunk, int reads_at_start, int threads) {
    graphvite::gvref_kernel_chunk = chunk;
    graphvite::gvref_kernel_reads = reads_at_start;
    graphvite::gvref_kernel_threads = threads;
}

int gvref_solver_dim() { return GVREF_DIM; }

// The optimizer the next gvref_solver_create builds with (core/optimizer.h:272-330; kAuto = the solver's default, SGD 0.025 / 5e-3
// linear, graph.cuh:634-636).  type: "" / "Default", "SGD", "Momentum", "AdaGrad", "RMSprop", "Adam"; the helper classes' own
// defaults for everything but lr and weight decay — and, gvref_set_optimizer_momentum, Momentum's coefficient (its third constructor
// argument, optimizer.h:283-289; 0 = the class's own default 0.999).
static std::string g_optimizer_type;
static float g_optimizer_lr = 0, g_optimizer_wd = 0, g_optimizer_momentum = 0;
void gvref_set_optimizer(const char *type, float lr, float weight_decay) {
    g_optimizer_type = type ? type : "";
    g_optimizer_lr = lr, g_optimizer_wd = weight_decay, g_optimizer_momentum = 0;
}
void gvref_set_optimizer_momentum(float momentum) { g_optimizer_momentum = momentum; }
static graphvite::Optimizer chosen_optimizer() {
    using namespace graphvite;
    if (g_optimizer_type == "SGD") return SGD(g_optimizer_lr, g_optimizer_wd);
    if (g_optimizer_type == "Momentum")
        return g_optimizer_momentum > 0 ? Momentum(g_optimizer_lr, g_optimizer_wd, g_optimizer_momentum) : Momentum(g_optimizer_lr, g_optimizer_wd);
    if (g_optimizer_type == "AdaGrad") return AdaGrad(g_optimizer_lr, g_optimizer_wd);
    if (g_optimizer_type == "RMSprop") return RMSprop(g_optimizer_lr, g_optimizer_wd);
    if (g_optimizer_type == "Adam") return Adam(g_optimizer_lr, g_optimizer_wd);
    return Optimizer(kAuto);
}

void *gvref_solver_create(const uint32_t *edges, const float *weights, uint64_t n, int as_undirected, int num_worker,
                          int num_sampler_per_worker, int num_partition, int num_negative, int batch_size,
                          int episode_size) {
    Handle *h = new Handle();
    if (weights) {
        std::vector<std::tuple<std::string, std::string, float>> list;
        for (uint64_t i = 0; i < n; i++)
            list.emplace_back(std::to_string(edges[2 * i]), std::to_string(edges[2 * i + 1]), weights[i]);
        h->graph.load_weighted_edge_list(list, as_undirected != 0, false);
    } else {
        std::vector<std::tuple<std::string, std::string>> list;
        for (uint64_t i = 0; i < n; i++) list.emplace_back(std::to_string(edges[2 * i]), std::to_string(edges[2 * i + 1]));
        h->graph.load_edge_list(list, as_undirected != 0, false);
    }
    std::vector<int> devices;
    for (int i = 0; i < num_worker; i++) devices.push_back(i);
    gvref_generator_count = 0;  // generator index == sampler index (SolverMixin constructor, solver.h:212-214)
    h->solver = new SolverT(devices, num_sampler_per_worker);
    h->solver->build(h->graph, chosen_optimizer(), num_partition, num_negative, batch_size, episode_size);
    return h;
}

void gvref_solver_destroy(void *handle) { delete static_cast<Handle *>(handle); }

// out: num_vertex, num_edge, num_directed_edge, num_partition, episode_size, head_partition_size, num_sampler, num_worker
void gvref_solver_info(void *handle, int64_t *out) {
    Handle *h = static_cast<Handle *>(handle);
    SolverT &s = *h->solver;
    h->graph.flatten();
    out[0] = s.num_vertex, out[1] = s.num_edge, out[2] = (int64_t)h->graph.edges.size(), out[3] = s.num_partition;
    out[4] = s.episode_size, out[5] = s.head_partition_size, out[6] = s.num_sampler, out[7] = s.num_worker;
}

// names[v] = the decimal label the vertex was loaded with; (part, local) = head_locations (solver.h:399-410)
void gvref_solver_partition(void *handle, uint32_t *labels, int32_t *part, uint32_t *local, float *vertex_weights) {
    Handle *h = static_cast<Handle *>(handle);
    SolverT &s = *h->solver;
    for (uint32_t v = 0; v < s.num_vertex; v++) {
        labels[v] = (uint32_t)std::stoul(h->graph.id2name[v]);
        part[v] = s.head_locations[v].first;
        local[v] = s.head_locations[v].second;
        vertex_weights[v] = h->graph.vertex_weights[v];
    }
}

// flattened directed edges {u, v} and weights in the order GraphMixin::flatten emits them (core/graph.h:87-101)
void gvref_solver_edges(void *handle, uint32_t *uv, float *weights) {
    Handle *h = static_cast<Handle *>(handle);
    h->graph.flatten();
    for (size_t e = 0; e < h->graph.edges.size(); e++) {
        uv[2 * e] = std::get<0>(h->graph.edges[e]);
        uv[2 * e + 1] = std::get<1>(h->graph.edges[e]);
        weights[e] = h->graph.edge_weights[e];
    }
}

// get_schedule() (solver.h:519-575): out[(step * num_worker + worker) * 2 + {0, 1}]; returns the number of steps
int gvref_solver_schedule(void *handle, int32_t *out, int capacity_steps) {
    SolverT &s = *static_cast<Handle *>(handle)->solver;
    auto schedule = s.get_schedule();
    int step = 0;
    for (auto &&assignment : schedule) {
        if (step >= capacity_steps) return -1;
        for (size_t w = 0; w < assignment.size(); w++) {
            out[(step * s.num_worker + w) * 2] = assignment[w].first;
            out[(step * s.num_worker + w) * 2 + 1] = assignment[w].second;
        }
        step++;
    }
    return step;
}

// What train() does before the first episode (graph.cuh:770-793, solver.h:588-625): configure, get_sample_function()
// (builds the alias tables), then every sampler fills its slice [work_load * i, ...) of the pools of pool_id ^ 1 —
// run one after the other here; the slices are disjoint.  pools: [P][P][pool_size] {tail, head} local ids.
int gvref_solver_sample(void *handle, const char *model, int augmentation_step, int walk_length, int walk_batch,
                        int shuffle_base, float p, float q, uint32_t *pools) {
    Handle *h = static_cast<Handle *>(handle);
    SolverT &s = *h->solver;
    s.model = model;
    s.augmentation_step = augmentation_step;
    s.random_walk_length = walk_length;
    s.random_walk_batch_size = walk_batch;
    s.shuffle_base = (s.model == "DeepWalk" || s.model == "node2vec") ? 1 : shuffle_base;
    s.p = p;
    s.q = q;
    s.sample_batch_size = walk_length * walk_batch;
    s.pool_id = 0;
    auto sample_function = s.get_sample_function();
    const int num_sample = s.episode_size * s.batch_size;
    const int work_load = (num_sample + s.num_sampler - 1) / s.num_sampler;
    for (int i = 0; i < s.num_sampler; i++)
        sample_function(s.samplers[i], work_load * i, std::min(work_load * (i + 1), num_sample));
    const int P = s.num_partition;
    auto &pool_set = s.sample_pools[s.pool_id ^ 1];
    for (int hp = 0; hp < P; hp++)
        for (int tp = 0; tp < P; tp++)
            for (int i = 0; i < num_sample; i++) {
                const auto &sample = pool_set[hp][tp][i];
                uint32_t *record = pools + (((size_t)hp * P + tp) * num_sample + i) * 2;
                record[0] = std::get<1>(sample);  // tail
                record[1] = std::get<0>(sample);  // head
            }
    return num_sample;
}

// GraphSolver::train (graph.cuh:770-793 -> solver.h:588-654) as written: sampler threads and worker threads, episode
// after episode; only the kernel launch and the negative draw inside a worker are the host loops above.  Afterwards
// vertex / context receive the embeddings (num_vertex x GVREF_DIM each, global ids).
int gvref_solver_train(void *handle, const char *model, int num_epoch, int augmentation_step, int walk_length,
                       int walk_batch, int shuffle_base, float p, float q, float negative_sample_exponent,
                       float negative_weight, float *vertex, float *context) {
    SolverT &s = *static_cast<Handle *>(handle)->solver;
    s.train(model, num_epoch, false, augmentation_step, walk_length, walk_batch, shuffle_base, p, q, 1,
            negative_sample_exponent, negative_weight, 1 << 30);
    for (uint32_t v = 0; v < s.num_vertex; v++) {
        memcpy(vertex + (size_t)v * GVREF_DIM, &(*s.vertex_embeddings)[v], GVREF_DIM * sizeof(float));
        memcpy(context + (size_t)v * GVREF_DIM, &(*s.context_embeddings)[v], GVREF_DIM * sizeof(float));
    }
    return (int)s.batch_id;
}

// WorkerMixin::load_partition -> build_negative_sampler (solver.h:1263-1278,1435-1496) of worker `worker` for block
// (head_partition, tail_partition): the negative sampler GraphSolver binds to the tail partition, degree^exponent in
// the partition's local order.  Returns the table size.
uint64_t gvref_solver_negative_table(void *handle, int worker, int head_partition, int tail_partition, float exponent,
                                     float *prob, uint32_t *alias, uint64_t capacity) {
    SolverT &s = *static_cast<Handle *>(handle)->solver;
    s.negative_sample_exponent = exponent;
    s.is_train = false;  // nothing to write back
    s.workers[worker]->load_partition(head_partition, tail_partition);
    auto &table = s.workers[worker]->negative_sampler;
    for (uint64_t i = 0; i < table.count && i < capacity; i++) prob[i] = table.prob_table[i], alias[i] = table.alias_table[i];
    return table.count;
}

// ---- the reference's text loaders on their own --------------------------------------------------------------------
// kind 0: Graph::load_file (graph.cuh:163-201); kind 1: WordGraph::load_file_compact (word_graph.cuh:73-181), for
// which a = window, b = min_count.  The graph is returned flattened (core/graph.h:87-101).
struct LoadedGraph {
    GraphT graph;
    graphvite::WordGraph<uint32_t> words;
    graphvite::Graph<uint32_t> *which = nullptr;
};

void *gvref_graph_load(int kind, const char *file_name, int a, int b, int normalization, const char *delimiters,
                       const char *comment) {
    LoadedGraph *g = new LoadedGraph();
    if (kind == 0) {
        g->graph.load_file(file_name, a != 0, normalization != 0, delimiters, comment);
        g->which = &g->graph;
    } else {
        g->words.load_file_compact(file_name, a, b, normalization != 0, delimiters, comment);
        g->which = &g->words;
    }
    g->which->flatten();
    return g;
}

void gvref_graph_destroy(void *handle) { delete static_cast<LoadedGraph *>(handle); }

// out: num_vertex, num_edge, number of flattened edges, total bytes of the names joined by '\n'
void gvref_graph_info(void *handle, int64_t *out) {
    auto *g = static_cast<LoadedGraph *>(handle)->which;
    size_t bytes = 0;
    for (auto &&name : g->id2name) bytes += name.size() + 1;
    out[0] = g->num_vertex, out[1] = g->num_edge, out[2] = (int64_t)g->edges.size(), out[3] = (int64_t)bytes;
}

void gvref_graph_data(void *handle, char *names, uint32_t *uv, float *edge_weights, float *vertex_weights) {
    auto *g = static_cast<LoadedGraph *>(handle)->which;
    for (auto &&name : g->id2name) {
        memcpy(names, name.data(), name.size());
        names += name.size();
        *names++ = '\n';
    }
    for (size_t e = 0; e < g->edges.size(); e++) {
        uv[2 * e] = std::get<0>(g->edges[e]);
        uv[2 * e + 1] = std::get<1>(g->edges[e]);
        edge_weights[e] = g->edge_weights[e];
    }
    for (size_t v = 0; v < g->vertex_weights.size(); v++) vertex_weights[v] = g->vertex_weights[v];
}

// alias tables get_sample_function() built: which = 0 the global edge table, 1 vertex_edge_tables[index],
// 2 edge_edge_tables[index]; returns the table size (0 when the vertex / edge has no table)
uint64_t gvref_solver_table(void *handle, int which, uint64_t index, float *prob, uint64_t *alias, uint64_t capacity) {
    SolverT &s = *static_cast<Handle *>(handle)->solver;
    uint64_t count = 0;
    if (which == 0) {
        count = s.edge_table.count;
        for (uint64_t i = 0; i < count && i < capacity; i++) prob[i] = s.edge_table.prob_table[i], alias[i] = s.edge_table.alias_table[i];
    } else {
        auto &table = which == 1 ? s.vertex_edge_tables[index] : s.edge_edge_tables[index];
        count = table.count;
        for (uint64_t i = 0; i < count && i < capacity; i++) prob[i] = table.prob_table[i], alias[i] = table.alias_table[i];
    }
    return count;
}

}  // extern "C"
