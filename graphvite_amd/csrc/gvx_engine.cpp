// gvx_engine.cpp — the native solver engine behind include/gvx.h: the orchestration of the reference's GraphSolver /
// SolverMixin / WorkerMixin (include/instance/graph.cuh:586-813, include/core/solver.h:87-888,1170-1623) as a C++ host
// runtime for MI355X.  Not a translation of it:
//   * one HOST thread issues everything; a worker is a set of HIP streams on its GPU, not an OS thread.  Every block is
//     one H2D copy + (optionally) one regrouping pass + episode_size back-to-back kernel launches, all asynchronous, so
//     the host is never the bottleneck and the reference's thread-per-worker joins per schedule step disappear;
//   * a worker keeps the WHOLE vertex table [P][1 + m][S][dim] and the context shards of the tail partitions it owns
//     for good (288 GB of HBM per GPU); nothing is evicted, reloaded or rebuilt between schedule steps
//     (WorkerMixin::load_partition / write_back, solver.h:1435-1504, have no counterpart);
//   * after a schedule step every worker copies the head shard it trained straight into the replicas of the other
//     workers (hipMemcpyPeerAsync over xGMI, on its exchange stream); consumers wait on events, the host does not;
//   * CPU sampler threads (gvs_sampler_fill) fill the next episode's pinned pools while the GPUs train this one.
// The arithmetic lives in the kernels (gvk.h); this file moves no embedding through the CPU during training.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "gvk_internal.h"
#include "gvx.h"

// Stream / event plumbing and clean-up calls are not checked one by one: a failure there is sticky in the HIP runtime
// and surfaces at the next checked call (every allocation, copy and launch is checked).
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"

namespace {

constexpr int kMaxPartition = 16;  // solver.h:51-57
constexpr int kMinBatchSize = 10000;
constexpr int kSamplePerVertex = 175;
constexpr int kMinEpisodeSample = 20000000;
constexpr int kExpectedDegree = 1600;  // graph.cuh:55
constexpr float kMaxNegativeWeight = 10;
constexpr size_t kChunkBytes = (size_t)256 << 20;  // host <-> device table traffic goes through chunks of this size

// ---- logging -----------------------------------------------------------------------------------------------------

int g_log_threshold = 0;
void (*g_log_sink)(int, const char *, void *) = nullptr;
void *g_log_user = nullptr;
std::mutex g_log_mutex;

void log_message(int severity, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void log_message(int severity, const char *fmt, ...) {
    if (severity < g_log_threshold) return;
    char text[4096];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(text, sizeof(text), fmt, ap);
    va_end(ap);
    std::lock_guard<std::mutex> lock(g_log_mutex);
    if (g_log_sink)
        g_log_sink(severity, text, g_log_user);
    else
        fprintf(stderr, "%s\n", text);
}

std::string size_string(double size) {  // util/io.h:41-58
    char buf[64];
    if (size >= (double)((size_t)1 << 40)) snprintf(buf, sizeof(buf), "%.3g TiB", size / (double)((size_t)1 << 40));
    else if (size >= (1 << 30)) snprintf(buf, sizeof(buf), "%.3g GiB", size / (1 << 30));
    else if (size >= (1 << 20)) snprintf(buf, sizeof(buf), "%.3g MiB", size / (1 << 20));
    else if (size >= (1 << 10)) snprintf(buf, sizeof(buf), "%.3g KiB", size / (1 << 10));
    else snprintf(buf, sizeof(buf), "%d B", (int)size);
    return buf;
}

std::string header(const std::string &content) {  // util/io.h:86-103
    const int width = 40, pad = std::max(width - (int)content.size() - 2, 0);
    return std::string(pad / 2, '-') + " " + content + " " + std::string(pad - pad / 2, '-');
}

int cpu_budget() { return gvk_cpu_budget(); }

struct Range {  // roctx range around a host phase (gvk_range_push / _pop): the reference's Timer scopes, time.h:28-60
    explicit Range(const char *name) { gvk_range_push(name); }
    ~Range() { gvk_range_pop(); }
};

#define HIP_TRY(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return gvk_fail(e_ == hipErrorOutOfMemory ? GVK_ENOMEM : GVK_EHIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)
#define GVK_TRY(call)            \
    do {                         \
        int rc_ = (call);        \
        if (rc_ != GVK_OK) return rc_; \
    } while (0)

struct Worker {
    int device = 0;
    hipStream_t compute = nullptr, copy = nullptr, exchange = nullptr;
    float *head = nullptr;     // [P slots][1 + m][S][dim]: every head partition, vertex rows then moment tables
    float *context = nullptr;  // [T][1 + m][S][dim]: the tail partitions this worker owns
    std::vector<int> tails;
    std::vector<gvk_alias_entry *> negative_tables;  // per owned tail: one alias slot per row, or null when ...
    std::vector<gvk_class_entry *> negative_classes; // ... the sampler draws by weight class (gvk.h): an alias table over
    std::vector<uint32_t> negative_class_counts;     // the classes of equal-degree rows, a few thousand entries
    float *loss = nullptr;
    uint32_t *pool[2] = {nullptr, nullptr}, *landing = nullptr;
    void *group_workspace = nullptr;
    size_t group_workspace_bytes = 0;
    hipEvent_t uploaded[2] = {nullptr, nullptr}, released[2] = {nullptr, nullptr}, trained = nullptr;
    bool released_valid[2] = {false, false};
    std::vector<hipEvent_t> copied;     // H2D copies of the current pool set still reading pinned memory
    std::vector<hipEvent_t> incoming;   // peer copies into this worker's replica not yet fenced on its compute stream
    hipEvent_t sent = nullptr;          // the peer copies of the head partition this worker trained last have left it
    int sending = -1;                   // ... that partition (-1: nothing in flight)
    uint64_t visits = 0;
    // positive samples drawn on the device (GVX_DEVICE_SAMPLING)
    struct EdgeBlock {
        gvk_edge_entry *table = nullptr;  // alias table over the weights of the block's edges + their {tail, head} records
        uint32_t count = 0;
    };
    hipStream_t sample = nullptr;
    hipEvent_t filled[2] = {nullptr, nullptr}, episode_end = nullptr;
    bool episode_end_valid = false;
    uint32_t *block_pools[2] = {nullptr, nullptr};  // [tails][P][n] records: the pools of every block it trains, two episodes
    uint32_t *slices = nullptr;                     // walks, several workers: [P * P][n / W], its slice of EVERY block
    std::vector<EdgeBlock> edge_blocks;             // edge mode: per block, index = tail index * P + head partition
    gvk_walk_graph walk{};                          // walk modes: the graph in HBM
    int32_t *walk_part = nullptr;
    uint64_t *walk_offsets = nullptr;
    uint32_t *walk_counters = nullptr;
    uint64_t sample_seed = 0, sample_index = 0;
};

}  // namespace

struct gvx_solver {
    // resources
    int dim = 0;
    std::vector<int> device_ids;
    int num_worker = 0, num_sampler_per_worker = 0, num_sampler = 0;
    size_t memory_request = 0, gpu_memory_limit = 0, gpu_memory_cost = 0;
    // build
    const gvs_graph *graph = nullptr;
    gvx_optimizer optimizer{};
    int num_moment = 0;
    uint32_t num_vertex = 0;
    uint64_t num_edge = 0;
    int num_partition = 0, num_negative = 0, batch_size = 0, episode_size = 0;
    // false: every worker keeps all head partitions and its context shards in HBM (the design of this engine);
    // true: the model does not fit that way — a worker holds ONE head and ONE tail partition and they travel through host
    // memory between blocks, the reference's load_partition / write_back scheme (solver.h:1435-1504)
    bool streamed = false;
    bool device_sampling = false;  // gvx_solver_set(GVX_DEVICE_SAMPLING)
    std::vector<uint64_t> sample_positions;  // per worker: draws (edge mode) / walks consumed since build()
    std::vector<int32_t> part;
    std::vector<uint32_t> local, part_sizes;
    uint32_t part_rows = 0;  // S
    std::vector<std::vector<uint32_t>> part_ids;  // global ids of a partition in local order
    std::vector<int32_t> schedule;                // [steps][W][2]
    int num_step = 0;
    std::vector<float> vertex, context;           // host embeddings, global order (the numpy views)
    std::vector<std::vector<float>> vertex_moments, context_moments;  // kept across train(resume=True)
    // train
    std::string model;
    gvx_train_config config{};
    int mode = 0;
    uint64_t batch_id = 0, num_batch = 0;
    double train_seconds = 0;
    gvs_sampler *sampler = nullptr;
    int sampler_mode = -1;
    float sampler_p = 0, sampler_q = 0;
    std::vector<Worker> workers;
    std::string info_text;

    ~gvx_solver() { release(); }

    void release_device() {
        if (device_sampling && !workers.empty()) {  // the device samplers go on where they stopped at the next train()
            sample_positions.assign(workers.size(), 0);
            for (size_t r = 0; r < workers.size(); r++) sample_positions[r] = workers[r].sample_index;
        }
        for (Worker &w : workers) {
            hipSetDevice(w.device);
            hipDeviceSynchronize();
            hipFree(w.head), hipFree(w.context), hipFree(w.loss), hipFree(w.pool[0]), hipFree(w.pool[1]);
            hipFree(w.landing), hipFree(w.group_workspace);
            for (auto *t : w.negative_tables) hipFree(t);
            for (auto *t : w.negative_classes) hipFree(t);
            hipFree(w.block_pools[0]), hipFree(w.block_pools[1]), hipFree(w.slices);
            for (auto &b : w.edge_blocks) hipFree(b.table);
            hipFree((void *)w.walk.flat_offsets), hipFree((void *)w.walk.edges_uv), hipFree((void *)w.walk.edge_table);
            hipFree((void *)w.walk.neighbor_table), hipFree((void *)w.walk.sorted_neighbors), hipFree((void *)w.walk.local);
            hipFree(w.walk_part), hipFree(w.walk_offsets), hipFree(w.walk_counters);
            for (hipEvent_t e : w.filled)
                if (e) hipEventDestroy(e);
            if (w.episode_end) hipEventDestroy(w.episode_end);
            if (w.sample) hipStreamDestroy(w.sample);
            for (int b = 0; b < 2; b++) {
                if (w.uploaded[b]) hipEventDestroy(w.uploaded[b]);
                if (w.released[b]) hipEventDestroy(w.released[b]);
            }
            if (w.trained) hipEventDestroy(w.trained);
            if (w.sent) hipEventDestroy(w.sent);
            for (auto e : w.copied) hipEventDestroy(e);
            for (auto e : w.incoming) hipEventDestroy(e);
            if (w.compute) hipStreamDestroy(w.compute);
            if (w.copy) hipStreamDestroy(w.copy);
            if (w.exchange) hipStreamDestroy(w.exchange);
        }
        workers.clear();
    }

    void release() {
        release_device();
        if (sampler) gvs_sampler_destroy(sampler);
        sampler = nullptr;
        sampler_mode = -1;
    }

    size_t table_floats() const { return (size_t)part_rows * dim; }
    size_t slot_floats() const { return table_floats() * (1 + num_moment); }
    float *head_table(Worker &w, int hp, int table) {
        return w.head + ((size_t)(streamed ? 0 : hp) * (1 + num_moment) + table) * table_floats();
    }
    float *context_table(Worker &w, int ti, int table) {
        return w.context + ((size_t)(streamed ? 0 : ti) * (1 + num_moment) + table) * table_floats();
    }

    size_t memory_demand(int P, int requested_episode, bool as_streamed = false) const;
    int load_block(Worker &w, int hp, int tp);
    int store_block(Worker &w, int hp, int tp);
    int configure(const gvx_train_config &c);
    int prepare_devices();
    int upload();
    int write_back();
    int move_table(bool to_device, Worker &w, float *device_table, std::vector<float> &host, int partition);
    int episode_loop();
    int train_block(Worker &w, int hp, int tp, uint32_t *pool);
    int fill(std::vector<uint32_t *> &pools);
    int prepare_device_sampling();
    int device_fill(int set);
    uint32_t *block_pool(Worker &w, int set, int hp, int tp) {
        const size_t ti = std::find(w.tails.begin(), w.tails.end(), tp) - w.tails.begin();
        return w.block_pools[set] + (ti * num_partition + hp) * (size_t)episode_size * batch_size * 2;
    }
    void make_info();
};

namespace {

const char *optimizer_name(int type) {
    static const char *const names[] = {"SGD", "Momentum", "AdaGrad", "RMSprop", "Adam"};
    return type >= 0 && type <= GVK_ADAM ? names[type] : "Default";
}

}  // namespace

size_t gvx_solver::memory_demand(int P, int requested_episode, bool as_streamed) const {
    const size_t S = (num_vertex + P - 1) / P, tails = std::max(P / num_worker, 1);
    // resident: all P head partitions + the owned context shards; streamed: one head + one tail partition (solver.h:365-376)
    size_t demand = (as_streamed ? 2 * S : P * S + tails * S) * (size_t)dim * 4 * (1 + num_moment);
    demand += (as_streamed ? (size_t)P : tails) * S * 8 + (size_t)batch_size * 4;
    size_t episode = requested_episode;
    if (requested_episode == GVX_AUTO) {
        episode = std::max<size_t>((size_t)((double)num_vertex * kSamplePerVertex / P / batch_size), 1);
        if (P == 1) episode = std::max<size_t>(episode, kMinEpisodeSample / batch_size);
    }
    demand += 3 * episode * batch_size * 8;  // two pool buffers + the regrouping landing buffer
    if (device_sampling && !as_streamed)  // the pools of every block a worker trains, two episodes, + its slices on their way to the owners
        demand += 3 * tails * P * episode * batch_size * 8;
    return demand;
}

void gvx_solver::make_info() {  // GraphSolver::info, graph.cuh:739-768 over SolverMixin::info, solver.h:768-825
    char buf[2048];
    std::string s;
    snprintf(buf, sizeof(buf), "GraphSolver<%d, float32, uint32>\n%s\n#worker: %d, #sampler: %d, #partition: %d\n"
             "tied weights: no, episode size: %d\ngpu memory limit: %s\ngpu memory cost: %s\n%s\n",
             dim, header("Resource").c_str(), num_worker, num_sampler, num_partition, episode_size,
             size_string((double)gpu_memory_limit).c_str(), size_string((double)gpu_memory_cost).c_str(),
             header("Sampling").c_str());
    s += buf;
    if (model == "LINE") snprintf(buf, sizeof(buf), "augmentation step: %d, shuffle base: %d\n", config.augmentation_step, config.shuffle_base);
    else if (model == "DeepWalk") snprintf(buf, sizeof(buf), "augmentation step: %d\n", config.augmentation_step);
    else if (model == "node2vec") snprintf(buf, sizeof(buf), "augmentation step: %d, p: %g, q: %g\n", config.augmentation_step, config.p, config.q);
    else buf[0] = 0;
    s += buf;
    snprintf(buf, sizeof(buf), "random walk length: %d\nrandom walk batch size: %d\n#negative: %d, negative sample exponent: %g\n"
             "%s\nmodel: %s\noptimizer: %s\nlearning rate: %g, lr schedule: %s\nweight decay: %g\n#epoch: %d, batch size: %d\n"
             "resume: %s\npositive reuse: %d, negative weight: %g",
             config.random_walk_length, config.random_walk_batch_size, num_negative, config.negative_sample_exponent,
             header("Training").c_str(), model.c_str(), optimizer_name(optimizer.type), optimizer.lr,
             optimizer.schedule == 0 ? "constant" : (optimizer.schedule == 1 ? "linear" : "custom"), optimizer.weight_decay,
             config.num_epoch, batch_size, config.resume ? "yes" : "no", config.positive_reuse, config.negative_weight);
    s += buf;
    info_text = s;
}

// ---- build -------------------------------------------------------------------------------------------------------

extern "C" gvx_solver *gvx_solver_create(int dim, const int *device_ids, int num_device, int num_sampler_per_worker,
                                         size_t gpu_memory_limit) {
    if (dim != 32 && dim != 64 && dim != 96 && dim != 128 && dim != 256 && dim != 512) {
        gvk_fail(GVK_EDIM, "GraphSolver: dim must be one of 32, 64, 96, 128, 256, 512");
        return nullptr;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        gvk_fail(GVK_EHIP, "No GPU devices found");  // solver.h:176
        return nullptr;
    }
    std::unique_ptr<gvx_solver> s(new gvx_solver());
    s->dim = dim;
    if (num_device <= 0)
        for (int i = 0; i < count; i++) s->device_ids.push_back(i);
    else
        for (int i = 0; i < num_device; i++) {
            if (device_ids[i] < 0 || device_ids[i] >= count) {
                gvk_fail(GVK_EINVAL, "Invalid GPU id `%d`: %d device(s) visible", device_ids[i], count);
                return nullptr;
            }
            s->device_ids.push_back(device_ids[i]);
        }
    s->num_worker = (int)s->device_ids.size();
    if (num_sampler_per_worker == GVX_AUTO)  // solver.h:193-194, over the CPUs this process may really use
        num_sampler_per_worker = std::max(cpu_budget() / s->num_worker - 1, 1);
    s->num_sampler_per_worker = num_sampler_per_worker;
    s->num_sampler = num_sampler_per_worker * s->num_worker;
    s->memory_request = gpu_memory_limit;
    s->gpu_memory_limit = gpu_memory_limit;
    s->config.random_walk_length = 40, s->config.random_walk_batch_size = 100, s->config.p = s->config.q = 1;
    s->config.positive_reuse = 1, s->config.log_frequency = 1000;
    return s.release();
}

extern "C" void gvx_solver_destroy(gvx_solver *s) { delete s; }

extern "C" int gvx_solver_set(gvx_solver *s, int option, int64_t value) {
    if (!s) return gvk_fail(GVK_EINVAL, "gvx_solver_set: null solver");
    if (option != GVX_DEVICE_SAMPLING || (value != 0 && value != 1))
        return gvk_fail(GVK_EINVAL, "gvx_solver_set: unknown option %d or unsupported value %lld", option, (long long)value);
    s->device_sampling = value != 0;
    return GVK_OK;
}

extern "C" int gvx_solver_build(gvx_solver *s, const gvs_graph *graph, const gvx_optimizer *optimizer, int num_partition,
                                int num_negative, int batch_size, int episode_size) {
    if (!s || !graph) return gvk_fail(GVK_EINVAL, "gvx_solver_build: null solver / graph");
    if (gvs_graph_num_vertex(graph) == 0 || gvs_graph_num_directed_edge(graph) == 0)
        return gvk_fail(GVK_EINVAL, "The graph is empty");
    if (batch_size < 1 || num_negative < 0 || episode_size < 0 || num_partition < 0)
        return gvk_fail(GVK_EINVAL, "batch_size must be positive; num_negative, num_partition, episode_size non-negative");
    s->release();
    s->graph = graph;
    gvx_optimizer opt{};
    if (optimizer) opt = *optimizer;
    else opt.type = -1;
    if (opt.type < 0) {  // solver.h:290-296 with GraphSolver::get_default_optimizer, graph.cuh:634-636
        const float lr = opt.lr > 0 ? opt.lr : 0.025f;
        opt = gvx_optimizer{};
        opt.type = GVK_SGD, opt.lr = lr, opt.weight_decay = 5e-3f, opt.schedule = 1;
    }
    if (opt.type > GVK_ADAM) return gvk_fail(GVK_EINVAL, "Unknown optimizer type %d", opt.type);
    if (opt.schedule == 2 && !opt.schedule_function) return gvk_fail(GVK_EINVAL, "custom lr schedule without a function");
    s->optimizer = opt;
    s->num_moment = opt.type == GVK_SGD ? 0 : (opt.type == GVK_ADAM ? 2 : 1);
    s->num_vertex = gvs_graph_num_vertex(graph);
    s->num_edge = gvs_graph_num_edge(graph);
    s->num_negative = num_negative, s->batch_size = batch_size;
    if (batch_size < kMinBatchSize)
        log_message(1, "It is recommended to a minimum batch size of %d, but %d is specified", kMinBatchSize, batch_size);
    s->batch_id = 0;
    s->sample_positions.clear();
    const int W = s->num_worker;
    size_t limit = s->memory_request;
    if (limit == GVX_AUTO) {
        limit = (size_t)-1;
        for (int d : s->device_ids) {
            size_t free_bytes = 0, total = 0;
            HIP_TRY(hipSetDevice(d));
            HIP_TRY(hipMemGetInfo(&free_bytes, &total));
            limit = std::min(limit, free_bytes);
        }
    }
    s->streamed = false;
    if (num_partition == GVX_AUTO) {
        num_partition = W;
        while (num_partition < kMaxPartition && s->memory_demand(num_partition, episode_size) >= limit) num_partition += W;
        if (s->memory_demand(num_partition, episode_size) >= limit) {
            // the resident design does not fit even with the most partitions: fall back to the reference's scheme — one
            // head and one tail partition per worker in HBM, everything else in host memory (solver.h:365-384)
            s->streamed = true;
            num_partition = W;
            while (num_partition < kMaxPartition && s->memory_demand(num_partition, episode_size, true) >= limit) num_partition += W;
            log_message(1, "The vertex table does not fit the GPU memory limit next to the sample pools: partitions will "
                           "travel through host memory between blocks (%d partitions)", num_partition);
        }
    } else {
        s->streamed = s->memory_demand(num_partition, episode_size) >= limit &&
                      s->memory_demand(num_partition, episode_size, true) < limit;
        if (num_partition < W) return gvk_fail(GVK_EINVAL, "#partition should be no less than %d", W);
        if (num_partition % W) return gvk_fail(GVK_EINVAL, "#partition (%d) must be a multiple of #worker (%d)", num_partition, W);
        if (num_partition > kMaxPartition)
            log_message(1, "It is recommended to use a maximum #partition of %d, but %d partitions are specified",
                        kMaxPartition, num_partition);
    }
    const int P = s->num_partition = num_partition;
    s->gpu_memory_limit = limit;
    s->gpu_memory_cost = s->memory_demand(P, episode_size, s->streamed);
    if (s->gpu_memory_cost >= limit) return gvk_fail(GVK_ENOMEM, "Can't satisfy the specified GPU memory limit");

    s->part.assign(s->num_vertex, 0), s->local.assign(s->num_vertex, 0), s->part_sizes.assign(P, 0);
    GVK_TRY(gvs_partition(gvs_graph_vertex_weights(graph), s->num_vertex, P, s->part.data(), s->local.data(),
                          s->part_sizes.data()));
    s->part_rows = *std::max_element(s->part_sizes.begin(), s->part_sizes.end());
    s->part_ids.assign(P, {});
    for (int p = 0; p < P; p++) s->part_ids[p].assign(s->part_sizes[p], 0);
    for (uint32_t v = 0; v < s->num_vertex; v++) s->part_ids[s->part[v]][s->local[v]] = v;
    s->schedule.assign((size_t)std::max((P / W) * (P / W) * W, 1) * W * 2 + 2, 0);
    s->num_step = gvs_schedule(P, W, s->schedule.data(), s->schedule.size());
    if (s->num_step < 0) return s->num_step;

    if (episode_size == GVX_AUTO) {  // solver.h:426-436
        episode_size = std::max((int)((double)s->num_vertex * kSamplePerVertex / P / batch_size), 1);
        if (P == 1) episode_size = std::max(episode_size, kMinEpisodeSample / batch_size);
    }
    s->episode_size = episode_size;
    s->vertex.assign((size_t)s->num_vertex * s->dim, 0.0f);
    s->context.assign((size_t)s->num_vertex * s->dim, 0.0f);
    s->vertex_moments.clear(), s->context_moments.clear();
    s->make_info();
    return GVK_OK;
}

// ---- train: configuration ------------------------------------------------------------------------------------------

int gvx_solver::configure(const gvx_train_config &in) {
    if (!graph) return gvk_fail(GVK_EINVAL, "The model must be built on a graph first");
    gvx_train_config c = in;
    const std::string m = c.model ? c.model : "";
    if (m != "DeepWalk" && m != "LINE" && m != "node2vec") return gvk_fail(GVK_EINVAL, "Invalid model `%s`", m.c_str());
    if (c.augmentation_step == GVX_AUTO) {  // graph.cuh:781-784
        const double density = std::log((double)num_edge / num_vertex);
        c.augmentation_step = density != 0 ? (int)(std::log((double)kExpectedDegree) / density) : c.random_walk_length + 1;
    }
    if (c.shuffle_base == GVX_AUTO) c.shuffle_base = c.augmentation_step;
    if (m == "DeepWalk" || m == "node2vec") c.shuffle_base = 1;  // graph.cuh:785-786
    if (c.augmentation_step < 1) return gvk_fail(GVK_EINVAL, "`augmentation_step` should be a positive integer");
    if (c.augmentation_step > c.random_walk_length)
        return gvk_fail(GVK_EINVAL, "`random_walk_length` should be no less than `augmentation_step`");
    if (c.positive_reuse < 1 || c.log_frequency < 1 || c.num_epoch < 0)
        return gvk_fail(GVK_EINVAL, "positive_reuse / log_frequency must be positive and num_epoch non-negative");
    if (c.negative_weight > kMaxNegativeWeight)
        log_message(1, "It is recommended to a maximum negative weight of %g, but %g is specified", kMaxNegativeWeight,
                    c.negative_weight);
    const size_t pool_size = (size_t)episode_size * batch_size;
    if (c.augmentation_step > 1 && pool_size % c.shuffle_base)
        return gvk_fail(GVK_EINVAL, "Can't perform pseudo shuffle on %zu elements by a shuffle base of %d. Try setting the "
                        "episode size to a multiple of the shuffle base", pool_size, c.shuffle_base);
    model = m;
    config = c;
    config.model = model.c_str();
    make_info();
    log_message(1, "\n<<<<<<<<<<<<<<<<<<<<<<<<<<<<<<<<<<<<<<<<\n%s\n>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>", info_text.c_str());
    if (!c.resume) {  // GraphSolver::init_embeddings, graph.cuh:724-731
        std::mt19937 seed(5489u);
        std::uniform_real_distribution<float> init(-0.5f / dim, 0.5f / dim);
        for (float &x : vertex) x = init(seed);
        std::fill(context.begin(), context.end(), 0.0f);
        vertex_moments.clear(), context_moments.clear();
        batch_id = 0;
    }
    num_batch = batch_id + (uint64_t)c.num_epoch * num_edge / batch_size;  // solver.h:611
    mode = c.augmentation_step == 1 ? GVS_MODE_EDGE : (m == "node2vec" ? GVS_MODE_BIASED_WALK : GVS_MODE_WALK);
    if (mode == GVS_MODE_BIASED_WALK) {
        // the per-edge alias tables need sum over edges of deg(head) entries (graph.cuh:656-677); past 2^30 entries the
        // same transition distribution is sampled by rejection over the per-vertex tables (gvs.h GVS_MODE_BIASED_REJECT)
        const uint64_t *offsets = gvs_graph_flat_offsets(graph);
        const uint32_t *uv = gvs_graph_edges(graph);
        const uint64_t D = gvs_graph_num_directed_edge(graph);
        uint64_t entries = 0;
        for (uint64_t e = 0; e < D; e++) entries += offsets[uv[2 * e + 1] + 1] - offsets[uv[2 * e + 1]];
        if (entries > ((uint64_t)1 << 30)) {
            log_message(1, "node2vec: %llu per-edge table entries exceed 2^30; sampling by rejection", (unsigned long long)entries);
            mode = GVS_MODE_BIASED_REJECT;
        }
    }
    if (device_sampling) {  // the GPUs draw the positives: no CPU sampler, none of its tables
        if (streamed)
            return gvk_fail(GVK_EINVAL, "device sampling needs every partition resident in GPU memory; raise gpu_memory_limit "
                                        "or use the CPU samplers");
        if (gvs_graph_num_directed_edge(graph) >= ((uint64_t)1 << 32))
            return gvk_fail(GVK_EINVAL, "device sampling supports graphs with fewer than 2^32 directed edges");
        if (mode != GVS_MODE_EDGE) {
            if (pool_size % num_worker)
                return gvk_fail(GVK_EINVAL, "episode_size * batch_size (%zu) must be a multiple of #worker (%d) for the "
                                "random-walk models", pool_size, num_worker);
            if (pool_size / num_worker % c.shuffle_base)
                return gvk_fail(GVK_EINVAL, "Can't perform pseudo shuffle on %zu elements by a shuffle base of %d",
                                pool_size / num_worker, c.shuffle_base);
        }
        return GVK_OK;
    }
    if (!sampler) {
        sampler = gvs_sampler_create(graph, part.data(), local.data(), num_partition, 0x9E3779B97F4A7C15ull);
        if (!sampler) return GVK_EINVAL;
    }
    if (sampler_mode != mode || sampler_p != c.p || sampler_q != c.q) {  // get_sample_function, graph.cuh:680-721
        GVK_TRY(gvs_sampler_prepare(sampler, mode, c.p, c.q, num_sampler + 1));
        sampler_mode = mode, sampler_p = c.p, sampler_q = c.q;
    }
    return GVK_OK;
}

// ---- device state --------------------------------------------------------------------------------------------------

int gvx_solver::prepare_devices() {
    release_device();
    const int P = num_partition, W = num_worker, nm = num_moment;
    workers.assign(W, Worker());
    for (int r = 0; r < W; r++) {
        Worker &w = workers[r];
        w.device = device_ids[r];
        for (int step = 0; step < num_step; step++) {
            const int tp = schedule[((size_t)step * W + r) * 2 + 1];
            if (std::find(w.tails.begin(), w.tails.end(), tp) == w.tails.end()) w.tails.push_back(tp);
        }
        std::sort(w.tails.begin(), w.tails.end());
        if (streamed) {  // any tail partition may come by: a negative sampler for each, tables for one at a time
            w.tails.clear();
            for (int p = 0; p < P; p++) w.tails.push_back(p);
        }
        HIP_TRY(hipSetDevice(w.device));
        for (int q = 0; q < W; q++)  // direct GPU-to-GPU copies for the exchange
            if (device_ids[q] != w.device) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, w.device, device_ids[q]) == hipSuccess && can) {
                    hipError_t e = hipDeviceEnablePeerAccess(device_ids[q], 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                }
            }
        HIP_TRY(hipStreamCreateWithFlags(&w.compute, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&w.copy, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&w.exchange, hipStreamNonBlocking));
        const size_t head_slots = streamed ? 1 : (size_t)P, context_slots = streamed ? 1 : w.tails.size();
        HIP_TRY(hipMalloc(&w.head, head_slots * slot_floats() * 4));
        HIP_TRY(hipMalloc(&w.context, context_slots * slot_floats() * 4));
        HIP_TRY(hipMemsetAsync(w.head, 0, head_slots * slot_floats() * 4, w.compute));
        HIP_TRY(hipMemsetAsync(w.context, 0, context_slots * slot_floats() * 4, w.compute));
        HIP_TRY(hipMalloc(&w.loss, (size_t)batch_size * 4));
        HIP_TRY(hipMemsetAsync(w.loss, 0, (size_t)batch_size * 4, w.compute));
        for (int b = 0; b < 2; b++) {
            HIP_TRY(hipEventCreateWithFlags(&w.uploaded[b], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&w.released[b], hipEventDisableTiming));
        }
        HIP_TRY(hipEventCreateWithFlags(&w.trained, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&w.sent, hipEventDisableTiming));
        // negative sampler per owned tail partition: degree^exponent in local order (solver.h:1263-1278)
        for (int tp : w.tails) {
            const std::vector<uint32_t> &ids = part_ids[tp];
            std::vector<float> weights(ids.size()), prob(ids.size());
            std::vector<uint32_t> alias(ids.size());
            std::vector<gvk_alias_entry> packed(ids.size());
            GVK_TRY(gvs_negative_weights(gvs_graph_vertex_weights(graph), ids.data(), ids.size(),
                                         config.negative_sample_exponent, weights.data()));
            // rows of equal weight (= equal degree: the partition is sorted by it) form a class; when the classes are at
            // least 8 times fewer than the rows (always, on an unweighted graph) the sampler draws class, then row
            std::vector<gvk_class_entry> classes(ids.size());
            uint32_t num_class = 0;
            GVK_TRY(gvk_class_table_build(weights.data(), weights.size(), classes.data(), &num_class));
            gvk_alias_entry *table = nullptr;
            gvk_class_entry *class_table = nullptr;
            if ((size_t)num_class * 8 <= ids.size()) {
                HIP_TRY(hipMalloc(&class_table, num_class * sizeof(gvk_class_entry)));
                HIP_TRY(hipMemcpy(class_table, classes.data(), num_class * sizeof(gvk_class_entry), hipMemcpyHostToDevice));
            } else {
                num_class = 0;
                GVK_TRY(gvk_alias_build(weights.data(), weights.size(), prob.data(), alias.data(), 4, packed.data()));
                HIP_TRY(hipMalloc(&table, packed.size() * sizeof(gvk_alias_entry)));
                HIP_TRY(hipMemcpy(table, packed.data(), packed.size() * sizeof(gvk_alias_entry), hipMemcpyHostToDevice));
            }
            w.negative_tables.push_back(table);
            w.negative_classes.push_back(class_table);
            w.negative_class_counts.push_back(num_class);
        }
        (void)nm;
    }
    // the pools are the elastic part, as in the reference (solver.h:437-455): halve the episode until they fit
    while (true) {
        bool ok = true;
        const size_t bytes = (size_t)episode_size * batch_size * 8;
        for (Worker &w : workers) {
            hipSetDevice(w.device);
            ok = ok && hipMalloc(&w.pool[0], bytes) == hipSuccess && hipMalloc(&w.pool[1], bytes) == hipSuccess &&
                 hipMalloc(&w.landing, bytes) == hipSuccess;
            if (!ok) break;
        }
        if (ok) break;
        (void)hipGetLastError();
        for (Worker &w : workers) {
            hipSetDevice(w.device);
            hipFree(w.pool[0]), hipFree(w.pool[1]), hipFree(w.landing);
            w.pool[0] = w.pool[1] = w.landing = nullptr;
        }
        if (episode_size <= 1)
            return gvk_fail(GVK_ENOMEM, "Out of GPU memory. Try to reduce the size of your graph or the dimension of your embeddings.");
        const int base = config.augmentation_step > 1 ? std::max(config.shuffle_base, 1) : 1;
        int half = episode_size / 2;
        while (half > 1 && ((size_t)half * batch_size) % base) half--;
        log_message(1, "Fail to allocate GPU memory for episode size of %d. Use %d instead.", episode_size, std::max(half, 1));
        episode_size = std::max(half, 1);
    }
    for (Worker &w : workers) {
        HIP_TRY(hipSetDevice(w.device));
        const int row_bits = std::max(32 - __builtin_clz(std::max(part_rows, 2u) - 1), 1);
        const int parts = gvk_train_launches(batch_size, part_rows);
        GVK_TRY(gvk_group_pairs(nullptr, nullptr, nullptr, nullptr, &w.group_workspace_bytes, batch_size / parts,
                                episode_size * parts, row_bits));
        HIP_TRY(hipMalloc(&w.group_workspace, std::max<size_t>(w.group_workspace_bytes, 16)));
    }
    return device_sampling ? prepare_device_sampling() : GVK_OK;
}

// ---- positive samples drawn on the device (GVX_DEVICE_SAMPLING) ------------------------------------------------------

namespace {

template <class T>
int to_device(T **out, const T *host, size_t count) {  // on the current device
    HIP_TRY(hipMalloc((void **)out, std::max<size_t>(count, 1) * sizeof(T)));
    if (count) HIP_TRY(hipMemcpy(*out, host, count * sizeof(T), hipMemcpyHostToDevice));
    return GVK_OK;
}

int largest_divisor(size_t n, int limit) {  // stripes of gvk_sample_walks_blocks: any divisor of the capacity
    for (int d = limit; d > 1; d--)
        if (n % d == 0) return d;
    return 1;
}

}  // namespace

// What the samplers of the device draw from: per block a worker trains, the block's edges and an alias table over their
// weights (augmentation_step 1: a positive sample of a block is one of its edges, gvk_sample_edges); for the walk modes
// the whole graph — CSR, per-vertex alias tables, the global edge table, the partition map (gvk_sample_walks_blocks).
int gvx_solver::prepare_device_sampling() {
    const int P = num_partition, W = num_worker;
    const uint64_t D = gvs_graph_num_directed_edge(graph);
    const uint32_t *uv = gvs_graph_edges(graph);
    const float *weights = gvs_graph_edge_weights(graph);
    const size_t n = (size_t)episode_size * batch_size;
    std::vector<std::vector<uint32_t>> of_block;
    std::vector<gvk_alias_entry> edge_table, neighbor_table;
    std::vector<uint32_t> sorted_neighbors;
    const bool biased = mode == GVS_MODE_BIASED_WALK || mode == GVS_MODE_BIASED_REJECT;
    if (mode == GVS_MODE_EDGE) {
        of_block.assign((size_t)P * P, {});
        for (uint64_t e = 0; e < D; e++) of_block[(size_t)part[uv[2 * e]] * P + part[uv[2 * e + 1]]].push_back((uint32_t)e);
    } else {
        std::vector<float> prob(D);
        std::vector<uint32_t> alias(D);
        edge_table.resize(D), neighbor_table.resize(D);
        GVK_TRY(gvk_alias_build(weights, D, prob.data(), alias.data(), 4, edge_table.data()));
        GVK_TRY(gvs_graph_neighbor_tables(graph, num_sampler + 1, neighbor_table.data()));
        if (biased) {  // out-neighbours ascending inside each vertex's CSR segment (the acceptance test searches them)
            const uint64_t *offsets = gvs_graph_flat_offsets(graph);
            sorted_neighbors.resize(D);
            for (uint64_t e = 0; e < D; e++) sorted_neighbors[e] = uv[2 * e + 1];
            for (uint32_t v = 0; v < num_vertex; v++)
                std::sort(sorted_neighbors.begin() + offsets[v], sorted_neighbors.begin() + offsets[v + 1]);
        }
    }
    for (int r = 0; r < W; r++) {
        Worker &w = workers[r];
        HIP_TRY(hipSetDevice(w.device));
        HIP_TRY(hipStreamCreateWithFlags(&w.sample, hipStreamNonBlocking));
        for (hipEvent_t &e : w.filled) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&w.episode_end, hipEventDisableTiming));
        w.sample_seed = 0x9E3779B97F4A7C15ull * (uint64_t)(r + 1) + 0x706f73;
        w.sample_index = (size_t)r < sample_positions.size() ? sample_positions[r] : 0;
        const size_t blocks = w.tails.size() * P;
        for (uint32_t *&pools : w.block_pools) HIP_TRY(hipMalloc(&pools, blocks * n * 8));
        if (mode == GVS_MODE_EDGE) {
            w.edge_blocks.assign(blocks, Worker::EdgeBlock());
            for (size_t ti = 0; ti < w.tails.size(); ti++)
                for (int hp = 0; hp < P; hp++) {
                    const std::vector<uint32_t> &ids = of_block[(size_t)hp * P + w.tails[ti]];
                    if (ids.empty())
                        return gvk_fail(GVK_EINVAL, "block (%d, %d) has no edges; use fewer partitions for this graph", hp,
                                        w.tails[ti]);
                    std::vector<float> block_weights(ids.size()), prob(ids.size());
                    std::vector<uint32_t> alias(ids.size());
                    std::vector<gvk_edge_entry> packed(ids.size());
                    for (size_t i = 0; i < ids.size(); i++) block_weights[i] = weights[ids[i]];
                    GVK_TRY(gvk_alias_build(block_weights.data(), ids.size(), prob.data(), alias.data(), 4, nullptr));
                    for (size_t i = 0; i < ids.size(); i++)  // records are {tail, head} in local ids
                        packed[i] = {prob[i], alias[i], local[uv[2 * (size_t)ids[i] + 1]], local[uv[2 * (size_t)ids[i]]]};
                    Worker::EdgeBlock &b = w.edge_blocks[ti * P + hp];
                    b.count = (uint32_t)ids.size();
                    GVK_TRY(to_device(&b.table, packed.data(), packed.size()));
                }
            continue;
        }
        gvk_walk_graph &g = w.walk;
        g.num_vertex = num_vertex, g.num_edge_entries = (uint32_t)D;
        g.biased = biased, g.p = config.p, g.q = config.q;
        GVK_TRY(to_device((uint64_t **)&g.flat_offsets, gvs_graph_flat_offsets(graph), (size_t)num_vertex + 1));
        GVK_TRY(to_device((uint32_t **)&g.edges_uv, uv, 2 * D));
        GVK_TRY(to_device((gvk_alias_entry **)&g.edge_table, edge_table.data(), D));
        GVK_TRY(to_device((gvk_alias_entry **)&g.neighbor_table, neighbor_table.data(), D));
        if (biased) GVK_TRY(to_device((uint32_t **)&g.sorted_neighbors, sorted_neighbors.data(), D));
        GVK_TRY(to_device((uint32_t **)&g.local, local.data(), num_vertex));
        GVK_TRY(to_device(&w.walk_part, part.data(), num_vertex));
        // one worker: the walks land in the block pools themselves; several: in this worker's slice of every block
        std::vector<uint64_t> offsets((size_t)P * P);
        const size_t n_slice = n / W;
        for (int hp = 0; hp < P; hp++)
            for (int tp = 0; tp < P; tp++) {
                const size_t ti = std::find(w.tails.begin(), w.tails.end(), tp) - w.tails.begin();
                offsets[(size_t)hp * P + tp] = W == 1 ? (ti * P + hp) * n : ((size_t)hp * P + tp) * n_slice;
            }
        GVK_TRY(to_device(&w.walk_offsets, offsets.data(), offsets.size()));
        if (W > 1) HIP_TRY(hipMalloc(&w.slices, (size_t)P * P * n_slice * 8));
        HIP_TRY(hipMalloc(&w.walk_counters, (size_t)P * P * largest_divisor(n_slice, 256) * 4));
    }
    return GVK_OK;
}

// The pools of one episode (set 0 / 1) for every worker, drawn by the GPUs on their sampling streams — what fill() does
// with CPU threads.  Edge mode: one gvk_sample_edges per block, nothing to wait for.  Walk modes: rounds of
// gvk_sample_walks_blocks until every stripe of every block is full (the host reads the counters between rounds —
// a handful of round trips, on a thread of its own while the GPUs train the episode before), then each slice goes to
// the worker that trains its block.
int gvx_solver::device_fill(int set) {
    Range range("Sample (device)");
    const int P = num_partition, W = num_worker;
    const size_t n = (size_t)episode_size * batch_size, n_slice = n / W;
    for (Worker &w : workers) {  // the pools of `set` were read by the episode before the one that trains now
        HIP_TRY(hipSetDevice(w.device));
        for (Worker &u : workers)
            if (u.episode_end_valid) HIP_TRY(hipStreamWaitEvent(w.sample, u.episode_end, 0));
    }
    if (mode == GVS_MODE_EDGE) {
        for (Worker &w : workers) {
            HIP_TRY(hipSetDevice(w.device));
            for (size_t i = 0; i < w.edge_blocks.size(); i++) {
                const Worker::EdgeBlock &b = w.edge_blocks[i];
                GVK_TRY(gvk_sample_edges(w.sample, b.table, b.count, w.sample_seed, w.sample_index,
                                         w.block_pools[set] + i * n * 2, n));
                w.sample_index += n;
            }
            HIP_TRY(hipEventRecord(w.filled[set], w.sample));
        }
        return GVK_OK;
    }
    const int L = config.random_walk_length, aug = config.augmentation_step;
    const uint64_t per_walk = (uint64_t)aug * L - (uint64_t)aug * (aug - 1) / 2;
    if (P == 1) {  // one block: every walk owns its slots of the pool, no binning, no rounds
        Worker &w = workers[0];
        GVK_TRY(gvk_sample_walks(w.sample, &w.walk, w.sample_seed, w.sample_index, w.block_pools[set], n, L, aug,
                                 config.shuffle_base));
        w.sample_index += (n + per_walk - 1) / per_walk;
        HIP_TRY(hipEventRecord(w.filled[set], w.sample));
        return GVK_OK;
    }
    const int stripes = largest_divisor(n_slice, 256);
    const uint64_t stripe_capacity = n_slice / stripes, every = 64ull * stripes;  // the same number of wavefronts per stripe
    const size_t num_counter = (size_t)P * P * stripes;
    std::vector<uint64_t> walks(W, ((n_slice * P * P + per_walk - 1) / per_walk / every + 1) * every), used(W, 0);
    std::vector<char> full(W, 0);
    std::vector<uint32_t> counters(num_counter);
    for (Worker &w : workers) {
        HIP_TRY(hipSetDevice(w.device));
        HIP_TRY(hipMemsetAsync(w.walk_counters, 0, num_counter * 4, w.sample));
    }
    for (int round = 0;; round++) {
        if (round == 64) return gvk_fail(GVK_EINVAL, "device sampling: the block pools are not full after 64 rounds of walks");
        for (int r = 0; r < W; r++) {
            Worker &w = workers[r];
            if (full[r]) continue;
            HIP_TRY(hipSetDevice(w.device));
            GVK_TRY(gvk_sample_walks_blocks(w.sample, &w.walk, w.walk_part, P, w.sample_seed, w.sample_index + used[r], walks[r],
                                            W == 1 ? w.block_pools[set] : w.slices, w.walk_offsets, w.walk_counters,
                                            (uint32_t)n_slice, stripes, L, aug, config.shuffle_base));
            used[r] += walks[r];
        }
        bool all_full = true;
        for (int r = 0; r < W; r++) {
            Worker &w = workers[r];
            if (full[r]) continue;
            HIP_TRY(hipSetDevice(w.device));
            HIP_TRY(hipMemcpyAsync(counters.data(), w.walk_counters, num_counter * 4, hipMemcpyDeviceToHost, w.sample));
            HIP_TRY(hipStreamSynchronize(w.sample));
            // next round from the share of all pairs each stripe still short has received so far
            double need = 0;
            for (size_t i = 0; i < num_counter; i++) {
                if (counters[i] >= stripe_capacity) continue;
                const double share = std::max((double)counters[i] / ((double)used[r] * per_walk), 1.0 / (64.0 * num_counter));
                if (counters[i] == 0 && used[r] * per_walk > 64ull * n_slice * P * P)
                    return gvk_fail(GVK_EINVAL, "block (%zu, %zu) of the partition grid receives no random-walk pairs; use "
                                    "fewer partitions", i / stripes / P, i / stripes % P);
                need = std::max(need, (double)(stripe_capacity - counters[i]) / share / per_walk);
            }
            full[r] = need == 0;
            walks[r] = ((uint64_t)(need * 1.1) / every + 1) * every;
            all_full = all_full && full[r];
        }
        if (all_full) break;
    }
    for (int r = 0; r < W; r++) {
        Worker &w = workers[r];
        w.sample_index += used[r];
        HIP_TRY(hipSetDevice(w.device));
        for (int b = 0; b < P * P && W > 1; b++) {  // slice r of block b to the worker that trains b
            const int hp = b / P, tp = b % P;
            for (Worker &u : workers) {
                const size_t ti = std::find(u.tails.begin(), u.tails.end(), tp) - u.tails.begin();
                if (ti == u.tails.size()) continue;
                uint32_t *to = u.block_pools[set] + ((ti * P + hp) * n + (size_t)r * n_slice) * 2;
                const uint32_t *from = w.slices + (size_t)b * n_slice * 2;
                hipError_t e = u.device == w.device
                                   ? hipMemcpyAsync(to, from, n_slice * 8, hipMemcpyDeviceToDevice, w.sample)
                                   : hipMemcpyPeerAsync(to, u.device, from, w.device, n_slice * 8, w.sample);
                if (e != hipSuccess) return gvk_fail(GVK_EHIP, "routing of block (%d, %d): %s", hp, tp, hipGetErrorString(e));
            }
        }
        HIP_TRY(hipEventRecord(w.filled[set], w.sample));
    }
    return GVK_OK;
}

// One [S][dim] table of one partition between the host array (global vertex order) and a device table, through a
// pinned staging buffer in chunks of at most 256 MiB; the row permutation is done by host threads.
int gvx_solver::move_table(bool to_device, Worker &w, float *device_table, std::vector<float> &host, int partition) {
    const std::vector<uint32_t> &ids = part_ids[partition];
    const size_t row_bytes = (size_t)dim * 4, chunk_rows = std::max<size_t>(kChunkBytes / row_bytes, 1);
    float *staging = nullptr;
    HIP_TRY(hipHostMalloc(&staging, std::min(chunk_rows, std::max<size_t>(ids.size(), 1)) * row_bytes, hipHostMallocDefault));
    const int threads = std::max(std::min(num_sampler, 16), 1);
    int rc = GVK_OK;
    for (size_t start = 0; start < ids.size() && rc == GVK_OK; start += chunk_rows) {
        const size_t n = std::min(chunk_rows, ids.size() - start);
        auto permute = [&](int t) {
            for (size_t i = start + (size_t)t * n / threads; i < start + (size_t)(t + 1) * n / threads; i++) {
                float *a = host.data() + (size_t)ids[i] * dim, *b = staging + (i - start) * dim;
                memcpy(to_device ? b : a, to_device ? a : b, row_bytes);
            }
        };
        auto run = [&]() {
            std::vector<std::thread> pool;
            for (int t = 1; t < threads; t++) pool.emplace_back(permute, t);
            permute(0);
            for (auto &th : pool) th.join();
        };
        if (to_device) run();
        hipError_t e = to_device ? hipMemcpy(device_table + start * dim, staging, n * row_bytes, hipMemcpyHostToDevice)
                                 : hipMemcpy(staging, device_table + start * dim, n * row_bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = gvk_fail(GVK_EHIP, "table copy: %s", hipGetErrorString(e));
        if (!to_device && rc == GVK_OK) run();
    }
    hipHostFree(staging);
    (void)w;
    return rc;
}

int gvx_solver::upload() {
    if (streamed) {  // tables stay in host memory; moments start there, too
        if (num_moment && !(config.resume && (int)vertex_moments.size() == num_moment)) {
            vertex_moments.assign(num_moment, std::vector<float>((size_t)num_vertex * dim, 0.0f));
            context_moments.assign(num_moment, std::vector<float>((size_t)num_vertex * dim, 0.0f));
        }
        for (Worker &w : workers) {
            HIP_TRY(hipSetDevice(w.device));
            HIP_TRY(hipStreamSynchronize(w.compute));
        }
        return GVK_OK;
    }
    const bool moments = config.resume && (int)vertex_moments.size() == num_moment && num_moment > 0;
    for (Worker &w : workers) {
        HIP_TRY(hipSetDevice(w.device));
        HIP_TRY(hipStreamSynchronize(w.compute));  // the memsets of prepare_devices
        for (int p = 0; p < num_partition; p++) {
            GVK_TRY(move_table(true, w, head_table(w, p, 0), vertex, p));
            for (int j = 0; moments && j < num_moment; j++) GVK_TRY(move_table(true, w, head_table(w, p, 1 + j), vertex_moments[j], p));
        }
        for (size_t ti = 0; ti < w.tails.size(); ti++) {
            GVK_TRY(move_table(true, w, context_table(w, (int)ti, 0), context, w.tails[ti]));
            for (int j = 0; moments && j < num_moment; j++)
                GVK_TRY(move_table(true, w, context_table(w, (int)ti, 1 + j), context_moments[j], w.tails[ti]));
        }
    }
    return GVK_OK;
}

int gvx_solver::write_back() {  // WorkerMixin::write_back, solver.h:1498-1504: every table once, from a worker that holds it
    for (Worker &w : workers) {
        HIP_TRY(hipSetDevice(w.device));
        HIP_TRY(hipDeviceSynchronize());
    }
    if (streamed) return GVK_OK;  // every block was stored when it was left
    if (num_moment) {
        vertex_moments.assign(num_moment, std::vector<float>((size_t)num_vertex * dim, 0.0f));
        context_moments.assign(num_moment, std::vector<float>((size_t)num_vertex * dim, 0.0f));
    }
    Worker &first = workers[0];
    HIP_TRY(hipSetDevice(first.device));
    for (int p = 0; p < num_partition; p++) {
        GVK_TRY(move_table(false, first, head_table(first, p, 0), vertex, p));
        for (int j = 0; j < num_moment; j++) GVK_TRY(move_table(false, first, head_table(first, p, 1 + j), vertex_moments[j], p));
    }
    for (Worker &w : workers) {
        HIP_TRY(hipSetDevice(w.device));
        for (size_t ti = 0; ti < w.tails.size(); ti++) {
            GVK_TRY(move_table(false, w, context_table(w, (int)ti, 0), context, w.tails[ti]));
            for (int j = 0; j < num_moment; j++)
                GVK_TRY(move_table(false, w, context_table(w, (int)ti, 1 + j), context_moments[j], w.tails[ti]));
        }
    }
    return GVK_OK;
}

// Streamed mode (WorkerMixin::load_partition / write_back, solver.h:1435-1504): the head and tail partition of a block come
// from the host tables before it trains and go back after.
int gvx_solver::load_block(Worker &w, int hp, int tp) {
    Range range("Load partition");
    HIP_TRY(hipSetDevice(w.device));
    GVK_TRY(move_table(true, w, head_table(w, hp, 0), vertex, hp));
    GVK_TRY(move_table(true, w, context_table(w, tp, 0), context, tp));
    for (int j = 0; j < num_moment; j++) {
        GVK_TRY(move_table(true, w, head_table(w, hp, 1 + j), vertex_moments[j], hp));
        GVK_TRY(move_table(true, w, context_table(w, tp, 1 + j), context_moments[j], tp));
    }
    return GVK_OK;
}

int gvx_solver::store_block(Worker &w, int hp, int tp) {
    Range range("Write back partition");
    HIP_TRY(hipSetDevice(w.device));
    HIP_TRY(hipStreamSynchronize(w.compute));
    GVK_TRY(move_table(false, w, head_table(w, hp, 0), vertex, hp));
    GVK_TRY(move_table(false, w, context_table(w, tp, 0), context, tp));
    for (int j = 0; j < num_moment; j++) {
        GVK_TRY(move_table(false, w, head_table(w, hp, 1 + j), vertex_moments[j], hp));
        GVK_TRY(move_table(false, w, context_table(w, tp, 1 + j), context_moments[j], tp));
    }
    return GVK_OK;
}

// ---- episode loop ---------------------------------------------------------------------------------------------------

int gvx_solver::fill(std::vector<uint32_t *> &pools) {
    Range range("Sample threads");  // solver.h:622
    gvs_fill_config f{};
    f.mode = mode;
    f.num_thread = 4 * num_sampler;  // 4 slices per OS thread: a descheduled thread delays a quarter-size slice
    f.sample_batch_size = config.random_walk_length * config.random_walk_batch_size;  // graph.cuh:791
    f.walk_length = config.random_walk_length, f.walk_batch = config.random_walk_batch_size;
    f.augmentation_step = config.augmentation_step, f.shuffle_base = config.shuffle_base;
    f.tail_partition = -1, f.os_threads = num_sampler, f.cpu_offset = -1;
    return gvs_sampler_fill(sampler, pools.data(), (uint64_t)episode_size * batch_size, &f);
}

// WorkerMixin::train (solver.h:1511-1522): positive_reuse x episode_size batches of one block on the worker's compute
// stream; batch ids interleave over the workers as the reference's shared atomic counter hands them out (solver.h:1520)
int gvx_solver::train_block(Worker &w, int hp, int tp, uint32_t *pool) {
    Range range("Train Batch");  // solver.h:1526 (one range per block: its batches are back-to-back launches)
    const int W = num_worker, r = (int)(&w - workers.data()), B = batch_size, nm = num_moment;
    const int ti = (int)(std::find(w.tails.begin(), w.tails.end(), tp) - w.tails.begin());
    gvk_tables t{};
    t.vertex = head_table(w, hp, 0), t.context = context_table(w, ti, 0);
    if (nm >= 1) t.vertex_moment1 = head_table(w, hp, 1), t.context_moment1 = context_table(w, ti, 1);
    if (nm >= 2) t.vertex_moment2 = head_table(w, hp, 2), t.context_moment2 = context_table(w, ti, 2);
    t.n_vertex = t.n_context = part_rows;
    gvk_negative_source neg{};
    neg.table = w.negative_tables[ti], neg.count = (uint32_t)part_ids[tp].size();
    neg.classes = w.negative_classes[ti], neg.class_count = w.negative_class_counts[ti];
    neg.seed = 0x100000001B3ull * 1 + (uint64_t)r;
    gvk_optimizer o{};
    o.type = optimizer.type, o.lr = optimizer.lr, o.weight_decay = optimizer.weight_decay;
    o.hp0 = optimizer.hp0, o.hp1 = optimizer.hp1, o.epsilon = optimizer.epsilon;
    for (int reuse = 0; reuse < config.positive_reuse; reuse++) {
        int done = 0;
        while (done < episode_size) {
            const uint64_t first = batch_id + ((uint64_t)reuse * episode_size + done) * W + r;
            if (first % config.log_frequency == 0) {  // solver.h:1527-1549 (the loss is the previous batch's)
                std::vector<float> host_loss(B);
                HIP_TRY(hipMemcpyAsync(host_loss.data(), w.loss, (size_t)B * 4, hipMemcpyDeviceToHost, w.compute));
                HIP_TRY(hipStreamSynchronize(w.compute));
                double sum = 0;
                for (float x : host_loss) sum += x;
                log_message(0, "Batch id: %llu / %llu", (unsigned long long)first, (unsigned long long)num_batch);
                log_message(0, "loss = %g", sum / B);
            }
            int n = 1;  // up to, not including, this worker's next logging batch
            while (n < episode_size - done && (first + (uint64_t)n * W) % config.log_frequency) n++;
            if (optimizer.schedule != 2) {
                GVK_TRY(gvk_train_episode(w.compute, dim, &o, optimizer.schedule == 1, &t, pool + (size_t)done * B * 2, &neg,
                                          (uint32_t)first, (uint32_t)W, (uint32_t)num_batch, n, w.loss, B, num_negative,
                                          config.negative_weight));
            } else {  // custom schedule: lr computed on the host per batch (optimizer.h:132-134)
                for (int b = 0; b < n; b++) {
                    const uint64_t id = first + (uint64_t)b * W;
                    gvk_optimizer ob = o;
                    ob.lr = optimizer.lr * optimizer.schedule_function((int)id, (int)num_batch, optimizer.user);
                    GVK_TRY(gvk_train(w.compute, dim, &ob, &t, pool + (size_t)(done + b) * B * 2, &neg, (uint32_t)id, w.loss, B,
                                      num_negative, config.negative_weight));
                }
            }
            done += n;
        }
    }
    return GVK_OK;
}

int gvx_solver::episode_loop() {
    const int P = num_partition, W = num_worker;
    const size_t pool_elems = (size_t)episode_size * batch_size * 2;
    // pinned host pools, two sets (the samplers fill one while the GPUs read the other), one pool per block;
    // with device sampling the two sets are Worker::block_pools, in HBM
    std::vector<uint32_t *> sets[2];
    auto free_sets = [&]() {
        for (auto &set : sets)
            for (uint32_t *p : set) hipHostFree(p);
    };
    for (auto &set : sets) {
        if (device_sampling) break;
        set.assign((size_t)P * P, nullptr);
        for (auto &p : set)
            if (hipHostMalloc(&p, pool_elems * 4, hipHostMallocDefault) != hipSuccess) {
                free_sets();
                return gvk_fail(GVK_ENOMEM, "Out of host memory for the sample pools (%d x %d blocks of %s)", P, P,
                                size_string((double)pool_elems * 4).c_str());
            }
    }
    // regroup by table size, at dim >= 64: cache-resident tables (< 16 MiB) always — adjacent
    // same-head samples are then trained as runs, which keeps training close to sequential (§7); shard-sized tables
    // (< 256 MiB) for independent edge draws — a shared head row becomes one fetch; larger tables keep the sampler's order
    const size_t table_bytes = (size_t)part_rows * dim * 4;
    const bool grouped = dim >= 64 && (table_bytes < ((size_t)16 << 20) || (table_bytes < ((size_t)256 << 20) && mode == GVS_MODE_EDGE));
    const int row_bits = std::max(32 - __builtin_clz(std::max(part_rows, 2u) - 1), 1);
    const int parts = gvk_train_launches(batch_size, part_rows);
    const uint64_t per_episode = (uint64_t)num_step * episode_size * config.positive_reuse * W;
    auto produce = [&](int set) { return device_sampling ? device_fill(set) : fill(sets[set]); };
    int rc = produce(0);
    int current = 0;
    while (rc == GVK_OK && batch_id < num_batch) {
        // the samplers may overwrite the other set once every copy out of it has landed
        for (Worker &w : workers) {
            for (hipEvent_t e : w.copied) {
                hipEventSynchronize(e);
                hipEventDestroy(e);
            }
            w.copied.clear();
        }
        int fill_rc = GVK_OK;
        std::thread filler;
        if (batch_id + per_episode < num_batch)  // no pools for an episode that will not run
            filler = std::thread([&, current]() { fill_rc = produce(current ^ 1); });
        for (Worker &w : workers)  // device sampling: the pools of this episode, slices from every worker included
            for (Worker &u : workers) {
                if (!device_sampling) break;
                hipSetDevice(w.device);
                hipStreamWaitEvent(w.copy, u.filled[current], 0);
                hipStreamWaitEvent(w.compute, u.filled[current], 0);
            }
        auto stage = [&](Worker &w, int step) -> int {  // H2D copy (+ regrouping) of the worker's block of `step`
            Range upload_range("Upload");
            const int r = (int)(&w - workers.data());
            const int hp = schedule[((size_t)step * W + r) * 2], tp = schedule[((size_t)step * W + r) * 2 + 1];
            const int b = (int)((w.visits + (uint64_t)step) & 1);
            HIP_TRY(hipSetDevice(w.device));
            if (w.released_valid[b]) HIP_TRY(hipStreamWaitEvent(w.copy, w.released[b], 0));
            const uint32_t *source = w.landing;
            if (device_sampling) {  // already in HBM: trained in place unless it is regrouped
                source = block_pool(w, current, hp, tp);
            } else {
                uint32_t *target = grouped ? w.landing : w.pool[b];
                HIP_TRY(hipMemcpyAsync(target, sets[current][(size_t)hp * P + tp], pool_elems * 4, hipMemcpyHostToDevice, w.copy));
                hipEvent_t copied;
                HIP_TRY(hipEventCreateWithFlags(&copied, hipEventDisableTiming));
                HIP_TRY(hipEventRecord(copied, w.copy));
                w.copied.push_back(copied);
            }
            Range regroup("Regroup");
            if (grouped)  // per part of a batch: a part is what one launch trains (gvk_train_launches)
                GVK_TRY(gvk_group_pairs(w.copy, source, w.pool[b], w.group_workspace, &w.group_workspace_bytes,
                                        batch_size / parts, episode_size * parts, row_bits));
            HIP_TRY(hipEventRecord(w.uploaded[b], w.copy));
            return GVK_OK;
        };
        for (Worker &w : workers)
            if ((rc = stage(w, 0)) != GVK_OK) break;
        for (int step = 0; step < num_step && rc == GVK_OK; step++) {
            for (int r = 0; r < W && rc == GVK_OK; r++) {
                Worker &w = workers[r];
                const int hp = schedule[((size_t)step * W + r) * 2], tp = schedule[((size_t)step * W + r) * 2 + 1];
                const int b = (int)((w.visits + (uint64_t)step) & 1);
                if (hipSetDevice(w.device) != hipSuccess) rc = gvk_fail(GVK_EHIP, "hipSetDevice");
                hipStreamWaitEvent(w.compute, w.uploaded[b], 0);
                for (hipEvent_t e : w.incoming) {  // head shards other workers trained in earlier steps
                    hipStreamWaitEvent(w.compute, e, 0);
                    hipEventDestroy(e);
                }
                w.incoming.clear();
                if (w.sending == hp) hipStreamWaitEvent(w.compute, w.sent, 0);  // its copies to the peers still read this slot
                if (rc == GVK_OK && step + 1 < num_step) rc = stage(w, step + 1);  // next block's pool while this one trains
                if (rc == GVK_OK && streamed) rc = load_block(w, hp, tp);
                if (rc == GVK_OK)
                    rc = train_block(w, hp, tp, device_sampling && !grouped ? block_pool(w, current, hp, tp) : w.pool[b]);
                hipEventRecord(w.released[b], w.compute);
                w.released_valid[b] = true;
                hipEventRecord(w.trained, w.compute);
            }
            // streamed: the blocks of this step go back to the host tables (distinct head and tail partitions per worker)
            for (int r = 0; r < W && rc == GVK_OK && streamed; r++)
                rc = store_block(workers[r], schedule[((size_t)step * W + r) * 2], schedule[((size_t)step * W + r) * 2 + 1]);
            // exchange: the head shard a worker just trained goes straight into every other worker's replica
            Range exchange_range("Exchange");
            for (int r = 0; r < W && rc == GVK_OK && W > 1 && !streamed; r++) {
                Worker &w = workers[r];
                const int hp = schedule[((size_t)step * W + r) * 2];
                hipSetDevice(w.device);
                hipStreamWaitEvent(w.exchange, w.trained, 0);
                for (int q = 0; q < W; q++) {
                    if (q == r) continue;
                    Worker &u = workers[q];
                    hipError_t e = u.device == w.device  // two workers sharing a GPU (tests): a plain device copy
                                       ? hipMemcpyAsync(head_table(u, hp, 0), head_table(w, hp, 0), slot_floats() * 4,
                                                        hipMemcpyDeviceToDevice, w.exchange)
                                       : hipMemcpyPeerAsync(head_table(u, hp, 0), u.device, head_table(w, hp, 0), w.device,
                                                            slot_floats() * 4, w.exchange);
                    if (e != hipSuccess) rc = gvk_fail(GVK_EHIP, "exchange of head partition %d: %s", hp, hipGetErrorString(e));
                    hipEvent_t arrived;
                    hipEventCreateWithFlags(&arrived, hipEventDisableTiming);
                    hipEventRecord(arrived, w.exchange);
                    u.incoming.push_back(arrived);
                }
                hipEventRecord(w.sent, w.exchange);
                w.sending = hp;
            }
            batch_id += (uint64_t)episode_size * config.positive_reuse * W;
        }
        for (Worker &w : workers) w.visits += (uint64_t)num_step;
        {
            Range wait("Wait for sample threads");  // solver.h:645
            if (filler.joinable()) filler.join();
        }
        if (rc == GVK_OK) rc = fill_rc;
        for (Worker &w : workers) {  // device sampling: the pools of this episode may be redrawn once it has trained
            if (!device_sampling) break;
            hipSetDevice(w.device);
            hipEventRecord(w.episode_end, w.compute);
            w.episode_end_valid = true;
        }
        current ^= 1;
    }
    for (Worker &w : workers) {
        hipSetDevice(w.device);
        hipDeviceSynchronize();
        for (hipEvent_t e : w.copied) hipEventDestroy(e);
        w.copied.clear();
    }
    free_sets();
    return rc;
}

extern "C" int gvx_solver_train(gvx_solver *s, const gvx_train_config *config) {
    if (!s || !config) return gvk_fail(GVK_EINVAL, "gvx_solver_train: null solver / config");
    GVK_TRY(s->configure(*config));
    GVK_TRY(s->prepare_devices());
    GVK_TRY(s->upload());
    const auto t0 = std::chrono::steady_clock::now();
    const uint64_t first = s->batch_id;
    int rc = s->episode_loop();
    s->train_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc == GVK_OK) {
        log_message(0, "[time] %llu batches in %.2f s (%.1f M edge-samples/s)", (unsigned long long)(s->batch_id - first),
                    s->train_seconds, (double)(s->batch_id - first) * s->batch_size / std::max(s->train_seconds, 1e-9) / 1e6);
        rc = s->write_back();
    }
    s->release_device();
    return rc;
}

extern "C" int gvx_solver_predict(gvx_solver *s, const int64_t *samples, size_t n, float *logits) {
    if (!s || !s->graph) return gvk_fail(GVK_EINVAL, "The model must be built on a graph first");
    if (n == 0) return GVK_OK;
    if (!samples || !logits) return gvk_fail(GVK_EINVAL, "gvx_solver_predict: null pointer");
    std::vector<uint32_t> records(2 * n);
    for (size_t i = 0; i < n; i++) {
        const int64_t v = samples[2 * i], c = samples[2 * i + 1];
        if (v < 0 || c < 0 || v >= (int64_t)s->num_vertex || c >= (int64_t)s->num_vertex)
            return gvk_fail(GVK_EINVAL, "node index out of range");
        records[2 * i] = (uint32_t)c, records[2 * i + 1] = (uint32_t)v;  // records are {tail, head} (solver.h:1127-1132)
    }
    HIP_TRY(hipSetDevice(s->device_ids[0]));
    float *dv = nullptr, *dc = nullptr, *dl = nullptr;
    uint32_t *dp = nullptr;
    const size_t table = (size_t)s->num_vertex * s->dim * 4, B = (size_t)std::max(s->batch_size, 1);
    auto done = [&](int rc) {
        hipFree(dv), hipFree(dc), hipFree(dl), hipFree(dp);
        return rc;
    };
    if (hipMalloc(&dv, table) != hipSuccess || hipMalloc(&dc, table) != hipSuccess || hipMalloc(&dl, n * 4) != hipSuccess ||
        hipMalloc(&dp, n * 8) != hipSuccess)
        return done(gvk_fail(GVK_ENOMEM, "predict: out of GPU memory"));
    hipMemcpy(dv, s->vertex.data(), table, hipMemcpyHostToDevice);
    hipMemcpy(dc, s->context.data(), table, hipMemcpyHostToDevice);
    hipMemcpy(dp, records.data(), n * 8, hipMemcpyHostToDevice);
    for (size_t start = 0; start < n; start += B) {
        const int rc = gvk_predict(nullptr, s->dim, dv, dc, dp + 2 * start, dl + start, (int)std::min(B, n - start));
        if (rc != GVK_OK) return done(rc);
    }
    if (hipMemcpy(logits, dl, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return done(gvk_fail(GVK_EHIP, "predict: copy back failed"));
    return done(GVK_OK);
}

extern "C" int gvx_solver_clear(gvx_solver *s) {
    if (!s) return gvk_fail(GVK_EINVAL, "gvx_solver_clear: null solver");
    s->release();
    s->vertex_moments.clear(), s->context_moments.clear();
    return GVK_OK;
}

extern "C" float *gvx_solver_embeddings(gvx_solver *s, int which, uint64_t *num_vertex) {
    if (!s) return nullptr;
    if (num_vertex) *num_vertex = s->num_vertex;
    return which == 0 ? s->vertex.data() : s->context.data();
}

extern "C" int gvx_solver_get(gvx_solver *s, gvx_solver_members *out) {
    if (!s || !out) return gvk_fail(GVK_EINVAL, "gvx_solver_get: null pointer");
    memset(out, 0, sizeof(*out));
    out->dim = s->dim, out->num_partition = s->num_partition, out->num_negative = s->num_negative;
    out->num_epoch = s->config.num_epoch, out->resume = s->config.resume, out->episode_size = s->episode_size;
    out->batch_size = s->batch_size, out->augmentation_step = s->config.augmentation_step;
    out->random_walk_length = s->config.random_walk_length, out->random_walk_batch_size = s->config.random_walk_batch_size;
    out->shuffle_base = s->config.shuffle_base, out->positive_reuse = s->config.positive_reuse;
    out->log_frequency = s->config.log_frequency, out->num_worker = s->num_worker, out->num_sampler = s->num_sampler;
    out->negative_sample_exponent = s->config.negative_sample_exponent, out->negative_weight = s->config.negative_weight;
    out->p = s->config.p, out->q = s->config.q;
    out->gpu_memory_limit = s->gpu_memory_limit, out->gpu_memory_cost = s->gpu_memory_cost;
    out->model = s->model.c_str();
    out->optimizer = s->optimizer;
    out->batch_id = s->batch_id, out->num_batch = s->num_batch, out->train_seconds = s->train_seconds;
    return GVK_OK;
}

extern "C" size_t gvx_solver_info(gvx_solver *s, char *buf, size_t capacity) {
    if (!s) return 0;
    s->make_info();
    if (buf && capacity) snprintf(buf, capacity, "%s", s->info_text.c_str());
    return s->info_text.size();
}

extern "C" int gvx_solver_save_embeddings(gvx_solver *s, const char *file_name) {
    if (!s || !s->graph || !file_name) return gvk_fail(GVK_EINVAL, "The model must be built on a graph first");
    FILE *f = fopen(file_name, "wb");
    if (!f) return gvk_fail(GVK_EINVAL, "Can't open file `%s`", file_name);
    fprintf(f, "%u %d\n", s->num_vertex, s->dim);
    std::vector<char> name(256);
    for (uint32_t v = 0; v < s->num_vertex; v++) {
        int64_t n = gvs_graph_id2name(s->graph, v, name.data(), name.size());
        if (n >= (int64_t)name.size()) {
            name.resize(n + 1);
            gvs_graph_id2name(s->graph, v, name.data(), name.size());
        }
        fprintf(f, "%s ", name.data());
        fwrite(s->vertex.data() + (size_t)v * s->dim, 4, s->dim, f);
        fputc('\n', f);
    }
    fclose(f);
    return GVK_OK;
}

extern "C" void gvx_set_logging(int threshold, void (*sink)(int, const char *, void *), void *user) {
    std::lock_guard<std::mutex> lock(g_log_mutex);
    g_log_threshold = threshold, g_log_sink = sink, g_log_user = user;
}
