// gvx_engine.cpp — the native solver engine behind include/gvx.h: the orchestration of the reference's GraphSolver /
// SolverMixin / WorkerMixin (include/instance/graph.cuh:586-813, include/core/solver.h:87-888,1170-1623) as a C++ host
// runtime for MI355X, and the ONLY orchestrator of this repository: the pybind11 module (bind/libgraphvite.cpp), the
// Python package (graphvite_amd/solver.py) and bench.py are bindings of it.  Not a translation of the reference's:
//   * one HOST thread issues everything; a worker is a set of HIP streams on its GPU, not an OS thread.  Every block is
//     one H2D copy + (optionally) one regrouping pass + episode_size back-to-back kernel launches, all asynchronous, so
//     the host is never the bottleneck and the reference's thread-per-worker joins per schedule step disappear;
//   * the W workers of a job live in one process (device_ids = [0, 1, ...], like the reference) or one per process
//     (gvx_solver_create_distributed: torchrun / mpirun); the code below only ever iterates over the LOCAL workers;
//   * a worker keeps the WHOLE vertex table — a slab [P slots][1 + m][S][dim] — and the context shards of the tail
//     partitions it owns for good (288 GB of HBM per GPU); nothing is evicted, reloaded or rebuilt between schedule steps
//     (WorkerMixin::load_partition / write_back, solver.h:1435-1504, only exist as the fallback for models that do not fit);
//   * after a schedule step ONE in-place all-gather of a head group's slab (RCCL over xGMI; gvx_comm.h) hands every worker
//     the shards the others trained; consumers wait on events, the host does not;
//   * CPU sampler threads (gvs_sampler_fill) fill the next episode's pinned pools while the GPUs train this one — or the
//     GPUs draw the positives themselves (GVX_DEVICE_SAMPLING).
// The arithmetic lives in the kernels (gvk.h); this file moves no embedding through the CPU during training.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
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
#include "gvx_comm.h"

// Stream / event plumbing and clean-up calls are not checked one by one: a failure there is sticky in the HIP runtime
// and surfaces at the next checked call (every allocation, copy and launch is checked).
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"

namespace {

constexpr int kMaxPartition = 16;  // solver.h:51-57
constexpr int kMinBatchSize = 10000;
constexpr int kSamplePerVertex = 175;
constexpr double kHubHits = 1.0;           // GVX_HUB_ROWS -1: a row a BATCH is expected to hit this often is a hub row (§7.10)
constexpr double kHubHitsPerPart = 0.125;  // ... and no row outside the chains is to be hit more often than this per PART of a batch: the parts follow (hub_parts_of)
constexpr int kHubListSlice = 0;            // GVX_LIST_SLICE: units of work lists per launch when the lists are built ahead (build_lists); 0: one launch (measured: slices are slower)
constexpr double kHubRoundShare = 0.02;    // GVX_HUB_ROUNDS -1: long chains work in rounds where the graph's largest vertex takes more than this share of its total degree (§7.11)
constexpr uint64_t kMaxHubRows = 16384;  // per table (gvk_hot_build counts the chains of both tables in LDS)
constexpr int kHubChunk = 128;          // batches whose work lists are built at once
constexpr int kHubEntriesPerPart = 250;  // with hub rows by chains a batch is trained as so many parts that its largest hub row meets about this many of its updates per part
constexpr int kHubMaxEntriesPerPart = 1000;  // ... and past this many per part chains are not used (configure): the entries of a part are worked side by side from its start state
constexpr int kHubMaxParts = 32;  // measured on the headline shape at P = 8 (a block's top hub holds 16 % of its samples: the rule asks for 64): 25 / 32 / 40 / 50
                                  // parts end -0.0013 / +0.0008 / +0.0013 / +0.0022 from the reference's loop — past 32 the parts only cost launches
constexpr int kHubMaxPartsResident = 50;  // cache-resident tables (< 16 MiB): a small partition's chains are feasible up to this many parts (§7.8)
constexpr double kWalkHitsPerPart = 0.022;  // walk-ordered pools: parts so that the rows outside the chains meet no more than this (hit-weighted mean of expected hits) per part
constexpr int kHubGroup = 2;             // GVX_HUB_GROUP 0: under executor 2 the chains of so many consecutive parts share a launch
constexpr int kHubLerp = 0;              // GVX_HUB_LERP -1: the pairs read hub rows as their part's chains left them
constexpr int kHubExecutor = 0;          // GVX_HUB_EXECUTOR -1: one launch per unit carries its pairs and the next unit's chains (gvk_train_episode_hot); 1: the chains as a stream of their own, a batch ahead of the pairs (gvk_train_episode_ahead: measured slower)
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
    int rank = 0;    // among the W workers of the job (the schedule's worker index)
    int device = 0;
    hipStream_t compute = nullptr, copy = nullptr, exchange = nullptr;
    float *head = nullptr;     // the slab [P slots][1 + m][S][dim]: every head partition, vertex rows then moment tables
    float *context = nullptr;  // [T][1 + m][S][dim]: the tail partitions this worker owns
    std::vector<int> tails;
    std::vector<gvk_alias_entry *> negative_tables;  // per owned tail: one alias slot per row, or null when ...
    std::vector<gvk_class_entry *> negative_classes; // ... the sampler draws by weight class (gvk.h): an alias table over
    std::vector<uint32_t> negative_class_counts;     // the classes of equal-degree rows, a few thousand entries
    float *loss = nullptr;
    int32_t *agreement = nullptr;  // [#worker] staging of allocate_pools' episode-size agreement: made with the streams, before anything can fail
    uint32_t *pool[2] = {nullptr, nullptr}, *landing = nullptr;
    void *group_workspace = nullptr;
    size_t group_workspace_bytes = 0;
    // hub rows by chains: the work lists of hub_chunk batches (gvk_hot_build) in two workspaces — the lists of the NEXT chunk are built
    // on the `lists` stream while this one trains out of the other; `chains`: the chains' stream of the chain-stream executor
    void *hub_workspaces[2] = {nullptr, nullptr};
    size_t hub_workspace_bytes = 0;
    hipStream_t chains = nullptr, lists = nullptr;
    hipEvent_t lists_built[2] = {nullptr, nullptr}, lists_trained[2] = {nullptr, nullptr};
    bool lists_trained_valid[2] = {false, false};
    uint64_t chunks_built = 0, chunks_trained = 0;
    // The first chunk of the NEXT visit's lists is built while this visit's last chunk trains (prefetch_lists): `staged` is the visit stage()
    // announced last, `prefetched` says what was built from it — train_block takes it when every field is what it would build itself.
    struct Staged {
        bool valid = false;
        int hp = 0, tp = 0, set = 0, b = 0;
    } staged;
    struct Prefetched {
        bool valid = false;
        const uint32_t *pool = nullptr;
        const void *workspace = nullptr;
        uint64_t first = 0;
        int m = 0, parts = 0, cap = 0, set = 0, b = 0;
        uint32_t kv = 0, kc = 0;
    } prefetched;
    hipEvent_t uploaded[2] = {nullptr, nullptr}, released[2] = {nullptr, nullptr}, trained = nullptr;
    bool released_valid[2] = {false, false};
    std::vector<hipEvent_t> copied;     // H2D copies of the current pool set still reading pinned memory
    std::vector<hipEvent_t> gathered;   // per head group: the all-gather of its slab has completed on this worker
    std::vector<char> gathered_valid;
    uint64_t visits = 0;
    // pools resident in HBM: positives drawn on the device (GVX_DEVICE_SAMPLING) or a session's resident pool sets
    struct EdgeBlock {
        gvk_edge_entry *table = nullptr;  // alias table over the weights of the block's edges + their {tail, head} records
        uint32_t count = 0;
    };
    hipStream_t sample = nullptr;
    hipEvent_t filled[2] = {nullptr, nullptr}, episode_end = nullptr;
    bool episode_end_valid = false;
    uint32_t *block_pools[2] = {nullptr, nullptr};  // [tails][P][n] records: the pools of every block it trains, two sets
    uint32_t *route_send = nullptr, *route_recv = nullptr;  // walks, several workers: [W][T * P][n / W], its slice of EVERY block
    uint32_t *route_host = nullptr;                 // ... sampled by this process's CPU samplers (pinned)
    std::vector<EdgeBlock> edge_blocks;             // edge mode: per block, index = tail index * P + head partition
    gvk_walk_graph walk{};                          // walk modes: the graph in HBM
    int32_t *walk_part = nullptr;
    uint64_t *walk_offsets = nullptr;
    uint32_t *walk_counters = nullptr;
    float *walk_accept = nullptr;          // [P * P] the blocks' thinning rates of the next launch (device_fill)
    std::vector<char> walk_collects;       // [P * P] this worker collects the block's pairs (walk_offsets != ~0)
    uint64_t sample_seed = 0, sample_index = 0;
};

}  // namespace

struct gvx_solver {
    // resources
    int dim = 0;
    std::vector<int> device_ids;  // of the LOCAL workers
    int num_worker = 0;           // W: workers of the whole job
    int first_rank = 0;           // rank of the first local worker (0 in one process; the process's rank otherwise)
    bool distributed = false;     // one process per GPU
    std::vector<char> unique_id;
    bool has_transport = false;
    gvx_transport transport{};
    int num_sampler_per_worker = 0, num_sampler = 0;
    size_t memory_request = 0, gpu_memory_limit = 0, gpu_memory_cost = 0;
    uint64_t seed = 0;
    int pair_order_request = 0, negative_table_request = 0;
    int fidelity = -1;              // GVX_FIDELITY: -1 = the default rule (hub rows by chains where chains exist), 0 = throughput (no chains), 1 = chains or an error
    int hub_parts_request = 0;      // GVX_HUB_PARTS: 0 the rule (gvk_train_launches when every row is a hub row, else 1), Q > 0 given
    int64_t hub_rows_request = -2;  // GVX_HUB_ROWS: -2 the default rule, -1 by expected hits per batch, 0 off, N > 0 the first N rows
    bool hogwild_said = false, order_said = false, rules_said = false;  // messages of configure() that are given once per solver
    int hub_rounds_request = -1;    // GVX_HUB_ROUNDS: -1 the rule (kHubRoundEntries), 0 / 1: long chains in one round / in rounds
    int hub_lerp_request = -1;      // GVX_HUB_LERP: -1 the rule, 0 / 1: the pairs read hub rows as their unit's chains left them / along the chains' way
    int hub_executor_request = -1;  // GVX_HUB_EXECUTOR: -1 the rule (kHubExecutor; the fused launches where lerp is asked for), 0 / 1
    int hub_pair_launches_request = 0;  // GVX_HUB_PAIR_LAUNCHES: launches the pairs of a batch are trained as under the chain-stream executor (0: one per part)
    int hub_group_request = 0;      // GVX_HUB_GROUP: parts whose chains share a launch under executor 2 (0: kHubGroup)
    int hub_executor() const {  // 0: a launch per part (gvk_train_episode_hot), 1: the chain stream, 2: the chains of a group of parts per launch (both gvk_train_episode_ahead)
        const int lerp = hub_lerp_request < 0 ? kHubLerp : hub_lerp_request;
        const int wanted = hub_executor_request < 0 ? (lerp ? 0 : kHubExecutor) : hub_executor_request;
        return optimizer.schedule != 2 && optimizer.type == GVK_SGD ? wanted : 0;  // a callback's schedule trains batch by batch, a moment optimizer's chains are one task each: fused
    }
    bool hub_ahead() const { return hub_executor() != 0; }  // the work lists carry versions and slots (gvk_ahead_build)
    int hub_group_of(int parts) const {  // the largest divisor of the parts up to the group asked for
        if (hub_executor() != 2) return 1;
        int group = std::max(hub_group_request > 0 ? hub_group_request : kHubGroup, 1);
        while (group > 1 && parts % group) group--;
        return group;
    }
    int hub_chunk = kHubChunk;      // batches whose work lists are built at once (fewer where memory is short)
    int hub_max_parts = kHubMaxParts;  // most parts a batch is trained as (kHubMaxPartsResident for cache-resident tables)
    int hub_chain_cap_request = 0;  // GVX_HUB_CHAIN_CAP: entries one chain task trains in sequence (0 = the kernels' default)
    uint64_t node2vec_table_limit = (uint64_t)1 << 30;
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
    std::vector<uint64_t> sample_positions;  // per local worker: draws (edge mode) / walks consumed since build()
    std::vector<int32_t> part;
    std::vector<uint32_t> local, part_sizes;
    uint32_t part_rows = 0;  // S
    std::vector<std::vector<uint32_t>> part_ids;  // global ids of a partition in local order
    std::vector<int32_t> schedule;                // [steps][W][2], the reference's order (solver.h:519-575)
    std::vector<int> order;                       // the episode's step order: head groups interleaved when P = m W, m > 1
    int num_step = 0;
    std::vector<float> vertex, context;           // host embeddings, global order (the numpy views)
    std::vector<std::vector<float>> vertex_moments, context_moments;  // kept across train(resume=True)
    // train
    std::string model;
    gvx_train_config config{};
    int mode = 0;
    uint64_t batch_id = 0, num_batch = 0;
    uint64_t lists_prefetched = 0;  // visits whose first chunk of work lists was built ahead and trained as built (prefetch_lists)
    double train_seconds = 0;
    gvs_sampler *sampler = nullptr;
    int sampler_mode = -1;
    float sampler_p = 0, sampler_q = 0;
    std::vector<Worker> workers;  // the LOCAL workers
    gvx::Comm *comm = nullptr, *route = nullptr;  // exchange of head shards / routing of walk pools (own communicators)
    std::string transport_name;
    // where every head partition sits in the slab, the same on every worker of the job (claim_slots)
    std::vector<int> slot_of, part_at;
    std::vector<std::vector<int>> claimed;  // per head group: the heads (rank order) it was last arranged for
    uint64_t exchanged_bytes = 0, exchanges = 0;
    bool grouped = false;  // this training regroups its pools (pair order)
    bool spread = false;   // this training spreads its walk-ordered pools over the launches that train them (gvk_spread_pairs, §3.1.3)
    bool reordered() const { return grouped || spread; }
    // hub rows trained by chains (gvk_train_episode_hot): per partition, how many of its first rows (they are
    // ordered by falling degree) are owned by a chain when the partition is a block's head / tail table; 0 everywhere = off
    std::vector<uint32_t> hub_rows;
    std::vector<int> hub_top_entries;  // per partition: updates a batch is expected to hold for its largest hub row (head role)
    bool hubs = false;
    bool resident_pools = false, session_open = false;
    std::vector<uint32_t *> host_sets[2];  // pinned host pools, two sets, one pool per block (index hp * P + tp)
    std::string info_text;

    ~gvx_solver() { release(); }

    int num_local() const { return (int)device_ids.size(); }
    bool walk_ordered() const {  // DeepWalk / node2vec pools: the pairs of a head node back to back (no pseudo shuffle)
        return mode != GVS_MODE_EDGE && config.shuffle_base == 1;
    }
    bool routed() const {  // walk pools are sampled in slices by every worker and routed to the worker that trains them
        return mode != GVS_MODE_EDGE && num_worker > 1 && (device_sampling || distributed);
    }

    void release_device() {
        if (device_sampling && !workers.empty()) {  // the device samplers go on where they stopped at the next train()
            sample_positions.assign(workers.size(), 0);
            for (size_t r = 0; r < workers.size(); r++) sample_positions[r] = workers[r].sample_index;
        }
        for (Worker &w : workers) {
            hipSetDevice(w.device);
            hipDeviceSynchronize();
        }
        delete comm, delete route;
        comm = route = nullptr;
        for (Worker &w : workers) {
            hipSetDevice(w.device);
            hipFree(w.head), hipFree(w.context), hipFree(w.loss), hipFree(w.pool[0]), hipFree(w.pool[1]);
            hipFree(w.landing), hipFree(w.group_workspace), hipFree(w.hub_workspaces[0]), hipFree(w.hub_workspaces[1]), hipFree(w.agreement);
            for (int b = 0; b < 2; b++) {
                if (w.lists_built[b]) hipEventDestroy(w.lists_built[b]);
                if (w.lists_trained[b]) hipEventDestroy(w.lists_trained[b]);
            }
            if (w.chains) gvk_ahead_release(w.chains), hipStreamDestroy(w.chains);
            if (w.lists) hipStreamDestroy(w.lists);
            for (auto *t : w.negative_tables) hipFree(t);
            for (auto *t : w.negative_classes) hipFree(t);
            hipFree(w.block_pools[0]), hipFree(w.block_pools[1]), hipFree(w.route_send), hipFree(w.route_recv);
            if (w.route_host) hipHostFree(w.route_host);
            for (auto &b : w.edge_blocks) hipFree(b.table);
            hipFree((void *)w.walk.flat_offsets), hipFree((void *)w.walk.edges_uv), hipFree((void *)w.walk.edge_table);
            hipFree((void *)w.walk.neighbor_table), hipFree((void *)w.walk.sorted_neighbors), hipFree((void *)w.walk.local);
            hipFree(w.walk_part), hipFree(w.walk_offsets), hipFree(w.walk_counters), hipFree(w.walk_accept);
            for (hipEvent_t e : w.filled)
                if (e) hipEventDestroy(e);
            if (w.episode_end) hipEventDestroy(w.episode_end);
            if (w.sample) hipStreamDestroy(w.sample);
            for (int b = 0; b < 2; b++) {
                if (w.uploaded[b]) hipEventDestroy(w.uploaded[b]);
                if (w.released[b]) hipEventDestroy(w.released[b]);
            }
            if (w.trained) hipEventDestroy(w.trained);
            for (auto e : w.copied) hipEventDestroy(e);
            for (auto e : w.gathered)
                if (e) hipEventDestroy(e);
            if (w.compute) hipStreamDestroy(w.compute);
            if (w.copy) hipStreamDestroy(w.copy);
            if (w.exchange) hipStreamDestroy(w.exchange);
        }
        workers.clear();
        for (auto &set : host_sets) {
            for (uint32_t *p : set)
                if (p) hipHostFree(p);
            set.clear();
        }
        session_open = false;
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
        return w.head + ((size_t)(streamed ? 0 : slot_of[hp]) * (1 + num_moment) + table) * table_floats();
    }
    float *context_table(Worker &w, int ti, int table) {
        return w.context + ((size_t)(streamed ? 0 : ti) * (1 + num_moment) + table) * table_floats();
    }
    int block_of(int step, const Worker &w, int which) const { return schedule[((size_t)step * num_worker + w.rank) * 2 + which]; }
    size_t tail_index(const Worker &w, int tp) const { return std::find(w.tails.begin(), w.tails.end(), tp) - w.tails.begin(); }

    size_t memory_demand(int P, int requested_episode, bool as_streamed = false) const;
    int load_block(Worker &w, int hp, int tp);
    int store_block(Worker &w, int hp, int tp);
    int configure(const gvx_train_config &c);
    int prepare_devices();
    int prepare_comm();
    int allocate_pools();
    int upload();
    int write_back();
    int move_table(bool to_device, Worker &w, float *device_table, std::vector<float> &host, int partition);
    int episode_loop();
    int train_block(Worker &w, int hp, int tp, const uint32_t *pool, int first, int count);
    int fill(int set);
    int host_fill(int set);
    int prepare_device_sampling();
    int device_fill(int set);
    int route_slices(int set);
    int hub_parts_of(int hp, int tp) const;
    int hub_workspace_for(Worker &w, size_t need);
    gvk_negative_source negative_source(Worker &w, int tp);
    int build_lists(Worker &w, int hp, int tp, const uint32_t *batches, uint64_t first_id, int m, int parts, bool ahead);
    int prefetch_lists(Worker &w, uint64_t next_batch_id);
    void discard_prefetched(Worker &w);
    bool hub_rounds_of(int hp, int tp) const;
    double hub_graph_share = 0;  // the largest vertex's share of the graph's total degree
    std::vector<double> hub_rest_hits;  // per partition: hit-weighted mean of the expected hits per batch of the rows that are not hub rows (as context rows)
    std::vector<double> hub_top_share, hub_next_hits;  // per partition: the largest row's share of the partition's degree; expected hits per batch of the first row that is not a hub row
    int stage(Worker &w, int step, int set, int b);
    int train_step(int step, int set, int first, int count, bool stage_next, int next_step, int next_set);
    int claim_slots(int step);
    int wait_exchange(Worker &w, int group);
    int exchange(int step);
    int allocate_host_sets();
    uint32_t *block_pool(Worker &w, int set, int hp, int tp) {
        return w.block_pools[set] + (tail_index(w, tp) * num_partition + hp) * (size_t)episode_size * batch_size * 2;
    }
    const uint32_t *trained_pool(Worker &w, int set, int b, int hp, int tp) {
        return w.block_pools[0] && !reordered() ? block_pool(w, set, hp, tp) : w.pool[b];
    }
    void make_info();
};

namespace {

const char *optimizer_name(int type) {
    static const char *const names[] = {"SGD", "Momentum", "AdaGrad", "RMSprop", "Adam"};
    return type >= 0 && type <= GVK_ADAM ? names[type] : "Default";
}

// The extern "C" entry points promise "nothing aborts": a C++ exception (std::bad_alloc from a large vector, an error a
// schedule callback threw) becomes an error code at the boundary.
template <class F>
int guarded(const char *what, F body) {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return gvk_fail(GVK_ENOMEM, "%s: out of host memory", what);
    } catch (const std::exception &e) {
        return gvk_fail(GVK_EINVAL, "%s: %s", what, e.what());
    } catch (...) {
        return gvk_fail(GVK_EINVAL, "%s: unknown exception", what);
    }
}

struct JoinOnExit {  // a filler thread never outlives the scope that started it, whatever leaves the scope
    std::thread &thread;
    ~JoinOnExit() {
        if (thread.joinable()) thread.join();
    }
};

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
    const size_t pool = episode * batch_size * 8;
    demand += 3 * pool + pool;  // two pool buffers + the regrouping landing buffer + the regrouping workspace
    // hub rows by chains: the work lists of up to kHubChunk batches (8 bytes per list entry, 2 (k + 1) entries per sample at
    // most, as much again for the chains' records) — of fewer batches where that would be more than a sixteenth of the memory
    // (prepare_devices).  The mirrors of the hub rows (at most 3 x 2 x kMaxHubRows rows: 50 MB at dim 128) are not counted.
    // — only where chains can exist: SGD, and not fidelity = "throughput"
    // Two such workspaces: the lists of the next chunk are built while this one trains.
    if (optimizer.type == GVK_SGD && (fidelity != 0 || hub_rows_request > -2) && hub_rows_request != 0)
        demand += std::min((size_t)2 * kHubChunk * batch_size * (num_negative + 1) * 32, gpu_memory_limit / 16);
    if (device_sampling && !as_streamed) {
        // the pools of every block a worker trains, two episodes, + its slices on their way to the owners (send + receive)
        demand += 4 * tails * P * pool;
        // what the device samplers draw from: ~16 B per directed edge (edge mode: packed block tables over the worker's
        // columns) or ~44 B (walk modes: CSR, two alias tables, sorted neighbours, maps), on every worker
        const uint64_t D = graph ? gvs_graph_num_directed_edge(graph) : 0;
        demand += D * 44 + (size_t)num_vertex * 16;
    }
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

namespace {

gvx_solver *new_solver(int dim, int num_sampler_per_worker, size_t gpu_memory_limit) {
    if (dim != 32 && dim != 64 && dim != 96 && dim != 128 && dim != 256 && dim != 512) {
        gvk_fail(GVK_EDIM, "GraphSolver: dim must be one of 32, 64, 96, 128, 256, 512");
        return nullptr;
    }
    std::unique_ptr<gvx_solver> s(new gvx_solver());
    s->dim = dim;
    s->num_sampler_per_worker = num_sampler_per_worker;
    s->memory_request = gpu_memory_limit;
    s->gpu_memory_limit = gpu_memory_limit;
    s->config.random_walk_length = 40, s->config.random_walk_batch_size = 100, s->config.p = s->config.q = 1;
    s->config.positive_reuse = 1, s->config.log_frequency = 1000;
    return s.release();
}

}  // namespace

extern "C" gvx_solver *gvx_solver_create(int dim, const int *device_ids, int num_device, int num_sampler_per_worker,
                                         size_t gpu_memory_limit) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        gvk_fail(GVK_EHIP, "No GPU devices found");  // solver.h:176
        return nullptr;
    }
    std::unique_ptr<gvx_solver> s(new_solver(dim, num_sampler_per_worker, gpu_memory_limit));
    if (!s) return nullptr;
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
        s->num_sampler_per_worker = std::max(cpu_budget() / s->num_worker - 1, 1);
    s->num_sampler = s->num_sampler_per_worker * s->num_worker;
    return s.release();
}

extern "C" gvx_solver *gvx_solver_create_distributed(int dim, int rank, int world_size, int device_id, const void *unique_id,
                                                     size_t unique_id_bytes, const gvx_transport *transport,
                                                     int num_sampler_per_worker, size_t gpu_memory_limit) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        gvk_fail(GVK_EHIP, "No GPU devices found");
        return nullptr;
    }
    if (world_size < 1 || rank < 0 || rank >= world_size || device_id < 0 || device_id >= count) {
        gvk_fail(GVK_EINVAL, "GraphSolver: rank %d of %d on GPU %d (%d device(s) visible)", rank, world_size, device_id, count);
        return nullptr;
    }
    if (transport && (!transport->all_gather || !transport->all_to_all)) {
        gvk_fail(GVK_EINVAL, "GraphSolver: a transport needs all_gather and all_to_all");
        return nullptr;
    }
    std::unique_ptr<gvx_solver> s(new_solver(dim, num_sampler_per_worker, gpu_memory_limit));
    if (!s) return nullptr;
    s->device_ids.assign(1, device_id);
    s->num_worker = world_size, s->first_rank = rank, s->distributed = world_size > 1;
    if (transport) s->has_transport = true, s->transport = *transport;
    if (unique_id && unique_id_bytes) s->unique_id.assign((const char *)unique_id, (const char *)unique_id + unique_id_bytes);
    if (num_sampler_per_worker == GVX_AUTO) {
        // the usable CPUs are shared by the processes of this node (LOCAL_WORLD_SIZE as torchrun / mpirun export it)
        const char *local_world = getenv("LOCAL_WORLD_SIZE");
        const int sharing = local_world && atoi(local_world) > 0 ? atoi(local_world) : world_size;
        s->num_sampler_per_worker = std::max(cpu_budget() / sharing - 1, 1);
    }
    s->num_sampler = s->num_sampler_per_worker * world_size;
    return s.release();
}

extern "C" int gvx_unique_id(void *out, size_t capacity) {
    if (!out || capacity < GVX_UNIQUE_ID_BYTES) return gvk_fail(GVK_EINVAL, "gvx_unique_id: %d bytes needed", GVX_UNIQUE_ID_BYTES);
    std::string why;
    for (int i = 0; i < 2; i++) {
        int rc = gvx::rccl_unique_id((char *)out + i * (GVX_UNIQUE_ID_BYTES / 2), GVX_UNIQUE_ID_BYTES / 2, &why);
        if (rc != GVK_OK) return gvk_fail(rc, "gvx_unique_id: %s", why.c_str());
    }
    return GVK_OK;
}

extern "C" int gvx_rccl_selftest(int device) {
    return guarded("gvx_rccl_selftest", [&]() -> int {
        std::string why;
        std::unique_ptr<gvx::Comm> comm(gvx::make_rccl_in_process({device}, &why));
        if (!comm) return gvk_fail(GVK_EHIP, "gvx_rccl_selftest: %s", why.c_str());
        HIP_TRY(hipSetDevice(device));
        const size_t n = 1 << 20;
        std::vector<uint32_t> host(n), back(n), other(n);
        for (size_t i = 0; i < n; i++) host[i] = (uint32_t)(i * 2654435761u);
        uint32_t *a = nullptr, *b = nullptr;
        hipStream_t stream = nullptr;
        HIP_TRY(hipMalloc(&a, n * 4));
        HIP_TRY(hipMalloc(&b, n * 4));
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        HIP_TRY(hipMemcpy(a, host.data(), n * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(b, 0, n * 4));
        int rc = comm->all_gather({{0, device, stream}}, {a}, n * 4);
        if (rc == GVK_OK) rc = comm->all_to_all({{0, device, stream}}, {a}, {b}, n * 4);
        if (rc == GVK_OK && hipStreamSynchronize(stream) != hipSuccess) rc = gvk_fail(GVK_EHIP, "gvx_rccl_selftest: stream failed");
        if (rc == GVK_OK) {
            hipMemcpy(back.data(), a, n * 4, hipMemcpyDeviceToHost);
            hipMemcpy(other.data(), b, n * 4, hipMemcpyDeviceToHost);
            if (back != host || other != host) rc = gvk_fail(GVK_EHIP, "gvx_rccl_selftest: a one-rank collective changed the data");
        }
        hipStreamDestroy(stream);
        hipFree(a), hipFree(b);
        return rc;
    });
}

extern "C" void gvx_solver_destroy(gvx_solver *s) { delete s; }

extern "C" int gvx_solver_set(gvx_solver *s, int option, int64_t value) {
    if (!s) return gvk_fail(GVK_EINVAL, "gvx_solver_set: null solver");
    if (option == GVX_DEVICE_SAMPLING && (value == 0 || value == 1)) {
        s->device_sampling = value != 0;
        return GVK_OK;
    }
    if (option == GVX_PAIR_ORDER && value >= 0 && value <= 2) {
        s->pair_order_request = (int)value;
        return GVK_OK;
    }
    if (option == GVX_NEGATIVE_TABLE && value >= 0 && value <= 2) {
        s->negative_table_request = (int)value;
        return GVK_OK;
    }
    if (option == GVX_SEED) {
        s->seed = (uint64_t)value;
        return GVK_OK;
    }
    if (option == GVX_FIDELITY && value >= -1 && value <= 1) {
        s->fidelity = (int)value;
        return GVK_OK;
    }
    if (option == GVX_HUB_PARTS && value >= 0 && value <= 1024) {
        s->hub_parts_request = (int)value;
        return GVK_OK;
    }
    if (option == GVX_HUB_ROWS && value >= -2) {
        s->hub_rows_request = value;
        return GVK_OK;
    }
    if (option == GVX_HUB_LERP && value >= -1 && value <= 1) {
        s->hub_lerp_request = (int)value;
        return GVK_OK;
    }
    if (option == GVX_HUB_GROUP && value >= 0 && value <= 127) {
        s->hub_group_request = (int)value;
        return GVK_OK;
    }
    if (option == GVX_HUB_EXECUTOR && value >= -1 && value <= 2) {
        s->hub_executor_request = (int)value;
        return GVK_OK;
    }
    if (option == GVX_HUB_PAIR_LAUNCHES && value >= 0 && value <= 127) {
        s->hub_pair_launches_request = (int)value;
        return GVK_OK;
    }
    if (option == GVX_HUB_ROUNDS && value >= -1 && value <= 1) {
        s->hub_rounds_request = (int)value;
        return GVK_OK;
    }
    if (option == GVX_HUB_CHAIN_CAP && value >= 0 && value <= (1 << 20)) {
        s->hub_chain_cap_request = (int)value;
        return GVK_OK;
    }
    if (option == GVX_NODE2VEC_TABLE_LIMIT && value >= 0) {
        s->node2vec_table_limit = (uint64_t)value;
        return GVK_OK;
    }
    return gvk_fail(GVK_EINVAL, "gvx_solver_set: unknown option %d or unsupported value %lld", option, (long long)value);
}

extern "C" int gvx_solver_build(gvx_solver *s, const gvs_graph *graph, const gvx_optimizer *optimizer, int num_partition,
                                int num_negative, int batch_size, int episode_size) {
    if (!s || !graph) return gvk_fail(GVK_EINVAL, "gvx_solver_build: null solver / graph");
    return guarded("GraphSolver.build", [&]() -> int {
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
    s->gpu_memory_limit = limit;  // memory_demand reads it (the hub workspace term): set before partitions are chosen
    if (num_partition == GVX_AUTO) {
        // resident, the design of this engine: N * dim * 4 * (1 + m) * (1 + 1 / W) bytes of tables whatever the partition
        // count — more partitions only shrink the pools.  When even the most partitions do not fit, the reference's
        // scheme: one head and one tail partition per worker in HBM, everything else in host memory (solver.h:365-384)
        num_partition = W;
        while (num_partition < kMaxPartition && s->memory_demand(num_partition, episode_size) >= limit) num_partition += W;
        if (s->memory_demand(num_partition, episode_size) >= limit) {
            if (s->distributed)
                return gvk_fail(GVK_ENOMEM, "The tables do not fit the GPU memory limit; partitions that travel through host "
                                            "memory need all workers in one process (device_ids=[...])");
            s->streamed = true;
            num_partition = W;
            while (num_partition < kMaxPartition && s->memory_demand(num_partition, episode_size, true) >= limit) num_partition += W;
            log_message(1, "The vertex table does not fit the GPU memory limit next to the sample pools: partitions will "
                           "travel through host memory between blocks (%d partitions)", num_partition);
        }
    } else {
        s->streamed = !s->distributed && s->memory_demand(num_partition, episode_size) >= limit &&
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
    // The reference walks the block groups x-major (solver.h:562-574): all steps that use head partitions x .. x + W - 1
    // come back to back, and each needs the exchange of the one before.  An episode may visit its P^2 blocks in any
    // order, so with several workers and P = m W (m > 1) the steps are interleaved across the m head groups: the all-gather
    // of group x's slab then runs on the exchange streams while the next m - 1 steps train on the other groups.  One worker
    // has no exchange to hide and keeps the reference's order (when a training is shorter than an episode — the automatic
    // episode size at P = 4 on the benchmark graph is 436 batches per block, 6 976 per episode — the order decides which
    // blocks meet the large learning rates).
    s->order.clear();
    const int m = P > 1 ? P / W : 1;
    if (m > 1 && W > 1 && !s->streamed) {
        for (int yi = 0; yi < m; yi++)
            for (int o = 0; o < W; o++)
                for (int xi = 0; xi < m; xi++) s->order.push_back((xi * m + yi) * W + o);
    } else {
        for (int step = 0; step < s->num_step; step++) s->order.push_back(step);
    }

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
    });
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
    for (Worker &w : workers) {  // lists built ahead for a visit of the training before this one (a session closed early): its pools are refilled
        discard_prefetched(w);
        w.staged.valid = false;
    }
    make_info();
    if (first_rank == 0)
        log_message(1, "\n<<<<<<<<<<<<<<<<<<<<<<<<<<<<<<<<<<<<<<<<\n%s\n>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>", info_text.c_str());
    if (!c.resume) {  // GraphSolver::init_embeddings, graph.cuh:724-731 — the same table on every process
        std::mt19937 generator(5489u + (uint32_t)seed);
        std::uniform_real_distribution<float> init(-0.5f / dim, 0.5f / dim);
        for (float &x : vertex) x = init(generator);
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
        if (entries > node2vec_table_limit) {
            log_message(1, "node2vec: %llu per-edge table entries exceed the limit of %llu; sampling by rejection",
                        (unsigned long long)entries, (unsigned long long)node2vec_table_limit);
            mode = GVS_MODE_BIASED_REJECT;
        }
    }
    // regroup by table size, at dim >= 64: cache-resident tables (< 16 MiB) always — adjacent
    // same-head samples are then trained as runs, which keeps training close to sequential (§7); shard-sized tables
    // (< 256 MiB) for independent edge draws — a shared head row becomes one fetch; larger tables keep the sampler's order
    const size_t table_bytes = (size_t)part_rows * dim * 4;
    // — and never the walk-ordered pools of DeepWalk / node2vec, which are trained pair by pair in the sampler's order
    // (gvk.h GVK_PAIRS_OF_WALKS)
    grouped = pair_order_request == 2 ||
              (pair_order_request == 0 && dim >= 64 && !walk_ordered() &&
               (table_bytes < ((size_t)16 << 20) || (table_bytes < ((size_t)256 << 20) && mode == GVS_MODE_EDGE)));
    // the walk-ordered pools spread over the units (below) — unless chains own every row: the pairs train nothing then
    auto final_spread = [&]() {
        bool on = walk_ordered() && pair_order_request != 1 && !grouped;
        for (int p = 0; p < num_partition && on; p++)
            if (hubs && hub_rows[p] == part_rows) on = false;
        return on;
    };
    spread = final_spread();  // hub_parts_of reads it (again below once the hub rows are known)
    // hub rows (GVX_HUB_ROWS): the rows a part of a batch is expected to hit kHubHitsPerPart times or more — as a head / tail (degree share of
    // the partition) or as a negative (share of degree^exponent) — are trained by chains; their batches keep the sampler's order
    hub_rows.assign(num_partition, 0);
    hub_top_entries.assign(num_partition, 0);
    hub_top_share.assign(num_partition, 0.0), hub_next_hits.assign(num_partition, 0.0), hub_rest_hits.assign(num_partition, 0.0);
    hubs = false;
    // the default rule (-2): every row of the walk-ordered pools of DeepWalk / node2vec on one partition of at most kMaxHubRows
    // rows; else the rows a part of a batch is expected to hit kHubHitsPerPart times — on tables that do not live in the caches
    // (smaller ones are regrouped and trained as runs of same-head samples, §3.1.1, pinned against the reference's loop the
    // same way); GVX_FIDELITY 0: none, every row pair by pair (Hogwild)
    int64_t request = hub_rows_request;
    // a cache-resident LINE table gets chains, too, where its hottest row leaves that feasible (below): else runs, as before
    const bool small_table = hub_rows_request == -2 && fidelity == -1 && !walk_ordered() && table_bytes < ((size_t)16 << 20);
    hub_max_parts = table_bytes < ((size_t)16 << 20) ? kHubMaxPartsResident : kHubMaxParts;
    if (request == -2) {
        if (fidelity == 0 || (fidelity == -1 && pair_order_request == 2)) request = 0;  // "grouped" asked for: regrouped batches, no chains
        else if (walk_ordered() && num_partition == 1 && part_rows <= kMaxHubRows) request = (int64_t)part_rows;
        else request = -1;
    }
    // every optimizer has chains: SGD's compose in closed form (tasks side by side), a moment optimizer's are one sequential task per
    // hub row and unit with the row's moment rows in registers (gvk_chains.hip train_moment_chains) — correct, and as slow as the
    // largest hub row's updates per batch one after the other; fidelity='throughput' trains every row pair by pair instead
    const bool chains_exist = true;
    if (request != 0 && !chains_exist && (fidelity == 1 || hub_rows_request > -2))
        return gvk_fail(GVK_EINVAL, "hub rows are trained by chains for SGD only: fidelity='reference' / hub_rows cannot be "
                        "honoured for this optimizer (use fidelity='throughput')");
    if (hub_rows_request == -2 && fidelity == -1 && pair_order_request == 2 && first_rank == 0 && !order_said) {
        log_message(1, "pair_order='grouped': batches are regrouped and trained pair by pair; hub rows get no chains (fidelity='reference' asks for them)");
        order_said = true;
    }
    if (fidelity == 1 && pair_order_request == 2 && first_rank == 0 && !order_said) {
        log_message(1, "fidelity='reference' with pair_order='grouped': tables with hub rows keep the sampler's order (chains), the regrouping is dropped there");
        order_said = true;
    }
    if (request != 0) {
        const float *vertex_weights = gvs_graph_vertex_weights(graph);
        {
            double all = 0, largest = 0;
            for (int p = 0; p < num_partition; p++)
                for (uint32_t id : part_ids[p]) all += vertex_weights[id], largest = std::max(largest, (double)vertex_weights[id]);
            hub_graph_share = largest / std::max(all, 1e-30);
        }
        for (int p = 0; p < num_partition; p++) {
            const std::vector<uint32_t> &ids = part_ids[p];
            uint64_t rows = 0;
            if (request > 0) {
                rows = (uint64_t)request;
            } else {
                double total = 0, total_negative = 0;
                for (uint32_t id : ids) total += vertex_weights[id], total_negative += std::pow((double)vertex_weights[id], (double)c.negative_sample_exponent);
                // the largest row: the head of batch_size * share samples, (num_negative + 1) chain entries each -> the parts a
                // batch of this partition's blocks is trained as (hub_parts_of)
                if (!ids.empty())
                    hub_top_entries[p] = (int)std::min(1e9, (double)batch_size * (num_negative + 1) * vertex_weights[ids[0]] / std::max(total, 1e-30));
                hub_top_share[p] = ids.empty() ? 0.0 : vertex_weights[ids[0]] / std::max(total, 1e-30);
                // The rows a BATCH is expected to hit kHubHits = once or more — as a head / tail (its share of the partition's degree) or as
                // a negative (its share of degree^exponent) — are hub rows: of two concurrent updates of a row one is lost, and a row that
                // is hit h times per batch meets another of its hits inside a part with probability ~ h / parts.  What the chains do
                // not own (the table is larger than kMaxHubRows, or the rule stops first) decides the parts with the largest hub
                // row (hub_parts_of): no row outside the chains is to be hit more than kHubHitsPerPart times per part.
                auto hits_of = [&](uint32_t id) {
                    const double w = vertex_weights[id];
                    return std::max(batch_size * w / std::max(total, 1e-30),
                                    (double)batch_size * num_negative * std::pow(w, (double)c.negative_sample_exponent) / std::max(total_negative, 1e-30));
                };
                for (uint32_t id : ids) {  // falling degree: stop at the first row below the threshold
                    if (hits_of(id) < kHubHits || rows >= kMaxHubRows) break;
                    rows++;
                }
                hub_next_hits[p] = rows < ids.size() ? hits_of(ids[rows]) : 0.0;
            }
            hub_rows[p] = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(rows, ids.size()), kMaxHubRows);
            if (request > 0 && !ids.empty()) {
                double total = 0;
                for (uint32_t id : ids) total += vertex_weights[id];
                hub_top_entries[p] = (int)std::min(1e9, (double)batch_size * (num_negative + 1) * vertex_weights[ids[0]] / std::max(total, 1e-30));
                hub_top_share[p] = vertex_weights[ids[0]] / std::max(total, 1e-30);
                hub_next_hits[p] = 0;  // the caller chose the hub rows: the parts follow the largest of them
            }
            {   // what the rows the chains do NOT own meet per batch as context rows (tail or negative), weighted by the hits themselves:
                // the expected hits per batch of the row a random such hit lands on — walk-ordered pools take their parts from it (hub_parts_of)
                double total = 0, total_negative = 0, s1 = 0, s2 = 0;
                for (uint32_t id : ids) total += vertex_weights[id], total_negative += std::pow((double)vertex_weights[id], (double)c.negative_sample_exponent);
                for (size_t i = hub_rows[p]; i < ids.size(); i++) {
                    const double w = vertex_weights[ids[i]];
                    const double h = batch_size * w / std::max(total, 1e-30) +
                                     (double)batch_size * num_negative * std::pow(w, (double)c.negative_sample_exponent) / std::max(total_negative, 1e-30);
                    s1 += h, s2 += h * h;
                }
                hub_rest_hits[p] = s1 > 0 ? s2 / s1 : 0.0;
            }
            hubs = hubs || hub_rows[p] > 0;
        }
        // Chains need so many parts that the largest hub row meets about kHubEntriesPerPart of its updates per part (tasks of one
        // chain run side by side from the same start: with thousands of entries per part their steps add up and the row
        // overshoots — a 6 250-row partition of a 100k-node graph at P = 16 diverges).  Where that takes more than
        // kHubMaxPartsResident parts a cache-resident table keeps regrouping + runs (§3.1.1, pinned at P = 8 / 16 in §7.8).
        if (small_table && hubs) {
            const int worst = *std::max_element(hub_top_entries.begin(), hub_top_entries.end());
            if ((worst + kHubEntriesPerPart - 1) / kHubEntriesPerPart > kHubMaxPartsResident) {
                hub_rows.assign(num_partition, 0);
                hubs = false;
            }
        }
        if (hubs && !chains_exist) {  // the rule found hub rows and this optimizer has no chains for them: said once per solver
            if (first_rank == 0 && !hogwild_said)
                log_message(1, "WARNING: this optimizer has no chains for hub rows: every row is trained pair by pair "
                            "(Hogwild); on this graph the %u largest rows of a partition then keep a few of their updates per batch",
                            *std::max_element(hub_rows.begin(), hub_rows.end()));
            hogwild_said = true;
            hub_rows.assign(num_partition, 0);
            hubs = false;
        }
        if (hubs && hub_parts_request > 0 && batch_size % hub_parts_request)
            return gvk_fail(GVK_EINVAL, "hub_parts (%d) must divide the batch size (%d)", hub_parts_request, batch_size);
        if (hubs) grouped = false;
        spread = final_spread();  // hub_parts_of below asks for the parts training will use
        // ... and the same question for every table, with the parts a block can actually be given (hub_parts_of: a divisor of the
        // batch size near the rule's, at most hub_max_parts — a prime batch size has none): past kHubMaxEntriesPerPart updates of
        // its largest hub row per part the chains' entries, side by side from the part's start state, overshoot; such a block's
        // rows are trained pair by pair and the log says so (asked for explicitly: an error)
        if (hubs) {
            int worst = 0, worst_parts = 1;
            for (int hp = 0; hp < num_partition; hp++)
                for (int tp = 0; tp < num_partition; tp++) {
                    if (hub_rows[hp] + hub_rows[tp] == 0) continue;
                    const int parts = hub_parts_of(hp, tp), top = std::max(hub_top_entries[hp], hub_top_entries[tp]);
                    if ((top + parts - 1) / parts > worst) worst = (top + parts - 1) / parts, worst_parts = parts;
                }
            bool in_rounds = true;  // with rounds no more than 16 x 4 entries work side by side however long the chain: nothing to overshoot (a moment optimizer's chains are sequential: the same)
            for (int hp = 0; hp < num_partition && in_rounds; hp++)
                for (int tp = 0; tp < num_partition && in_rounds; tp++)
                    if (hub_rows[hp] + hub_rows[tp] > 0 && !hub_rounds_of(hp, tp) && optimizer.type == GVK_SGD) in_rounds = false;
            if (worst > kHubMaxEntriesPerPart && !in_rounds) {
                if (fidelity == 1 || hub_rows_request > -2)
                    return gvk_fail(GVK_EINVAL, "hub rows by chains: the largest hub row would meet %d of its updates per part (a batch as %d "
                                    "parts; at most %d keep the chains stable): use a batch size with more divisors or set hub_parts",
                                    worst, worst_parts, kHubMaxEntriesPerPart);
                if (first_rank == 0)
                    log_message(1, "WARNING: the largest hub row would meet %d of its updates per part of a batch (%d parts can be given, "
                                "at most %d updates per part keep chains stable): every row is trained pair by pair (Hogwild)",
                                worst, worst_parts, kHubMaxEntriesPerPart);
                hub_rows.assign(num_partition, 0);
                hubs = false;
            }
        }
        if (hubs) grouped = false;
    }
    // The walk-ordered pools of DeepWalk / node2vec: a walk emits the pairs of a head node back to back and meets a tail node
    // in pairs a few records apart (graph.cuh:320-348), so consecutive records share rows — side by side in one launch all
    // but one of those updates are lost.  Unless chains own every row (a small partition, above), the pool is spread: record i
    // to the launch i % units (gvk_spread_pairs), so that what the reference's sequential loop trains one after the other is
    // trained by consecutive launches.  pair_order = "sampled" keeps the sampler's order.
    spread = final_spread();
    if (hubs && first_rank == 0 && !rules_said) {  // which of the three rules fired and what they cost: once per solver, at INFO
        int most_parts = 1, least_parts = 1 << 30;
        bool any_rounds = false;
        for (int hp = 0; hp < num_partition; hp++)
            for (int tp = 0; tp < num_partition; tp++) {
                if (hub_rows[hp] + hub_rows[tp] == 0) continue;
                most_parts = std::max(most_parts, hub_parts_of(hp, tp)), least_parts = std::min(least_parts, hub_parts_of(hp, tp));
                any_rounds = any_rounds || hub_rounds_of(hp, tp);
            }
        const int top = *std::max_element(hub_top_entries.begin(), hub_top_entries.end());
        log_message(0, "hub rows by chains: up to %u rows per table (a batch is expected to hit them once or more), a batch as %d%s%d parts = launches "
                    "(the largest hub row meets %d updates per batch: about %d per part%s%s), long chains %s%s",
                    *std::max_element(hub_rows.begin(), hub_rows.end()), least_parts, least_parts == most_parts ? " = " : " .. ", most_parts, top,
                    top / std::max(most_parts, 1), spread ? "; walk-ordered pools: parts also by the hits of the rows outside the chains" : "",
                    most_parts > 8 ? ": every launch costs its longest chain, 9-13 us, however few pairs it holds" : "",
                    optimizer.type != GVK_SGD ? "as ONE sequential task each (a moment optimizer: a batch then takes as long as its largest hub row's updates one after the other)"
                    : (any_rounds ? "in rounds of 4 entries per task (the largest vertex takes more than 2 % of the degree: about 28 % slower)" : "in one round"),
                    fidelity == 0 ? "" : "; fidelity='throughput' trains every row pair by pair instead (2-3 x the rate, hub rows then keep a few of their updates per batch)");
        rules_said = true;
    }
    if (routed()) {
        if (pool_size % num_worker)
            return gvk_fail(GVK_EINVAL, "episode_size * batch_size (%zu) must be a multiple of #worker (%d) for the "
                            "random-walk models", pool_size, num_worker);
        if (pool_size / num_worker % c.shuffle_base)
            return gvk_fail(GVK_EINVAL, "Can't perform pseudo shuffle on %zu elements by a shuffle base of %d",
                            pool_size / num_worker, c.shuffle_base);
    }
    if (device_sampling) {  // the GPUs draw the positives: no CPU sampler, none of its tables
        if (streamed)
            return gvk_fail(GVK_EINVAL, "device sampling needs every partition resident in GPU memory; raise gpu_memory_limit "
                                        "or use the CPU samplers");
        if (gvs_graph_num_directed_edge(graph) >= ((uint64_t)1 << 32))
            return gvk_fail(GVK_EINVAL, "device sampling supports graphs with fewer than 2^32 directed edges");
        return GVK_OK;
    }
    if (!sampler) {
        sampler = gvs_sampler_create(graph, part.data(), local.data(), num_partition,
                                     (seed + 0x9E3779B97F4A7C15ull * (uint64_t)(first_rank + 1)));
        if (!sampler) return GVK_EINVAL;
    }
    const int threads = num_sampler_per_worker * num_local();
    if (sampler_mode != mode || sampler_p != c.p || sampler_q != c.q) {  // get_sample_function, graph.cuh:680-721
        GVK_TRY(gvs_sampler_prepare(sampler, mode, c.p, c.q, threads + 1));
        sampler_mode = mode, sampler_p = c.p, sampler_q = c.q;
    }
    return GVK_OK;
}

// ---- device state --------------------------------------------------------------------------------------------------

// The carriers of the exchange and of the walk-pool routing (gvx_comm.h): a caller-supplied transport; RCCL — one
// communicator per GPU, a second set for the routing, whose all-to-all runs on its own stream and host thread beside the
// all-gathers; or, when the workers of one process share GPUs, event-ordered device copies.
int gvx_solver::prepare_comm() {
    if (num_worker == 1 || streamed) return GVK_OK;
    std::string why;
    if (has_transport) {
        comm = gvx::make_callbacks(transport, first_rank, num_worker);
        route = gvx::make_callbacks(transport, first_rank, num_worker);
    } else if (distributed) {
        const size_t half = GVX_UNIQUE_ID_BYTES / 2;
        if (unique_id.size() < GVX_UNIQUE_ID_BYTES)
            return gvk_fail(GVK_EINVAL, "GraphSolver: %d processes need the %d bytes of gvx_unique_id() from rank 0", num_worker,
                            GVX_UNIQUE_ID_BYTES);
        comm = gvx::make_rccl_rank(first_rank, num_worker, device_ids[0], unique_id.data(), half, &why);
        if (comm) route = gvx::make_rccl_rank(first_rank, num_worker, device_ids[0], unique_id.data() + half, half, &why);
        if (!comm || !route) return gvk_fail(GVK_EHIP, "GraphSolver: no RCCL communicator: %s", why.c_str());
    } else {
        comm = gvx::make_rccl_in_process(device_ids, &why);
        if (comm) route = gvx::make_rccl_in_process(device_ids, &why);
        if (!comm || !route) {
            delete comm, delete route;
            log_message(0, "exchange by device copies (%s)", why.c_str());
            comm = gvx::make_copies(num_worker);
            route = gvx::make_copies(num_worker);
        }
    }
    transport_name = comm->name();
    return GVK_OK;
}

int gvx_solver::prepare_devices() {
    release_device();
    const int P = num_partition, W = num_worker;
    workers.assign(num_local(), Worker());
    slot_of.resize(P), part_at.resize(P);
    for (int p = 0; p < P; p++) slot_of[p] = part_at[p] = p;
    claimed.assign(std::max(P / W, 1), {});
    exchanged_bytes = exchanges = 0;
    for (int l = 0; l < num_local(); l++) {
        Worker &w = workers[l];
        w.rank = first_rank + l;
        w.device = device_ids[l];
        for (int step = 0; step < num_step; step++) {
            const int tp = block_of(step, w, 1);
            if (std::find(w.tails.begin(), w.tails.end(), tp) == w.tails.end()) w.tails.push_back(tp);
        }
        std::sort(w.tails.begin(), w.tails.end());
        if (streamed) {  // any tail partition may come by: a negative sampler for each, tables for one at a time
            w.tails.clear();
            for (int p = 0; p < P; p++) w.tails.push_back(p);
        }
        HIP_TRY(hipSetDevice(w.device));
        for (int q = 0; q < num_local(); q++)  // direct GPU-to-GPU copies (the copies carrier)
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
        // GVX_CHAIN_CU_SHARE=N (measurement): the chain stream on every N-th compute unit only, the compute stream on the others
        if (const char *share = getenv("GVX_CHAIN_CU_SHARE")) {
            const int every = std::max(atoi(share), 2);
            hipDeviceProp_t prop;
            HIP_TRY(hipGetDeviceProperties(&prop, w.device));
            const int words = (prop.multiProcessorCount + 31) / 32;
            std::vector<uint32_t> chain_mask(words, 0), pair_mask(words, 0);
            for (int cu = 0; cu < prop.multiProcessorCount; cu++) ((cu / 8) % every == 0 ? chain_mask : pair_mask)[cu / 32] |= 1u << (cu % 32);
            HIP_TRY(hipStreamDestroy(w.compute));
            HIP_TRY(hipExtStreamCreateWithCUMask(&w.compute, (uint32_t)words, pair_mask.data()));
            HIP_TRY(hipExtStreamCreateWithCUMask(&w.chains, (uint32_t)words, chain_mask.data()));
        } else
        HIP_TRY(hipStreamCreateWithFlags(&w.chains, hipStreamNonBlocking));
        {  // GVX_LISTS_PRIORITY=low / high (measurement): the lists stream at the device's least / greatest priority — measured: 337-366 / 299 against 808 M/s
           // (profiles/r6/experiments/r6_lists_priority_ab.txt): an ordinary stream stays
            const char *knob = getenv("GVX_LISTS_PRIORITY");
            int least = 0, greatest = 0;
            HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
            if (knob && (!strcmp(knob, "low") || !strcmp(knob, "high")))
                HIP_TRY(hipStreamCreateWithPriority(&w.lists, hipStreamNonBlocking, !strcmp(knob, "low") ? least : greatest));
            else
                HIP_TRY(hipStreamCreateWithFlags(&w.lists, hipStreamNonBlocking));
        }
        for (int b = 0; b < 2; b++) {
            HIP_TRY(hipEventCreateWithFlags(&w.lists_built[b], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&w.lists_trained[b], hipEventDisableTiming));
        }
        const size_t head_slots = streamed ? 1 : (size_t)P, context_slots = streamed ? 1 : w.tails.size();
        HIP_TRY(hipMalloc(&w.head, head_slots * slot_floats() * 4));
        HIP_TRY(hipMalloc(&w.context, context_slots * slot_floats() * 4));
        HIP_TRY(hipMemsetAsync(w.head, 0, head_slots * slot_floats() * 4, w.compute));
        HIP_TRY(hipMemsetAsync(w.context, 0, context_slots * slot_floats() * 4, w.compute));
        HIP_TRY(hipMalloc(&w.loss, (size_t)batch_size * 4));
        HIP_TRY(hipMalloc(&w.agreement, (size_t)std::max(num_worker, 1) * 4));
        HIP_TRY(hipMemsetAsync(w.loss, 0, (size_t)batch_size * 4, w.compute));
        for (int b = 0; b < 2; b++) {
            HIP_TRY(hipEventCreateWithFlags(&w.uploaded[b], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&w.released[b], hipEventDisableTiming));
        }
        HIP_TRY(hipEventCreateWithFlags(&w.trained, hipEventDisableTiming));
        w.gathered.assign(std::max(P / W, 1), nullptr);
        w.gathered_valid.assign(w.gathered.size(), 0);
        for (hipEvent_t &e : w.gathered) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
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
            // the table joins the worker's lists BEFORE the copy that may fail, so that release_device() always frees it
            w.negative_tables.push_back(nullptr);
            w.negative_classes.push_back(nullptr);
            w.negative_class_counts.push_back(0);
            const bool by_class = negative_table_request == 2 || (negative_table_request == 0 && (size_t)num_class * 8 <= ids.size());
            if (by_class) {
                HIP_TRY(hipMalloc(&w.negative_classes.back(), num_class * sizeof(gvk_class_entry)));
                HIP_TRY(hipMemcpy(w.negative_classes.back(), classes.data(), num_class * sizeof(gvk_class_entry), hipMemcpyHostToDevice));
                w.negative_class_counts.back() = num_class;
            } else {
                GVK_TRY(gvk_alias_build(weights.data(), weights.size(), prob.data(), alias.data(), 4, packed.data()));
                HIP_TRY(hipMalloc(&w.negative_tables.back(), packed.size() * sizeof(gvk_alias_entry)));
                HIP_TRY(hipMemcpy(w.negative_tables.back(), packed.data(), packed.size() * sizeof(gvk_alias_entry), hipMemcpyHostToDevice));
            }
        }
    }
    GVK_TRY(prepare_comm());  // before the pools: ranks agree on the episode size through it
    GVK_TRY(allocate_pools());
    const size_t n = (size_t)episode_size * batch_size;
    for (Worker &w : workers) {
        if (!w.block_pools[0]) break;
        HIP_TRY(hipSetDevice(w.device));
        HIP_TRY(hipStreamCreateWithFlags(&w.sample, hipStreamNonBlocking));
        for (hipEvent_t &e : w.filled) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&w.episode_end, hipEventDisableTiming));
        if (!routed()) continue;  // its slice of every block pool: [owner rank][the owner's blocks][n / W] out, the same back
        HIP_TRY(hipMalloc(&w.route_send, (size_t)P * P * (n / W) * 8));
        HIP_TRY(hipMalloc(&w.route_recv, (size_t)P * P * (n / W) * 8));
        if (!device_sampling) HIP_TRY(hipHostMalloc(&w.route_host, (size_t)P * P * (n / W) * 8, hipHostMallocDefault));
    }
    return device_sampling ? prepare_device_sampling() : GVK_OK;
}

// The pools are the elastic part, as in the reference (solver.h:437-455): halve the episode until they fit — the two
// device buffers and the regrouping landing buffer, and the pool sets kept in HBM (device sampling, resident sessions).
int gvx_solver::allocate_pools() {
    const int W = num_worker;
    const bool in_hbm = device_sampling || resident_pools || routed();
    // one process per GPU: every rank must end with the same episode size (ranks with different episode sizes would issue
    // collectives of different sizes and hang).  The agreement is itself collective, so every rank takes part in it whatever
    // happened to it locally: two all-gathers — the size a rank fits (-1: none), then whether the smallest of them was
    // allocated — and every rank returns the same verdict.  (With a transport supplied by the embedding program, gvx.h
    // gvx_transport, these two gathers of 4 bytes per rank run through its all_gather at build time.)
    const bool agree = distributed && W > 1 && comm;
    auto release_pools = [&]() {
        for (Worker &w : workers) {
            hipSetDevice(w.device);
            hipFree(w.pool[0]), hipFree(w.pool[1]), hipFree(w.landing), hipFree(w.block_pools[0]), hipFree(w.block_pools[1]);
            w.pool[0] = w.pool[1] = w.landing = w.block_pools[0] = w.block_pools[1] = nullptr;
        }
    };
    auto try_allocate = [&]() {  // every pool at the current episode size, or none
        bool ok = true;
        const size_t bytes = (size_t)episode_size * batch_size * 8;
        for (Worker &w : workers) {
            hipSetDevice(w.device);
            ok = ok && hipMalloc(&w.pool[0], bytes) == hipSuccess && hipMalloc(&w.pool[1], bytes) == hipSuccess &&
                 hipMalloc(&w.landing, bytes) == hipSuccess;
            for (uint32_t *&pools : w.block_pools)
                ok = ok && (!in_hbm || hipMalloc(&pools, w.tails.size() * num_partition * bytes) == hipSuccess);
            if (!ok) break;
        }
        if (!ok) {
            (void)hipGetLastError();
            release_pools();
        }
        return ok;
    };
    auto smallest_of = [&](int32_t mine, int32_t *smallest) {  // collective: every rank enters it whatever happened to it locally; its staging buffer exists since prepare_devices
        Worker &w = workers[0];
        int32_t *sizes = w.agreement;
        std::vector<int32_t> host(W, 0);
        host[w.rank] = mine;
        int rc = hipSetDevice(w.device) == hipSuccess &&
                 hipMemcpyAsync(sizes, host.data(), (size_t)W * 4, hipMemcpyHostToDevice, w.exchange) == hipSuccess ? GVK_OK : GVK_EHIP;
        const int gathered = comm->all_gather({{w.rank, w.device, w.exchange}}, {sizes}, 4);  // (a transport whose all_gather fails must fail on every rank: gvx.h)
        if (rc == GVK_OK) rc = gathered;
        if (rc == GVK_OK && (hipMemcpyAsync(host.data(), sizes, (size_t)W * 4, hipMemcpyDeviceToHost, w.exchange) != hipSuccess ||
                             hipStreamSynchronize(w.exchange) != hipSuccess))
            rc = GVK_EHIP;
        if (rc != GVK_OK) {
            release_pools();  // nothing stays allocated on a rank that could not agree
            return gvk_fail(rc, "GraphSolver: the workers could not agree on an episode size (%s)", gvk_last_error());
        }
        *smallest = *std::min_element(host.begin(), host.end());
        return GVK_OK;
    };
    int local = GVK_OK;  // what this rank's own memory says
    while (!try_allocate()) {
        if (episode_size <= 1) {
            local = gvk_fail(GVK_ENOMEM, "Out of GPU memory. Try to reduce the size of your graph or the dimension of your embeddings.");
            break;
        }
        // the halved episode keeps what configure() checked: a whole number of shuffle bases per pool — per slice, when
        // every worker samples a slice of every pool
        const size_t base = (size_t)(config.augmentation_step > 1 ? std::max(config.shuffle_base, 1) : 1) * (routed() ? W : 1);
        int half = episode_size / 2;
        while (half > 1 && ((size_t)half * batch_size) % base) half--;
        if (((size_t)std::max(half, 1) * batch_size) % base) {
            local = gvk_fail(GVK_ENOMEM, "Out of GPU memory for an episode of %d batches, and no smaller episode is a multiple of "
                             "the shuffle base times the number of workers", episode_size);
            break;
        }
        log_message(1, "Fail to allocate GPU memory for episode size of %d. Use %d instead.", episode_size, std::max(half, 1));
        episode_size = std::max(half, 1);
        make_info();
    }
    if (!agree) {
        if (local != GVK_OK) return local;
    } else {
        const std::string mine = local != GVK_OK ? gvk_last_error() : "";
        int32_t smallest = 0;
        GVK_TRY(smallest_of(local == GVK_OK ? episode_size : -1, &smallest));
        bool ok = smallest > 0;
        if (ok && smallest < episode_size) {  // a rank with less free memory has halved further: its size is everyone's
            log_message(1, "Another worker fits an episode size of %d only. Use %d instead of %d.", smallest, smallest, episode_size);
            release_pools();
            episode_size = smallest;
            make_info();
            ok = try_allocate();
        }
        int32_t all_fit = 0;
        GVK_TRY(smallest_of(ok ? 1 : -1, &all_fit));
        if (smallest <= 0 || all_fit <= 0) {
            release_pools();
            if (local != GVK_OK) return gvk_fail(local, "%s", mine.c_str());
            return gvk_fail(GVK_ENOMEM, smallest <= 0 ? "Out of GPU memory on another worker" : "Out of GPU memory for the episode size of %d the workers agreed on", episode_size);
        }
    }
    for (Worker &w : workers) {
        HIP_TRY(hipSetDevice(w.device));
        const int row_bits = std::max(32 - __builtin_clz(std::max(part_rows, 2u) - 1), 1);
        const int parts = gvk_train_launches(batch_size, part_rows);
        GVK_TRY(gvk_group_pairs(nullptr, nullptr, nullptr, nullptr, &w.group_workspace_bytes, batch_size / parts,
                                episode_size * parts, row_bits));
        HIP_TRY(hipMalloc(&w.group_workspace, std::max<size_t>(w.group_workspace_bytes, 16)));
        for (uint32_t *pools : w.block_pools)  // never train what nothing wrote
            if (pools) HIP_TRY(hipMemsetAsync(pools, 0, w.tails.size() * num_partition * (size_t)episode_size * batch_size * 8, w.compute));
        if (hubs) {  // the chains' work lists and mirrors, sized for the largest block: hub_chunk batches at a time
            size_t need = 0;
            for (hub_chunk = kHubChunk;; hub_chunk /= 2) {
                need = 0;
                for (int hp = 0; hp < num_partition; hp++)
                    for (int tp = 0; tp < num_partition; tp++) {
                        if (hub_rows[hp] + hub_rows[tp] == 0) continue;
                        size_t bytes = 0;
                        GVK_TRY((hub_ahead() ? gvk_ahead_plan : gvk_hot_plan)(dim, batch_size, num_negative, hub_rows[hp], hub_rows[tp], hub_chunk,
                                                                              hub_parts_of(hp, tp), hub_chain_cap_request, &bytes));
                        need = std::max(need, bytes);
                    }
                if (need <= gpu_memory_limit / 32 || hub_chunk == 1) break;  // two workspaces
            }
            GVK_TRY(hub_workspace_for(w, need));
        }
    }
    return GVK_OK;
}

// ---- positive samples drawn on the device (GVX_DEVICE_SAMPLING) ------------------------------------------------------

namespace {

template <class T>
int to_device(T **out, const T *host, size_t count) {  // on the current device; *out is owned by the caller's Worker
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
    const int threads = num_sampler_per_worker * num_local();
    if (mode == GVS_MODE_EDGE) {
        of_block.assign((size_t)P * P, {});
        for (uint64_t e = 0; e < D; e++) of_block[(size_t)part[uv[2 * e]] * P + part[uv[2 * e + 1]]].push_back((uint32_t)e);
    } else {
        std::vector<float> prob(D);
        std::vector<uint32_t> alias(D);
        edge_table.resize(D), neighbor_table.resize(D);
        GVK_TRY(gvk_alias_build(weights, D, prob.data(), alias.data(), 4, edge_table.data()));
        GVK_TRY(gvs_graph_neighbor_tables(graph, threads + 1, neighbor_table.data()));
        if (biased) {  // out-neighbours ascending inside each vertex's CSR segment (the acceptance test searches them)
            const uint64_t *offsets = gvs_graph_flat_offsets(graph);
            sorted_neighbors.resize(D);
            for (uint64_t e = 0; e < D; e++) sorted_neighbors[e] = uv[2 * e + 1];
            for (uint32_t v = 0; v < num_vertex; v++)
                std::sort(sorted_neighbors.begin() + offsets[v], sorted_neighbors.begin() + offsets[v + 1]);
        }
    }
    for (size_t l = 0; l < workers.size(); l++) {
        Worker &w = workers[l];
        HIP_TRY(hipSetDevice(w.device));
        w.sample_seed = seed * 0x9E3779B1ull + 0x9E3779B97F4A7C15ull * (uint64_t)(w.rank + 1) + 0x706f73;
        w.sample_index = l < sample_positions.size() ? sample_positions[l] : 0;
        const size_t blocks = w.tails.size() * P;
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
        // one worker: the walks land in the block pools themselves; several: in this worker's slice of every block, laid
        // out for the all-to-all — [owner rank][the owner's blocks, tail-major][n / W]
        std::vector<uint64_t> offsets((size_t)P * P);
        const size_t n_slice = n / W, per_rank = (size_t)(P / W) * P;
        for (int hp = 0; hp < P; hp++)
            for (int tp = 0; tp < P; tp++) {
                const size_t owner = tp % W, ti = tp / W;  // worker i owns tails i, W + i, ... (gvs_schedule)
                offsets[(size_t)hp * P + tp] = W == 1 ? (ti * P + hp) * n : ((owner * per_rank + ti * P + hp) * n_slice);
            }
        GVK_TRY(to_device(&w.walk_offsets, offsets.data(), offsets.size()));
        HIP_TRY(hipMalloc(&w.walk_counters, (size_t)P * P * largest_divisor(n_slice, 256) * 4));
        HIP_TRY(hipMalloc(&w.walk_accept, (size_t)P * P * 4));
        w.walk_collects.assign((size_t)P * P, 1);  // every worker collects its slice of every block
    }
    return GVK_OK;
}

// The slices every worker drew of every block pool go to the workers that train the blocks: one all-to-all per episode
// on the sampling streams (its own communicator), then each block's W slices are laid side by side in the block's pool.
int gvx_solver::route_slices(int set) {
    const int P = num_partition, W = num_worker;
    const size_t n = (size_t)episode_size * batch_size, n_slice = n / W, per_rank = (size_t)(P / W) * P;
    std::vector<gvx::Peer> peers;
    std::vector<const void *> send;
    std::vector<void *> recv;
    for (Worker &w : workers) {
        peers.push_back({w.rank, w.device, w.sample});
        send.push_back(w.route_send), recv.push_back(w.route_recv);
    }
    GVK_TRY(route->all_to_all(peers, send, recv, per_rank * n_slice * 8));
    for (Worker &w : workers) {
        HIP_TRY(hipSetDevice(w.device));
        for (int q = 0; q < W; q++)  // rank q's slices of this worker's blocks: block i goes to pool i, W slices per pool
            HIP_TRY(hipMemcpy2DAsync(w.block_pools[set] + (size_t)q * n_slice * 2, n * 8,
                                     w.route_recv + (size_t)q * per_rank * n_slice * 2, n_slice * 8, n_slice * 8, per_rank,
                                     hipMemcpyDeviceToDevice, w.sample));
        HIP_TRY(hipEventRecord(w.filled[set], w.sample));
    }
    return GVK_OK;
}

// The pools of one episode (set 0 / 1) for every local worker, drawn by the GPUs on their sampling streams — what
// host_fill() does with CPU threads.  Edge mode: one gvk_sample_edges per block, nothing to wait for.  Walk modes: rounds
// of gvk_sample_walks_blocks until every stripe of every block is full (the host reads the counters between rounds — a
// handful of round trips, on a thread of its own while the GPUs train the episode before), then the slices are routed.
int gvx_solver::device_fill(int set) {
    Range range("Sample (device)");
    const int P = num_partition, W = num_worker, L = num_local();
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
    const int length = config.random_walk_length, aug = config.augmentation_step;
    const uint64_t per_walk = (uint64_t)aug * length - (uint64_t)aug * (aug - 1) / 2;
    if (P == 1) {  // one block: every walk owns its slots of the pool, no binning, no rounds
        Worker &w = workers[0];
        GVK_TRY(gvk_sample_walks(w.sample, &w.walk, w.sample_seed, w.sample_index, w.block_pools[set], n, length, aug,
                                 config.shuffle_base));
        w.sample_index += (n + per_walk - 1) / per_walk;
        HIP_TRY(hipEventRecord(w.filled[set], w.sample));
        return GVK_OK;
    }
    const int stripes = largest_divisor(n_slice, 256);
    const uint64_t stripe_capacity = n_slice / stripes, every = 64ull * stripes;  // the same number of wavefronts per stripe
    const size_t num_counter = (size_t)P * P * stripes;
    // The base of the sampler's stripe rule (pair i of a walk to stripe group i % base): the caller's shuffle base, but at least 8-16 groups.  With
    // the two groups of LINE's augmentation_step 2 / shuffle_base 2 the Friendster-like shape in 8 partitions ended +0.0020 above the reference's
    // loop (four seeds, SE 0.0002) where the CPU samplers' pools end +0.0011; with 8 / 16 / 40 / 80 groups +0.0012 / +0.0011 / +0.0011 / +0.0011
    // (profiles/r6/experiments/r6_fs_device_stripe_base.txt).  Youtube-size node2vec (shuffle base 1) does not see the base: 1 / 5 / 10 within 0.0004.
    const int stripe_base = std::max(std::max(config.shuffle_base, 1), largest_divisor(stripes, 16));
    // Blocks receive unequal shares of the walks' pairs, and a pool that is full drops what arrives later in the launch — with node2vec's
    // rejection sampling the late walks are those that rejected most, a selection the AUC sees (DESIGN.md section 7.11 f).  So: a first small
    // launch shows every block's share; from then on every block is THINNED (gvk_sample_walks_blocks_thinned: by a hash of walk and pair, not by
    // arrival) to the rate at which its pool fills together with the slowest one's, and only the last per cent of a pool is first come, first served.
    const uint64_t all_walks = ((n_slice * P * P + per_walk - 1) / per_walk / every + 1) * every;
    std::vector<uint64_t> walks(L, std::max<uint64_t>(all_walks / 8 / every, 1) * every), used(L, 0);
    std::vector<char> full(L, 0);
    std::vector<uint32_t> counters(num_counter);
    std::vector<std::vector<double>> share(L);
    std::vector<std::vector<float>> accept(L, std::vector<float>((size_t)P * P, 1.0f));
    for (Worker &w : workers) {
        HIP_TRY(hipSetDevice(w.device));
        HIP_TRY(hipMemsetAsync(w.walk_counters, 0, num_counter * 4, w.sample));
    }
    for (int round = 0;; round++) {
        if (round == 64) return gvk_fail(GVK_EINVAL, "device sampling: the block pools are not full after 64 rounds of walks");
        for (int l = 0; l < L; l++) {
            Worker &w = workers[l];
            if (full[l]) continue;
            HIP_TRY(hipSetDevice(w.device));
            if (round > 0) HIP_TRY(hipMemcpyAsync(w.walk_accept, accept[l].data(), (size_t)P * P * 4, hipMemcpyHostToDevice, w.sample));
            GVK_TRY(gvk_sample_walks_blocks_thinned(w.sample, &w.walk, w.walk_part, P, w.sample_seed, w.sample_index + used[l], walks[l],
                                                    W == 1 ? w.block_pools[set] : w.route_send, w.walk_offsets, w.walk_counters,
                                                    (uint32_t)n_slice, stripes, length, aug, stripe_base, round > 0 ? w.walk_accept : nullptr));
            used[l] += walks[l];
        }
        bool all_full = true;
        for (int l = 0; l < L; l++) {
            Worker &w = workers[l];
            if (full[l]) continue;
            HIP_TRY(hipSetDevice(w.device));
            HIP_TRY(hipMemcpyAsync(counters.data(), w.walk_counters, num_counter * 4, hipMemcpyDeviceToHost, w.sample));
            HIP_TRY(hipStreamSynchronize(w.sample));
            if (round == 0) {  // unthinned: the counters are the blocks' shares of everything the walks emit
                share[l].assign((size_t)P * P, 0.0);
                for (size_t i = 0; i < num_counter; i++) share[l][i / stripes] += (double)counters[i] / ((double)used[l] * per_walk);
            }
            // how many emitted pairs each block still needs at its share: the slowest sets the next launch, the others are thinned to its pace
            double longest = 0;
            std::vector<double> time_to_fill((size_t)P * P, 0.0);
            for (size_t blk = 0; blk < (size_t)P * P; blk++) {
                double missing = 0;
                bool collected = false;
                for (int k = 0; k < stripes; k++) {
                    const uint32_t have = counters[blk * stripes + k];
                    collected = collected || have > 0;
                    if (have < stripe_capacity) missing += (double)(stripe_capacity - have);
                }
                if (missing == 0) continue;
                if (!collected && share[l][blk] == 0) {  // a block this worker does not collect — or one that receives nothing at all
                    if (w.walk_collects[blk] && used[l] * per_walk > 64ull * n_slice * P * P)
                        return gvk_fail(GVK_EINVAL, "block (%zu, %zu) of the partition grid receives no random-walk pairs; use "
                                        "fewer partitions", blk / P, blk % P);
                    if (!w.walk_collects[blk]) continue;
                }
                time_to_fill[blk] = missing / std::max(share[l][blk], 1.0 / (64.0 * P * P * stripes));
                longest = std::max(longest, time_to_fill[blk]);
            }
            full[l] = longest == 0;
            for (size_t blk = 0; blk < (size_t)P * P; blk++) accept[l][blk] = longest > 0 ? (float)std::min(1.0, std::max(time_to_fill[blk] / longest, 0.0)) : 1.0f;
            walks[l] = ((uint64_t)(longest * 1.02 / per_walk) / every + 1) * every;
            all_full = all_full && full[l];
        }
        if (all_full) break;
    }
    for (int l = 0; l < L; l++) workers[l].sample_index += used[l];
    if (W > 1) return route_slices(set);
    HIP_TRY(hipSetDevice(workers[0].device));
    HIP_TRY(hipEventRecord(workers[0].filled[set], workers[0].sample));
    return GVK_OK;
}

// One [S][dim] table of one partition between the host array (global vertex order) and a device table, through a
// pinned staging buffer in chunks of at most 256 MiB; the row permutation is done by host threads.
int gvx_solver::move_table(bool to_device, Worker &w, float *device_table, std::vector<float> &host, int partition) {
    const std::vector<uint32_t> &ids = part_ids[partition];
    const size_t row_bytes = (size_t)dim * 4, chunk_rows = std::max<size_t>(kChunkBytes / row_bytes, 1);
    float *staging = nullptr;
    HIP_TRY(hipHostMalloc(&staging, std::min(chunk_rows, std::max<size_t>(ids.size(), 1)) * row_bytes, hipHostMallocDefault));
    const int threads = std::max(std::min(num_sampler_per_worker * num_local(), 16), 1);
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

// WorkerMixin::write_back, solver.h:1498-1504: every table once, from a worker that holds it.  One process per GPU: the
// context shards of the other ranks arrive by all-gather (one per owned tail slot: shard i of every rank), so that every
// rank's host arrays are complete.
int gvx_solver::write_back() {
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
    if (!distributed) {
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
    const int W = num_worker;
    float *all = nullptr;  // [W][1 + m][S][dim]: slot ti of every rank
    HIP_TRY(hipMalloc(&all, (size_t)W * slot_floats() * 4));
    int rc = GVK_OK;
    for (size_t ti = 0; ti < first.tails.size() && rc == GVK_OK; ti++) {
        hipMemcpyAsync(all + (size_t)first.rank * slot_floats(), context_table(first, (int)ti, 0), slot_floats() * 4,
                       hipMemcpyDeviceToDevice, first.exchange);
        rc = comm->all_gather({{first.rank, first.device, first.exchange}}, {all}, slot_floats() * 4);
        if (rc == GVK_OK && hipStreamSynchronize(first.exchange) != hipSuccess) rc = gvk_fail(GVK_EHIP, "write back: all-gather failed");
        for (int q = 0; q < W && rc == GVK_OK; q++) {
            const int tp = (int)ti * W + q;  // worker q owns tails q, W + q, ... (gvs_schedule)
            rc = move_table(false, first, all + ((size_t)q * (1 + num_moment)) * table_floats(), context, tp);
            for (int j = 0; j < num_moment && rc == GVK_OK; j++)
                rc = move_table(false, first, all + ((size_t)q * (1 + num_moment) + 1 + j) * table_floats(), context_moments[j], tp);
        }
    }
    hipFree(all);
    return rc;
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

// ---- pools -----------------------------------------------------------------------------------------------------------

int gvx_solver::allocate_host_sets() {
    if (device_sampling || !host_sets[0].empty()) return GVK_OK;
    const int P = num_partition;
    const size_t pool_bytes = (size_t)episode_size * batch_size * 8;
    for (auto &set : host_sets) {
        set.assign((size_t)P * P, nullptr);
        if (routed()) continue;  // this process samples slices (Worker::route_host), not whole pools
        for (Worker &w : workers)
            for (int tp : w.tails)
                for (int hp = 0; hp < P; hp++)
                    if (hipHostMalloc(&set[(size_t)hp * P + tp], pool_bytes, hipHostMallocDefault) != hipSuccess)
                        return gvk_fail(GVK_ENOMEM, "Out of host memory for the sample pools (%d x %d blocks of %s)", P, P,
                                        size_string((double)pool_bytes).c_str());
    }
    return GVK_OK;
}

// The CPU samplers (SamplerMixin::sample, solver.h:1012-1055; GraphSampler::sample_random_walk, graph.cuh:298-450) fill the
// pools of every block the LOCAL workers train.  Independent edge draws with several partitions: column by column, from
// an alias table over exactly the edges that end in the column's tail partition (the exact conditional distribution,
// nothing dropped; the reference draws from all edges and drops what does not fit).  Random walks yield pairs for every
// block: one process fills all P x P pools at once; with one process per GPU each process fills its 1 / W slice of every
// pool and the slices are routed (route_slices).
int gvx_solver::host_fill(int set) {
    Range range("Sample threads");  // solver.h:622
    const int P = num_partition, threads = num_sampler_per_worker * num_local();
    gvs_fill_config f{};
    f.mode = mode;
    f.num_thread = 4 * threads;  // 4 slices per OS thread: a descheduled thread delays a quarter-size slice
    f.sample_batch_size = config.random_walk_length * config.random_walk_batch_size;  // graph.cuh:791
    f.walk_length = config.random_walk_length, f.walk_batch = config.random_walk_batch_size;
    f.augmentation_step = config.augmentation_step, f.shuffle_base = config.shuffle_base;
    f.tail_partition = -1, f.os_threads = threads, f.cpu_offset = -1;
    const uint64_t n = (uint64_t)episode_size * batch_size;
    if (routed()) {  // this process's slice of every block pool, laid out for the all-to-all
        Worker &w = workers[0];
        const int W = num_worker;
        const size_t n_slice = n / W, per_rank = (size_t)(P / W) * P;
        std::vector<uint32_t *> slices((size_t)P * P);
        for (int hp = 0; hp < P; hp++)
            for (int tp = 0; tp < P; tp++)
                slices[(size_t)hp * P + tp] = w.route_host + (((size_t)(tp % W) * per_rank + (size_t)(tp / W) * P + hp) * n_slice) * 2;
        GVK_TRY(gvs_sampler_fill(sampler, slices.data(), n_slice, &f));
        HIP_TRY(hipSetDevice(w.device));
        for (Worker &u : workers)
            if (u.episode_end_valid) HIP_TRY(hipStreamWaitEvent(w.sample, u.episode_end, 0));
        HIP_TRY(hipMemcpyAsync(w.route_send, w.route_host, (size_t)P * P * n_slice * 8, hipMemcpyHostToDevice, w.sample));
        GVK_TRY(route_slices(set));
        HIP_TRY(hipStreamSynchronize(w.sample));  // the pinned slices may be refilled
        return GVK_OK;
    }
    if (mode == GVS_MODE_EDGE && P > 1) {
        for (Worker &w : workers)
            for (int tp : w.tails) {
                std::vector<uint32_t *> column((size_t)P * P, nullptr);
                for (int hp = 0; hp < P; hp++) column[(size_t)hp * P + tp] = host_sets[set][(size_t)hp * P + tp];
                f.tail_partition = tp;
                GVK_TRY(gvs_sampler_fill(sampler, column.data(), n, &f));
            }
    } else {
        GVK_TRY(gvs_sampler_fill(sampler, host_sets[set].data(), n, &f));
    }
    if (!resident_pools) return GVK_OK;
    for (Worker &w : workers) {  // a session that keeps its pool sets in HBM: one H2D copy per block, now
        HIP_TRY(hipSetDevice(w.device));
        for (Worker &u : workers)
            if (u.episode_end_valid) HIP_TRY(hipStreamWaitEvent(w.sample, u.episode_end, 0));
        for (int tp : w.tails)
            for (int hp = 0; hp < P; hp++)
                HIP_TRY(hipMemcpyAsync(block_pool(w, set, hp, tp), host_sets[set][(size_t)hp * P + tp], n * 8, hipMemcpyHostToDevice,
                                       w.sample));
        HIP_TRY(hipEventRecord(w.filled[set], w.sample));
    }
    for (Worker &w : workers) HIP_TRY(hipStreamSynchronize(w.sample));
    return GVK_OK;
}

int gvx_solver::fill(int set) { return device_sampling ? device_fill(set) : host_fill(set); }

// ---- one block -------------------------------------------------------------------------------------------------------

// WorkerMixin::train (solver.h:1511-1522): batches [first, first + count) of one block's pool on the worker's compute
// stream; batch ids interleave over the workers as the reference's shared atomic counter hands them out (solver.h:1520):
// `base` is the id of the block visit's first batch on worker 0.
// With hub rows trained by chains, the parts a batch of block (hp, tp) is trained as (GVX_HUB_PARTS; gvk.h `parts`).
bool gvx_solver::hub_rounds_of(int hp, int tp) const {
    if (!hubs || hub_rows[hp] + hub_rows[tp] == 0) return false;
    if (hub_rounds_request >= 0) return hub_rounds_request != 0;
    // The gradient steps of entries that work side by side from one state add up where the sequential loop's see each other; how much
    // that matters grows with the norms of the rows a hub row meets, and those grow with how large a share of ALL training the
    // largest hubs take — a property of the graph, not of the block (a block of the headline shape at P = 8 has a row that heads
    // 8 % of its samples, yet that row is trained no more often than at P = 1: one round is within 0.001 AUC there, rounds add
    // nothing).  Measured: one round of 16 x 16 entries stays with the reference's loop where the largest hub
    // takes 1 % of the graph's degree (the headline shape, any P) and overshoots it by 0.004-0.008 AUC where it takes 6 % (even at
    // 250 entries per part); rounds of four are within 0.001 on both.
    (void)hp, (void)tp;
    return hub_graph_share > kHubRoundShare;
}

// The two workspaces of a worker's chains hold at least `need` bytes each
int gvx_solver::hub_workspace_for(Worker &w, size_t need) {
    if (need <= w.hub_workspace_bytes) return GVK_OK;
    HIP_TRY(hipSetDevice(w.device));
    HIP_TRY(hipStreamSynchronize(w.compute));
    HIP_TRY(hipStreamSynchronize(w.lists));
    HIP_TRY(hipStreamSynchronize(w.chains));
    for (int b = 0; b < 2; b++) {
        hipFree(w.hub_workspaces[b]);
        w.hub_workspaces[b] = nullptr;
        w.lists_trained_valid[b] = false;
    }
    w.hub_workspace_bytes = 0;
    for (int b = 0; b < 2; b++) HIP_TRY(hipMalloc(&w.hub_workspaces[b], need));
    w.hub_workspace_bytes = need;
    return GVK_OK;
}

int gvx_solver::hub_parts_of(int hp, int tp) const {
    const int B = batch_size;
    const uint32_t kv = hubs ? hub_rows[hp] : 0, kc = hubs ? hub_rows[tp] : 0;
    if (kv + kc == 0) return 1;
    if (hub_parts_request > 0) return hub_parts_request;  // divides the batch size (configure)
    // so many parts that the largest hub row meets about kHubEntriesPerPart of its updates per part —
    // and, where every row is a hub row (a small table: many samples per row and batch), at least the parts
    // gvk_train_launches prescribes for it (§7.8: a chain then sees its partners at most a part old) —, a divisor of the
    // batch size, at most hub_max_parts (32; 50 for cache-resident tables)
    int want = std::max((std::max(hub_top_entries[hp], hub_top_entries[tp]) + kHubEntriesPerPart / 2) / kHubEntriesPerPart, 1);
    // ... and so many that no row the chains do not own is expected to be hit more than kHubHitsPerPart times per part
    want = std::max(want, (int)std::ceil(std::min(1e6, std::max(hub_next_hits[hp], hub_next_hits[tp]) / kHubHitsPerPart)));
    // ... and, for the walk-ordered pools of DeepWalk / node2vec (spread over the units: record i to unit i % units, so the pairs of one walk
    // never share a launch), so many that the rows the chains do not own collide as rarely inside a launch as they do on the headline
    // shape: the hit-weighted mean of their expected hits per batch / kWalkHitsPerPart.  Measured at Youtube size (DESIGN.md section 7.11 e): one
    // partition (0.33 hits) 8 / 16 / 32 parts -0.0014 / -0.0008 / -0.0001; four partitions (0.72 hits) 16 / 32 parts -0.0025 / -0.0009.
    // (Until round 6 this term was augmentation_step^2 + 1 = 26 -> 32 parts for every walk pool.)
    static const bool walk_term = !(getenv("GVX_WALK_PARTS_TERM") && !strcmp(getenv("GVX_WALK_PARTS_TERM"), "0"));  // measurement: the rule without this term
    if (spread && walk_term) want = std::max(want, (int)std::ceil(std::min(1e6, std::max(hub_rest_hits[hp], hub_rest_hits[tp]) / kWalkHitsPerPart)));
    // ... and, under a moment optimizer, as many as a batch may have: a hub row's chain is one sequential task whatever the parts (they cost its
    // launches 8 % at 32 against 8), and what the parts buy is fewer concurrent updates of the rows outside the chains — to which Adam, whose
    // every step has the same length, is more sensitive than SGD.  Measured on the headline shape, eight seeds (DESIGN.md section 7.11 b): Adam
    // 8 / 16 / 32 / 100 parts -0.0043 / -0.0034 / -0.0028 / -0.0024 from the reference's loop.
    if (optimizer.type != GVK_SGD) want = std::max(want, hub_max_parts);
    want = std::min(want, hub_max_parts);
    if (kv == part_rows && kc == part_rows) want = std::max(want, gvk_train_launches(B, part_rows));
    int parts = 1;
    for (int q = want; q <= 2 * want && parts == 1 && want > 1; q++)
        if (B % q == 0) parts = q;
    for (int q = want; q >= 2 && parts == 1; q--)
        if (B % q == 0) parts = q;
    return parts;
}

gvk_negative_source gvx_solver::negative_source(Worker &w, int tp) {
    const int ti = (int)tail_index(w, tp);
    gvk_negative_source neg{};
    neg.table = w.negative_tables[ti], neg.count = (uint32_t)part_ids[tp].size();
    neg.classes = w.negative_classes[ti], neg.class_count = w.negative_class_counts[ti];
    neg.seed = seed * 0x100000001B3ull + 0x100000001B3ull + (uint64_t)w.rank;
    return neg;
}

// The work lists of m batches of block (hp, tp) — the batches at `batches`, ids first_id, first_id + W, ... — on the lists stream into the
// workspace whose turn it is (two in rotation; it waits until the chunk that trained out of that workspace last has trained).
int gvx_solver::build_lists(Worker &w, int hp, int tp, const uint32_t *batches, uint64_t first_id, int m, int parts, bool ahead) {
    const int slot = (int)(w.chunks_built & 1);
    const uint32_t kv = hub_rows[hp], kc = hub_rows[tp];
    gvk_negative_source neg = negative_source(w, tp);
    if (w.lists_trained_valid[slot]) HIP_TRY(hipStreamWaitEvent(w.lists, w.lists_trained[slot], 0));
    if (hub_ahead())
        GVK_TRY(gvk_ahead_build(w.lists, dim, w.hub_workspaces[slot], w.hub_workspace_bytes, batches, batch_size, m, num_negative, &neg, (uint32_t)first_id,
                                (uint32_t)num_worker, kv, kc, parts, hub_chain_cap_request, hub_group_of(parts)));
    else {
        // Lists built AHEAD run beside the launches that train the chunk before them, and a visit's list workgroups at once slow seven of those
        // launches (headline shape: 20-41 us from 14.5, kernel trace of profiles/r6).  Building them a few units per launch (gvk_hot_build_sliced) was
        // measured and is SLOWER — 16 / 32 / 64 units per launch 723 / 775-782 / 779 against 789-813 M/s in one launch, 221 against 231 at the shard
        // size of an 8-GPU run (profiles/r6/experiments/r6_list_slice_ab.txt): the disturbance lasts as many launches longer as it is thinner.  One
        // launch stays; GVX_LIST_SLICE = units per launch is the measurement knob.
        const char *knob = getenv("GVX_LIST_SLICE");
        const int slice = ahead ? (knob ? std::max(atoi(knob), 0) : kHubListSlice) : 0;
        GVK_TRY(gvk_hot_build_sliced(w.lists, dim, w.hub_workspaces[slot], w.hub_workspace_bytes, batches, batch_size, m, num_negative, &neg, (uint32_t)first_id,
                                     (uint32_t)num_worker, kv, kc, parts, hub_chain_cap_request, slice));
    }
    HIP_TRY(hipEventRecord(w.lists_built[slot], w.lists));
    w.chunks_built++;
    return GVK_OK;
}

// Lists that were built ahead and will not be trained (the next call trained something else): their workspace is the next to be built into again.
void gvx_solver::discard_prefetched(Worker &w) {
    if (w.prefetched.valid && w.chunks_built > w.chunks_trained) w.chunks_built--;
    w.prefetched.valid = false;
}

// The lists of a visit's FIRST chunk depend on its pool and the ids of its batches, never on the embeddings, and a work list kernel is one
// workgroup's latency (about 100 us) however few batches it covers: built when the visit begins, it is time in which the GPU trains nothing
// (4 % of a 20-batch visit on the headline shape, 3 % of an 8-batch visit at the shard size of an 8-GPU run).  So once a visit's last
// launches are enqueued, the first chunk of the visit stage() announced last is built on the lists stream — beside those launches — under the
// assumption that the next call trains that visit from its first batch with the ids that follow this call's; train_block checks the assumption
// field by field and builds as before where it fails.  GVX_LISTS_PREFETCH=0: never (measurement).
int gvx_solver::prefetch_lists(Worker &w, uint64_t next_batch_id) {
    const char *knob = getenv("GVX_LISTS_PREFETCH");
    const bool enabled = !(knob && !strcmp(knob, "0"));
    if (!w.staged.valid) return GVK_OK;
    const Worker::Staged s = w.staged;
    w.staged.valid = false;
    if (!enabled || !hubs || w.prefetched.valid || next_batch_id >= num_batch) return GVK_OK;
    const uint32_t kv = hub_rows[s.hp], kc = hub_rows[s.tp];
    if (kv + kc == 0 || w.chunks_built != w.chunks_trained) return GVK_OK;
    const int W = num_worker, parts = hub_parts_of(s.hp, s.tp);
    size_t need = 0;
    GVK_TRY((hub_ahead() ? gvk_ahead_plan : gvk_hot_plan)(dim, batch_size, num_negative, kv, kc, hub_chunk, parts, hub_chain_cap_request, &need));
    if (need > w.hub_workspace_bytes) return GVK_OK;  // a larger block than any before it: its visit makes room (hub_workspace_for)
    const uint64_t first = next_batch_id + (uint64_t)w.rank;
    int n = 1;  // as train_block: up to, not including, the worker's next logging batch
    while (n < episode_size && (first + (uint64_t)n * W) % config.log_frequency) n++;
    const int m = std::min(optimizer.schedule == 2 ? 1 : hub_chunk, n);
    const uint32_t *pool = trained_pool(w, s.set, s.b, s.hp, s.tp);
    HIP_TRY(hipSetDevice(w.device));
    HIP_TRY(hipStreamWaitEvent(w.lists, w.uploaded[s.b], 0));
    if (w.block_pools[0]) HIP_TRY(hipStreamWaitEvent(w.lists, w.filled[s.set], 0));
    const void *workspace = w.hub_workspaces[w.chunks_built & 1];
    GVK_TRY(build_lists(w, s.hp, s.tp, pool, first, m, parts, true));
    w.prefetched.valid = true;
    w.prefetched.pool = pool, w.prefetched.workspace = workspace, w.prefetched.first = first;
    w.prefetched.m = m, w.prefetched.parts = parts, w.prefetched.cap = hub_chain_cap_request, w.prefetched.set = s.set, w.prefetched.b = s.b;
    w.prefetched.kv = kv, w.prefetched.kc = kc;
    return GVK_OK;
}

int gvx_solver::train_block(Worker &w, int hp, int tp, const uint32_t *pool, int first_batch, int count) {
    Range range("Train Batch");  // solver.h:1526 (one range per block: its batches are back-to-back launches)
    const int W = num_worker, r = w.rank, B = batch_size, nm = num_moment;
    const int ti = (int)tail_index(w, tp);
    gvk_tables t{};
    t.vertex = head_table(w, hp, 0), t.context = context_table(w, ti, 0);
    if (nm >= 1) t.vertex_moment1 = head_table(w, hp, 1), t.context_moment1 = context_table(w, ti, 1);
    if (nm >= 2) t.vertex_moment2 = head_table(w, hp, 2), t.context_moment2 = context_table(w, ti, 2);
    t.n_vertex = t.n_context = part_rows;
    t.flags = walk_ordered() ? GVK_PAIRS_OF_WALKS : 0;
    gvk_negative_source neg = negative_source(w, tp);
    gvk_optimizer o{};
    o.type = optimizer.type, o.lr = optimizer.lr, o.weight_decay = optimizer.weight_decay;
    o.hp0 = optimizer.hp0, o.hp1 = optimizer.hp1, o.epsilon = optimizer.epsilon;
    int done = first_batch;
    const int end = first_batch + count;
    while (done < end) {
        const uint64_t first = batch_id + (uint64_t)(done - first_batch) * W + r;
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
        while (n < end - done && (first + (uint64_t)n * W) % config.log_frequency) n++;
        const uint32_t kv = hubs ? hub_rows[hp] : 0, kc = hubs ? hub_rows[tp] : 0;
        if (kv + kc > 0) {
            // hub rows by chains: the work lists of up to kHubChunk batches, then their launches, on the same stream
            // a small table — every row a hub row, many samples per row and batch — is trained as the parts gvk_train_launches
            // prescribes for it (§7.8): a chain then sees its partners at most a part old
            const int parts = hub_parts_of(hp, tp);
            const int chain_cap = hub_chain_cap_request;
            // rounds (gvk.h GVK_HOT_ROUNDS): where the parts a block can be given leave its largest hub row more updates per part than
            // one round holds side by side — the parts rule aims at kHubEntriesPerPart = 250 = what 16 tasks of 16 entries hold; beyond
            // kHubRoundEntries the gradient steps of entries that start from one state add up past the reference's loop
            const bool rounds = hub_rounds_of(hp, tp);
            const int form = optimizer.type != GVK_SGD ? 0 : ((hub_lerp_request < 0 ? kHubLerp : hub_lerp_request) ? GVK_HOT_LERP : 0) | (rounds ? GVK_HOT_ROUNDS : 0);
            const bool ahead = hub_ahead();
            size_t need = 0;
            GVK_TRY((ahead ? gvk_ahead_plan : gvk_hot_plan)(dim, B, num_negative, kv, kc, hub_chunk, parts, chain_cap, &need));
            GVK_TRY(hub_workspace_for(w, need));  // first block, or a block with more hub rows / parts than any before it
            // a schedule computed by a callback (optimizer.h:132-134) gives every batch its learning rate on the host: a call per batch
            const int chunk = optimizer.schedule == 2 ? 1 : hub_chunk;
            // The work lists of a chunk depend on the pool and the draws' seed only, never on the embeddings: they are built on the
            // lists stream into one of two workspaces while the chunk before trains out of the other (the lists stream has waited for
            // the pool: train_step) — nothing but training launches on the compute stream.
            const int pair_launches = hub_pair_launches_request > 0 && parts % hub_pair_launches_request == 0 ? hub_pair_launches_request : 0;
            auto build = [&](int at) -> int {
                return build_lists(w, hp, tp, pool + (size_t)(done + at) * B * 2, first + (uint64_t)at * W, std::min(chunk, n - at), parts, at > 0);
            };
            // the first chunk may be there already: built while the visit before this one trained its last chunk (prefetch_lists)
            bool built = false;
            if (w.prefetched.valid) {
                const Worker::Prefetched &p = w.prefetched;
                built = p.pool == pool + (size_t)done * B * 2 && p.first == first && p.m == std::min(chunk, n) && p.parts == parts && p.cap == chain_cap &&
                        p.kv == kv && p.kc == kc && p.workspace == w.hub_workspaces[w.chunks_trained & 1] && w.chunks_built == w.chunks_trained + 1;
                if (built)
                    w.prefetched.valid = false, lists_prefetched++;
                else
                    discard_prefetched(w);
            }
            if (!built) GVK_TRY(build(0));
            for (int at = 0; at < n; at += chunk) {
                const int slot = (int)(w.chunks_trained & 1), m = std::min(chunk, n - at);
                const uint64_t id = first + (uint64_t)at * W;
                const uint32_t *batches = pool + (size_t)(done + at) * B * 2;
                gvk_optimizer ob = o;
                if (optimizer.schedule == 2) ob.lr = optimizer.lr * optimizer.schedule_function((int)id, (int)num_batch, optimizer.user);
                HIP_TRY(hipStreamWaitEvent(w.compute, w.lists_built[slot], 0));
                if (ahead)
                    GVK_TRY(gvk_train_episode_ahead(w.compute, w.chains, dim, &ob, optimizer.schedule == 1, &t, batches, &neg, (uint32_t)id, (uint32_t)W,
                                                    (uint32_t)num_batch, m, w.loss, B, num_negative, config.negative_weight, w.hub_workspaces[slot],
                                                    w.hub_workspace_bytes, kv, kc, m, parts, chain_cap, pair_launches, hub_group_of(parts),
                                                    rounds ? GVK_HOT_ROUNDS : 0));
                else
                    GVK_TRY(gvk_train_episode_hot(w.compute, dim, &ob, optimizer.schedule == 1, &t, batches, &neg, (uint32_t)id, (uint32_t)W,
                                                  (uint32_t)num_batch, m, w.loss, B, num_negative, config.negative_weight,
                                                  w.hub_workspaces[slot], w.hub_workspace_bytes, kv, kc, m, parts, chain_cap, form));
                HIP_TRY(hipEventRecord(w.lists_trained[slot], w.compute));
                w.lists_trained_valid[slot] = true;
                w.chunks_trained++;
                if (at + chunk < n) GVK_TRY(build(at + chunk));
            }
        } else if (optimizer.schedule != 2) {
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
    return GVK_OK;
}

// The pool of the worker's block at `step`, from pool set `set`, into device buffer b on the copy stream: the H2D copy
// (pools in pinned host memory) and / or the regrouping pass — per part of a batch, a part being what one launch trains
// (gvk_train_launches).  Pools that live in HBM and are not regrouped are trained in place: nothing to do.
int gvx_solver::stage(Worker &w, int step, int set, int b) {
    Range upload_range("Upload");
    const int P = num_partition;
    const size_t pool_elems = (size_t)episode_size * batch_size * 2;
    const int hp = block_of(step, w, 0), tp = block_of(step, w, 1);
    HIP_TRY(hipSetDevice(w.device));
    if (w.released_valid[b]) HIP_TRY(hipStreamWaitEvent(w.copy, w.released[b], 0));
    const uint32_t *source = w.landing;
    if (w.block_pools[0]) {  // already in HBM
        HIP_TRY(hipStreamWaitEvent(w.copy, w.filled[set], 0));
        source = block_pool(w, set, hp, tp);
    } else {
        uint32_t *target = reordered() ? w.landing : w.pool[b];
        HIP_TRY(hipMemcpyAsync(target, host_sets[set][(size_t)hp * P + tp], pool_elems * 4, hipMemcpyHostToDevice, w.copy));
        // copies that have landed need no event any more (a session stages block after block without the episode loop's drain)
        while (!w.copied.empty() && hipEventQuery(w.copied.front()) == hipSuccess) {
            hipEventDestroy(w.copied.front());
            w.copied.erase(w.copied.begin());
        }
        hipEvent_t copied;
        HIP_TRY(hipEventCreateWithFlags(&copied, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(copied, w.copy));
        w.copied.push_back(copied);
    }
    if (grouped) {
        Range regroup("Regroup");
        const int row_bits = std::max(32 - __builtin_clz(std::max(part_rows, 2u) - 1), 1);
        const int parts = gvk_train_launches(batch_size, part_rows);
        GVK_TRY(gvk_group_pairs(w.copy, source, w.pool[b], w.group_workspace, &w.group_workspace_bytes, batch_size / parts,
                                episode_size * parts, row_bits));
    } else if (spread) {  // consecutive records of a walk-ordered pool to consecutive launches (one per part of a batch)
        Range regroup("Spread");
        GVK_TRY(gvk_spread_pairs(w.copy, source, w.pool[b], (size_t)episode_size * batch_size, episode_size * hub_parts_of(hp, tp)));
    }
    HIP_TRY(hipEventRecord(w.uploaded[b], w.copy));
    if (w.prefetched.valid && w.prefetched.b == b) discard_prefetched(w);  // lists of what this buffer held
    w.staged.valid = true, w.staged.hp = hp, w.staged.tp = tp, w.staged.set = set, w.staged.b = b;
    return GVK_OK;
}

// ---- the exchange ----------------------------------------------------------------------------------------------------

int gvx_solver::wait_exchange(Worker &w, int group) {
    if (num_worker == 1 || streamed || !w.gathered_valid[group]) return GVK_OK;
    HIP_TRY(hipStreamWaitEvent(w.compute, w.gathered[group], 0));
    return GVK_OK;
}

// Before worker r trains head partition hp it makes sure hp sits in ITS slot of hp's head group (slot x W + r): at most
// one device-local copy of a shard — whatever partition sat in that slot is trained by another worker in this very step
// and comes back with the gather.  After the step the group's slab is [what rank 0 trained][what rank 1 trained]..., so
// the exchange is ONE in-place all-gather.  Every worker of the job keeps the same partition -> slot map.
int gvx_solver::claim_slots(int step) {
    const int W = num_worker;
    if (W == 1 || streamed) return GVK_OK;
    std::vector<int> heads(W);
    for (int q = 0; q < W; q++) heads[q] = schedule[((size_t)step * W + q) * 2];
    const int group = heads[0] / W;
    if (claimed[group] == heads) return GVK_OK;
    for (Worker &w : workers) {
        const int hp = heads[w.rank], target = group * W + w.rank, source = slot_of[hp];
        if (source == target) continue;
        HIP_TRY(hipSetDevice(w.device));
        HIP_TRY(hipMemcpyAsync(w.head + (size_t)target * slot_floats(), w.head + (size_t)source * slot_floats(), slot_floats() * 4,
                               hipMemcpyDeviceToDevice, w.compute));
    }
    for (int q = 0; q < W; q++) {
        slot_of[heads[q]] = group * W + q;
        part_at[group * W + q] = heads[q];
    }
    claimed[group] = heads;
    return GVK_OK;
}

// After a schedule step every worker has trained a different head partition of one head group, each in its own slot of
// the group's slab: ONE in-place all-gather — vertex rows and moment tables together — gives every worker the whole,
// current group again (RCCL over xGMI).  Asynchronous, on the exchange streams; the next block that reads the group
// waits for it (wait_exchange), blocks of other groups train meanwhile.
int gvx_solver::exchange(int step) {
    const int W = num_worker;
    if (W == 1 || streamed) return GVK_OK;
    Range range("Exchange");
    const int group = schedule[(size_t)step * W * 2] / W;
    std::vector<gvx::Peer> peers;
    std::vector<void *> slabs;
    for (Worker &w : workers) {
        HIP_TRY(hipSetDevice(w.device));
        HIP_TRY(hipStreamWaitEvent(w.exchange, w.trained, 0));
        peers.push_back({w.rank, w.device, w.exchange});
        slabs.push_back(w.head + (size_t)group * W * slot_floats());
    }
    GVK_TRY(comm->all_gather(peers, slabs, slot_floats() * 4));
    for (Worker &w : workers) {
        HIP_TRY(hipSetDevice(w.device));
        HIP_TRY(hipEventRecord(w.gathered[group], w.exchange));
        w.gathered_valid[group] = 1;
    }
    exchanged_bytes += slot_floats() * 4 * (uint64_t)(W - 1);
    exchanges++;
    return GVK_OK;
}

// Batches [first, first + count) of every local worker's block at `step`: wait for the staged pool and for the exchange
// the block depends on, bring the head partition to the worker's slot, stage the next visit's pool while this one trains.
int gvx_solver::train_step(int step, int set, int first, int count, bool stage_next, int next_step, int next_set) {
    const int W = num_worker;
    const int group = schedule[(size_t)step * W * 2] / W;
    for (Worker &w : workers) {
        HIP_TRY(hipSetDevice(w.device));
        GVK_TRY(wait_exchange(w, group));
    }
    GVK_TRY(claim_slots(step));
    for (Worker &w : workers) {
        const int hp = block_of(step, w, 0), tp = block_of(step, w, 1);
        const int b = (int)(w.visits & 1);
        HIP_TRY(hipSetDevice(w.device));
        HIP_TRY(hipStreamWaitEvent(w.compute, w.uploaded[b], 0));
        HIP_TRY(hipStreamWaitEvent(w.lists, w.uploaded[b], 0));  // the chains' work lists are built from the pool (train_block)
        if (w.block_pools[0]) {
            HIP_TRY(hipStreamWaitEvent(w.compute, w.filled[set], 0));
            HIP_TRY(hipStreamWaitEvent(w.lists, w.filled[set], 0));
        }
        if (stage_next) {
            w.visits++;
            const int rc = stage(w, next_step, next_set, (int)(w.visits & 1));
            w.visits--;
            if (rc != GVK_OK) return rc;
        }
        if (streamed) GVK_TRY(load_block(w, hp, tp));
        GVK_TRY(train_block(w, hp, tp, trained_pool(w, set, b, hp, tp), first, count));
        // the visit announced by stage() — unless it is this one — gets the lists of its first chunk now, beside this visit's last launches
        if (w.staged.valid && w.staged.b == b && w.staged.hp == hp && w.staged.tp == tp && w.staged.set == set) w.staged.valid = false;
        GVK_TRY(prefetch_lists(w, batch_id + (uint64_t)count * W));
        HIP_TRY(hipEventRecord(w.released[b], w.compute));
        w.released_valid[b] = true;
        HIP_TRY(hipEventRecord(w.trained, w.compute));
    }
    // streamed: the blocks of this step go back to the host tables (distinct head and tail partitions per worker)
    for (Worker &w : workers)
        if (streamed) GVK_TRY(store_block(w, block_of(step, w, 0), block_of(step, w, 1)));
    batch_id += (uint64_t)count * W;
    return GVK_OK;
}

// ---- episode loop ---------------------------------------------------------------------------------------------------

int gvx_solver::episode_loop() {
    const int W = num_worker;
    GVK_TRY(allocate_host_sets());
    const uint64_t per_episode = (uint64_t)num_step * episode_size * config.positive_reuse * W;
    // Routed walk pools end their fill with an all-to-all (route_slices).  Every collective of a rank — the exchange's
    // all-gathers and that all-to-all, whatever carries them (RCCL communicators, the embedding program's transport) — is
    // issued from THIS thread, in the same order on every rank: two communicators driven from two host threads may enqueue
    // their collectives in different orders on different ranks and deadlock.  Such a fill therefore runs here, after the
    // episode's last step has been enqueued (the GPUs train that episode meanwhile); every other fill has a thread of its own.
    const bool overlap_fill = !routed();
    int rc = fill(0);
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
        JoinOnExit join{filler};
        const bool more = batch_id + per_episode < num_batch;  // no pools for an episode that will not run
        if (more && overlap_fill) filler = std::thread([&, current]() { fill_rc = guarded("sampling", [&]() { return fill(current ^ 1); }); });
        for (Worker &w : workers) {
            if ((rc = stage(w, order[0], current, (int)(w.visits & 1))) != GVK_OK) break;
        }
        for (int k = 0; k < num_step && rc == GVK_OK; k++) {
            const int step = order[k];
            for (int reuse = 0; reuse < config.positive_reuse && rc == GVK_OK; reuse++)
                rc = train_step(step, current, 0, episode_size, reuse == config.positive_reuse - 1 && k + 1 < num_step,
                                k + 1 < num_step ? order[k + 1] : 0, current);
            for (Worker &w : workers) w.visits++;
            if (rc == GVK_OK) rc = exchange(step);
        }
        {
            Range wait("Wait for sample threads");  // solver.h:645
            if (filler.joinable()) filler.join();
        }
        if (rc == GVK_OK) rc = fill_rc;
        for (Worker &w : workers) {  // pool sets in HBM may be refilled once this episode has trained
            if (!w.block_pools[0] || rc != GVK_OK) break;
            hipSetDevice(w.device);
            hipEventRecord(w.episode_end, w.compute);
            w.episode_end_valid = true;
        }
        if (rc == GVK_OK && more && !overlap_fill) rc = fill(current ^ 1);
        current ^= 1;
    }
    for (Worker &w : workers) {
        hipSetDevice(w.device);
        hipDeviceSynchronize();
        for (hipEvent_t e : w.copied) hipEventDestroy(e);
        w.copied.clear();
    }
    return rc;
}

extern "C" int gvx_solver_train(gvx_solver *s, const gvx_train_config *config) {
    if (!s || !config) return gvk_fail(GVK_EINVAL, "gvx_solver_train: null solver / config");
    return guarded("GraphSolver.train", [&]() -> int {
        s->resident_pools = false;
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
    });
}

// ---- a training run step by step (gvx.h "session") -------------------------------------------------------------------

#define SESSION(s, what)                                                                            \
    if (!(s) || !(s)->session_open) return gvk_fail(GVK_EINVAL, "%s: no open session (gvx_session_open)", what)

extern "C" int gvx_session_open(gvx_solver *s, const gvx_train_config *config, int resident_pools) {
    if (!s || !config) return gvk_fail(GVK_EINVAL, "gvx_session_open: null solver / config");
    return guarded("gvx_session_open", [&]() -> int {
        s->resident_pools = resident_pools != 0;
        GVK_TRY(s->configure(*config));
        if (s->streamed) return gvk_fail(GVK_EINVAL, "a session needs the tables resident in GPU memory");
        GVK_TRY(s->prepare_devices());
        GVK_TRY(s->upload());
        GVK_TRY(s->allocate_host_sets());
        s->session_open = true;
        return GVK_OK;
    });
}

extern "C" int gvx_session_steps(gvx_solver *s) { return s && s->session_open ? s->num_step : 0; }

extern "C" int gvx_session_block(gvx_solver *s, int step, int worker, int *head_partition, int *tail_partition) {
    SESSION(s, "gvx_session_block");
    if (step < 0 || step >= s->num_step || worker < 0 || worker >= s->num_local())
        return gvk_fail(GVK_EINVAL, "gvx_session_block: step %d / worker %d out of range", step, worker);
    const Worker &w = s->workers[worker];
    if (head_partition) *head_partition = s->block_of(s->order[step], w, 0);
    if (tail_partition) *tail_partition = s->block_of(s->order[step], w, 1);
    return GVK_OK;
}

extern "C" int gvx_session_fill(gvx_solver *s, int set) {
    SESSION(s, "gvx_session_fill");
    if (set != 0 && set != 1) return gvk_fail(GVK_EINVAL, "gvx_session_fill: set must be 0 or 1");
    return guarded("gvx_session_fill", [&]() {
        for (Worker &w : s->workers) {  // lists built ahead from the pools this call overwrites
            if (w.prefetched.valid && w.prefetched.set == set) s->discard_prefetched(w);
            if (w.staged.set == set) w.staged.valid = false;
        }
        return s->fill(set);
    });
}

extern "C" int gvx_session_stage(gvx_solver *s, int step, int set, int buffer) {
    SESSION(s, "gvx_session_stage");
    if (step < 0 || step >= s->num_step || (set | buffer) < 0 || set > 1 || buffer > 1)
        return gvk_fail(GVK_EINVAL, "gvx_session_stage: step %d, set %d, buffer %d", step, set, buffer);
    return guarded("gvx_session_stage", [&]() -> int {
        for (Worker &w : s->workers) GVK_TRY(s->stage(w, s->order[step], set, buffer));
        return GVK_OK;
    });
}

extern "C" int gvx_session_train(gvx_solver *s, int step, int set, int buffer, int first, int count) {
    SESSION(s, "gvx_session_train");
    if (step < 0 || step >= s->num_step || (set | buffer) < 0 || set > 1 || buffer > 1 || first < 0 || count < 0 ||
        first + count > s->episode_size)
        return gvk_fail(GVK_EINVAL, "gvx_session_train: step %d, set %d, buffer %d, batches [%d, %d) of %d", step, set, buffer,
                        first, first + count, s->episode_size);
    return guarded("gvx_session_train", [&]() -> int {
        for (Worker &w : s->workers) w.visits = (uint64_t)buffer;  // the buffer this visit reads
        return s->train_step(s->order[step], set, first, count, false, 0, 0);
    });
}

extern "C" int gvx_session_exchange(gvx_solver *s, int step) {
    SESSION(s, "gvx_session_exchange");
    if (step < 0 || step >= s->num_step) return gvk_fail(GVK_EINVAL, "gvx_session_exchange: step %d out of range", step);
    return guarded("gvx_session_exchange", [&]() { return s->exchange(s->order[step]); });
}

extern "C" int gvx_session_wait(gvx_solver *s) {
    SESSION(s, "gvx_session_wait");
    for (Worker &w : s->workers) {
        HIP_TRY(hipSetDevice(w.device));
        for (size_t g = 0; g < w.gathered.size(); g++) GVK_TRY(s->wait_exchange(w, (int)g));
    }
    return GVK_OK;
}

extern "C" int gvx_session_synchronize(gvx_solver *s) {
    SESSION(s, "gvx_session_synchronize");
    for (Worker &w : s->workers) {
        HIP_TRY(hipSetDevice(w.device));
        HIP_TRY(hipDeviceSynchronize());
    }
    return GVK_OK;
}

extern "C" void *gvx_session_stream(gvx_solver *s, int worker) {
    if (!s || !s->session_open || worker < 0 || worker >= s->num_local()) return nullptr;
    return s->workers[worker].compute;
}

extern "C" int gvx_session_loss(gvx_solver *s, int worker, float *mean_loss) {
    SESSION(s, "gvx_session_loss");
    if (worker < 0 || worker >= s->num_local() || !mean_loss) return gvk_fail(GVK_EINVAL, "gvx_session_loss: bad argument");
    return guarded("gvx_session_loss", [&]() -> int {
        Worker &w = s->workers[worker];
        std::vector<float> host((size_t)s->batch_size);
        HIP_TRY(hipSetDevice(w.device));
        HIP_TRY(hipMemcpyAsync(host.data(), w.loss, host.size() * 4, hipMemcpyDeviceToHost, w.compute));
        HIP_TRY(hipStreamSynchronize(w.compute));
        double sum = 0;
        for (float x : host) sum += x;
        *mean_loss = (float)(sum / (double)host.size());
        return GVK_OK;
    });
}

extern "C" int gvx_session_probe(gvx_solver *s, int step, int set, int buffer, int launches, float *ms_per_launch) {
    SESSION(s, "gvx_session_probe");
    if (step < 0 || step >= s->num_step || launches < 1 || !ms_per_launch || s->num_negative != 1 || s->num_moment != 0)
        return gvk_fail(GVK_EINVAL, "gvx_session_probe: needs SGD with one negative, a valid step and launches >= 1");
    return guarded("gvx_session_probe", [&]() -> int {
        Worker &w = s->workers[0];
        const int hp = s->block_of(s->order[step], w, 0), tp = s->block_of(s->order[step], w, 1), B = s->batch_size;
        const int ti = (int)s->tail_index(w, tp), batches = std::min(s->episode_size, launches);
        HIP_TRY(hipSetDevice(w.device));
        HIP_TRY(hipStreamWaitEvent(w.compute, w.uploaded[buffer], 0));
        if (w.block_pools[0]) HIP_TRY(hipStreamWaitEvent(w.compute, w.filled[set], 0));
        const uint32_t *pool = s->trained_pool(w, set, buffer, hp, tp);
        uint32_t *negatives = nullptr;
        HIP_TRY(hipMalloc(&negatives, (size_t)batches * B * 4));
        hipEvent_t e0 = nullptr, e1 = nullptr;
        auto done = [&](int rc) {
            hipFree(negatives);
            if (e0) hipEventDestroy(e0);
            if (e1) hipEventDestroy(e1);
            return rc;
        };
        for (int b = 0; b < batches; b++) {  // fresh negatives per launch, drawn as the training kernel draws them
            const int rc = w.negative_classes[ti]
                               ? gvk_negative_draw_classes(w.compute, w.negative_classes[ti], w.negative_class_counts[ti], 0x51ed, b,
                                                           negatives + (size_t)b * B, B, 1)
                               : gvk_negative_draw(w.compute, w.negative_tables[ti], (uint32_t)s->part_ids[tp].size(), 0x51ed, b,
                                                   negatives + (size_t)b * B, B, 1);
            if (rc != GVK_OK) return done(rc);
        }
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return done(gvk_fail(GVK_EHIP, "probe: events"));
        for (int pass = 0; pass < 2; pass++) {  // an untimed sweep, then the timed one
            if (pass == 1) hipEventRecord(e0, w.compute);
            for (int i = 0; i < launches; i++) {
                const int b = i % batches;
                const int rc = gvk_probe_row_traffic(w.compute, s->dim, s->head_table(w, hp, 0), s->context_table(w, ti, 0),
                                                     pool + (size_t)b * B * 2, negatives + (size_t)b * B, 0.0f, B);
                if (rc != GVK_OK) return done(rc);
            }
        }
        hipEventRecord(e1, w.compute);
        if (hipEventSynchronize(e1) != hipSuccess) return done(gvk_fail(GVK_EHIP, "probe: synchronize"));
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        *ms_per_launch = ms / launches;
        return done(GVK_OK);
    });
}

extern "C" int gvx_session_exchange_stats(gvx_solver *s, uint64_t *bytes_sent_per_worker, uint64_t *exchanges) {
    if (!s) return gvk_fail(GVK_EINVAL, "gvx_session_exchange_stats: null solver");
    if (bytes_sent_per_worker) *bytes_sent_per_worker = s->exchanged_bytes;
    if (exchanges) *exchanges = s->exchanges;
    return GVK_OK;
}

extern "C" int gvx_session_close(gvx_solver *s) {
    SESSION(s, "gvx_session_close");
    return guarded("gvx_session_close", [&]() -> int {
        const int rc = s->write_back();
        s->release_device();
        return rc;
    });
}

extern "C" int gvx_solver_predict(gvx_solver *s, const int64_t *samples, size_t n, float *logits) {
    if (!s || !s->graph) return gvk_fail(GVK_EINVAL, "The model must be built on a graph first");
    if (n == 0) return GVK_OK;
    if (!samples || !logits) return gvk_fail(GVK_EINVAL, "gvx_solver_predict: null pointer");
    if (s->session_open)  // the tables live in HBM until the session closes: the host tables are those of before the session
        return gvk_fail(GVK_EINVAL, "predict: a training session is open; close it first (its tables are written back then)");
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
    if (hipMemcpy(dv, s->vertex.data(), table, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dc, s->context.data(), table, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dp, records.data(), n * 8, hipMemcpyHostToDevice) != hipSuccess)
        return done(gvk_fail(GVK_EHIP, "predict: copy to the GPU failed"));
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
    out->rank = s->first_rank, out->num_local_worker = s->num_local();
    out->pair_order = s->grouped ? 2 : (s->spread ? 3 : 1);
    out->sampler_mode = s->mode, out->device_sampling = s->device_sampling;
    out->partition_rows = s->part_rows;
    out->transport = s->transport_name.c_str();
    out->hub_rows = s->hubs && !s->hub_rows.empty() ? *std::max_element(s->hub_rows.begin(), s->hub_rows.end()) : 0;
    out->hub_parts = 0, out->hub_rounds = 0;
    if (out->hub_rows)
        for (int hp = 0; hp < s->num_partition; hp++)
            for (int tp = 0; tp < s->num_partition; tp++) {
                out->hub_parts = std::max(out->hub_parts, s->hub_parts_of(hp, tp));
                out->hub_rounds = out->hub_rounds || s->hub_rounds_of(hp, tp);
            }
    out->hub_lerp = out->hub_rows ? (s->hub_lerp_request < 0 ? kHubLerp : s->hub_lerp_request) : 0;
    out->lists_prefetched = (uint32_t)s->lists_prefetched;
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
