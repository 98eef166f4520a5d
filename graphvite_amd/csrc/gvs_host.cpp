// gvs_host.cpp — host runtime behind include/gvs.h: graph store, partition / schedule, and the multi-threaded
// CPU samplers (edge, random walk, node2vec) that fill the episode sample pools the HIP kernels consume.
//
// Design notes (what differs from the reference's host code, and why):
//   * the graph is kept as an insertion-ordered directed edge list while loading and turned into CSR by one
//     stable counting sort — the same neighbour order as the reference's vector<vector<>> adjacency lists
//     (include/instance/graph.cuh:124-153, include/core/graph.h:87-101) without 1M small heap blocks;
//   * (partition, local id) of a vertex is one packed 8-byte word, and an edge-table slot is one 16-byte
//     {prob, alias} record, so a positive sample costs 4 random DRAM lines instead of 7;
//   * uniforms come from a counter-based Philox stream per sampler thread (include/gvk.h "RNG contract")
//     instead of 40 MB cuRAND buffers copied back from the GPU (include/core/solver.h:943-967,1015-1016);
//   * per-vertex / per-edge alias tables live in two flat CSR-aligned arrays, not in one AliasTable object
//     (two heap blocks) per vertex or per edge (include/instance/graph.cuh:645-677).
//   * every sampler stage (alias slot -> edge -> locations; per walk step: table slot -> edge -> location) runs over
//     a whole inner round with the next stage's cache lines prefetched: tens of DRAM misses in flight per thread
//     instead of one dependent chain at a time (5x on the edge sampler end to end).
// Pool slots are written in the reference's order; uniforms are consumed in the reference's order by the edge sampler
// and in lockstep order (see fill_walks) by the walk samplers.

#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <immintrin.h>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "gvk.h"
#include "gvk_internal.h"
#include "gvs.h"

// Big randomly-accessed tables (edge slots, CSR arrays, locations) live in 2 MiB pages where the kernel grants them:
// with 4 KiB pages every draw is also a TLB miss, and the kernel's periodic page scans (NUMA balancing) turn a
// 1.3 GB table into hundreds of thousands of hinting faults — observed as every other fill running 10x slower.
template <class T>
struct HugeAllocator {
    typedef T value_type;
    static constexpr size_t kHuge = (size_t)2 << 20, kThreshold = (size_t)4 << 20;
    HugeAllocator() = default;
    template <class U>
    HugeAllocator(const HugeAllocator<U> &) {}
    T *allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes < kThreshold) {
            void *p = malloc(bytes ? bytes : 1);
            if (!p) throw std::bad_alloc();
            return static_cast<T *>(p);
        }
        const size_t rounded = (bytes + kHuge - 1) / kHuge * kHuge;
        void *p = mmap(nullptr, rounded, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) throw std::bad_alloc();
        static const bool huge = getenv("GVS_NO_HUGEPAGE") == nullptr;
        if (huge) madvise(p, rounded, MADV_HUGEPAGE);
        return static_cast<T *>(p);
    }
    void deallocate(T *p, size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes < kThreshold)
            free(p);
        else
            munmap(p, (bytes + kHuge - 1) / kHuge * kHuge);
    }
    template <class U>
    bool operator==(const HugeAllocator<U> &) const { return true; }
    template <class U>
    bool operator!=(const HugeAllocator<U> &) const { return false; }
};
template <class T>
using HugeVector = std::vector<T, HugeAllocator<T>>;

// The same storage for arrays whose every element is written right after they are sized: resize(n) leaves the new
// elements uninitialized instead of zeroing them on one thread, so the threads that fill the array are also the ones
// that first touch (fault in) its pages.
template <class T>
struct RawHugeAllocator : HugeAllocator<T> {
    typedef T value_type;
    template <class U>
    struct rebind {
        typedef RawHugeAllocator<U> other;
    };
    RawHugeAllocator() = default;
    template <class U>
    RawHugeAllocator(const RawHugeAllocator<U> &) {}
    template <class U>
    void construct(U *p) {
        ::new ((void *)p) U;
    }
    template <class U, class... Args>
    void construct(U *p, Args &&...args) {
        ::new ((void *)p) U(std::forward<Args>(args)...);
    }
};
template <class T>
using RawHugeVector = std::vector<T, RawHugeAllocator<T>>;

// ---- graph ---------------------------------------------------------------------------------------------

// name -> id for the text loaders.  A billion-edge list is two lookups per line into a table of up to 10^8 names, each
// a cache miss: std::unordered_map<std::string, ...> pays three or four of them (bucket, node, key bytes — 0.8 us per
// lookup measured on 1M names) plus a temporary std::string.  Here a probe touches ONE 32-byte slot that holds the
// 64-bit hash, the id and the name itself when it has at most 18 bytes (node names almost always do); longer names
// are compared against id2name.  Linear probing, power-of-two capacity, at most half full.
class NameTable {
public:
    static constexpr uint32_t kNone = 0xffffffffu;

    void clear() {
        decltype(slots_)().swap(slots_);
        count_ = 0;
    }
    size_t size() const { return count_; }

    static uint64_t hash_of(const char *name, size_t len) { return hash(name, len); }
    // start fetching the slot a name with this hash would probe first (the table may still grow before the lookup —
    // then the prefetch was for nothing, which is harmless)
    void prefetch(uint64_t h) const {
        if (!slots_.empty()) __builtin_prefetch(&slots_[h & (slots_.size() - 1)]);
    }

    uint32_t find(const char *name, size_t len, const std::vector<std::string> &id2name) const {
        return find(name, len, hash(name, len), id2name);
    }
    uint32_t find(const char *name, size_t len, uint64_t h, const std::vector<std::string> &id2name) const {
        if (slots_.empty()) return kNone;
        for (size_t i = h & (slots_.size() - 1);; i = (i + 1) & (slots_.size() - 1)) {
            const Slot &s = slots_[i];
            if (s.id == kNone) return kNone;
            if (s.hash == h && equal(s, name, len, id2name)) return s.id;
        }
    }

    // name must not be present
    void insert(const char *name, size_t len, uint32_t id) { insert(name, len, hash(name, len), id); }
    void insert(const char *name, size_t len, uint64_t h, uint32_t id) {
        if ((count_ + 1) * 2 > slots_.size()) grow();
        Slot s;
        s.hash = h;
        s.id = id;
        s.len = (uint8_t)(len <= kInline ? len : 255);
        memset(s.text, 0, sizeof(s.text));
        if (len <= kInline) memcpy(s.text, name, len);
        place(s);
        count_++;
    }

private:
    static constexpr size_t kInline = 18;
    struct Slot {
        uint64_t hash;
        uint32_t id = kNone;
        uint8_t len;
        char text[kInline + 1];
    };
    static_assert(sizeof(Slot) == 32, "one slot = half a cache line");

    HugeVector<Slot> slots_;
    size_t count_ = 0;

    static uint64_t hash(const char *p, size_t len) {  // 8 bytes at a time, multiply-fold (wyhash-style mixing)
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (len * 0xff51afd7ed558ccdull);
        while (len >= 8) {
            uint64_t k;
            memcpy(&k, p, 8);
            h = mix(h ^ k);
            p += 8, len -= 8;
        }
        uint64_t k = 0;
        memcpy(&k, p, len);
        return mix(h ^ k ^ ((uint64_t)len << 56));
    }
    static uint64_t mix(uint64_t x) {
        const __uint128_t m = (__uint128_t)x * 0xD6E8FEB86659FD93ull;
        return (uint64_t)m ^ (uint64_t)(m >> 64);
    }
    static bool equal(const Slot &s, const char *name, size_t len, const std::vector<std::string> &id2name) {
        if (s.len != 255) return s.len == len && memcmp(s.text, name, len) == 0;
        const std::string &full = id2name[s.id];
        return full.size() == len && memcmp(full.data(), name, len) == 0;
    }
    void place(const Slot &s) {
        size_t i = s.hash & (slots_.size() - 1);
        while (slots_[i].id != kNone) i = (i + 1) & (slots_.size() - 1);
        slots_[i] = s;
    }
    void grow() {
        HugeVector<Slot> old;
        old.swap(slots_);
        slots_.resize(old.empty() ? 1024 : old.size() * 2);
        for (const Slot &s : old)
            if (s.id != kNone) place(s);
    }
};

struct gvs_graph {
    uint32_t num_vertex = 0;
    uint64_t num_edge = 0;
    bool as_undirected = true, normalization = false;
    bool label_mode = false;
    NameTable name2id;
    std::vector<std::string> id2name;
    std::unordered_map<uint32_t, uint32_t> label2id;
    std::vector<uint32_t> labels;
    // insertion-ordered directed edges (only while loading)
    std::vector<uint32_t> src, dst;
    std::vector<float> w;
    // flattened
    std::vector<float> vertex_weights;
    RawHugeVector<uint32_t> edges_uv;   // both filled, every element, by finalize()
    RawHugeVector<float> edge_weights;
    HugeVector<uint64_t> flat_offsets;

    void clear() {
        num_vertex = 0;
        num_edge = 0;
        label_mode = false;
        name2id.clear();
        decltype(id2name)().swap(id2name);
        decltype(label2id)().swap(label2id);
        decltype(dense_label2id)().swap(dense_label2id);
        decltype(labels)().swap(labels);
        decltype(src)().swap(src);
        decltype(dst)().swap(dst);
        decltype(w)().swap(w);
        decltype(vertex_weights)().swap(vertex_weights);
        decltype(edges_uv)().swap(edges_uv);
        decltype(edge_weights)().swap(edge_weights);
        decltype(flat_offsets)().swap(flat_offsets);
    }

    uint32_t id_of_name(const char *name) {
        const size_t len = strlen(name);
        return id_of_name(name, len, NameTable::hash_of(name, len));
    }
    uint32_t id_of_name(const char *name, size_t len, uint64_t hash) {
        const uint32_t known = name2id.find(name, len, hash, id2name);
        if (known != NameTable::kNone) return known;
        uint32_t id = num_vertex++;
        id2name.emplace_back(name, len);
        name2id.insert(name, len, hash, id);
        vertex_weights.push_back(0);
        return id;
    }

    // labels below 2^28 are resolved through a flat table (4 B per possible label, grown on demand): integer edge
    // lists are usually dense, and a hash lookup per endpoint dominates the load time of a billion-edge list
    static constexpr uint32_t kDenseLabels = 1u << 28, kNoId = 0xffffffffu;
    std::vector<uint32_t> dense_label2id;

    uint32_t id_of_label(uint32_t label) {
        if (label < kDenseLabels) {
            if (label >= dense_label2id.size())
                dense_label2id.resize(std::max<size_t>((size_t)label + 1, dense_label2id.size() * 2), kNoId);
            uint32_t &slot = dense_label2id[label];
            if (slot == kNoId) {
                slot = num_vertex++;
                labels.push_back(label);
                vertex_weights.push_back(0);
            }
            return slot;
        }
        auto it = label2id.find(label);
        if (it != label2id.end()) return it->second;
        uint32_t id = num_vertex++;
        label2id.emplace(label, id);
        labels.push_back(label);
        vertex_weights.push_back(0);
        return id;
    }

    // Graph::add_edge, graph.cuh:124-153 (ids already resolved, u before v)
    void add_edge(uint32_t u, uint32_t v, float weight) {
        src.push_back(u);
        dst.push_back(v);
        w.push_back(weight);
        vertex_weights[u] += weight;
        if (as_undirected && u != v) {
            src.push_back(v);
            dst.push_back(u);
            w.push_back(weight);
            vertex_weights[v] += weight;
        }
        num_edge++;
    }

    // GraphMixin::flatten (core/graph.h:87-101) + Graph::normalize (graph.cuh:103-121).  The flattening is a stable
    // counting sort of the directed edges by source; its scatter runs on `threads` host threads that each own a range
    // of SOURCE VERTICES (equal shares of the edges): every thread scans the whole edge list in order and places the
    // edges of its range, so a vertex's edges keep their insertion order, nobody shares a cursor, and a thread's
    // random writes stay inside its own slice of the output.
    void finalize(int threads = 0) {
        const size_t D = src.size();
        const auto t_begin = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (getenv("GVS_TIMING"))
                fprintf(stderr, "[gvs] finalize: %s at %.3f s\n", what,
                        std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
        };
        if (threads <= 0) threads = gvk_cpu_budget();
        const size_t T = (size_t)std::max(1, std::min<int>(threads, (int)std::max<size_t>(D >> 18, 1)));
        flat_offsets.assign((size_t)num_vertex + 1, 0);
        for (size_t e = 0; e < D; e++) flat_offsets[src[e] + 1]++;  // (sequential reads: as fast on one thread as on many)
        lap("histogram");
        for (uint32_t u = 0; u < num_vertex; u++) flat_offsets[u + 1] += flat_offsets[u];
        edges_uv.resize(2 * D);
        edge_weights.resize(D);
        std::vector<uint64_t> cursor(flat_offsets.begin(), flat_offsets.end() - 1);
        lap("allocation");
        {  // scatter: vertex ranges holding equal shares of the edges
            std::vector<uint32_t> bound(T + 1, num_vertex);
            bound[0] = 0;
            for (size_t t = 1; t < T; t++)
                bound[t] = (uint32_t)(std::lower_bound(flat_offsets.begin(), flat_offsets.end() - 1, (uint64_t)(D / T * t)) -
                                      flat_offsets.begin());
            std::vector<std::thread> pool;
            for (size_t t = 0; t < T; t++)
                pool.emplace_back([&, t]() {
                    const uint32_t lo = bound[t], hi = bound[t + 1];
                    if (lo >= hi) return;
                    for (size_t e = 0; e < D; e++) {
                        const uint32_t u = src[e];
                        if (u < lo || u >= hi) continue;
                        const uint64_t slot = cursor[u]++;
                        edges_uv[2 * slot] = u;
                        edges_uv[2 * slot + 1] = dst[e];
                        edge_weights[slot] = w[e];
                    }
                });
            for (auto &th : pool) th.join();
        }
        lap("scatter");
        decltype(src)().swap(src);
        decltype(dst)().swap(dst);
        decltype(w)().swap(w);
        if (normalization) {
            std::vector<float> context_weights(num_vertex, 0.f);
            for (size_t e = 0; e < D; e++) context_weights[edges_uv[2 * e + 1]] += edge_weights[e];
            for (uint32_t u = 0; u < num_vertex; u++) {
                float weight = 0;
                for (uint64_t e = flat_offsets[u]; e < flat_offsets[u + 1]; e++) {
                    edge_weights[e] /= sqrtf(vertex_weights[u] * context_weights[edges_uv[2 * e + 1]]);
                    weight += edge_weights[e];
                }
                vertex_weights[u] = weight;
            }
        }
    }
};

namespace {

template <class F>
int guarded(const char *what, F &&f) {
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return gvk_fail(GVK_ENOMEM, "%s: out of host memory", what);
    } catch (const std::exception &e) {
        return gvk_fail(GVK_EINVAL, "%s: %s", what, e.what());
    }
}

// strtok-compatible tokenizer over a mutable line
char *next_token(char **cursor, const char *delimiters) {
    char *p = *cursor;
    p += strspn(p, delimiters);
    if (!*p) {
        *cursor = p;
        return nullptr;
    }
    char *end = p + strcspn(p, delimiters);
    if (*end) *end++ = 0;
    *cursor = end;
    return p;
}

}  // namespace

extern "C" {

gvs_graph *gvs_graph_create(void) { return new (std::nothrow) gvs_graph(); }

void gvs_graph_destroy(gvs_graph *g) { delete g; }

int gvs_graph_load_file(gvs_graph *g, const char *file_name, int as_undirected, int normalization,
                        const char *delimiters, const char *comment) {
    if (!g || !file_name) return gvk_fail(GVK_EINVAL, "gvs_graph_load_file: null argument");
    if (!delimiters) delimiters = " \t\r\n";
    if (!comment) comment = "#";
    return guarded("gvs_graph_load_file", [&]() {
        FILE *fin = fopen(file_name, "r");
        if (!fin) return gvk_fail(GVK_EINVAL, "File `%s` doesn't exist", file_name);
        const auto t_start = std::chrono::steady_clock::now();
        g->clear();
        g->as_undirected = as_undirected != 0;
        g->normalization = normalization != 0;
        // Two stages on two threads.  A reader takes the file in chunks of whole lines, tokenises them in place and hashes
        // the names (Graph::load_file's getline / strtok loop, graph.cuh:156-201, is all of that on one thread); this
        // thread resolves the names in the order of the file — ids are first-seen order — with the table slots of the
        // names 16 lines ahead being prefetched: two lookups per line into a table of up to 10^8 names are cache misses.
        struct Record {
            const char *u_name, *v_name;
            uint32_t u_len, v_len;
            uint64_t u_hash, v_hash;
            float weight;
        };
        struct Batch {
            std::vector<char> text;       // whole lines; the tokens are terminated in place
            std::vector<Record> records;
            size_t bad_line = 0;          // first malformed line of the batch (1-based line number), 0 = none
            bool last = false;
        };
        constexpr int kBatches = 3;
        constexpr size_t kChunk = (size_t)4 << 20;
        Batch batches[kBatches];
        std::mutex lock;
        std::condition_variable changed;
        size_t produced = 0, consumed = 0;  // batches[produced % k] is being filled, batches[consumed % k] resolved
        bool abandon = false, reader_failed = false;
        std::thread reader([&]() {
          try {
            std::string carry;
            size_t line_number = 0;
            bool eof = false;
            while (!eof) {
                {
                    std::unique_lock<std::mutex> hold(lock);
                    changed.wait(hold, [&]() { return abandon || produced - consumed < (size_t)kBatches; });
                    if (abandon) return;
                }
                Batch &batch = batches[produced % kBatches];
                batch.records.clear();
                batch.bad_line = 0;
                batch.text.assign(carry.begin(), carry.end());
                carry.clear();
                size_t complete = 0;  // bytes of whole lines in batch.text
                while (true) {
                    const size_t have = batch.text.size();
                    batch.text.resize(have + kChunk);
                    const size_t got = fread(batch.text.data() + have, 1, kChunk, fin);
                    batch.text.resize(have + got);
                    if (got < kChunk) eof = true;
                    for (size_t i = batch.text.size(); i > have; i--)
                        if (batch.text[i - 1] == '\n') {
                            complete = i;
                            break;
                        }
                    if (complete || eof) break;  // (a line longer than a chunk: keep reading)
                }
                if (eof) {
                    complete = batch.text.size();  // the last line may lack its newline
                } else {
                    carry.assign(batch.text.begin() + complete, batch.text.end());
                }
                batch.text.resize(complete);
                batch.text.push_back(0);
                char *line = batch.text.data(), *const end = batch.text.data() + complete;
                while (line < end && !batch.bad_line) {
                    // getline keeps the newline in the line (it is a token byte unless it is a delimiter)
                    char *stop = static_cast<char *>(memchr(line, '\n', end - line));
                    stop = stop ? stop + 1 : end;
                    const char saved = *stop;
                    *stop = 0;
                    line_number++;
                    if (*comment) {
                        char *c = strstr(line, comment);
                        if (c) *c = 0;
                    }
                    char *cursor = line;
                    Record r;
                    char *u_name = next_token(&cursor, delimiters);
                    if (u_name) {
                        char *v_name = next_token(&cursor, delimiters);
                        char *w_str = next_token(&cursor, delimiters);
                        char *more = next_token(&cursor, delimiters);
                        if (!v_name || more) {
                            batch.bad_line = line_number;
                        } else {
                            r.u_name = u_name, r.v_name = v_name;
                            r.weight = w_str ? (float)atof(w_str) : 1.f;
                            r.u_len = (uint32_t)strlen(u_name), r.v_len = (uint32_t)strlen(v_name);
                            r.u_hash = NameTable::hash_of(u_name, r.u_len), r.v_hash = NameTable::hash_of(v_name, r.v_len);
                            batch.records.push_back(r);
                        }
                    }
                    *stop = saved;
                    line = stop;
                }
                batch.last = eof || batch.bad_line;
                {
                    std::lock_guard<std::mutex> hold(lock);
                    produced++;
                }
                changed.notify_all();
                if (batch.bad_line) return;
            }
          } catch (...) {  // out of memory while reading: hand the consumer a final, empty batch
            std::lock_guard<std::mutex> hold(lock);
            Batch &batch = batches[produced % kBatches];
            batch.records.clear();
            batch.bad_line = 0, batch.last = true;
            reader_failed = true;
            produced++;
            changed.notify_all();
          }
        });
        struct Joiner {  // whatever way this scope is left, the reader is told to stop and joined
            std::thread &thread;
            std::mutex &lock;
            std::condition_variable &changed;
            bool &abandon;
            ~Joiner() {
                {
                    std::lock_guard<std::mutex> hold(lock);
                    abandon = true;
                }
                changed.notify_all();
                if (thread.joinable()) thread.join();
            }
        } joiner{reader, lock, changed, abandon};
        int rc = GVK_OK;
        constexpr size_t kLookahead = 16;
        double waited = 0;  // seconds this thread waited for the reader (GVS_TIMING)
        while (true) {
            {
                const auto w0 = std::chrono::steady_clock::now();
                std::unique_lock<std::mutex> hold(lock);
                changed.wait(hold, [&]() { return produced > consumed; });
                waited += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
            }
            Batch &batch = batches[consumed % kBatches];
            const std::vector<Record> &rec = batch.records;
            for (size_t i = 0; i < std::min(kLookahead, rec.size()); i++)
                g->name2id.prefetch(rec[i].u_hash), g->name2id.prefetch(rec[i].v_hash);
            for (size_t i = 0; i < rec.size(); i++) {
                if (i + kLookahead < rec.size())
                    g->name2id.prefetch(rec[i + kLookahead].u_hash), g->name2id.prefetch(rec[i + kLookahead].v_hash);
                const uint32_t u = g->id_of_name(rec[i].u_name, rec[i].u_len, rec[i].u_hash);
                const uint32_t v = g->id_of_name(rec[i].v_name, rec[i].v_len, rec[i].v_hash);
                g->add_edge(u, v, rec[i].weight);
            }
            if (batch.bad_line) rc = gvk_fail(GVK_EINVAL, "Invalid format at line %zu of `%s`", batch.bad_line, file_name);
            const bool last = batch.last;
            {
                std::lock_guard<std::mutex> hold(lock);
                consumed++;
            }
            changed.notify_all();
            if (last) break;
        }
        reader.join();
        fclose(fin);
        if (reader_failed) rc = gvk_fail(GVK_ENOMEM, "gvs_graph_load_file: out of host memory");
        if (rc != GVK_OK) {
            g->clear();
            return rc;
        }
        const auto t_parsed = std::chrono::steady_clock::now();
        g->finalize();
        if (getenv("GVS_TIMING"))
            fprintf(stderr, "[gvs] load_file: parse %.2f s (%.2f s of it waiting for the reader thread), flatten %.2f s\n",
                    std::chrono::duration<double>(t_parsed - t_start).count(), waited,
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t_parsed).count());
        return GVK_OK;
    });
}

// WordGraph::load_file_compact (include/instance/word_graph.cuh:73-181): a corpus becomes the graph of word
// co-occurrences within `window` tokens of a line; repeated co-occurrences add up in the edge weight.  Two passes over
// the file like the reference (frequencies, then pairs), but the pair counts live in ONE hash map keyed by (u, v) next
// to an insertion-ordered edge list — finalize() then groups them by source — instead of one hash map per vertex.
// The reference leaves the order of a vertex's neighbours to unordered_map iteration; here it is first co-occurrence.
int gvs_graph_load_corpus(gvs_graph *g, const char *file_name, int window, int min_count, int normalization,
                          const char *delimiters, const char *comment) {
    if (!g || !file_name) return gvk_fail(GVK_EINVAL, "gvs_graph_load_corpus: null argument");
    if (window < 0) return gvk_fail(GVK_EINVAL, "gvs_graph_load_corpus: negative window");
    if (!delimiters) delimiters = " \t\r\n";
    if (!comment) comment = "#";
    return guarded("gvs_graph_load_corpus", [&]() {
        FILE *fin = fopen(file_name, "r");
        if (!fin) return gvk_fail(GVK_EINVAL, "File `%s` doesn't exist", file_name);
        g->clear();
        g->as_undirected = true;
        g->normalization = normalization != 0;
        char *line = nullptr;
        size_t cap = 0;
        auto strip_comment = [&](char *text) {
            if (*comment) {
                char *c = strstr(text, comment);
                if (c) *c = 0;
            }
        };
        // pass 1: word frequencies, ids in first-seen order
        std::unordered_map<std::string, uint32_t> seen;
        std::vector<std::string> words;
        std::vector<uint64_t> frequency;
        while (getline(&line, &cap, fin) >= 0) {
            strip_comment(line);
            char *cursor = line;
            while (char *word = next_token(&cursor, delimiters)) {
                auto it = seen.find(word);
                if (it != seen.end()) {
                    frequency[it->second]++;
                } else {
                    seen.emplace(word, (uint32_t)words.size());
                    words.emplace_back(word);
                    frequency.push_back(1);
                }
            }
        }
        decltype(seen)().swap(seen);
        for (size_t i = 0; i < words.size(); i++)
            if (frequency[i] >= (uint64_t)std::max(min_count, 0)) g->id_of_name(words[i].c_str());
        decltype(words)().swap(words);
        decltype(frequency)().swap(frequency);
        // pass 2: co-occurrence counts
        std::unordered_map<uint64_t, uint64_t> edge_of;  // (u << 32 | v) -> index into src / dst / w
        auto count = [&](uint32_t u, uint32_t v) {
            const uint64_t key = ((uint64_t)u << 32) | v;
            auto it = edge_of.find(key);
            if (it == edge_of.end()) {
                edge_of.emplace(key, g->src.size());
                g->src.push_back(u);
                g->dst.push_back(v);
                g->w.push_back(1.f);
            } else {
                g->w[it->second] += 1.f;
            }
        };
        rewind(fin);
        std::vector<uint32_t> sentence;
        while (getline(&line, &cap, fin) >= 0) {
            strip_comment(line);
            sentence.clear();
            char *cursor = line;
            while (char *word = next_token(&cursor, delimiters)) {
                const uint32_t id = g->name2id.find(word, strlen(word), g->id2name);
                if (id != NameTable::kNone) sentence.push_back(id);
            }
            for (size_t i = 0; i < sentence.size(); i++)
                for (size_t j = 1; j <= (size_t)window && i + j < sentence.size(); j++) {
                    const uint32_t u = sentence[i], v = sentence[i + j];
                    count(u, v);
                    count(v, u);
                    g->vertex_weights[u] += 1.f;
                    g->vertex_weights[v] += 1.f;
                }
        }
        free(line);
        fclose(fin);
        g->num_edge = g->src.size();  // every (u, v) entry counts, both directions (word_graph.cuh:156-160)
        g->finalize();
        return GVK_OK;
    });
}

int gvs_graph_load_names(gvs_graph *g, const char *const *u_names, const char *const *v_names, const float *weights,
                         size_t n, int as_undirected, int normalization) {
    if (!g || (n && (!u_names || !v_names))) return gvk_fail(GVK_EINVAL, "gvs_graph_load_names: null argument");
    return guarded("gvs_graph_load_names", [&]() {
        g->clear();
        g->as_undirected = as_undirected != 0;
        g->normalization = normalization != 0;
        for (size_t i = 0; i < n; i++) {
            if (!u_names[i] || !v_names[i]) {
                g->clear();
                return gvk_fail(GVK_EINVAL, "gvs_graph_load_names: null name at edge %zu", i);
            }
            const uint32_t u = g->id_of_name(u_names[i]);
            const uint32_t v = g->id_of_name(v_names[i]);
            g->add_edge(u, v, weights ? weights[i] : 1.f);
        }
        g->finalize();
        return GVK_OK;
    });
}

int gvs_graph_load_labels(gvs_graph *g, const uint32_t *u_labels, const uint32_t *v_labels, const float *weights,
                          size_t n, int as_undirected, int normalization) {
    if (!g || (n && (!u_labels || !v_labels))) return gvk_fail(GVK_EINVAL, "gvs_graph_load_labels: null argument");
    return guarded("gvs_graph_load_labels", [&]() {
        g->clear();
        g->label_mode = true;
        g->as_undirected = as_undirected != 0;
        g->normalization = normalization != 0;
        g->src.reserve(as_undirected ? 2 * n : n);
        g->dst.reserve(as_undirected ? 2 * n : n);
        g->w.reserve(as_undirected ? 2 * n : n);
        for (size_t i = 0; i < n; i++) {
            const uint32_t u = g->id_of_label(u_labels[i]);
            const uint32_t v = g->id_of_label(v_labels[i]);
            g->add_edge(u, v, weights ? weights[i] : 1.f);
        }
        g->finalize();
        return GVK_OK;
    });
}

int gvs_graph_save(const gvs_graph *g, const char *file_name, int weighted, int anonymous) {
    if (!g || !file_name) return gvk_fail(GVK_EINVAL, "gvs_graph_save: null argument");
    FILE *fout = fopen(file_name, "w");
    if (!fout) return gvk_fail(GVK_EINVAL, "gvs_graph_save: cannot open `%s`", file_name);
    const size_t D = g->edge_weights.size();
    for (size_t e = 0; e < D; e++) {
        const uint32_t i = g->edges_uv[2 * e], j = g->edges_uv[2 * e + 1];
        if (anonymous)
            fprintf(fout, "%llu\t%llu", (unsigned long long)i, (unsigned long long)j);
        else if (g->label_mode)
            fprintf(fout, "%u\t%u", g->labels[i], g->labels[j]);
        else
            fprintf(fout, "%s\t%s", g->id2name[i].c_str(), g->id2name[j].c_str());
        if (weighted) fprintf(fout, "\t%f", g->edge_weights[e]);
        fputc('\n', fout);
    }
    fclose(fout);
    return GVK_OK;
}

uint32_t gvs_graph_num_vertex(const gvs_graph *g) { return g ? g->num_vertex : 0; }
uint64_t gvs_graph_num_edge(const gvs_graph *g) { return g ? g->num_edge : 0; }
uint64_t gvs_graph_num_directed_edge(const gvs_graph *g) { return g ? g->edge_weights.size() : 0; }
int gvs_graph_as_undirected(const gvs_graph *g) { return g && g->as_undirected; }
int gvs_graph_normalization(const gvs_graph *g) { return g && g->normalization; }

int64_t gvs_graph_name2id(const gvs_graph *g, const char *name) {
    if (!g || !name) return -1;
    if (g->label_mode) {
        char *end = nullptr;
        unsigned long long label = strtoull(name, &end, 10);
        if (end == name || *end || label > UINT32_MAX) return -1;
        if (label < gvs_graph::kDenseLabels) {
            if (label >= g->dense_label2id.size() || g->dense_label2id[label] == gvs_graph::kNoId) return -1;
            return (int64_t)g->dense_label2id[label];
        }
        auto it = g->label2id.find((uint32_t)label);
        return it == g->label2id.end() ? -1 : (int64_t)it->second;
    }
    const uint32_t id = g->name2id.find(name, strlen(name), g->id2name);
    return id == NameTable::kNone ? -1 : (int64_t)id;
}

int64_t gvs_graph_id2name(const gvs_graph *g, uint32_t id, char *buf, size_t buflen) {
    if (!g || id >= g->num_vertex) return -1;
    char tmp[16];
    const char *name;
    if (g->label_mode) {
        snprintf(tmp, sizeof(tmp), "%u", g->labels[id]);
        name = tmp;
    } else {
        name = g->id2name[id].c_str();
    }
    const size_t len = strlen(name);
    if (buf && buflen) {
        const size_t n = std::min(len, buflen - 1);
        memcpy(buf, name, n);
        buf[n] = 0;
    }
    return (int64_t)len;
}

const uint32_t *gvs_graph_edges(const gvs_graph *g) { return g ? g->edges_uv.data() : nullptr; }
const float *gvs_graph_edge_weights(const gvs_graph *g) { return g ? g->edge_weights.data() : nullptr; }
const uint64_t *gvs_graph_flat_offsets(const gvs_graph *g) { return g ? g->flat_offsets.data() : nullptr; }
const float *gvs_graph_vertex_weights(const gvs_graph *g) { return g ? g->vertex_weights.data() : nullptr; }

// ---- partition / schedule -----------------------------------------------------------------------------------

int gvs_partition(const float *weights, uint32_t n, int P, int32_t *part, uint32_t *local, uint32_t *part_sizes) {
    if (P < 1) return gvk_fail(GVK_EINVAL, "gvs_partition: num_partition must be >= 1");
    if (n && (!weights || !part || !local)) return gvk_fail(GVK_EINVAL, "gvs_partition: null argument");
    if (!part_sizes) return gvk_fail(GVK_EINVAL, "gvs_partition: null part_sizes");
    return guarded("gvs_partition", [&]() {
        std::vector<uint32_t> order(n);
        for (uint32_t i = 0; i < n; i++) order[i] = i;
        // the reference's std::sort leaves ties in unspecified order; stable_sort on ascending ids pins them
        std::stable_sort(order.begin(), order.end(),
                         [weights](uint32_t x, uint32_t y) { return weights[x] > weights[y]; });
        for (int p = 0; p < P; p++) part_sizes[p] = 0;
        for (uint32_t i = 0; i < n; i++) {
            int pid = (int)(i % (uint32_t)(P * 2));
            pid = std::min(pid, P * 2 - 1 - pid);
            part[order[i]] = pid;
            local[order[i]] = part_sizes[pid]++;
        }
        return GVK_OK;
    });
}

int gvs_schedule(int P, int W, int32_t *out, size_t out_len) {
    if (P < 1 || W < 1 || !out) return gvk_fail(GVK_EINVAL, "gvs_schedule: bad argument");
    if (P == 1) {
        if (out_len < 2) return gvk_fail(GVK_EINVAL, "gvs_schedule: output too small");
        out[0] = out[1] = 0;
        return 1;
    }
    if (P % W) return gvk_fail(GVK_EINVAL, "gvs_schedule: #partition (%d) must be a multiple of #worker (%d)", P, W);
    const size_t need = (size_t)(P / W) * (P / W) * W * W * 2;
    if (out_len < need) return gvk_fail(GVK_EINVAL, "gvs_schedule: output too small");
    int steps = 0;
    for (int x = 0; x < P; x += W)
        for (int y = 0; y < P; y += W)
            for (int offset = 0; offset < W; offset++) {
                for (int i = 0; i < W; i++) {
                    out[((size_t)steps * W + i) * 2] = x + (i + offset) % W;
                    out[((size_t)steps * W + i) * 2 + 1] = y + i;
                }
                steps++;
            }
    return steps;
}

}  // extern "C"

// ---- samplers ---------------------------------------------------------------------------------------------

namespace {

inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                          uint32_t out[4]) {
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0;
        c1 = (uint32_t)p1;
        c2 = n2;
        c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

constexpr uint32_t kTagHost = 0x686f7374u;

// Philox is most of the cost of a draw once the table misses are pipelined (~30 cycles per call scalar, one call per
// edge sample), and calls are independent by construction: run 8 of them per iteration in 64-bit AVX2 lanes
// (vpmuludq reads the low halves only, so the high halves may hold garbage until the final mask).  Same words, same
// doubles as the scalar routine above (tests pin both against the oracle).
// (((hi << 32) | lo) >> 11) * 2^-53 for 4 lanes, exactly as the scalar conversion (both halves via the 2^52 / 2^84 trick)
__attribute__((target("avx2"))) inline __m256d words_to_double_avx2(__m256i w_lo, __m256i w_hi) {
    const __m256i low = _mm256_set1_epi64x(0xffffffffll);
    const __m256i magic_lo = _mm256_set1_epi64x(0x4330000000000000ll), magic_hi = _mm256_set1_epi64x(0x4530000000000000ll);
    const __m256d bias = _mm256_set1_pd(19342813118337666422669312.0 /* 2^84 + 2^52 */);
    const __m256i x = _mm256_srli_epi64(_mm256_or_si256(_mm256_slli_epi64(w_hi, 32), _mm256_and_si256(w_lo, low)), 11);
    const __m256d lo = _mm256_castsi256_pd(_mm256_or_si256(_mm256_and_si256(x, low), magic_lo));
    const __m256d hi = _mm256_castsi256_pd(_mm256_or_si256(_mm256_srli_epi64(x, 32), magic_hi));
    return _mm256_mul_pd(_mm256_add_pd(_mm256_sub_pd(hi, bias), lo), _mm256_set1_pd(1.0 / 9007199254740992.0));
}

__attribute__((target("avx2"))) void philox_doubles_avx2(uint64_t call0, int num_call, uint32_t stream, uint32_t k0,
                                                         uint32_t k1, double *out) {
    const __m256i M0 = _mm256_set1_epi64x(0xD2511F53ll), M1 = _mm256_set1_epi64x(0xCD9E8D57ll);
    __m256i K0[10], K1[10];
    for (int r = 0; r < 10; r++) {
        K0[r] = _mm256_set1_epi64x((long long)(uint32_t)(k0 + (uint32_t)r * 0x9E3779B9u));
        K1[r] = _mm256_set1_epi64x((long long)(uint32_t)(k1 + (uint32_t)r * 0xBB67AE85u));
    }
    const __m256i c2_init = _mm256_set1_epi64x((long long)stream), c3_init = _mm256_set1_epi64x((long long)kTagHost);
    for (int j = 0; j < num_call; j += 8) {
        const __m256i ia = _mm256_add_epi64(_mm256_set1_epi64x((long long)(call0 + (uint64_t)j)), _mm256_set_epi64x(3, 2, 1, 0));
        const __m256i ib = _mm256_add_epi64(ia, _mm256_set1_epi64x(4));
        __m256i a0 = ia, a1 = _mm256_srli_epi64(ia, 32), a2 = c2_init, a3 = c3_init;
        __m256i b0 = ib, b1 = _mm256_srli_epi64(ib, 32), b2 = c2_init, b3 = c3_init;
        for (int r = 0; r < 10; r++) {
            const __m256i pa0 = _mm256_mul_epu32(M0, a0), pa1 = _mm256_mul_epu32(M1, a2);
            const __m256i pb0 = _mm256_mul_epu32(M0, b0), pb1 = _mm256_mul_epu32(M1, b2);
            a0 = _mm256_xor_si256(_mm256_xor_si256(_mm256_srli_epi64(pa1, 32), a1), K0[r]);
            a2 = _mm256_xor_si256(_mm256_xor_si256(_mm256_srli_epi64(pa0, 32), a3), K1[r]);
            a1 = pa1, a3 = pa0;
            b0 = _mm256_xor_si256(_mm256_xor_si256(_mm256_srli_epi64(pb1, 32), b1), K0[r]);
            b2 = _mm256_xor_si256(_mm256_xor_si256(_mm256_srli_epi64(pb0, 32), b3), K1[r]);
            b1 = pb1, b3 = pb0;
        }
        const __m256d ad0 = words_to_double_avx2(a0, a1), ad1 = words_to_double_avx2(a2, a3),
                      bd0 = words_to_double_avx2(b0, b1), bd1 = words_to_double_avx2(b2, b3);
        const __m256d al = _mm256_unpacklo_pd(ad0, ad1), ah = _mm256_unpackhi_pd(ad0, ad1);
        const __m256d bl = _mm256_unpacklo_pd(bd0, bd1), bh = _mm256_unpackhi_pd(bd0, bd1);
        _mm256_storeu_pd(out + 2 * j, _mm256_permute2f128_pd(al, ah, 0x20));
        _mm256_storeu_pd(out + 2 * j + 4, _mm256_permute2f128_pd(al, ah, 0x31));
        _mm256_storeu_pd(out + 2 * j + 8, _mm256_permute2f128_pd(bl, bh, 0x20));
        _mm256_storeu_pd(out + 2 * j + 12, _mm256_permute2f128_pd(bl, bh, 0x31));
    }
}

// the same, 16 calls per iteration, where the CPU has AVX-512 (F + DQ for the u64 -> f64 conversion)
__attribute__((target("avx512f,avx512dq"))) void philox_doubles_avx512(uint64_t call0, int num_call, uint32_t stream,
                                                                       uint32_t k0, uint32_t k1, double *out) {
    const __m512i M0 = _mm512_set1_epi64(0xD2511F53ll), M1 = _mm512_set1_epi64(0xCD9E8D57ll);
    const __m512i low = _mm512_set1_epi64(0xffffffffll);
    const __m512d scale = _mm512_set1_pd(1.0 / 9007199254740992.0);
    const __m512i even = _mm512_set_epi64(11, 3, 10, 2, 9, 1, 8, 0), odd = _mm512_set_epi64(15, 7, 14, 6, 13, 5, 12, 4);
    __m512i K0[10], K1[10];
    for (int r = 0; r < 10; r++) {
        K0[r] = _mm512_set1_epi64((long long)(uint32_t)(k0 + (uint32_t)r * 0x9E3779B9u));
        K1[r] = _mm512_set1_epi64((long long)(uint32_t)(k1 + (uint32_t)r * 0xBB67AE85u));
    }
    const __m512i c2_init = _mm512_set1_epi64((long long)stream), c3_init = _mm512_set1_epi64((long long)kTagHost);
    for (int j = 0; j < num_call; j += 16) {
        const __m512i ia = _mm512_add_epi64(_mm512_set1_epi64((long long)(call0 + (uint64_t)j)),
                                            _mm512_set_epi64(7, 6, 5, 4, 3, 2, 1, 0));
        const __m512i ib = _mm512_add_epi64(ia, _mm512_set1_epi64(8));
        __m512i a0 = ia, a1 = _mm512_srli_epi64(ia, 32), a2 = c2_init, a3 = c3_init;
        __m512i b0 = ib, b1 = _mm512_srli_epi64(ib, 32), b2 = c2_init, b3 = c3_init;
        for (int r = 0; r < 10; r++) {
            const __m512i pa0 = _mm512_mul_epu32(M0, a0), pa1 = _mm512_mul_epu32(M1, a2);
            const __m512i pb0 = _mm512_mul_epu32(M0, b0), pb1 = _mm512_mul_epu32(M1, b2);
            a0 = _mm512_xor_si512(_mm512_xor_si512(_mm512_srli_epi64(pa1, 32), a1), K0[r]);
            a2 = _mm512_xor_si512(_mm512_xor_si512(_mm512_srli_epi64(pa0, 32), a3), K1[r]);
            a1 = pa1, a3 = pa0;
            b0 = _mm512_xor_si512(_mm512_xor_si512(_mm512_srli_epi64(pb1, 32), b1), K0[r]);
            b2 = _mm512_xor_si512(_mm512_xor_si512(_mm512_srli_epi64(pb0, 32), b3), K1[r]);
            b1 = pb1, b3 = pb0;
        }
#define GVS_TO_DOUBLE(w_lo, w_hi)                                                                                    \
    _mm512_mul_pd(_mm512_cvtepu64_pd(_mm512_srli_epi64(                                                              \
                      _mm512_or_si512(_mm512_slli_epi64(w_hi, 32), _mm512_and_si512(w_lo, low)), 11)),               \
                  scale)
        const __m512d ad0 = GVS_TO_DOUBLE(a0, a1), ad1 = GVS_TO_DOUBLE(a2, a3);
        const __m512d bd0 = GVS_TO_DOUBLE(b0, b1), bd1 = GVS_TO_DOUBLE(b2, b3);
#undef GVS_TO_DOUBLE
        _mm512_storeu_pd(out + 2 * j, _mm512_permutex2var_pd(ad0, even, ad1));
        _mm512_storeu_pd(out + 2 * j + 8, _mm512_permutex2var_pd(ad0, odd, ad1));
        _mm512_storeu_pd(out + 2 * j + 16, _mm512_permutex2var_pd(bd0, even, bd1));
        _mm512_storeu_pd(out + 2 * j + 24, _mm512_permutex2var_pd(bd0, odd, bd1));
    }
}

inline double words_to_double(uint32_t hi, uint32_t lo) {
    return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}

void philox_doubles(uint64_t call0, int num_call, uint32_t stream, uint32_t k0, uint32_t k1, double *out) {
    static const bool avx2 = __builtin_cpu_supports("avx2") && getenv("GVS_NO_AVX2") == nullptr;
    static const bool avx512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") &&
                               getenv("GVS_NO_AVX512") == nullptr;
    if (avx512 && num_call % 16 == 0) return philox_doubles_avx512(call0, num_call, stream, k0, k1, out);
    if (avx2 && num_call % 8 == 0) return philox_doubles_avx2(call0, num_call, stream, k0, k1, out);
    for (int j = 0; j < num_call; j++) {
        const uint64_t i = call0 + (uint64_t)j;
        uint32_t w[4];
        philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), stream, kTagHost, k0, k1, w);
        out[2 * j] = words_to_double(w[1], w[0]);
        out[2 * j + 1] = words_to_double(w[3], w[2]);
    }
}

// Per-thread stream of doubles: call i of the thread's Philox stream yields doubles 2i and 2i+1.  `pos` (doubles
// consumed) is the whole state; doubles are produced a block at a time and handed out in order.
struct HostRng {
    static constexpr int kBlockCalls = 128;
    uint32_t k0, k1, stream;
    uint64_t pos;        // doubles consumed
    uint64_t block_pos;  // double index of block[0]; pos + 1 = no block yet (the distance below wraps to 2^64 - 1)
    double block[2 * kBlockCalls];

    HostRng(uint64_t seed, uint32_t stream_, uint64_t pos_)
        : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)), stream(stream_), pos(pos_), block_pos(pos_ + 1) {}

    inline double next() {
        uint64_t at = pos - block_pos;
        if (__builtin_expect(at >= (uint64_t)(2 * kBlockCalls), 0)) {  // also taken for the first draw
            block_pos = pos & ~(uint64_t)1;
            philox_doubles(block_pos >> 1, kBlockCalls, stream, k0, k1, block);
            at = pos - block_pos;
        }
        pos++;
        return block[at];
    }
};

struct EdgeSlot {
    float prob;
    uint32_t pad;
    uint64_t alias;
};

inline uint64_t pack_location(int32_t part, uint32_t local) { return ((uint64_t)(uint32_t)part << 32) | local; }

// One cache line per edge draw: the slot's probability plus the packed (partition, local id) locations of both
// outcomes (keep the slot's own edge / take its alias).  The draw itself is unchanged — same slot, same comparison —
// but the dependent misses into edges[] and the location table disappear: at 255 threads the edge sampler is bound by
// random DRAM lines per second, and this is one line per sample instead of two plus two cache-resident lookups.
struct alignas(64) FatSlot {
    float prob;
    uint32_t pad;
    uint64_t self_head, self_tail, alias_head, alias_tail;
};

// What a random walk reads when it steps onto a vertex: its packed location (for the pool record) and the range of its
// out-edges (for the next draw) — one line instead of one miss into location[] and one into flat_offsets[].
struct alignas(32) WalkVertex {
    uint64_t location, first;
    uint64_t degree;
    uint64_t pad;
};

// Walks on graphs below the fat-slot limit read the outcome of a draw from the slot itself: the vertices of both
// outcomes (keep the slot's edge / take its alias) sit next to the probability, so a step touches the slot and the
// WalkVertex of where it lands — two lines instead of slot, edge list, location and offsets.
struct alignas(32) WalkStartSlot {  // over the global edge table: a walk starts with a weighted edge (graph.cuh:322-333)
    float prob;
    uint32_t self_from, self_to, alias_from, alias_to;
    uint32_t pad[3];
};
struct WalkStepSlot {  // per-vertex tables, CSR-aligned
    float prob;
    uint32_t alias;  // kept for completeness: local index of the alias neighbour
    uint32_t self_to, alias_to;
};

// tables above this many entries keep the 16-byte slots (64 B x 2^27 = 8 GiB)
constexpr size_t kFatSlotLimit = (size_t)1 << 27;

// GVS_FAT_SLOT_LIMIT=0 forces the thin tables (tests run both forms against the oracle)
inline size_t fat_slot_limit() {
    const char *env = getenv("GVS_FAT_SLOT_LIMIT");
    return env ? (size_t)strtoull(env, nullptr, 10) : kFatSlotLimit;
}

}  // namespace

// A persistent pool of OS threads that runs the "virtual" sampler threads of a fill.  Creating hundreds of
// std::threads per fill (the reference does, solver.h:633-635) costs little on average but stalls for tens of
// milliseconds every few fills when the new threads fight over the process's mmap lock (stacks, malloc arenas);
// with the pool a fill is a wake-up.  Slice t and uniform stream t still belong to virtual thread t, whichever OS
// thread happens to run it, so a fill remains a pure function of (seed, num_thread, stream positions).
class WorkerPool {
public:
    ~WorkerPool() { shutdown(); }

    void run(int total, int os_threads, int cpu_offset, const std::function<void(int)> &job) {
        const int hw = std::max(1, (int)std::thread::hardware_concurrency());
        resize(std::min(total, os_threads > 0 ? os_threads : hw));
        pin(cpu_offset);
        {
            std::lock_guard<std::mutex> lock(mutex_);
            job_ = &job;
            total_ = total;
            next_.store(0);
            pending_ = (int)threads_.size();
            generation_++;
        }
        wake_.notify_all();
        std::unique_lock<std::mutex> lock(mutex_);
        done_.wait(lock, [this] { return pending_ == 0; });
        job_ = nullptr;
    }

private:
    // thread k -> the (offset + k)-th CPU of the process's affinity mask; done once per offset
    void pin(int offset) {
        if (offset < 0 || (offset == pinned_offset_ && pinned_count_ == threads_.size())) return;
        cpu_set_t allowed;
        if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
        std::vector<int> cpus;
        for (int c = 0; c < CPU_SETSIZE; c++)
            if (CPU_ISSET(c, &allowed)) cpus.push_back(c);
        if (cpus.empty()) return;
        for (size_t k = 0; k < threads_.size(); k++) {
            cpu_set_t one;
            CPU_ZERO(&one);
            CPU_SET(cpus[((size_t)offset + k) % cpus.size()], &one);
            pthread_setaffinity_np(threads_[k].native_handle(), sizeof(one), &one);
        }
        pinned_offset_ = offset;
        pinned_count_ = threads_.size();
    }

    void resize(int n) {
        if ((int)threads_.size() >= n) return;
        while ((int)threads_.size() < n) {
            const uint64_t seen = generation_;
            threads_.emplace_back([this, seen] { loop(seen); });
        }
    }

    void loop(uint64_t seen) {
        for (;;) {
            const std::function<void(int)> *job;
            int total;
            {
                std::unique_lock<std::mutex> lock(mutex_);
                wake_.wait(lock, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
                job = job_;
                total = total_;
            }
            for (int i; (i = next_.fetch_add(1)) < total;) (*job)(i);
            {
                std::lock_guard<std::mutex> lock(mutex_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }

    void shutdown() {
        {
            std::lock_guard<std::mutex> lock(mutex_);
            stop_ = true;
        }
        wake_.notify_all();
        for (auto &t : threads_) t.join();
        threads_.clear();
    }

    std::vector<std::thread> threads_;
    std::mutex mutex_;
    std::condition_variable wake_, done_;
    const std::function<void(int)> *job_ = nullptr;
    std::atomic<int> next_{0};
    int total_ = 0, pending_ = 0;
    uint64_t generation_ = 0;
    bool stop_ = false;
    int pinned_offset_ = -1;
    size_t pinned_count_ = 0;
};

struct gvs_sampler {
    const gvs_graph *g = nullptr;
    int P = 1;
    uint64_t seed = 0;
    HugeVector<uint64_t> location;  // (part << 32) | local, per vertex
    std::vector<float> edge_prob;
    std::vector<uint64_t> edge_alias;
    HugeVector<EdgeSlot> edge_slots;  // the same table, one cache line touch per draw
    int prepared = GVS_MODE_EDGE;
    float p = 1, q = 1;
    HugeVector<gvk_alias_entry> nb_slots;  // per-vertex (WALK / REJECT) or per-edge (BIASED) tables, {prob, alias} per slot
    HugeVector<WalkVertex> walk_vertex;    // what a walk needs when it arrives at a vertex, in one cache line
    HugeVector<WalkStartSlot> walk_start;  // WALK / REJECT below the fat-slot limit
    HugeVector<WalkStepSlot> walk_step;
    std::vector<float> nb_prob;            // split copies of nb_slots for hosts that ask for them (tests, device upload)
    std::vector<uint32_t> nb_alias;
    HugeVector<uint64_t> ee_offsets;
    HugeVector<uint32_t> sorted_nb;  // BIASED_REJECT: out-neighbours of every vertex, ascending
    std::vector<uint64_t> positions;
    // EDGE mode with a tail-partition filter: a table over just the edges whose tail lives in that partition
    // (the exact conditional distribution) instead of drawing from all edges and dropping (P - 1) / P of them
    struct Column {
        std::vector<uint64_t> edge_ids;
        std::vector<float> prob;
        std::vector<uint64_t> alias;
        HugeVector<EdgeSlot> slots;
        HugeVector<FatSlot> fat;
    };
    std::vector<Column> columns;
    HugeVector<FatSlot> edge_fat;
    WorkerPool pool;

    // fat[i] for slot i of a table whose entry j stands for flattened edge ids[j] (ids == nullptr: entry j is edge j)
    void build_fat(const HugeVector<EdgeSlot> &slots, const uint64_t *ids, HugeVector<FatSlot> *fat,
                   int num_thread) const {
        if (!fat->empty() || slots.size() > fat_slot_limit()) return;
        const uint32_t *edges = g->edges_uv.data();
        fat->resize(slots.size());
        FatSlot *out = fat->data();
        const size_t n = slots.size(), T = (size_t)std::max(1, num_thread), work = (n + T - 1) / T;
        std::vector<std::thread> threads;
        for (size_t t = 0; t < T && t * work < n; t++)
            threads.emplace_back([&, t]() {
                for (size_t i = t * work; i < std::min(n, (t + 1) * work); i++) {
                    const uint64_t self = ids ? ids[i] : i, other = ids ? ids[slots[i].alias] : slots[i].alias;
                    out[i] = FatSlot{slots[i].prob, 0, location[edges[2 * self]], location[edges[2 * self + 1]],
                                     location[edges[2 * other]], location[edges[2 * other + 1]]};
                }
            });
        for (auto &t : threads) t.join();
    }

    // Column table of tail partition r: alias table over exactly the edges that end in r (+ its fat form).
    int ensure_column(int r, int num_thread) {
        if (columns.empty()) columns.resize(P);
        Column &col = columns[r];
        if (col.slots.empty()) {
            const size_t D = g->edge_weights.size();
            std::vector<float> weights;
            for (size_t e = 0; e < D; e++)
                if ((int)(location[g->edges_uv[2 * e + 1]] >> 32) == r) {
                    col.edge_ids.push_back(e);
                    weights.push_back(g->edge_weights[e]);
                }
            if (weights.empty()) return gvk_fail(GVK_EINVAL, "no edge ends in partition %d", r);
            col.prob.resize(weights.size());
            col.alias.resize(weights.size());
            const int rc = gvk_alias_build(weights.data(), weights.size(), col.prob.data(), col.alias.data(), 8, nullptr);
            if (rc != GVK_OK) return rc;
            col.slots.resize(weights.size());
            for (size_t i = 0; i < weights.size(); i++) col.slots[i] = EdgeSlot{col.prob[i], 0, col.alias[i]};
        }
        build_fat(col.slots, col.edge_ids.data(), &col.fat, num_thread);
        return GVK_OK;
    }

    // The alias table over ALL flattened edges (the reference's edge_table, solver.h:123,259-260).  Built on first
    // use: a multi-GPU LINE run only ever draws from its column tables and never needs it.
    int ensure_edge_table() {
        if (!edge_slots.empty()) return GVK_OK;
        const size_t D = g->edge_weights.size();
        std::vector<float> prob(D);
        std::vector<uint64_t> alias(D);
        const int rc = gvk_alias_build(g->edge_weights.data(), D, prob.data(), alias.data(), 8, nullptr);
        if (rc != GVK_OK) return rc;
        edge_slots.resize(D);
        for (size_t e = 0; e < D; e++) edge_slots[e] = EdgeSlot{prob[e], 0, alias[e]};
        return GVK_OK;
    }

    // split copies for hosts that want the reference's two-array form (tests)
    void materialize_split_table() {
        if (edge_prob.size() == edge_slots.size()) return;
        edge_prob.resize(edge_slots.size());
        edge_alias.resize(edge_slots.size());
        for (size_t e = 0; e < edge_slots.size(); e++) {
            edge_prob[e] = edge_slots[e].prob;
            edge_alias[e] = edge_slots[e].alias;
        }
    }
};

namespace {

struct FillShared {
    const gvs_sampler *s;
    uint32_t *const *pools;
    uint64_t pool_size;
    gvs_fill_config c;
    std::atomic<int> error{0};
};

// a block that no positive sample can reach would spin forever (the reference would, too): give up after this
// many consecutive inner rounds in which this thread wrote nothing
constexpr int kMaxIdleRounds = 1 << 16;

struct BlockCursor {
    std::vector<int64_t> offsets;
    int num_complete = 0, target = 0;
    int P, tail_filter;
    int64_t end;
    BlockCursor(int P_, int tail_filter_, int64_t start, int64_t end_)
        : offsets((size_t)P_ * P_, start), P(P_), tail_filter(tail_filter_), end(end_) {
        target = tail_filter < 0 ? P * P : P;
    }
    // returns the slot to write, or -1 when the block is full / not ours
    inline int64_t claim(int hp, int tp) {
        if (tail_filter >= 0 && tp != tail_filter) return -1;
        int64_t &offset = offsets[(size_t)hp * P + tp];
        if (offset >= end) return -1;
        const int64_t slot = offset;
        if (++offset == end) num_complete++;
        return slot;
    }
    inline bool done() const { return num_complete >= target; }
};

void fill_edges(FillShared *sh, int thread, int64_t start, int64_t end, uint64_t *position) {
    const gvs_sampler &s = *sh->s;
    HostRng rng(s.seed, (uint32_t)thread, *position);
    if (start >= end) return;
    BlockCursor cur(s.P, sh->c.tail_partition, start, end);
    const int n = sh->c.sample_batch_size;
    const uint32_t *edges = s.g->edges_uv.data();
    const gvs_sampler::Column *column =
        sh->c.tail_partition >= 0 && s.P > 1 ? &s.columns[sh->c.tail_partition] : nullptr;
    const EdgeSlot *slots = column ? column->slots.data() : s.edge_slots.data();
    const double count = (double)(column ? column->slots.size() : s.edge_slots.size());
    // A draw is a chain of dependent cache misses (slot -> edge -> two locations) into tables far larger than the
    // caches.  The reference walks that chain one sample at a time (solver.h:1022-1035); here every stage runs over
    // the whole inner round with the next stage's lines prefetched, so a thread keeps tens of misses in flight.
    // Uniforms are still consumed two per sample, in sample order.
    // scratch lives with the (persistent) OS thread: per-fill allocation of these buffers by hundreds of threads at once
    // makes malloc trim and re-fault its arenas, which serialises on the process's mmap lock (fills 10x slower)
    static thread_local std::vector<uint64_t> index, edge, heads, tails;
    static thread_local std::vector<float> u;
    index.resize(n), edge.resize(n), heads.resize(n), tails.resize(n), u.resize(n);
    const HugeVector<FatSlot> &fat_table = column ? column->fat : s.edge_fat;
    const FatSlot *fat = fat_table.empty() ? nullptr : fat_table.data();
    int idle = 0;
    while (!cur.done() && !sh->error.load(std::memory_order_relaxed)) {
        for (int i = 0; i < n; i++) {
            const double r1 = rng.next(), r2 = rng.next();
            index[i] = (uint64_t)(r1 * count);
            u[i] = (float)r2;
            if (fat)
                __builtin_prefetch(&fat[index[i]]);
            else
                __builtin_prefetch(&slots[index[i]]);
        }
        if (fat) {
            for (int i = 0; i < n; i++) {
                const FatSlot &slot = fat[index[i]];
                const bool self = u[i] < slot.prob;
                heads[i] = self ? slot.self_head : slot.alias_head;
                tails[i] = self ? slot.self_tail : slot.alias_tail;
            }
        } else {
            for (int i = 0; i < n; i++) {
                const EdgeSlot &slot = slots[index[i]];
                const uint64_t pick = u[i] < slot.prob ? index[i] : slot.alias;
                edge[i] = column ? column->edge_ids[pick] : pick;
                __builtin_prefetch(&edges[2 * edge[i]]);
            }
            for (int i = 0; i < n; i++) {
                heads[i] = edges[2 * edge[i]];
                tails[i] = edges[2 * edge[i] + 1];
                __builtin_prefetch(&s.location[heads[i]]);
                __builtin_prefetch(&s.location[tails[i]]);
            }
            for (int i = 0; i < n; i++) {
                heads[i] = s.location[heads[i]];
                tails[i] = s.location[tails[i]];
            }
        }
        bool wrote = false;
        for (int i = 0; i < n; i++) {
            const int hp = (int)(heads[i] >> 32), tp = (int)(tails[i] >> 32);
            const int64_t slot = cur.claim(hp, tp);
            if (slot >= 0) {
                uint32_t *pool = sh->pools[(size_t)hp * s.P + tp];
                pool[2 * slot] = (uint32_t)tails[i];
                pool[2 * slot + 1] = (uint32_t)heads[i];
                wrote = true;
            }
        }
        idle = wrote ? 0 : idle + 1;
        if (idle > kMaxIdleRounds) sh->error.store(1);
    }
    *position = rng.pos;
}

// FAT: WALK / REJECT with walk_start / walk_step built.  Same draws, same uniforms, same pools as the thin form.
template <bool FAT>
void fill_walks(FillShared *sh, int thread, int64_t start, int64_t end, uint64_t *position) {
    const gvs_sampler &s = *sh->s;
    HostRng rng(s.seed, (uint32_t)thread, *position);
    if (start >= end) return;
    const bool biased = sh->c.mode == GVS_MODE_BIASED_WALK, reject = sh->c.mode == GVS_MODE_BIASED_REJECT;
    const float fmax = std::max(1.0f, std::max(1.0f / s.p, 1.0f / s.q));
    BlockCursor cur(s.P, sh->c.tail_partition, start, end);
    const int L = sh->c.walk_length, nb = sh->c.walk_batch, aug = sh->c.augmentation_step;
    const int64_t sb = sh->c.shuffle_base, stride = (int64_t)(sh->pool_size / (uint64_t)sb);
    static thread_local std::vector<uint64_t> chains, index, edge_id, base, proposal;
    static thread_local std::vector<int> lengths, live, pending;
    static thread_local std::vector<uint32_t> current, previous, next;
    static thread_local std::vector<float> u, accept;
    chains.resize((size_t)nb * (L + 1));
    lengths.resize(nb), live.resize(nb), pending.resize(nb), current.resize(nb), u.resize(nb), accept.resize(nb);
    index.resize(nb), edge_id.resize(nb), base.resize(nb), proposal.resize(nb), previous.resize(nb), next.resize(nb);
    const uint32_t *edges = s.g->edges_uv.data();
    const EdgeSlot *slots = s.edge_slots.data();
    const double edge_count = (double)s.edge_slots.size();
    const WalkVertex *wv = s.walk_vertex.data();
    const gvk_alias_entry *nb_slots = s.nb_slots.data();
    const WalkStartSlot *walk_start = s.walk_start.data();
    const WalkStepSlot *walk_step = s.walk_step.data();
    const uint32_t *sorted_nb = s.sorted_nb.data();
    // The walks of one inner round advance in LOCKSTEP: all start edges, then step 2 of every live walk, step 3, ...
    // (the reference finishes one walk before it starts the next, graph.cuh:322-350,400-425).  Each stage runs over
    // the whole round with the next stage's cache lines prefetched, so a thread overlaps ~walk_batch misses instead
    // of paying every one of them serially.  Uniforms are consumed in that lockstep order, two per draw.
    int idle = 0;
    while (!cur.done() && !sh->error.load(std::memory_order_relaxed)) {
        for (int i = 0; i < nb; i++) {
            const double r1 = rng.next(), r2 = rng.next();
            index[i] = (uint64_t)(r1 * edge_count);
            u[i] = (float)r2;
            if (FAT)
                __builtin_prefetch(&walk_start[index[i]]);
            else
                __builtin_prefetch(&slots[index[i]]);
        }
        if (FAT) {
            for (int i = 0; i < nb; i++) {
                const WalkStartSlot &slot = walk_start[index[i]];
                const bool self = u[i] < slot.prob;
                previous[i] = self ? slot.self_from : slot.alias_from;
                current[i] = self ? slot.self_to : slot.alias_to;
                __builtin_prefetch(&wv[previous[i]]);
                __builtin_prefetch(&wv[current[i]]);
            }
        } else {
            for (int i = 0; i < nb; i++) {
                const EdgeSlot &slot = slots[index[i]];
                edge_id[i] = u[i] < slot.prob ? index[i] : slot.alias;
                __builtin_prefetch(&edges[2 * edge_id[i]]);
            }
            for (int i = 0; i < nb; i++) {
                previous[i] = edges[2 * edge_id[i]];
                current[i] = edges[2 * edge_id[i] + 1];
                __builtin_prefetch(&wv[previous[i]]);
                __builtin_prefetch(&wv[current[i]]);
            }
        }
        int num_live = 0;
        for (int i = 0; i < nb; i++) {
            uint64_t *chain = chains.data() + (size_t)i * (L + 1);
            chain[0] = wv[previous[i]].location;
            chain[1] = wv[current[i]].location;
            lengths[i] = L;
            live[num_live++] = i;
        }
        for (int j = 2; j <= L && num_live; j++) {
            int kept = 0;
            for (int n = 0; n < num_live; n++) {  // walks standing on a node without out-edges stop here
                const int i = live[n];
                if (wv[current[i]].degree == 0)
                    lengths[i] = j - 1;  // graph.cuh:346-349,421-424
                else
                    live[kept++] = i;
            }
            num_live = kept;
            // proposal rounds: one for the table-driven walks; node2vec by rejection repeats for the rejected walks
            int num_pending = num_live;
            for (int n = 0; n < num_live; n++) pending[n] = live[n];
            while (num_pending) {
                for (int n = 0; n < num_pending; n++) {
                    const int i = pending[n];
                    const WalkVertex &at = wv[current[i]];
                    const double r1 = rng.next(), r2 = rng.next();
                    if (reject) accept[i] = (float)rng.next();
                    base[i] = biased ? s.ee_offsets[edge_id[i]] : at.first;
                    index[i] = (uint64_t)(r1 * (double)at.degree);
                    u[i] = (float)r2;
                    if (FAT)
                        __builtin_prefetch(&walk_step[base[i] + index[i]]);
                    else
                        __builtin_prefetch(&nb_slots[base[i] + index[i]]);
                }
                if (FAT) {
                    for (int n = 0; n < num_pending; n++) {
                        const int i = pending[n];
                        const WalkStepSlot &slot = walk_step[base[i] + index[i]];
                        next[i] = u[i] < slot.prob ? slot.self_to : slot.alias_to;
                        __builtin_prefetch(&wv[next[i]]);
                    }
                } else {
                    for (int n = 0; n < num_pending; n++) {
                        const int i = pending[n];
                        const gvk_alias_entry &slot = nb_slots[base[i] + index[i]];
                        const uint32_t neighbor = u[i] < slot.prob ? (uint32_t)index[i] : slot.alias;
                        proposal[i] = wv[current[i]].first + neighbor;
                        __builtin_prefetch(&edges[2 * proposal[i] + 1]);
                    }
                    for (int n = 0; n < num_pending; n++) {
                        const int i = pending[n];
                        next[i] = edges[2 * proposal[i] + 1];
                        __builtin_prefetch(&wv[next[i]]);
                    }
                }
                if (!reject) break;
                int rejected = 0;
                for (int n = 0; n < num_pending; n++) {
                    const int i = pending[n];
                    const uint32_t x = next[i], prev = previous[i];
                    float f;
                    if (x == prev) {
                        f = 1.0f / s.p;
                    } else {
                        const uint32_t *first = sorted_nb + wv[x].first;
                        f = std::binary_search(first, first + wv[x].degree, prev) ? 1.0f : 1.0f / s.q;
                    }
                    if (!(accept[i] * fmax < f)) pending[rejected++] = i;
                }
                num_pending = rejected;
            }
            for (int n = 0; n < num_live; n++) {
                const int i = live[n];
                edge_id[i] = proposal[i];  // only the per-edge tables of BIASED_WALK (never FAT) read it
                previous[i] = current[i];
                current[i] = next[i];
                chains[(size_t)i * (L + 1) + j] = wv[current[i]].location;
            }
        }
        bool wrote = false;
        for (int i = 0; i < nb; i++) {
            const uint64_t *chain = chains.data() + (size_t)i * (L + 1);
            const int len = lengths[i];
            for (int j = 0; j < len; j++)
                for (int k = 1; k <= aug; k++) {
                    if (j + k > len) break;
                    const uint64_t h = chain[j], t = chain[j + k];
                    const int hp = (int)(h >> 32), tp = (int)(t >> 32);
                    const int64_t offset = cur.claim(hp, tp);
                    if (offset >= 0) {
                        const int64_t slot = offset % sb * stride + offset / sb;  // pseudo shuffle, graph.cuh:440-442
                        uint32_t *pool = sh->pools[(size_t)hp * s.P + tp];
                        pool[2 * slot] = (uint32_t)t;
                        pool[2 * slot + 1] = (uint32_t)h;
                        wrote = true;
                    }
                }
        }
        idle = wrote ? 0 : idle + 1;
        if (idle > kMaxIdleRounds) sh->error.store(1);
    }
    *position = rng.pos;
}

// CSR-aligned alias tables: one table per vertex over its out-edge weights (graph.cuh:645-653)
void build_vertex_tables(const gvs_graph *g, gvk_alias_entry *out, uint32_t begin, uint32_t end, std::atomic<int> *error) {
    const uint64_t *flat = g->flat_offsets.data();
    std::vector<float> prob;
    std::vector<uint32_t> alias;
    for (uint32_t u = begin; u < end; u++) {
        const uint64_t off = flat[u], deg = flat[u + 1] - off;
        if (!deg) continue;
        prob.resize(deg), alias.resize(deg);
        if (gvk_alias_build(g->edge_weights.data() + off, deg, prob.data(), alias.data(), 4, out + off) != GVK_OK)
            error->store(1);
    }
}

// node2vec: one table per directed edge (u -> v) over v's out-edges (graph.cuh:656-677)
void build_edge_tables(gvs_sampler *s, const std::vector<uint32_t> *sorted_nb, uint64_t begin, uint64_t end,
                       std::atomic<int> *error) {
    const uint32_t *edges = s->g->edges_uv.data();
    const uint64_t *flat = s->g->flat_offsets.data();
    const float *ew = s->g->edge_weights.data();
    std::vector<float> weights, prob;
    std::vector<uint32_t> alias;
    for (uint64_t e = begin; e < end; e++) {
        const uint32_t u = edges[2 * e], v = edges[2 * e + 1];
        const uint64_t off = flat[v], deg = flat[v + 1] - off;
        if (!deg) continue;
        weights.resize(deg), prob.resize(deg), alias.resize(deg);
        for (uint64_t f = 0; f < deg; f++) {
            const uint32_t x = edges[2 * (off + f) + 1];
            const float w = ew[off + f];
            if (x == u) {
                weights[f] = w / s->p;
            } else {
                const uint32_t *b = sorted_nb->data() + flat[x], *en = sorted_nb->data() + flat[x + 1];
                weights[f] = std::binary_search(b, en, u) ? w : w / s->q;
            }
        }
        const uint64_t base = s->ee_offsets[e];
        if (gvk_alias_build(weights.data(), deg, prob.data(), alias.data(), 4, s->nb_slots.data() + base) != GVK_OK)
            error->store(1);
    }
}

template <class F>
void parallel_ranges(uint64_t n, int num_thread, F &&f) {
    num_thread = std::max(1, num_thread);
    const uint64_t work = (n + num_thread - 1) / num_thread;
    std::vector<std::thread> threads;
    for (int t = 0; t < num_thread; t++) {
        const uint64_t b = work * t, e = std::min(work * (t + 1), n);
        if (b >= e) break;
        threads.emplace_back(f, b, e);
    }
    for (auto &t : threads) t.join();
}

}  // namespace

extern "C" {

gvs_sampler *gvs_sampler_create(const gvs_graph *g, const int32_t *part, const uint32_t *local, int P, uint64_t seed) {
    if (!g || !part || !local || P < 1) {
        gvk_fail(GVK_EINVAL, "gvs_sampler_create: bad argument");
        return nullptr;
    }
    const size_t D = g->edge_weights.size();
    if (!D) {
        gvk_fail(GVK_EINVAL, "gvs_sampler_create: the graph has no edges");
        return nullptr;
    }
    gvs_sampler *s = nullptr;
    try {
        s = new gvs_sampler();
        s->g = g;
        s->P = P;
        s->seed = seed;
        s->location.resize(g->num_vertex);
        for (uint32_t v = 0; v < g->num_vertex; v++) {
            if (part[v] < 0 || part[v] >= P) {
                delete s;
                gvk_fail(GVK_EINVAL, "gvs_sampler_create: part[%u] = %d is outside [0, %d)", v, part[v], P);
                return nullptr;
            }
            s->location[v] = pack_location(part[v], local[v]);
        }
    } catch (const std::bad_alloc &) {
        delete s;
        gvk_fail(GVK_ENOMEM, "gvs_sampler_create: out of host memory");
        return nullptr;
    }
    return s;
}

void gvs_sampler_destroy(gvs_sampler *s) { delete s; }

// WorkerMixin::build_negative_sampler (solver.h:1263-1278): weight of vertex ids[i] in a partition's negative sampler =
// std::pow(vertex_weight, exponent) in single precision — libm's powf, which is what the reference's host code calls
// (numpy's float32 power is a different implementation and differs from it in the last bit for some inputs).
int gvs_negative_weights(const float *vertex_weights, const uint32_t *ids, uint64_t n, float exponent, float *out) {
    if (n && (!vertex_weights || !ids || !out)) return gvk_fail(GVK_EINVAL, "gvs_negative_weights: null argument");
    for (uint64_t i = 0; i < n; i++) out[i] = powf(vertex_weights[ids[i]], exponent);
    return GVK_OK;
}

int gvs_graph_neighbor_tables(const gvs_graph *g, int num_thread, gvk_alias_entry *out) {
    if (!g || !out) return gvk_fail(GVK_EINVAL, "gvs_graph_neighbor_tables: null argument");
    return guarded("gvs_graph_neighbor_tables", [&]() {
        std::atomic<int> error{0};
        parallel_ranges(g->num_vertex, num_thread,
                        [&](uint64_t b, uint64_t e) { build_vertex_tables(g, out, (uint32_t)b, (uint32_t)e, &error); });
        return error.load() ? gvk_fail(GVK_EINVAL, "gvs_graph_neighbor_tables: alias table construction failed") : GVK_OK;
    });
}

int gvs_sampler_prepare(gvs_sampler *s, int mode, float p, float q, int num_thread) {
    if (!s) return gvk_fail(GVK_EINVAL, "gvs_sampler_prepare: null sampler");
    if (mode != GVS_MODE_EDGE && mode != GVS_MODE_WALK && mode != GVS_MODE_BIASED_WALK && mode != GVS_MODE_BIASED_REJECT)
        return gvk_fail(GVK_EINVAL, "gvs_sampler_prepare: unknown mode %d", mode);
    return guarded("gvs_sampler_prepare", [&]() {
        const gvs_graph *g = s->g;
        const uint64_t D = g->edge_weights.size();
        const uint64_t *flat = g->flat_offsets.data();
        std::atomic<int> error{0};
        decltype(s->nb_slots)().swap(s->nb_slots);
        decltype(s->walk_vertex)().swap(s->walk_vertex);
        decltype(s->walk_start)().swap(s->walk_start);
        decltype(s->walk_step)().swap(s->walk_step);
        decltype(s->nb_prob)().swap(s->nb_prob);
        decltype(s->nb_alias)().swap(s->nb_alias);
        decltype(s->ee_offsets)().swap(s->ee_offsets);
        decltype(s->sorted_nb)().swap(s->sorted_nb);
        s->prepared = GVS_MODE_EDGE;
        if (mode == GVS_MODE_EDGE && s->P == 1) {  // single partition: every fill draws from the global table
            const int rc = s->ensure_edge_table();
            if (rc != GVK_OK) return rc;
            s->build_fat(s->edge_slots, nullptr, &s->edge_fat, num_thread);
        }
        if (mode != GVS_MODE_EDGE) {
            const int rc = s->ensure_edge_table();  // walks start from a weighted edge draw
            if (rc != GVK_OK) return rc;
            s->walk_vertex.resize(g->num_vertex);
            parallel_ranges(g->num_vertex, num_thread, [&](uint64_t b, uint64_t e) {
                for (uint64_t v = b; v < e; v++)
                    s->walk_vertex[v] = WalkVertex{s->location[v], flat[v], flat[v + 1] - flat[v], 0};
            });
        }
        if (mode == GVS_MODE_BIASED_REJECT) {
            if (!(p > 0) || !(q > 0)) return gvk_fail(GVK_EINVAL, "gvs_sampler_prepare: p and q must be positive");
            s->p = p;
            s->q = q;
            s->sorted_nb.resize(D);
            for (uint64_t e = 0; e < D; e++) s->sorted_nb[e] = g->edges_uv[2 * e + 1];
            parallel_ranges(g->num_vertex, num_thread, [&](uint64_t b, uint64_t e) {
                for (uint64_t v = b; v < e; v++)
                    std::sort(s->sorted_nb.begin() + flat[v], s->sorted_nb.begin() + flat[v + 1]);
            });
        }
        if (mode == GVS_MODE_WALK || mode == GVS_MODE_BIASED_REJECT) {
            s->nb_slots.resize(D);
            parallel_ranges(g->num_vertex, num_thread, [&](uint64_t b, uint64_t e) {
                build_vertex_tables(g, s->nb_slots.data(), (uint32_t)b, (uint32_t)e, &error);
            });
        } else if (mode == GVS_MODE_BIASED_WALK) {
            if (!(p > 0) || !(q > 0)) return gvk_fail(GVK_EINVAL, "gvs_sampler_prepare: p and q must be positive");
            s->p = p;
            s->q = q;
            s->ee_offsets.resize(D + 1);
            s->ee_offsets[0] = 0;
            for (uint64_t e = 0; e < D; e++) {
                const uint32_t v = g->edges_uv[2 * e + 1];
                s->ee_offsets[e + 1] = s->ee_offsets[e] + (flat[v + 1] - flat[v]);
            }
            const uint64_t total = s->ee_offsets[D];
            s->nb_slots.resize(total);  // sum over edges of deg(head): the reference's node2vec memory hog
            std::vector<uint32_t> sorted_nb(D);
            for (uint64_t e = 0; e < D; e++) sorted_nb[e] = g->edges_uv[2 * e + 1];
            parallel_ranges(g->num_vertex, num_thread, [&](uint64_t b, uint64_t e) {
                for (uint64_t u = b; u < e; u++) std::sort(sorted_nb.begin() + flat[u], sorted_nb.begin() + flat[u + 1]);
            });
            parallel_ranges(D, num_thread,
                            [&](uint64_t b, uint64_t e) { build_edge_tables(s, &sorted_nb, b, e, &error); });
        }
        if (error.load()) return gvk_fail(GVK_EINVAL, "gvs_sampler_prepare: alias table construction failed");
        if ((mode == GVS_MODE_WALK || mode == GVS_MODE_BIASED_REJECT) && D <= fat_slot_limit()) {
            const uint32_t *edges = g->edges_uv.data();
            s->walk_start.resize(D);
            s->walk_step.resize(D);
            parallel_ranges(D, num_thread, [&](uint64_t b, uint64_t e) {
                for (uint64_t i = b; i < e; i++) {
                    const uint64_t other = s->edge_slots[i].alias;
                    s->walk_start[i] = WalkStartSlot{s->edge_slots[i].prob, edges[2 * i], edges[2 * i + 1],
                                                     edges[2 * other], edges[2 * other + 1], {0, 0, 0}};
                }
            });
            parallel_ranges(g->num_vertex, num_thread, [&](uint64_t b, uint64_t e) {
                for (uint64_t v = b; v < e; v++)
                    for (uint64_t i = flat[v]; i < flat[v + 1]; i++) {
                        const gvk_alias_entry &slot = s->nb_slots[i];
                        s->walk_step[i] = WalkStepSlot{slot.prob, slot.alias, edges[2 * i + 1],
                                                       edges[2 * (flat[v] + slot.alias) + 1]};
                    }
            });
        }
        s->prepared = mode;
        return GVK_OK;
    });
}

int gvs_sampler_prepare_column(gvs_sampler *s, int tail_partition, int num_thread) {
    if (!s || tail_partition < 0 || tail_partition >= s->P)
        return gvk_fail(GVK_EINVAL, "gvs_sampler_prepare_column: bad argument");
    return guarded("gvs_sampler_prepare_column", [&]() { return s->ensure_column(tail_partition, num_thread); });
}

int gvs_sampler_fill(gvs_sampler *s, uint32_t *const *pools, uint64_t pool_size, const gvs_fill_config *c) {
    if (!s || !pools || !c) return gvk_fail(GVK_EINVAL, "gvs_sampler_fill: null argument");
    if (c->num_thread < 1) return gvk_fail(GVK_EINVAL, "gvs_sampler_fill: num_thread must be >= 1");
    if (pool_size > (uint64_t)INT32_MAX) return gvk_fail(GVK_EINVAL, "gvs_sampler_fill: pool too large");
    if (c->tail_partition >= s->P) return gvk_fail(GVK_EINVAL, "gvs_sampler_fill: tail_partition out of range");
    if (c->mode == GVS_MODE_EDGE) {
        if (c->sample_batch_size < 1) return gvk_fail(GVK_EINVAL, "gvs_sampler_fill: sample_batch_size must be >= 1");
    } else if (c->mode == GVS_MODE_WALK || c->mode == GVS_MODE_BIASED_WALK || c->mode == GVS_MODE_BIASED_REJECT) {
        if (s->prepared != c->mode)
            return gvk_fail(GVK_EINVAL, "gvs_sampler_fill: call gvs_sampler_prepare for this mode first");
        if (c->augmentation_step < 1)
            return gvk_fail(GVK_EINVAL, "`augmentation_step` should be a positive integer");
        if (c->augmentation_step > c->walk_length)
            return gvk_fail(GVK_EINVAL, "`random_walk_length` should be no less than `augmentation_step`");
        if (c->walk_batch < 1) return gvk_fail(GVK_EINVAL, "gvs_sampler_fill: walk_batch must be >= 1");
        if (c->shuffle_base < 1 || pool_size % (uint64_t)c->shuffle_base)
            return gvk_fail(GVK_EINVAL,
                            "Can't perform pseudo shuffle on %llu elements by a shuffle base of %d. Try setting the "
                            "episode size to a multiple of the shuffle base",
                            (unsigned long long)pool_size, c->shuffle_base);
    } else {
        return gvk_fail(GVK_EINVAL, "gvs_sampler_fill: unknown mode %d", c->mode);
    }
    for (int hp = 0; hp < s->P; hp++)
        for (int tp = 0; tp < s->P; tp++)
            if ((c->tail_partition < 0 || tp == c->tail_partition) && !pools[(size_t)hp * s->P + tp])
                return gvk_fail(GVK_EINVAL, "gvs_sampler_fill: pool (%d, %d) is null", hp, tp);
    return guarded("gvs_sampler_fill", [&]() {
        if ((int)s->positions.size() < c->num_thread) s->positions.resize(c->num_thread, 0);
        const bool column_mode = c->mode == GVS_MODE_EDGE && c->tail_partition >= 0 && s->P > 1;
        if (!column_mode) {
            const int rc = s->ensure_edge_table();
            if (rc != GVK_OK) return rc;
            if (c->mode == GVS_MODE_EDGE) s->build_fat(s->edge_slots, nullptr, &s->edge_fat, c->num_thread);
        }
        if (column_mode) {
            const int rc = s->ensure_column(c->tail_partition, c->num_thread);
            if (rc != GVK_OK) return rc;
        }
        FillShared sh;
        sh.s = s;
        sh.pools = pools;
        sh.pool_size = pool_size;
        sh.c = *c;
        const int64_t work = ((int64_t)pool_size + c->num_thread - 1) / c->num_thread;  // solver.h:616-617
        const bool timing = getenv("GVS_TIMING") != nullptr;
        std::vector<double> slice_ms(timing ? c->num_thread : 0);
        const auto now = [] { return std::chrono::duration<double, std::milli>(
                                         std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const std::function<void(int)> job = [&](int t) {
            const int64_t b = work * t, e = std::min(work * (t + 1), (int64_t)pool_size);
            uint64_t *position = &s->positions[t];
            const double t0 = timing ? now() : 0;
            if (c->mode == GVS_MODE_EDGE)
                fill_edges(&sh, t, b, e, position);
            else
                if (s->walk_step.empty())
                    fill_walks<false>(&sh, t, b, e, position);
                else
                    fill_walks<true>(&sh, t, b, e, position);
            if (timing) slice_ms[t] = now() - t0;
        };
        const double t0 = timing ? now() : 0;
        s->pool.run(c->num_thread, c->os_threads, c->cpu_offset, job);
        if (timing) {
            double total = now() - t0, sum = 0, mx = 0;
            int slow = 0;
            for (double x : slice_ms) {
                sum += x;
                mx = std::max(mx, x);
                slow += x > 5;
            }
            fprintf(stderr, "[gvs] fill %.1f ms: %d slices, sum %.1f ms, max %.2f ms, %d slices > 5 ms\n", total,
                    c->num_thread, sum, mx, slow);
        }
        if (sh.error.load())
            return gvk_fail(GVK_EINVAL,
                            "gvs_sampler_fill: a block pool cannot be filled (no positive sample falls into it); "
                            "use fewer partitions for this graph");
        return GVK_OK;
    });
}

uint64_t gvs_sampler_stream_position(const gvs_sampler *s, int thread) {
    return s && thread >= 0 && thread < (int)s->positions.size() ? s->positions[thread] : 0;
}

int gvs_sampler_set_stream_position(gvs_sampler *s, int thread, uint64_t position) {
    if (!s || thread < 0) return gvk_fail(GVK_EINVAL, "gvs_sampler_set_stream_position: bad argument");
    if ((int)s->positions.size() <= thread) s->positions.resize(thread + 1, 0);
    s->positions[thread] = position;
    return GVK_OK;
}

const float *gvs_sampler_edge_prob(const gvs_sampler *s) {
    gvs_sampler *m = const_cast<gvs_sampler *>(s);
    if (!m || m->ensure_edge_table() != GVK_OK) return nullptr;
    m->materialize_split_table();
    return m->edge_prob.data();
}
const uint64_t *gvs_sampler_edge_alias(const gvs_sampler *s) {
    gvs_sampler *m = const_cast<gvs_sampler *>(s);
    if (!m || m->ensure_edge_table() != GVK_OK) return nullptr;
    m->materialize_split_table();
    return m->edge_alias.data();
}
static void materialize_split_neighbor_tables(gvs_sampler *m) {
    if (m->nb_prob.size() == m->nb_slots.size()) return;
    m->nb_prob.resize(m->nb_slots.size());
    m->nb_alias.resize(m->nb_slots.size());
    for (size_t i = 0; i < m->nb_slots.size(); i++) {
        m->nb_prob[i] = m->nb_slots[i].prob;
        m->nb_alias[i] = m->nb_slots[i].alias;
    }
}
const float *gvs_sampler_neighbor_prob(const gvs_sampler *s) {
    if (!s) return nullptr;
    materialize_split_neighbor_tables(const_cast<gvs_sampler *>(s));
    return s->nb_prob.data();
}
const uint32_t *gvs_sampler_neighbor_alias(const gvs_sampler *s) {
    if (!s) return nullptr;
    materialize_split_neighbor_tables(const_cast<gvs_sampler *>(s));
    return s->nb_alias.data();
}
const gvk_alias_entry *gvs_sampler_neighbor_slots(const gvs_sampler *s) { return s ? s->nb_slots.data() : nullptr; }
const uint64_t *gvs_sampler_edge_edge_offsets(const gvs_sampler *s) { return s ? s->ee_offsets.data() : nullptr; }

int gvs_sampler_column(const gvs_sampler *s, int tail_partition, uint64_t *count, const uint64_t **edge_ids,
                       const float **prob, const uint64_t **alias) {
    if (!s || tail_partition < 0 || tail_partition >= (int)s->columns.size() ||
        s->columns[tail_partition].slots.empty())
        return gvk_fail(GVK_EINVAL, "gvs_sampler_column: column %d has not been built", tail_partition);
    const gvs_sampler::Column &col = s->columns[tail_partition];
    if (count) *count = col.edge_ids.size();
    if (edge_ids) *edge_ids = col.edge_ids.data();
    if (prob) *prob = col.prob.data();
    if (alias) *alias = col.alias.data();
    return GVK_OK;
}

void gvs_host_uniforms(uint64_t seed, uint32_t stream, uint64_t first, size_t n, double *out) {
    HostRng rng(seed, stream, first);
    for (size_t i = 0; i < n; i++) out[i] = rng.next();
}

}  // extern "C"
