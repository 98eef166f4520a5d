"""TEST INFRASTRUCTURE: builds tests/hostdev/build/libsimt_chains.so — the SOURCE of the hub chains' device functions
(train_long_chains_one_round, train_long_chains_in_rounds, train_short_chains and the cross-lane helpers they use) cut out of
graphvite_amd/csrc/gvk_chains.hip (and gvk_device.hpp) as written, compiled for the host over tests/hostdev/simt.h (one host thread per lane,
cross-lane operations as rendezvous of a wavefront's 64 threads, __syncthreads as a barrier of 256)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = os.path.join(ROOT, "graphvite_amd", "csrc", "gvk_device.hpp")  # TrainArgs, cross-lane helpers, rows in registers, sigmoid
SOURCE = os.path.join(ROOT, "graphvite_amd", "csrc", "gvk_chains.hip")  # HotArgs and the chains' device functions
HOSTDEV = os.path.join(ROOT, "tests", "hostdev")
OUT = os.path.join(HOSTDEV, "build", "libsimt_chains.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"

WRAPPER = r'''
}  // namespace

simt::Group *simt::group = nullptr;

// ---- the chain side of one unit as train_hot_kernel runs it: train_long_chains for every long-chain workgroup, then
// train_short_chains for every workgroup of short chains; one workgroup at a time
namespace {
struct Block {
    unsigned index;
    int dim, role;  // role 0: long chains, 1: short chains
    uint32_t block;
    const TrainArgs *a;
    const HotArgs *h;
};

template <int DIM, int G>
void block_of(const Block &t) {
    if (t.role == 1) train_short_chains<DIM, G>(*t.a, *t.h, t.block);
    else if (t.h->round_steps) train_long_chains_in_rounds<DIM, G>(*t.a, *t.h, t.block);
    else train_long_chains_one_round<DIM, G>(*t.a, *t.h, t.block);
}

void *block_main(void *p) {
    const Block &t = *static_cast<const Block *>(p);
    threadIdx.x = t.index;
    switch (t.dim) {
        case 32: block_of<32, 8>(t); break;
        case 64: block_of<64, 16>(t); break;
        case 96: block_of<96, 8>(t); break;
        case 128: block_of<128, 16>(t); break;
        case 256: block_of<256, 16>(t); break;
        case 512: block_of<512, 32>(t); break;
    }
    return nullptr;
}
}  // namespace

extern "C" int simt_unit_chains(int dim, float *vertex, float *context, uint32_t hot_vertex, uint32_t hot_context, float wd,
                                float neg_weight, const uint32_t *chain_start, const uint32_t *entries, const uint32_t *long_list,
                                const uint32_t *short_list, uint32_t long_capacity, uint32_t cap, uint32_t round_steps, const float *from, float *to, float lr,
                                float log2_decay_positive, float log2_decay_negative, int long_blocks, int short_blocks) {
    if (dim != 32 && dim != 64 && dim != 96 && dim != 128 && dim != 256 && dim != 512) return -1;
    static simt::Group g;
    simt::group = &g;
    TrainArgs a;
    HotArgs h;
    memset(&a, 0, sizeof(a)), memset(&h, 0, sizeof(h));
    a.vertex = vertex, a.context = context, a.hot_vertex = hot_vertex, a.hot_context = hot_context, a.wd = wd, a.neg_weight = neg_weight;
    h.chain_start = chain_start, h.entries = entries, h.long_list = long_list, h.short_list = short_list;
    h.from = from, h.to = to, h.chains = hot_vertex + hot_context, h.long_capacity = long_capacity, h.cap = cap, h.round_steps = round_steps, h.lr = lr;
    h.log2_decay_positive = log2_decay_positive, h.log2_decay_negative = log2_decay_negative;
    h.long_blocks = long_blocks, h.short_blocks = short_blocks;
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setstacksize(&attr, 1 << 20);
    for (int role = 0; role < 2; role++)
        for (int b = 0; b < (role ? short_blocks : long_blocks); b++) {
            pthread_barrier_init(&g.barrier, nullptr, simt::kThreads);
            for (auto &w : g.wave) pthread_barrier_init(&w.barrier, nullptr, simt::kWave);
            static Block blocks[simt::kThreads];
            pthread_t threads[simt::kThreads];
            for (int i = 0; i < simt::kThreads; i++) {
                blocks[i] = Block{(unsigned)i, dim, role, (uint32_t)b, &a, &h};
                if (pthread_create(&threads[i], &attr, block_main, &blocks[i]) != 0) return -2;
            }
            for (int i = 0; i < simt::kThreads; i++) pthread_join(threads[i], nullptr);
            pthread_barrier_destroy(&g.barrier);
            for (auto &w : g.wave) pthread_barrier_destroy(&w.barrier);
        }
    return 0;
}
'''


def cut(text, begin, end, include_end=True):
    a = text.index(begin)
    b = text.index(end, a)
    return text[a:b + (len(end) if include_end else 0)]


def host_source():
    common, text = open(COMMON).read(), open(SOURCE).read()
    pieces = [
        '#include "simt.h"\n#include <algorithm>\nstruct gvk_alias_entry;\nstruct gvk_class_entry;\nnamespace {\nconstexpr int kBlock = 256;\n#define GVK_SIMT_HOST 1\n',
        cut(common, "struct TrainArgs {", "\n};\n"),
        cut(common, "template <int CTRL>\n__device__ __forceinline__ float dpp(float x) {", "// ---- Philox4x32-10", include_end=False),
        cut(common, "template <int DIM, int G>\nstruct Layout {", "// ---- arithmetic", include_end=False),
        cut(common, "__device__ __forceinline__ float sigmoidf(float x) {", "\n}\n"),
        # HotArgs, the chains of 1 .. 7 entries, the idle rows, train_long_chains: everything up to the kernel itself
        cut(text, "// The logistic function of a chain step", "// ---- moment optimizers (Momentum, AdaGrad, RMSprop, Adam", include_end=False),
    ]
    body = "\n".join(pieces)
    assert "asm volatile" not in body and "train_long_chains_in_rounds" in body
    return body + "\n" + WRAPPER


def build():
    """Rebuilds when the kernel source, the stand-in or this file is newer than the library; returns its path."""
    inputs = [COMMON, SOURCE, os.path.join(HOSTDEV, "simt.h"), os.path.abspath(__file__)]
    if os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(p) for p in inputs):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    generated = os.path.join(os.path.dirname(OUT), "simt_chains.cpp")
    with open(generated, "w") as f:
        f.write(host_source())
    compiler = CLANG if os.path.exists(CLANG) else "clang++"
    subprocess.check_call([compiler, "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-I" + HOSTDEV, generated, "-o", OUT])
    return OUT
