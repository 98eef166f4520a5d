// TEST INFRASTRUCTURE ONLY — never imported, linked or executed by the product path.
//
// Host build of the REFERENCE's own arithmetic, compiled from the sources where they lie
// under /root/reference/include (nothing is copied into this repo):
//   instance/model/graph.h:40-85   LINE<Vector>::forward / backward<optimizer_type>
//   core/optimizer.h:77-79,132-134 LRSchedule::linear_schedule, Optimizer::apply_schedule
//   core/optimizer.h:161-210       sgd / momentum / adagrad / rmsprop / adam update rules
//   util/math.h:30-33              sigmoid
//   base/vector.h:31-69            Vector<dim, float>
//   util/gpu.cuh:24-37             FOR / SUM (host branch: plain loop / identity)
// The reference has no host trainer: its per-sample loop exists only inside the CUDA
// kernels (instance/gpu/graph.cuh:54-94, 122-166, 196-241, 265-278).  The loops below
// RESTATE that control flow (negatives first, then the positive; progressive vertex
// buffer; loss / (1 + k * negative_weight)) and call the reference's own model and
// optimizer code for every arithmetic step, so a result from this harness is "the
// reference's arithmetic in the reference's order, executed sequentially".
//
// Built by oracle/Makefile into oracle/_ref/libgvref.so (git-ignored, travels with gpurun).
// Used (a) to validate oracle/gv_oracle.c, (b) to generate tests/golden/*.npz
// (tests/golden/make_golden.py), (c) as bench.py's cpu_baseline (kind = "reference").

#define __host__
#define __device__
#define __global__
#include <math.h>  // global float overloads of exp/log/sqrt, as nvcc provides on the host side
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>
#include <glog/logging.h>  // oracle/ref_stubs

namespace graphvite {
// util/common.h:28-29 (that header cannot be included: it drags io.h -> glog flags)
const float kEpsilon = 1e-15;
const int kAuto = 0;
}  // namespace graphvite

#include "util/math.h"
#include "util/gpu.cuh"
#include "base/vector.h"
#include "core/optimizer.h"
#include "instance/model/graph.h"

using namespace graphvite;

namespace {

struct Args {
    float *vertex, *context;
    float *vm1, *cm1, *vm2, *cm2;  // moment tables (may be null)
    const uint32_t *batch;         // {tail, head} records, solver.h:911 + gpu/graph.cuh:55-57
    const uint32_t *negatives;     // [B * k]
    float *loss;                   // [B]
    int begin, end, k;
    float negative_weight;
};

// One pass of the kernel body over samples [begin, end) — gpu/graph.cuh:54-94 (0 moment),
// :122-166 (1 moment), :196-241 (2 moments).
template <size_t dim, OptimizerType type>
void train_range(const Args &a, const Optimizer &optimizer) {
    typedef Vector<dim, float> Vec;
    typedef LINE<Vec> Model;
    Vec *vertex_embeddings = reinterpret_cast<Vec *>(a.vertex);
    Vec *context_embeddings = reinterpret_cast<Vec *>(a.context);
    Vec *vm1 = reinterpret_cast<Vec *>(a.vm1), *cm1 = reinterpret_cast<Vec *>(a.cm1);
    Vec *vm2 = reinterpret_cast<Vec *>(a.vm2), *cm2 = reinterpret_cast<Vec *>(a.cm2);
    Vec vertex_buffer;
    for (int sample_id = a.begin; sample_id < a.end; sample_id++) {
        uint32_t head_id = a.batch[sample_id * 2 + 1];
        Vec &vertex = vertex_embeddings[head_id];
        vertex_buffer = vertex;
        float sample_loss = 0;
        for (int s = 0; s <= a.k; s++) {
            uint32_t tail_id;
            int label;
            if (s < a.k) {
                tail_id = a.negatives[(size_t)sample_id * a.k + s];
                label = 0;
            } else {
                tail_id = a.batch[sample_id * 2];
                label = 1;
            }
            Vec &context = context_embeddings[tail_id];
            float logit;
            Model::forward(vertex_buffer, context, logit);
            float prob = sigmoid(logit);
            float gradient, weight;
            if (label) {
                gradient = prob - 1;
                weight = 1;
                sample_loss += weight * -log(prob + kEpsilon);
            } else {
                gradient = prob;
                weight = a.negative_weight;
                sample_loss += weight * -log(1 - prob + kEpsilon);
            }
            if (type == kSGD)
                Model::template backward<kSGD>(vertex_buffer, context, gradient, optimizer, weight);
            else if (type == kAdam)
                Model::template backward<kAdam>(vertex_buffer, context, vm1[head_id], cm1[tail_id], vm2[head_id],
                                                cm2[tail_id], gradient, optimizer, weight);
            else
                Model::template backward<(type == kSGD || type == kAdam) ? kMomentum : type>(
                    vertex_buffer, context, vm1[head_id], cm1[tail_id], gradient, optimizer, weight);
        }
        a.loss[sample_id] = sample_loss / (1 + a.k * a.negative_weight);
        vertex = vertex_buffer;
    }
}

template <size_t dim>
void train_dim(int type, const Args &a, const Optimizer &o) {
    switch (type) {
        case kSGD: train_range<dim, kSGD>(a, o); break;
        case kMomentum: train_range<dim, kMomentum>(a, o); break;
        case kAdaGrad: train_range<dim, kAdaGrad>(a, o); break;
        case kRMSprop: train_range<dim, kRMSprop>(a, o); break;
        case kAdam: train_range<dim, kAdam>(a, o); break;
    }
}

bool train_any(int dim, int type, const Args &a, const Optimizer &o) {
    switch (dim) {
        case 32: train_dim<32>(type, a, o); return true;
        case 64: train_dim<64>(type, a, o); return true;
        case 96: train_dim<96>(type, a, o); return true;
        case 128: train_dim<128>(type, a, o); return true;
        case 256: train_dim<256>(type, a, o); return true;
        case 512: train_dim<512>(type, a, o); return true;
    }
    return false;
}

// Helper classes core/optimizer.h:272-319; hp = {momentum | alpha | beta1, beta2, epsilon}
Optimizer make_optimizer(int type, float lr, float weight_decay, const float *hp) {
    switch (type) {
        case kMomentum: return Momentum(lr, weight_decay, hp[0], "constant");
        case kAdaGrad: return AdaGrad(lr, weight_decay, hp[2], "constant");
        case kRMSprop: return RMSprop(lr, weight_decay, hp[0], hp[2], "constant");
        case kAdam: return Adam(lr, weight_decay, hp[0], hp[1], hp[2], "constant");
        default: return SGD(lr, weight_decay, "constant");
    }
}

}  // namespace

extern "C" {

// Sequential batch. Returns 0, or -1 for an unsupported dim / optimizer type.
int gvref_train(int dim, int optimizer_type, float *vertex, float *context, float *vm1, float *cm1, float *vm2,
                float *cm2, const uint32_t *batch, const uint32_t *negatives, float *loss, int batch_size,
                int num_negative, float lr, float weight_decay, float negative_weight, const float *hp) {
    if (optimizer_type < 0 || optimizer_type >= kNumOptimizer) return -1;
    Optimizer o = make_optimizer(optimizer_type, lr, weight_decay, hp);
    Args a = {vertex, context, vm1, cm1, vm2, cm2, batch, negatives, loss, 0, batch_size, num_negative,
              negative_weight};
    return train_any(dim, optimizer_type, a, o) ? 0 : -1;
}

// Hogwild over std::threads, each thread sequential on a contiguous slice of the batch —
// the CPU analogue of the reference's racing warps. SGD only.
int gvref_train_mt(int dim, float *vertex, float *context, const uint32_t *batch, const uint32_t *negatives,
                   float *loss, int batch_size, int num_negative, float lr, float weight_decay,
                   float negative_weight, int num_thread) {
    if (num_thread < 1) return -1;
    float hp[3] = {0, 0, 0};
    Optimizer o = make_optimizer(kSGD, lr, weight_decay, hp);
    std::vector<std::thread> threads;
    int work = (batch_size + num_thread - 1) / num_thread;
    bool ok = true;
    for (int t = 0; t < num_thread; t++) {
        int b = work * t, e = std::min(work * (t + 1), batch_size);
        if (b >= e) break;
        threads.emplace_back([=, &o, &ok]() {
            Args a = {vertex, context, nullptr, nullptr, nullptr, nullptr, batch, negatives, loss, b, e,
                      num_negative, negative_weight};
            if (!train_any(dim, kSGD, a, o)) ok = false;
        });
    }
    for (auto &t : threads) t.join();
    return ok ? 0 : -1;
}

// gpu/graph.cuh:265-278 with LINE::forward
int gvref_predict(int dim, const float *vertex, const float *context, const uint32_t *batch, float *logits,
                  int batch_size) {
#define GVREF_PREDICT(D)                                                                         \
    case D: {                                                                                    \
        typedef Vector<D, float> Vec;                                                            \
        const Vec *v = reinterpret_cast<const Vec *>(vertex);                                    \
        const Vec *c = reinterpret_cast<const Vec *>(context);                                   \
        for (int s = 0; s < batch_size; s++)                                                     \
            LINE<Vec>::forward(v[batch[s * 2 + 1]], c[batch[s * 2]], logits[s]);                 \
        return 0;                                                                                \
    }
    switch (dim) {
        GVREF_PREDICT(32) GVREF_PREDICT(64) GVREF_PREDICT(96) GVREF_PREDICT(128) GVREF_PREDICT(256)
        GVREF_PREDICT(512)
    }
#undef GVREF_PREDICT
    return -1;
}

float gvref_sigmoid(float x) { return sigmoid(x); }

// optimizer.h:132-134 with the named schedule (0 = constant, 1 = linear)
float gvref_lr(float init_lr, int linear, int batch_id, int num_batch) {
    SGD o(init_lr, 0, linear ? "linear" : "constant");
    o.apply_schedule(batch_id, num_batch);
    return o.lr;
}

}  // extern "C"
