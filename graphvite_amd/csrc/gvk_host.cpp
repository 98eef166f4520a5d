// gvk_host.cpp — host-side entry points of include/gvk.h: error plumbing and the alias-table builder.
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include <roctracer/roctx.h>

#include "gvk.h"
#include "gvk_internal.h"

namespace {
thread_local char g_error[512] = "";
}

// Named ranges on the calling thread for rocprofv3 (--marker-trace): the reference's USE_TIMER scopes
// (include/util/time.h:28-60; "Sample threads", "Train Batch", "Train Kernel" at include/core/solver.h:622,645,1526-1552)
// become roctx ranges, so a trace shows which host phase issued which kernels and where the pipeline overlaps.
extern "C" void gvk_range_push(const char *name) { roctxRangePushA(name ? name : ""); }
extern "C" void gvk_range_pop(void) { roctxRangePop(); }

int gvk_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code;
}

namespace {

// Vose's method in the reference's order (include/base/alias_table.cuh:84-128): normalise by the mean
// computed in double, two FIFO queues filled in index order, `large` donates to `little`, leftovers
// alias themselves.  `little` only ever grows by indices that left `large`, so each index is queued
// there once: a flat array with a moving head is enough.  `large` re-queues its head, so it is a ring.
template <class Index>
void alias_build(const float *w, size_t n, float *prob, Index *alias) {
    double norm = 0;
    for (size_t i = 0; i < n; i++) norm += w[i];
    norm /= n;
    for (size_t i = 0; i < n; i++) {
        prob[i] = w[i];
        prob[i] /= norm;
    }
    std::vector<Index> little(n), large(n);
    size_t little_head = 0, little_tail = 0, large_head = 0, large_tail = 0, large_count = 0;
    for (size_t i = 0; i < n; i++) {
        if (prob[i] < 1) {
            little[little_tail++] = (Index)i;
        } else {
            large[large_tail] = (Index)i;
            large_tail = large_tail + 1 == n ? 0 : large_tail + 1;
            large_count++;
        }
    }
    while (little_head < little_tail && large_count) {
        const Index i = little[little_head++], j = large[large_head];
        large_head = large_head + 1 == n ? 0 : large_head + 1;
        large_count--;
        alias[i] = j;
        prob[j] = prob[i] + prob[j] - 1;
        if (prob[j] < 1) {
            little[little_tail++] = j;
        } else {
            large[large_tail] = j;
            large_tail = large_tail + 1 == n ? 0 : large_tail + 1;
            large_count++;
        }
    }
    for (; little_head < little_tail; little_head++) alias[little[little_head]] = little[little_head];
    for (; large_count; large_count--) {
        alias[large[large_head]] = large[large_head];
        large_head = large_head + 1 == n ? 0 : large_head + 1;
    }
}

}  // namespace

int gvk_cpu_budget(void) {
    unsigned n = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = (unsigned)CPU_COUNT(&set);
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[64];
        double period = 0;
        if (fscanf(f, "%63s %lf", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0)
            n = std::min<unsigned>(n, (unsigned)std::max(1, (int)(atof(quota) / period)));
        fclose(f);
    }
    return (int)std::max(1u, n);
}

extern "C" {

int gvk_alias_build(const float *weights, size_t n, float *prob, void *alias, int index_bytes,
                    gvk_alias_entry *packed) {
    if (!weights || !prob || !alias) return gvk_fail(GVK_EINVAL, "gvk_alias_build: null pointer");
    if (n == 0) return gvk_fail(GVK_EINVAL, "gvk_alias_build: invalid sampling distribution (empty)");
    if (index_bytes == 4 && n > 0xffffffffu) return gvk_fail(GVK_EINVAL, "gvk_alias_build: n needs 8-byte indexes");
    if (index_bytes != 4 && index_bytes != 8) return gvk_fail(GVK_EINVAL, "gvk_alias_build: index_bytes must be 4 or 8");
    if (packed && index_bytes != 4) return gvk_fail(GVK_EINVAL, "gvk_alias_build: packed form needs 4-byte indexes");
    try {
        if (index_bytes == 4)
            alias_build(weights, n, prob, static_cast<uint32_t *>(alias));
        else
            alias_build(weights, n, prob, static_cast<uint64_t *>(alias));
    } catch (const std::bad_alloc &) {
        return gvk_fail(GVK_ENOMEM, "gvk_alias_build: out of host memory");
    }
    if (packed) {
        const uint32_t *a = static_cast<const uint32_t *>(alias);
        for (size_t i = 0; i < n; i++) {
            packed[i].prob = prob[i];
            packed[i].alias = a[i];
        }
    }
    return GVK_OK;
}

int gvk_class_table_build(const float *weights, size_t n, gvk_class_entry *out, uint32_t *num_class) {
    if (!weights || !out || !num_class) return gvk_fail(GVK_EINVAL, "gvk_class_table_build: null pointer");
    if (n == 0 || n > 0xffffffffu) return gvk_fail(GVK_EINVAL, "gvk_class_table_build: 1 .. 2^32 - 1 rows");
    try {
        // classes: maximal runs of consecutive rows of equal weight (rows sorted by degree: one run per distinct weight)
        std::vector<float> mass;
        size_t classes = 0;
        for (size_t i = 0; i < n;) {
            size_t j = i + 1;
            while (j < n && weights[j] == weights[i]) j++;
            out[classes].first = (uint32_t)i, out[classes].count = (uint32_t)(j - i);
            mass.push_back((float)((double)(j - i) * (double)weights[i]));
            classes++;
            i = j;
        }
        std::vector<float> prob(classes);
        std::vector<uint32_t> alias(classes);
        alias_build(mass.data(), classes, prob.data(), alias.data());
        for (size_t c = 0; c < classes; c++) out[c].prob = prob[c], out[c].alias = alias[c];
        *num_class = (uint32_t)classes;
    } catch (const std::bad_alloc &) {
        return gvk_fail(GVK_ENOMEM, "gvk_class_table_build: out of host memory");
    }
    return GVK_OK;
}

const char *gvk_last_error(void) { return g_error; }

const char *gvk_version(void) { return "gvk 0.1 (gfx950)"; }

}  // extern "C"
