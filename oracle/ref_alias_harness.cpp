// TEST INFRASTRUCTURE ONLY — never linked into or loaded by the product.
//
// Host build of the reference's OWN alias table (include/base/alias_table.cuh:48-175 with base/memory.h), compiled from
// /root/reference where it lies — nothing is copied into this repo — against an emulated CUDA runtime
// (ref_stubs/cuda_runtime.h: pinned / device allocations become malloc).  It pins oracle/gv_oracle.c's restatement
// (gvo_alias_build, gvo_alias_sample) and the product's gvk_alias_build against the reference's code itself:
// tests/test_oracle_cpu.py, fixtures in tests/golden/ for the GPU box.
//
//   hipcc -x hip --cuda-host-only ... (oracle/Makefile) -> oracle/_ref/libgvref_alias.so
#include <cstdint>
#include <vector>

#include "base/alias_table.cuh"

extern "C" {

// AliasTable<float, uint32_t>::build on `n` weights -> prob[n], alias[n]
int gvref_alias_build(const float *weights, uint32_t n, float *prob, uint32_t *alias) {
    if (!n) return -1;
    graphvite::AliasTable<float, uint32_t> table(-1);
    table.build(std::vector<float>(weights, weights + n));
    for (uint32_t i = 0; i < n; i++) {
        prob[i] = table.prob_table[i];
        alias[i] = table.alias_table[i];
    }
    return 0;
}

// AliasTable<float, uint32_t>::sample(rand1, rand2) for m pairs of uniforms over a table built from `weights`
int gvref_alias_sample(const float *weights, uint32_t n, const double *rand, uint32_t m, uint32_t *out) {
    if (!n) return -1;
    graphvite::AliasTable<float, uint32_t> table(-1);
    table.build(std::vector<float>(weights, weights + n));
    for (uint32_t i = 0; i < m; i++) out[i] = table.sample(rand[2 * i], rand[2 * i + 1]);
    return 0;
}

// The 64-bit instantiation the edge sampler uses (AliasTable<Float, Index = size_t>, solver.h:330)
int gvref_alias_build64(const float *weights, uint64_t n, float *prob, uint64_t *alias) {
    if (!n) return -1;
    graphvite::AliasTable<float, size_t> table(-1);
    table.build(std::vector<float>(weights, weights + n));
    for (uint64_t i = 0; i < n; i++) {
        prob[i] = table.prob_table[i];
        alias[i] = table.alias_table[i];
    }
    return 0;
}

}  // extern "C"
