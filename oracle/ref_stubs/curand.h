// TEST INFRASTRUCTURE ONLY: emulated cuRAND host API for the reference's sampler threads (core/solver.h:921-967).
// A "generator" is a counter; the uniforms themselves come from a source the harness installs (the same per-thread
// Philox stream the oracle and the product's samplers consume), so that the reference's sampler code — compiled as
// written — can be compared with them draw for draw.
#pragma once
#include <cstddef>
typedef int curandStatus_t;
enum { CURAND_STATUS_SUCCESS = 0 };
enum curandRngType_t { CURAND_RNG_PSEUDO_DEFAULT = 100 };
struct gvref_generator { int index; unsigned long long position; };
typedef gvref_generator *curandGenerator_t;
typedef void (*gvref_uniform_source_t)(int generator_index, unsigned long long position, double *out, size_t n);
extern gvref_uniform_source_t gvref_uniform_source;
extern int gvref_generator_count;
inline curandStatus_t curandCreateGenerator(curandGenerator_t *g, curandRngType_t) {
    *g = new gvref_generator{gvref_generator_count++, 0};
    return CURAND_STATUS_SUCCESS;
}
inline curandStatus_t curandDestroyGenerator(curandGenerator_t g) { delete g; return CURAND_STATUS_SUCCESS; }
inline curandStatus_t curandSetPseudoRandomGeneratorSeed(curandGenerator_t, unsigned long long) { return CURAND_STATUS_SUCCESS; }
template <class S>
inline curandStatus_t curandSetStream(curandGenerator_t, S) { return CURAND_STATUS_SUCCESS; }
inline curandStatus_t curandGenerateUniformDouble(curandGenerator_t g, double *out, size_t n) {
    gvref_uniform_source(g->index, g->position, out, n);
    g->position += n;
    return CURAND_STATUS_SUCCESS;
}
