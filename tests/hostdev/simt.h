// simt.h — TEST INFRASTRUCTURE: a stand-in for one 256-thread workgroup of wave64 hardware on the CPU, so that the SOURCE of a
// device function (extracted from graphvite_amd/csrc/gvk_chains.hip by tests/simt_build.py) runs as written: one host thread per
// lane, the cross-lane operations (wavefront shuffles, DPP, ballot, the fp32 matrix instruction) as rendezvous of the 64 threads
// of a wavefront, __syncthreads as a barrier of all 256.  Valid for code whose branches around cross-lane operations are the same
// for a whole wavefront (long_chain_gram is).  Lane maps: /opt/skills/guides/cdna_hip_programming.md (v_mfma_f32_16x16x4_f32:
// "A[l&15][k=l>>4] / B[k=l>>4][l&15]", "col=lane&15, row=(lane>>4)*4+reg_idx"); DPP controls: quad_perm 0x00-0xFF,
// row_half_mirror 0x141, row_mirror 0x140, row_newbcast 0x150 + n.  Nothing in graphvite_amd/ includes this file.
#pragma once
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <string.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define __device__
#define __global__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

namespace simt {
constexpr int kThreads = 256, kWave = 64;
struct Wave {
    pthread_barrier_t barrier;
    uint32_t a[kWave], b[kWave];
};
struct Group {
    pthread_barrier_t barrier;
    Wave wave[kThreads / kWave];
};
extern Group *group;
struct Index { unsigned x; };
}  // namespace simt
static thread_local simt::Index threadIdx;

namespace simt {
inline Wave &mine() { return group->wave[threadIdx.x / kWave]; }
inline int lane() { return (int)(threadIdx.x % kWave); }
// every lane hands in a word and reads the word of lane `source`
inline uint32_t exchange(uint32_t value, int source) {
    Wave &w = mine();
    w.a[lane()] = value;
    pthread_barrier_wait(&w.barrier);
    const uint32_t r = w.a[source];
    pthread_barrier_wait(&w.barrier);
    return r;
}
inline uint32_t bits(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
inline float real(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }
inline int dpp_source(int l, int ctrl) {
    if (ctrl >= 0 && ctrl <= 0xFF) return (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);  // quad_perm
    if (ctrl == 0x141) return (l & ~7) | (7 - (l & 7));                               // row_half_mirror
    if (ctrl == 0x140) return (l & ~15) | (15 - (l & 15));                            // row_mirror
    if (ctrl >= 0x150 && ctrl <= 0x15F) return (l & ~15) | (ctrl - 0x150);            // row_newbcast
    __builtin_trap();
}
inline int update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    (void)old, (void)bound_ctrl;
    if (row_mask != 0xf || bank_mask != 0xf) __builtin_trap();
    return (int)exchange((uint32_t)src, dpp_source(lane(), ctrl));
}
inline unsigned long long ballot(bool predicate) {
    Wave &w = mine();
    w.a[lane()] = predicate ? 1u : 0u;
    pthread_barrier_wait(&w.barrier);
    unsigned long long mask = 0;
    for (int i = 0; i < kWave; i++) mask |= (unsigned long long)w.a[i] << i;
    pthread_barrier_wait(&w.barrier);
    return mask;
}
// v_mfma_f32_16x16x4_f32: D = A (16 x 4) B (4 x 16) + C, each product and sum in fp32 in the order k = 0 .. 3
inline f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
    Wave &w = mine();
    const int l = lane();
    w.a[l] = bits(a), w.b[l] = bits(b);
    pthread_barrier_wait(&w.barrier);
    f32x4 d = c;
    for (int v = 0; v < 4; v++) {
        const int i = 4 * (l >> 4) + v, j = l & 15;
        float sum = d[v];
        for (int k = 0; k < 4; k++) sum = fmaf(real(w.a[i + 16 * k]), real(w.b[j + 16 * k]), sum);
        d[v] = sum;
    }
    pthread_barrier_wait(&w.barrier);
    return d;
}
}  // namespace simt

inline void __syncthreads() { pthread_barrier_wait(&simt::group->barrier); }
// __shfl(value, source, width): lane `source` of the caller's own group of `width` lanes (hip/amd_detail/amd_warp_functions.h)
inline int shfl_source(int source, int width) { return (simt::lane() & ~(width - 1)) | (source & (width - 1)); }
inline int __shfl(int v, int source, int width = 64) { return (int)simt::exchange((uint32_t)v, shfl_source(source, width)); }
inline float __shfl(float v, int source, int width = 64) { return simt::real(simt::exchange(simt::bits(v), shfl_source(source, width))); }
inline float __shfl_xor(float v, int mask, int width = 64) { (void)width; return simt::real(simt::exchange(simt::bits(v), simt::lane() ^ mask)); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline float __expf(float x) { return expf(x); }
#define __builtin_amdgcn_readfirstlane(x) (x)  /* only ever given a value that is the same for the wavefront */
#define __builtin_amdgcn_update_dpp simt::update_dpp
#define __builtin_amdgcn_ballot_w64 simt::ballot
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) simt::mfma_16x16x4(a, b, c)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
/* the grouped launches' coherent loads / stores and their flag (gvk_chains.hip await_row, load_row_coherent): plain accesses here — the
 * stand-in runs one workgroup at a time, nothing is ever waited for */
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
