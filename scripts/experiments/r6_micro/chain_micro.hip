// Round 6 microbenchmarks ("next round" item 2: measure first, build second).  Standalone: hipcc --offload-arch=gfx950 -O3.
//
// (i)  One workgroup of 256 threads trains ONE long chain of N entries as 16 tasks (lane groups of 16, dim 128: 8 floats per lane) the
//      way train_long_chains does — dot, DPP butterfly, v_exp / v_rcp, update — with the partner rows
//        ring   : fetched per step through a ring of D rows in registers (today's chain_steps, D = 4),
//        lds    : all rows of a task landed in LDS up front by global_load_lds_dwordx4 (no registers in flight), steps read LDS,
//        regs16 : all 16 rows of a task in registers up front (what a kernel of its own with 168+ registers could do).
//      Reported: microseconds from the first row request to the last step, isolated and beside a bandwidth hog on a second stream.
// (ii) Hand-off between the units of a chain engine: 256 workgroups (one per CU) pass an 8 MB mirror from unit to unit — every
//      workgroup writes its slice, then reads a slice another XCD wrote —
//        boundary   : one kernel launch per unit (what the stream does today),
//        persistent : one launch, a grid barrier per unit (release: __threadfence + agent-scope atomic; acquire: agent-scope load),
//      verified (every unit checks the values it reads), microseconds per unit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <cmath>

#include <algorithm>
#include <chrono>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int DIM = 128, G = 16, V = DIM / G, NG = 16, BLOCK = 256;

template <int CTRL>
__device__ __forceinline__ float dpp(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float group_sum(float x) {
    x += dpp<0xB1>(x); x += dpp<0x4E>(x); x += dpp<0x141>(x); x += dpp<0x140>(x);
    return x;
}
__device__ __forceinline__ float chain_sigmoid(float x) {
    const float t = __builtin_amdgcn_exp2f(-fabsf(x) * 1.44269504088896340736f);
    return (x > 0 ? 1.0f : t) * __builtin_amdgcn_rcpf(1.0f + t);
}
__device__ __forceinline__ void load_row(const float *row, int lane, float (&r)[V]) {
    const f32x4 a = *reinterpret_cast<const f32x4 *>(row + lane * 4), b = *reinterpret_cast<const f32x4 *>(row + 64 + lane * 4);
    r[0] = a.x, r[1] = a.y, r[2] = a.z, r[3] = a.w, r[4] = b.x, r[5] = b.y, r[6] = b.z, r[7] = b.w;
}
__device__ __forceinline__ void step(float (&own)[V], const float (&c)[V], bool positive, float weight_on) {
    float partial = 0;
#pragma unroll
    for (int x = 0; x < V; x++) partial += own[x] * c[x];
    const float prob = chain_sigmoid(group_sum(partial));
    const float gradient = positive ? prob - 1 : prob, weight = weight_on * (positive ? 1.0f : 5.0f);
#pragma unroll
    for (int x = 0; x < V; x++) own[x] -= 0.025f * weight * (gradient * c[x] + 0.005f * own[x]);
}

// MODE 0 ring of D = 4 rows in registers, 1 every row of the task landed in LDS first, 2 every row of the task in registers first
template <int MODE>
__global__ void __launch_bounds__(BLOCK) chain_kernel(const float *table, const uint32_t *entries, int n, float *out, unsigned long long *stamps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x % G, group = threadIdx.x / G, wave = threadIdx.x / 64, wl = threadIdx.x % 64;
    const int per = (n + NG - 1) / NG, begin = min(group * per, n), end = min(begin + per, n);
    float own[V];
    load_row(table, lane, own);
    uint32_t mine = begin + lane < end ? entries[begin + lane] : 0;  // per <= 16 here
    auto entry = [&](int i) { return (uint32_t)__shfl((int)mine, i, G); };
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    unsigned long long t1 = t0;
    if (MODE == 0) {
        constexpr int D = 4;
        float ring[D][V];
#pragma unroll
        for (int i = 0; i < D; i++) load_row(table + (size_t)(entry(i) & 0x7fffffffu) * DIM, lane, ring[i]);
        for (int base = 0; base < per; base += D) {
#pragma unroll
            for (int i = 0; i < D; i++) {
                const int p = base + i;
                const uint32_t e = entry(p < 16 ? p : 15);
                step(own, ring[i], (e >> 31) != 0, begin + p < end ? 1.0f : 0.0f);
                const uint32_t f = entry(p + D < 16 ? p + D : 15);
                load_row(table + (size_t)(f & 0x7fffffffu) * DIM, lane, ring[i]);
            }
        }
    } else if (MODE == 1) {
        // slot (wave, i, half): 1 KiB, lane-linear — lane wl of the wave lands its 16 bytes at wl * 16
        for (int i = 0; i < per; i++) {
            const float *row = table + (size_t)(entry(i) & 0x7fffffffu) * DIM;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                char *slot = lds + ((size_t)(wave * per + i) * 2 + h) * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(row + h * 64 + lane * 4),
                                                 (__attribute__((address_space(3))) void *)(slot), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t1 = wall_clock64();
        for (int i = 0; i < per; i++) {
            float c[V];
            const char *slot = lds + ((size_t)(wave * per + i) * 2) * 1024 + wl * 16;
            const f32x4 a = *reinterpret_cast<const f32x4 *>(slot), b = *reinterpret_cast<const f32x4 *>(slot + 1024);
            c[0] = a.x, c[1] = a.y, c[2] = a.z, c[3] = a.w, c[4] = b.x, c[5] = b.y, c[6] = b.z, c[7] = b.w;
            const uint32_t e = entry(i);
            step(own, c, (e >> 31) != 0, begin + i < end ? 1.0f : 0.0f);
        }
    } else {
        float rows[16][V];
#pragma unroll
        for (int i = 0; i < 16; i++) load_row(table + (size_t)(entry(i) & 0x7fffffffu) * DIM, lane, rows[i]);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t e = entry(i);
            step(own, rows[i], (e >> 31) != 0, begin + i < end ? 1.0f : 0.0f);
        }
    }
    const unsigned long long t2 = wall_clock64();
    __syncthreads();
    if (threadIdx.x == 0) stamps[0] = t0, stamps[1] = t1, stamps[2] = wall_clock64();
    if (wl == 0) stamps[4 + wave] = t2;
#pragma unroll
    for (int x = 0; x < V; x++) out[(size_t)group * DIM + x * G + lane] = own[x];
}

__global__ void hog_kernel(const f32x4 *in, f32x4 *out, size_t n, int rounds) {
    for (int r = 0; r < rounds; r++)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i] + (float)r;
}

// ---- (ii) ---------------------------------------------------------------------------------------------------------------------
constexpr int UNITS = 200;
constexpr size_t MIRROR_FLOATS = 2u << 20;  // 8 MB

// every workgroup reads `floats` floats of the slice workgroup (block + 37) % wg wrote in unit u - 1 (another XCD under round-robin
// dispatch) and writes its own.  COHERENT: loads and stores as agent-scope atomics (sc1: through the XCD's L2 to memory), no fences
template <int COHERENT>
__device__ __forceinline__ void unit_work(const float *from, float *to, int u, int block, int wg, size_t floats, unsigned *errors) {
    const size_t slice = MIRROR_FLOATS / wg;
    const int other = (block + 37) % wg;
    const float *in = from + (size_t)other * slice;
    float *out = to + (size_t)block * slice;
    for (size_t i = threadIdx.x; i < floats; i += BLOCK) {
        float v;
        if (COHERENT) v = __hip_atomic_load(in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else v = in[i];
        if (v != (float)(u - 1)) atomicAdd(errors, 1u);
        if (COHERENT) __hip_atomic_store(out + i, (float)u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else out[i] = (float)u;
    }
}

__global__ void __launch_bounds__(BLOCK) boundary_kernel(const float *from, float *to, int u, size_t floats, unsigned *errors) {
    unit_work<0>(from, to, u, blockIdx.x, gridDim.x, floats, errors);
}

template <int COHERENT>
__global__ void __launch_bounds__(BLOCK) persistent_kernel(float *a, float *b, unsigned *barrier, unsigned *errors, int units, size_t floats) {
    for (int u = 1; u <= units; u++) {
        unit_work<COHERENT>((u & 1) ? a : b, (u & 1) ? b : a, u, blockIdx.x, gridDim.x, floats, errors);
        // grid barrier: the workgroup's stores have left the CU (barrier), thread 0 releases them (one L2 write-back), arrives and
        // waits for all (bounded spin: a bug must not hang the GPU); then every wavefront drops its stale lines
        __syncthreads();
        if (threadIdx.x == 0) {
            if (COHERENT) __hip_atomic_fetch_add(barrier, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(barrier, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)u * gridDim.x;
            unsigned spins = 0;
            while (__hip_atomic_load(barrier, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);
            if (spins >= (1u << 22)) atomicAdd(errors, 1u << 16);
        }
        __syncthreads();
        if (!COHERENT) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
}

struct BigArgs { char bytes[400]; };
// a kernel that lasts about `us` microseconds on one workgroup per CU and leaves its start / end times (100 MHz clock)
__global__ void timed_kernel(BigArgs a, unsigned long long *stamps, int index, int us) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)us * 100) __builtin_amdgcn_s_sleep(8);
    if (blockIdx.x == 0 && threadIdx.x == 0) stamps[2 * index] = t0, stamps[2 * index + 1] = wall_clock64();
    if (a.bytes[0] == 77) stamps[0] = 0;
}
// (iv) the executor's own pattern with kernels of known length: stream A runs groups of 8 kernels of 10 us (the chains of a batch),
// stream B groups of 8 of 6 us (its pairs); B's group i waits for A's group i, A's group i + 2 for B's group i.  Ideal: 80 us per group.
static void pattern(hipStream_t sa, hipStream_t sb, unsigned flags, const char *what) {
    const int groups = 40;
    unsigned long long *stamps;
    CHECK(hipMalloc(&stamps, groups * 16 * 16));
    BigArgs a = {};
    hipEvent_t ea[4], eb[4];
    for (auto &e : ea) CHECK(hipEventCreateWithFlags(&e, flags));
    for (auto &e : eb) CHECK(hipEventCreateWithFlags(&e, flags));
    for (int trial = 0; trial < 2; trial++) {
        CHECK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < groups; i++) {
            if (i >= 2) CHECK(hipStreamWaitEvent(sa, eb[(i - 2) & 3], 0));
            for (int j = 0; j < 8; j++) hipLaunchKernelGGL(timed_kernel, dim3(64), dim3(64), 0, sa, a, stamps, i * 16 + j, 10);
            CHECK(hipEventRecord(ea[i & 3], sa));
            CHECK(hipStreamWaitEvent(sb, ea[i & 3], 0));
            for (int j = 0; j < 8; j++) hipLaunchKernelGGL(timed_kernel, dim3(64), dim3(64), 0, sb, a, stamps, i * 16 + 8 + j, 6);
            CHECK(hipEventRecord(eb[i & 3], sb));
        }
        auto t1 = std::chrono::steady_clock::now();
        CHECK(hipDeviceSynchronize());
        auto t2 = std::chrono::steady_clock::now();
        std::vector<unsigned long long> h(groups * 32);
        CHECK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        double a_gap = 0, ab = 0, ba = 0;
        for (int i = 10; i < 30; i++) {
            a_gap += (h[2 * ((i + 1) * 16)] - h[2 * (i * 16 + 7) + 1]) * 0.01;      // A: end of group i -> start of group i + 1
            ab += (h[2 * (i * 16 + 8)] - h[2 * (i * 16 + 7) + 1]) * 0.01;           // end of A's group i -> start of B's group i
        }
        printf("(iv) [%s] two streams, events per group of 8 + 8 kernels: host %.1f us per group enqueued, %.1f us per group until done (ideal 80); on the GPU: "
               "A idle between its groups %.1f us, B starts %.1f us after A's group ends\n", what, std::chrono::duration<double, std::micro>(t1 - t0).count() / groups,
               std::chrono::duration<double, std::micro>(t2 - t0).count() / groups, a_gap / 20, ab / 20);
        (void)ba;
    }
}

__global__ void tiny_kernel(BigArgs a, int *out) { if (threadIdx.x == 0 && a.bytes[0] == 77) out[0] = 1; }

// (iii) what the HOST pays per call: launches of a tiny kernel with 400 bytes of arguments, and the event record + cross-stream wait
// pairs the chain stream needs per batch
static void host_costs(hipStream_t s1, hipStream_t s2) {
    int *out;
    CHECK(hipMalloc(&out, 4));
    BigArgs a = {};
    hipEvent_t ev[8];
    for (auto &e : ev) CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int trial = 0; trial < 2; trial++) {
        CHECK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 2000; i++) hipLaunchKernelGGL(tiny_kernel, dim3(300), dim3(256), 0, s1, a, out);
        auto t1 = std::chrono::steady_clock::now();
        CHECK(hipDeviceSynchronize());
        auto t2 = std::chrono::steady_clock::now();
        printf("(iii) 2000 launches (400 B of arguments) on one stream: host %.2f us per launch enqueued, %.2f us per launch until all done\n",
               std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000, std::chrono::duration<double, std::micro>(t2 - t0).count() / 2000);
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 500; i++) {
            for (int j = 0; j < 4; j++) hipLaunchKernelGGL(tiny_kernel, dim3(300), dim3(256), 0, s2, a, out);
            CHECK(hipEventRecord(ev[i & 3], s2));
            CHECK(hipStreamWaitEvent(s1, ev[i & 3], 0));
            for (int j = 0; j < 4; j++) hipLaunchKernelGGL(tiny_kernel, dim3(300), dim3(256), 0, s1, a, out);
            CHECK(hipEventRecord(ev[4 + (i & 3)], s1));
            CHECK(hipStreamWaitEvent(s2, ev[4 + (i & 3)], 0));
        }
        t1 = std::chrono::steady_clock::now();
        CHECK(hipDeviceSynchronize());
        t2 = std::chrono::steady_clock::now();
        printf("(iii) 500 x {4 launches on A, record, B waits, 4 launches on B, record, A waits}: host %.2f us per iteration enqueued, %.2f us until all done\n",
               std::chrono::duration<double, std::micro>(t1 - t0).count() / 500, std::chrono::duration<double, std::micro>(t2 - t0).count() / 500);
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 2000; i++) CHECK(hipEventRecord(ev[i & 3], s2));
        t1 = std::chrono::steady_clock::now();
        for (int i = 0; i < 2000; i++) CHECK(hipStreamWaitEvent(s1, ev[i & 3], 0));
        t2 = std::chrono::steady_clock::now();
        CHECK(hipDeviceSynchronize());
        printf("(iii) hipEventRecord %.2f us, hipStreamWaitEvent %.2f us (host, each)\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000,
               std::chrono::duration<double, std::micro>(t2 - t1).count() / 2000);
    }
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs, wall clock %d kHz\n", prop.name, prop.multiProcessorCount, 100000);
    // ---- (i) ----
    const size_t rows = 1u << 20;
    float *table, *out;
    uint32_t *entries;
    unsigned long long *stamps;
    CHECK(hipMalloc(&table, rows * DIM * 4));
    CHECK(hipMalloc(&out, 16 * DIM * 4));
    CHECK(hipMalloc(&entries, 4096 * 4));
    CHECK(hipMalloc(&stamps, 64 * 8));
    {
        std::vector<float> host(rows * DIM);
        srand(1);
        for (auto &x : host) x = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
        CHECK(hipMemcpy(table, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    }
    f32x4 *hog_in, *hog_out;
    const size_t hog_n = (size_t)64 << 20;  // 1 GiB each
    CHECK(hipMalloc(&hog_in, hog_n * 16));
    CHECK(hipMalloc(&hog_out, hog_n * 16));
    CHECK(hipMemset(hog_in, 0, hog_n * 16));
    hipStream_t s1, s2;
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    float reference[16 * DIM];
    for (int n : {250, 112}) {
        for (int loaded = 0; loaded < 2; loaded++) {
            for (int mode = 0; mode < 3; mode++) {
                double sum_all = 0, sum_rows = 0, sum_steps = 0, worst = 0;
                const int reps = 40;
                float max_diff = 0;
                for (int rep = 0; rep < reps; rep++) {
                    std::vector<uint32_t> host(4096);
                    srand(100 + rep);
                    for (auto &e : host) e = ((uint32_t)rand() % (rows - 1) + 1) | ((rand() & 1) ? 0x80000000u : 0u);
                    CHECK(hipMemcpy(entries, host.data(), 4096 * 4, hipMemcpyHostToDevice));
                    CHECK(hipDeviceSynchronize());
                    if (loaded) hipLaunchKernelGGL(hog_kernel, dim3(4096), dim3(256), 0, s2, hog_in, hog_out, hog_n, 2);
                    const size_t lds = mode == 1 ? (size_t)4 * 16 * 2048 : 0;
                    if (mode == 0) hipLaunchKernelGGL(chain_kernel<0>, dim3(1), dim3(BLOCK), lds, s1, table, entries, n, out, stamps);
                    if (mode == 1) hipLaunchKernelGGL(chain_kernel<1>, dim3(1), dim3(BLOCK), lds, s1, table, entries, n, out, stamps);
                    if (mode == 2) hipLaunchKernelGGL(chain_kernel<2>, dim3(1), dim3(BLOCK), lds, s1, table, entries, n, out, stamps);
                    CHECK(hipGetLastError());
                    CHECK(hipDeviceSynchronize());
                    unsigned long long st[8];
                    CHECK(hipMemcpy(st, stamps, sizeof st, hipMemcpyDeviceToHost));
                    const unsigned long long last = std::max(std::max(st[4], st[5]), std::max(st[6], st[7]));
                    const double all = (last - st[0]) * 0.01, rows_us = (st[1] - st[0]) * 0.01;
                    sum_all += all, sum_rows += rows_us, sum_steps += all - rows_us, worst = std::max(worst, all);
                    float got[16 * DIM];
                    CHECK(hipMemcpy(got, out, sizeof got, hipMemcpyDeviceToHost));
                    if (mode == 0 && rep == reps - 1) std::copy(got, got + 16 * DIM, reference);
                    if (mode != 0 && rep == reps - 1)
                        for (int i = 0; i < 16 * DIM; i++) max_diff = std::max(max_diff, fabsf(got[i] - reference[i]));
                }
                printf("(i) chain of %3d entries = 16 tasks of %2d, %-8s %-7s: first request -> last step %6.2f us (max %6.2f)%s", n, (n + 15) / 16,
                       loaded ? "loaded" : "isolated", mode == 0 ? "ring4" : (mode == 1 ? "lds" : "regs16"), sum_all / reps, worst,
                       mode == 1 ? "" : "\n");
                if (mode == 1) printf("  [rows landed %5.2f us, steps %5.2f us = %.3f us per step]\n", sum_rows / reps, sum_steps / reps, sum_steps / reps / ((n + 15) / 16));
                if (mode != 0) printf("      max |difference| to ring4 on the last repetition: %g\n", max_diff);
            }
        }
    }
    CHECK(hipDeviceSynchronize());
    host_costs(s1, s2);
    pattern(s1, s2, hipEventDisableTiming, "hipEventDisableTiming");
    pattern(s1, s2, hipEventDisableTiming | hipEventDisableSystemFence, "hipEventDisableTiming | hipEventDisableSystemFence");
    pattern(s1, s2, hipEventDefault, "hipEventDefault");
    // ---- (ii) ----
    float *a, *b;
    unsigned *barrier, *errors;
    CHECK(hipMalloc(&a, MIRROR_FLOATS * 4));
    CHECK(hipMalloc(&b, MIRROR_FLOATS * 4));
    CHECK(hipMalloc(&barrier, 4));
    CHECK(hipMalloc(&errors, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<float> init(MIRROR_FLOATS, 0.0f);
    for (int wg : {256, 64}) {
        for (size_t floats : {MIRROR_FLOATS / wg, (size_t)2048, (size_t)256}) {
            for (int form = 0; form < 3; form++) {
                CHECK(hipMemcpy(a, init.data(), MIRROR_FLOATS * 4, hipMemcpyHostToDevice));  // unit 1 reads a = 0 = the value of unit 0
                CHECK(hipMemset(errors, 0, 4));
                CHECK(hipMemset(barrier, 0, 4));
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0, s1));
                if (form == 0)
                    for (int u = 1; u <= UNITS; u++)
                        hipLaunchKernelGGL(boundary_kernel, dim3(wg), dim3(BLOCK), 0, s1, (u & 1) ? a : b, (u & 1) ? b : a, u, floats, errors);
                if (form == 1) hipLaunchKernelGGL(persistent_kernel<0>, dim3(wg), dim3(BLOCK), 0, s1, a, b, barrier, errors, UNITS, floats);
                if (form == 2) hipLaunchKernelGGL(persistent_kernel<1>, dim3(wg), dim3(BLOCK), 0, s1, a, b, barrier, errors, UNITS, floats);
                CHECK(hipEventRecord(e1, s1));
                CHECK(hipDeviceSynchronize());
                float ms;
                unsigned err;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                CHECK(hipMemcpy(&err, errors, 4, hipMemcpyDeviceToHost));
                printf("(ii) %3d workgroups hand %7zu floats each from unit to unit (%d units): %-62s %6.2f us per unit, errors %u%s\n", wg, floats, UNITS,
                       form == 0 ? "a launch per unit" : (form == 1 ? "one launch, grid barrier, release / acquire fences" : "one launch, grid barrier, sc1 loads and stores, no fences"),
                       ms * 1000 / UNITS, err & 0xffffu, (err >> 16) ? " (a barrier TIMED OUT)" : "");
            }
        }
    }
    return 0;
}
