// gvk_group_pairs: inside every batch of a device-resident pool, make the pairs that share a head row adjacent.
//
// Why: the samples of a batch are i.i.d. draws that the training kernel processes concurrently, so their order
// inside the batch carries no meaning — but adjacent pairs run in the same workgroup at the same time, so a head
// row that several pairs of a batch share (30 % of the head rows of a 100k batch on a power-law graph) is fetched
// from HBM once.
//
// How: a stable LSD radix sort of each batch on the low row_bits bits of the head row, one workgroup per batch,
// ceil(row_bits / 10) counting passes (two for tables of up to 2^20 rows).  A batch is 800 KB — it lives in L2 for
// the whole sort; the only state is one histogram per wavefront in LDS (16 x 1024 counters = 64 KB):
//   A  wave w counts the digits of ITS contiguous 1/16th of the batch into its own histogram (LDS atomics);
//   B  thread d turns column d of the 16 histograms into exclusive offsets in (digit, wave) order;
//   C  wave w walks its slice again in order, 64 records at a time; lanes holding the same digit find each other
//      with one ballot per digit bit, take consecutive slots from the wave's cursor for that digit, and the lowest
//      of them moves the cursor on.  A wave's cursors are its own, so C needs no barrier and the order is the input
//      order: stable, deterministic, no global atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gvk.h"
#include "gvk_internal.h"

namespace {

constexpr int kThreads = 1024, kWaves = kThreads / 64, kMaxDigitBits = 10, kDigits = 1 << kMaxDigitBits;
constexpr int kAhead = 8;  // 64-record steps of a wave's walk whose loads are in flight together

__global__ void __launch_bounds__(kThreads) group_pass_kernel(const uint64_t *__restrict__ in, uint64_t *__restrict__ out,
                                                              int batch_size, int shift, int digit_bits) {
    __shared__ uint32_t cursor[kWaves][kDigits];
    __shared__ uint32_t wave_total[kWaves];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t base = (size_t)blockIdx.x * (size_t)batch_size;
    const uint32_t digit_mask = (1u << digit_bits) - 1;
    // slice of this wave: a multiple of 64 records, so that every step of the walk is a full wavefront but the last
    const int per_wave = ((batch_size + kWaves - 1) / kWaves + 63) / 64 * 64;
    const int begin = min(wave * per_wave, batch_size), end = min(begin + per_wave, batch_size);

    for (int i = threadIdx.x; i < kWaves * kDigits; i += kThreads) (&cursor[0][0])[i] = 0;
    __syncthreads();

    // A: per-wave digit histogram (kAhead steps of the walk loaded before the first is used: a wave is otherwise
    // serialised on the latency of its own loads)
    for (int at = begin; at < end; at += 64 * kAhead) {
        uint64_t record[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; u++) {
            const int i = at + u * 64 + lane;
            record[u] = i < end ? in[base + i] : 0;
        }
#pragma unroll
        for (int u = 0; u < kAhead; u++)
            if (at + u * 64 + lane < end) atomicAdd(&cursor[wave][((uint32_t)(record[u] >> 32) >> shift) & digit_mask], 1u);
    }
    __syncthreads();

    // B: exclusive offsets in (digit, wave) order; thread d owns digit d
    {
        const int d = threadIdx.x;
        uint32_t running = 0;
        for (int w = 0; w < kWaves; w++) {
            const uint32_t count = cursor[w][d];
            cursor[w][d] = running;
            running += count;
        }
        // block-wide exclusive scan of the per-digit totals: inclusive scan inside the wave, then across waves
        uint32_t inclusive = running;
        for (int step = 1; step < 64; step <<= 1) {
            const uint32_t up = __shfl_up(inclusive, step);
            if (lane >= step) inclusive += up;
        }
        if (lane == 63) wave_total[wave] = inclusive;
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < wave; w++) before += wave_total[w];
        const uint32_t digit_base = before + inclusive - running;
        for (int w = 0; w < kWaves; w++) cursor[w][d] += digit_base;
    }
    __syncthreads();

    // C: stable scatter
    const uint64_t lower_lanes = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int at = begin; at < end; at += 64 * kAhead) {
        uint64_t record[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; u++) {
            const int i = at + u * 64 + lane;
            record[u] = i < end ? in[base + i] : 0;
        }
#pragma unroll
        for (int u = 0; u < kAhead; u++) {
            const bool valid = at + u * 64 + lane < end;
            const uint32_t digit = ((uint32_t)(record[u] >> 32) >> shift) & digit_mask;
            uint64_t peers = __ballot(valid);
            for (int bit = 0; bit < digit_bits; bit++) {
                const bool set = (digit >> bit) & 1;
                const uint64_t with = __ballot(valid && set);
                peers &= set ? with : ~with;
            }
            if (valid) {
                const uint32_t rank = (uint32_t)__popcll(peers & lower_lanes);
                const uint32_t slot = cursor[wave][digit] + rank;  // every peer reads the cursor before its leader moves it
                if (rank == 0) cursor[wave][digit] = slot + (uint32_t)__popcll(peers);
                out[base + slot] = record[u];
            }
        }
    }
}

// gvk_spread_pairs: record i of the pool to place (i % units) * (n / units) + i / units
__global__ void __launch_bounds__(256) spread_kernel(const uint64_t *in, uint64_t *out, const uint64_t n, const uint64_t units) {
    const uint64_t per = n / units;
    // thread t writes output record t (coalesced stores; the loads stride by `units` records)
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const uint64_t unit = t / per, slot = t % per;
    out[t] = in[slot * units + unit];
}

}  // namespace

extern "C" int gvk_spread_pairs(void *stream, const uint32_t *pool_in, uint32_t *pool_out, size_t num_pair, int units) {
    if (units < 1 || num_pair % (size_t)units) return gvk_fail(GVK_EINVAL, "gvk_spread_pairs: %d units do not divide %zu pairs", units, num_pair);
    if (num_pair == 0) return GVK_OK;
    if (!pool_in || !pool_out || pool_in == pool_out) return gvk_fail(GVK_EINVAL, "gvk_spread_pairs: needs distinct input and output pools");
    hipLaunchKernelGGL(spread_kernel, dim3((unsigned)((num_pair + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const uint64_t *>(pool_in), reinterpret_cast<uint64_t *>(pool_out), (uint64_t)num_pair, (uint64_t)units);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return gvk_fail(GVK_EHIP, "gvk_spread_pairs: %s", hipGetErrorString(err));
    return GVK_OK;
}

extern "C" int gvk_group_pairs(void *stream, const uint32_t *pool_in, uint32_t *pool_out, void *workspace,
                               size_t *workspace_bytes, int batch_size, int num_batch, int row_bits) {
    if (!workspace_bytes) return gvk_fail(GVK_EINVAL, "gvk_group_pairs: workspace_bytes is null");
    if (batch_size < 0 || num_batch < 0 || row_bits < 1 || row_bits > 32)
        return gvk_fail(GVK_EINVAL, "gvk_group_pairs: bad sizes (batch_size %d, num_batch %d, row_bits %d)", batch_size,
                        num_batch, row_bits);
    const size_t n = (size_t)batch_size * (size_t)num_batch;
    const int passes = (row_bits + kMaxDigitBits - 1) / kMaxDigitBits;
    const size_t need = passes > 1 ? n * sizeof(uint64_t) : 0;
    if (!workspace) {  // size query
        *workspace_bytes = need ? need : 1;
        return GVK_OK;
    }
    if (n == 0) return GVK_OK;
    if (!pool_in || !pool_out || pool_in == pool_out)
        return gvk_fail(GVK_EINVAL, "gvk_group_pairs: needs distinct input and output pools");
    if (*workspace_bytes < need)
        return gvk_fail(GVK_EINVAL, "gvk_group_pairs: workspace holds %zu bytes, %zu needed", *workspace_bytes, need);
    const int digit_bits = (row_bits + passes - 1) / passes;
    // ping-pong between the workspace and pool_out so that the last pass lands in pool_out
    const uint64_t *source = reinterpret_cast<const uint64_t *>(pool_in);
    uint64_t *out = reinterpret_cast<uint64_t *>(pool_out), *scratch = reinterpret_cast<uint64_t *>(workspace);
    for (int pass = 0; pass < passes; pass++) {
        uint64_t *target = (passes - pass) % 2 ? out : scratch;
        const int shift = pass * digit_bits, bits = pass == passes - 1 ? row_bits - shift : digit_bits;
        hipLaunchKernelGGL(group_pass_kernel, dim3((unsigned)num_batch), dim3(kThreads), 0, (hipStream_t)stream, source,
                           target, batch_size, shift, bits);
        source = target;
    }
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return gvk_fail(GVK_EHIP, "gvk_group_pairs: %s", hipGetErrorString(err));
    return GVK_OK;
}
