// gvk_group_pairs: inside every batch of a device-resident pool, make the pairs that share a head row adjacent.
//
// Why: the samples of a batch are i.i.d. draws that the training kernel processes concurrently, so their order
// inside the batch carries no meaning — but adjacent pairs run in the same workgroup at the same time, so a head
// row that several pairs of a batch share (30 % of the head rows of a 100k batch on a power-law graph) is fetched
// from HBM once.  A pool record {tail, head} is one little-endian 64-bit word with the head row in the high half:
// a stable segmented radix sort of those words on bits [32, 32 + row_bits) is exactly the regrouping, one segment
// per batch.  The sort itself is rocPRIM's (a library sort, like a library GEMM); nothing here allocates.
#include <hip/hip_runtime.h>
#include <string.h>

#include <cstring>
#include <rocprim/rocprim.hpp>

#include "gvk.h"
#include "gvk_internal.h"

namespace {
struct BatchOffset {
    unsigned int batch_size;
    __host__ __device__ unsigned int operator()(unsigned int i) const { return i * batch_size; }
};
}  // namespace

extern "C" int gvk_group_pairs(void *stream, const uint32_t *pool_in, uint32_t *pool_out, void *workspace,
                               size_t *workspace_bytes, int batch_size, int num_batch, int row_bits) {
    if (!workspace_bytes) return gvk_fail(GVK_EINVAL, "gvk_group_pairs: workspace_bytes is null");
    if (batch_size < 0 || num_batch < 0 || row_bits < 1 || row_bits > 32)
        return gvk_fail(GVK_EINVAL, "gvk_group_pairs: bad sizes (batch_size %d, num_batch %d, row_bits %d)", batch_size,
                        num_batch, row_bits);
    if ((uint64_t)batch_size * (uint64_t)num_batch > 0xffffffffull)
        return gvk_fail(GVK_EINVAL, "gvk_group_pairs: more than 2^32 - 1 pairs in one pool");
    const unsigned int n = (unsigned int)batch_size * (unsigned int)num_batch;
    if (workspace && n == 0) return GVK_OK;
    if (workspace && (!pool_in || !pool_out || pool_in == pool_out))
        return gvk_fail(GVK_EINVAL, "gvk_group_pairs: needs distinct input and output pools");
    auto begin = rocprim::make_transform_iterator(rocprim::make_counting_iterator(0u), BatchOffset{(unsigned int)batch_size});
    size_t bytes = workspace ? *workspace_bytes : 0;
    const hipError_t err = rocprim::segmented_radix_sort_keys(
        workspace, bytes, reinterpret_cast<const uint64_t *>(pool_in), reinterpret_cast<uint64_t *>(pool_out), n,
        (unsigned int)num_batch, begin, begin + 1, 32u, 32u + (unsigned int)row_bits, (hipStream_t)stream);
    if (err != hipSuccess) return gvk_fail(GVK_EHIP, "gvk_group_pairs: %s", hipGetErrorString(err));
    if (!workspace) *workspace_bytes = bytes ? bytes : 1;
    return GVK_OK;
}
