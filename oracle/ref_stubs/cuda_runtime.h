// TEST INFRASTRUCTURE ONLY (oracle/ref_alias_harness.cpp).  Host emulation of the few CUDA runtime calls the reference's
// base/memory.h makes, so that its AliasTable (include/base/alias_table.cuh) can be compiled and run on the CPU exactly
// as written.  hip_runtime.h supplies __global__, blockIdx and the <<< >>> launch syntax under `hipcc --cuda-host-only`;
// the launch in AliasTable::device_sample is compiled but never executed.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

typedef hipStream_t cudaStream_t;
typedef int cudaError_t;
enum { cudaSuccess = 0 };
enum cudaMemcpyKind { cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2 };

inline const char *cudaGetErrorString(cudaError_t) { return "emulated CUDA runtime"; }
template <class T>
inline cudaError_t cudaMallocHost(T **ptr, size_t bytes) {
    *ptr = static_cast<T *>(malloc(bytes ? bytes : 1));
    return *ptr ? cudaSuccess : 1;
}
template <class T>
inline cudaError_t cudaMalloc(T **ptr, size_t bytes) { return cudaMallocHost(ptr, bytes); }
inline cudaError_t cudaFreeHost(void *ptr) { free(ptr); return cudaSuccess; }
inline cudaError_t cudaFree(void *ptr) { free(ptr); return cudaSuccess; }
inline cudaError_t cudaSetDevice(int) { return cudaSuccess; }
extern int gvref_device_count;           // what cudaGetDeviceCount reports (set by the harness)
extern size_t gvref_device_memory;       // what cudaMemGetInfo reports as free and total
inline cudaError_t cudaGetDeviceCount(int *count) { *count = gvref_device_count; return cudaSuccess; }
inline cudaError_t cudaMemGetInfo(size_t *free_bytes, size_t *total) {
    if (free_bytes) *free_bytes = gvref_device_memory;
    if (total) *total = gvref_device_memory;
    return cudaSuccess;
}
inline cudaError_t cudaStreamCreate(cudaStream_t *stream) { *stream = nullptr; return cudaSuccess; }
inline cudaError_t cudaStreamDestroy(cudaStream_t) { return cudaSuccess; }
inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
inline cudaError_t cudaStreamSynchronize(cudaStream_t) { return cudaSuccess; }
inline cudaError_t cudaMemcpyAsync(void *dst, const void *src, size_t bytes, cudaMemcpyKind, cudaStream_t) {
    memcpy(dst, src, bytes);
    return cudaSuccess;
}
