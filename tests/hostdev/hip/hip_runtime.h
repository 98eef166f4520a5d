// TEST INFRASTRUCTURE ONLY — never part of the product.
//
// A stand-in for <hip/hip_runtime.h> that lets the solver engine (graphvite_amd/csrc/gvx_engine.cpp, gvx_comm.cpp) be
// compiled by g++ and run on a machine WITHOUT a GPU (tests/hostdev/Makefile -> tests/hostdev/build/libgvk_host.so): "device"
// memory is host memory, streams execute immediately, events are trivially complete.  What the CPU tests exercise through
// it is the engine's host logic — partitions, schedule, slot claims, the exchange, pool routing, batch-id / lr accounting,
// write-back — with the kernels of tests/hostdev/host_kernels.cpp (the CPU oracle behind the gvk.h entry points).
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1, hipErrorPeerAccessAlreadyEnabled = 704 };
typedef struct gvh_stream *hipStream_t;
typedef struct gvh_event *hipEvent_t;
typedef enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 } hipMemcpyKind;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };

extern "C" size_t gvh_memory_limit;  // bytes hipMalloc may hand out in total (tests lower it to exercise the fallbacks)
extern "C" size_t gvh_memory_used;

hipError_t gvh_malloc(void **p, size_t bytes);
hipError_t gvh_free(void *p);

inline hipError_t hipGetDeviceCount(int *count) { *count = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : (e == hipErrorOutOfMemory ? "out of memory" : "error"); }
inline hipError_t hipMemGetInfo(size_t *free_bytes, size_t *total) {
    *total = gvh_memory_limit, *free_bytes = gvh_memory_limit > gvh_memory_used ? gvh_memory_limit - gvh_memory_used : 0;
    return hipSuccess;
}
template <class T> inline hipError_t hipMalloc(T **p, size_t bytes) { return gvh_malloc((void **)p, bytes); }
inline hipError_t hipFree(void *p) { return gvh_free(p); }
template <class T> inline hipError_t hipHostMalloc(T **p, size_t bytes, unsigned) { *p = (T *)malloc(bytes ? bytes : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemset(void *p, int value, size_t bytes) { memset(p, value, bytes); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *p, int value, size_t bytes, hipStream_t) { memset(p, value, bytes); return hipSuccess; }
inline hipError_t hipMemcpy(void *to, const void *from, size_t bytes, hipMemcpyKind) { memmove(to, from, bytes); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *to, const void *from, size_t bytes, hipMemcpyKind, hipStream_t) { memmove(to, from, bytes); return hipSuccess; }
inline hipError_t hipMemcpyPeerAsync(void *to, int, const void *from, int, size_t bytes, hipStream_t) { memmove(to, from, bytes); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void *to, size_t to_pitch, const void *from, size_t from_pitch, size_t width, size_t height,
                                   hipMemcpyKind, hipStream_t) {
    for (size_t r = 0; r < height; r++) memmove((char *)to + r * to_pitch, (const char *)from + r * from_pitch, width);
    return hipSuccess;
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)malloc(1); return hipSuccess; }
struct hipDeviceProp_t { int multiProcessorCount = 256; };
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *, int) { return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned flags, int) { return hipStreamCreateWithFlags(s, flags); }
inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0, *greatest = 0; return hipSuccess; }
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, uint32_t, const uint32_t *) { return hipStreamCreateWithFlags(s, 0); }
inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (hipEvent_t)malloc(1); return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { return hipEventCreateWithFlags(e, 0); }
inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 1.0f; return hipSuccess; }
inline hipError_t hipDeviceCanAccessPeer(int *can, int, int) { *can = 1; return hipSuccess; }
inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
