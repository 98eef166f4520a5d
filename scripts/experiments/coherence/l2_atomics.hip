// Bring-up (GPU): is a value changed by a device-scope float atomic (or a plain store) in one kernel seen by the plain loads
// of EVERY block of the next kernel, when every XCD's L2 held the old line?   hipcc --offload-arch=gfx950 -O2 -o l2_atomics l2_atomics.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void read_all(const float *x, float *seen, int n) {  // every block reads the n floats (plain loads)
    float s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) seen[blockIdx.x] = s;
}
__global__ void bump_atomic(float *x, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) __hip_atomic_fetch_add(x + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void bump_store(float *x, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) x[i] += 1.0f;
}
// one launch: block 0 bumps (atomic or store) while the others keep reading; reports the last value each block saw
__global__ void same_launch(float *x, float *seen, int n, int atomic, int rounds) {
    if (blockIdx.x == 0) {
        for (int r = 0; r < rounds; r++)
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                if (atomic) __hip_atomic_fetch_add(x + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else x[i] += 1.0f;
            }
        return;
    }
    float s = 0;
    for (int r = 0; r < rounds * 50; r++) {
        s = 0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
    }
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) seen[blockIdx.x] = s;
}

int main() {
    const int n = 128, blocks = 1024;
    float *x, *seen;
    hipMalloc(&x, n * 4);
    hipMalloc(&seen, blocks * 4);
    std::vector<float> host(blocks);
    for (int mode = 0; mode < 2; mode++) {
        hipMemset(x, 0, n * 4);
        int stale = 0;
        for (int round = 1; round <= 200; round++) {
            hipLaunchKernelGGL(read_all, dim3(blocks), dim3(64), 0, 0, x, seen, n);   // every L2 holds the line
            if (mode == 0) hipLaunchKernelGGL(bump_atomic, dim3(1), dim3(64), 0, 0, x, n);
            else hipLaunchKernelGGL(bump_store, dim3(1), dim3(64), 0, 0, x, n);
            hipLaunchKernelGGL(read_all, dim3(blocks), dim3(64), 0, 0, x, seen, n);
            hipMemcpy(host.data(), seen, blocks * 4, hipMemcpyDeviceToHost);
            for (int b = 0; b < blocks; b++) stale += host[b] != (float)round * n;
        }
        printf("%s in one kernel, plain loads in the next: %d of %d block reads saw an old value\n", mode == 0 ? "atomic add" : "plain store", stale, 200 * blocks);
    }
    for (int atomic = 1; atomic >= 0; atomic--) {
        hipMemset(x, 0, n * 4);
        hipLaunchKernelGGL(same_launch, dim3(blocks), dim3(64), 0, 0, x, seen, n, atomic, 20);
        hipMemcpy(host.data(), seen, blocks * 4, hipMemcpyDeviceToHost);
        int final_ = 0, zero = 0;
        for (int b = 1; b < blocks; b++) final_ += host[b] == 20.0f * n, zero += host[b] == 0;
        printf("same launch, %s: %d of %d reader blocks ended on the final value, %d never saw a change\n", atomic ? "atomic add" : "plain store", final_, blocks - 1, zero);
    }
    return 0;
}
