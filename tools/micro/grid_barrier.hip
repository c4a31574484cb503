// Micro-benchmark: cost of a grid-wide barrier (one atomic per workgroup + spin) vs a kernel boundary, on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22)) break;          // never hang the device
        }
    }
    __syncthreads();
}
__global__ void __launch_bounds__(256) k_barriers(unsigned* counter, int rounds, float* sink) {
    float x = threadIdx.x;
    for (int r = 0; r < rounds; r++) {
        x = x * 1.0001f + 1.f;
        grid_barrier(counter, (unsigned)(r + 1) * gridDim.x);
    }
    if (x == -1.f) sink[0] = x;
}
__global__ void __launch_bounds__(256) k_empty(float* sink) { if (threadIdx.x == 9999) sink[0] = 1.f; }

int main() {
    unsigned* counter; float* sink;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&sink, 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int G : {256, 489, 1024, 2048}) {
        const int rounds = 200;
        CK(hipMemset(counter, 0, 4));
        void* args[] = {&counter, (void*)&rounds, &sink};
        hipError_t e = hipLaunchCooperativeKernel((void*)k_barriers, dim3(G), dim3(256), args, 0, 0);   // warm-up + residency check
        if (e != hipSuccess) { printf("G=%d: cooperative launch refused (%s)\n", G, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        CK(hipDeviceSynchronize());
        CK(hipMemset(counter, 0, 4));
        CK(hipEventRecord(a, 0));
        CK(hipLaunchCooperativeKernel((void*)k_barriers, dim3(G), dim3(256), args, 0, 0));
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("G=%4d blocks: %.2f us per grid barrier\n", G, ms * 1e3f / rounds);
    }
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_empty, dim3(489), dim3(256), 0, 0, sink);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("empty dependent kernels (489 blocks): %.2f us per launch\n", ms * 1e3f / 200);
    return 0;
}
