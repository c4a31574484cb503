// How much does a launch pay for workgroups that read one range and retire?  (the blend launches carry
// patches x 45 segment slots, of which ~3/4 have nothing to walk.)  Grid of G single-wave workgroups; a workgroup whose slot index
// (blockIdx % 45) is >= live[blockIdx / 45] exits after one 8-byte load; the others spin for `work` dependent FMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/empty_wg.hip -o /tmp/empty_wg && /tmp/empty_wg
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(64) k_slots(const uint2* __restrict__ ranges, int slots, int work, float* out) {
    const int patch = blockIdx.x / slots, slot = blockIdx.x - patch * slots;
    const uint2 r = ranges[patch];
    if ((uint32_t)slot >= r.y - r.x) return;
    float x = (float)threadIdx.x;
    for (int i = 0; i < work; i++) x = fmaf(x, 1.0001f, 0.5f);
    if (x == 12345.f) out[blockIdx.x] = x;
}

__global__ void __launch_bounds__(64) k_list(const uint32_t* __restrict__ list, const uint32_t* __restrict__ n, int work, float* out) {
    if (blockIdx.x >= *n) return;
    const uint32_t item = list[blockIdx.x];
    float x = (float)threadIdx.x + (float)item;
    for (int i = 0; i < work; i++) x = fmaf(x, 1.0001f, 0.5f);
    if (x == 12345.f) out[blockIdx.x] = x;
}

// persistent: `grid` workgroups pull items off a counter
__global__ void __launch_bounds__(64) k_persist(const uint32_t* __restrict__ list, const uint32_t* __restrict__ n, uint32_t* counter, int work, float* out) {
    const uint32_t total = *n;
    for (;;) {
        uint32_t i = 0;
        if (threadIdx.x == 0) i = atomicAdd(counter, 1u);
        i = __builtin_amdgcn_readfirstlane(i);
        if (i >= total) return;
        const uint32_t item = list[i];
        float x = (float)threadIdx.x + (float)item;
        for (int k = 0; k < work; k++) x = fmaf(x, 1.0001f, 0.5f);
        if (x == 12345.f) out[i] = x;
    }
}

// the same slot grid with W waves per workgroup: wave w of workgroup b is slot b * W + w (each wave on its own, no barrier)
template <int W>
__global__ void __launch_bounds__(64 * W) k_slots_w(const uint2* __restrict__ ranges, int slots, int work, float* out) {
    const int gslot = blockIdx.x * W + (threadIdx.x >> 6);
    const int patch = gslot / slots, slot = gslot - patch * slots;
    const uint2 r = ranges[patch];
    if ((uint32_t)slot >= r.y - r.x) return;
    float x = (float)threadIdx.x;
    for (int i = 0; i < work; i++) x = fmaf(x, 1.0001f, 0.5f);
    if (x == 12345.f) out[gslot] = x;
}

static float time_ms(hipStream_t s, int reps, auto&& launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; i++) launch();
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const int patches = 2656, slots = 45;
    hipStream_t s; hipStreamCreate(&s);
    uint2* ranges; float* out; uint32_t* list; uint32_t* n; uint32_t* counter;
    hipMalloc(&ranges, patches * sizeof(uint2)); hipMalloc(&out, patches * slots * sizeof(float));
    hipMalloc(&list, patches * slots * 4); hipMalloc(&n, 4); hipMalloc(&counter, 4);
    printf("{\"grid\": %d, \"rows\": [\n", patches * slots);
    bool first = true;
    for (int live : {0, 5, 10}) {
        std::vector<uint2> h(patches); std::vector<uint32_t> hl;
        for (int p = 0; p < patches; p++) { h[p] = make_uint2(0u, (uint32_t)live); for (int q = 0; q < live; q++) hl.push_back(p * slots + q); }
        hipMemcpy(ranges, h.data(), patches * sizeof(uint2), hipMemcpyHostToDevice);
        if (!hl.empty()) hipMemcpy(list, hl.data(), hl.size() * 4, hipMemcpyHostToDevice);
        const uint32_t cnt = (uint32_t)hl.size();
        hipMemcpy(n, &cnt, 4, hipMemcpyHostToDevice);
        for (int work : {0, 2000}) {
            const float t_slots = time_ms(s, 50, [&] { hipLaunchKernelGGL(k_slots, dim3(patches * slots), dim3(64), 0, s, ranges, slots, work, out); });
            const float t_list_full = time_ms(s, 50, [&] { hipLaunchKernelGGL(k_list, dim3(patches * slots), dim3(64), 0, s, list, n, work, out); });
            const float t_list_exact = cnt ? time_ms(s, 50, [&] { hipLaunchKernelGGL(k_list, dim3(cnt), dim3(64), 0, s, list, n, work, out); }) : 0.f;
            const float t_persist = time_ms(s, 50, [&] { hipMemsetAsync(counter, 0, 4, s); hipLaunchKernelGGL(k_persist, dim3(256 * 8), dim3(64), 0, s, list, n, counter, work, out); });
            const int G = patches * slots;
            const float t_w2 = time_ms(s, 50, [&] { hipLaunchKernelGGL(k_slots_w<2>, dim3(G / 2), dim3(128), 0, s, ranges, slots, work, out); });
            const float t_w4 = time_ms(s, 50, [&] { hipLaunchKernelGGL(k_slots_w<4>, dim3(G / 4), dim3(256), 0, s, ranges, slots, work, out); });
            const float t_w8 = time_ms(s, 50, [&] { hipLaunchKernelGGL(k_slots_w<8>, dim3(G / 8), dim3(512), 0, s, ranges, slots, work, out); });
            const float t_w16 = time_ms(s, 50, [&] { hipLaunchKernelGGL(k_slots_w<16>, dim3(G / 16), dim3(1024), 0, s, ranges, slots, work, out); });
            printf("%s  {\"live_slots_per_patch\": %d, \"work_fma\": %d, \"slots_grid_us\": %.2f, \"list_full_grid_us\": %.2f, \"list_exact_grid_us\": %.2f, \"persistent_2048wg_us\": %.2f, "
                   "\"slots_2_waves_per_wg_us\": %.2f, \"slots_4_waves_per_wg_us\": %.2f, \"slots_8_waves_per_wg_us\": %.2f, \"slots_16_waves_per_wg_us\": %.2f}",
                   first ? "" : ",\n", live, work, t_slots * 1e3f, t_list_full * 1e3f, t_list_exact * 1e3f, t_persist * 1e3f, t_w2 * 1e3f, t_w4 * 1e3f, t_w8 * 1e3f, t_w16 * 1e3f);
            first = false;
        }
    }
    printf("\n]}\n");
    return 0;
}
