// Times lg::launch_radix_sort_pairs / _pairs16 on small inputs (the single-launch k_radix_sort_small path) and checks them against
// std::stable_sort.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I lidar-gs_amd/csrc tools/micro/small_sort_bench.cpp \
//                          lidar-gs_amd/csrc/binning.hip -o /tmp/ssb && /tmp/ssb
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>
#include "lidargs_common.h"

template <typename KT>
static void run(size_t n, int bits, int reps) {
    std::mt19937 rng(7);
    std::vector<KT> hk(n); std::vector<uint32_t> hv(n);
    for (size_t i = 0; i < n; i++) { hk[i] = (KT)(rng() & ((1u << bits) - 1u)); hv[i] = (uint32_t)i; }
    KT *ka, *kb; uint32_t *va, *vb, *scratch;
    hipMalloc(&ka, n * sizeof(KT) + 64); hipMalloc(&kb, n * sizeof(KT) + 64); hipMalloc(&va, n * 4 + 64); hipMalloc(&vb, n * 4 + 64);
    const size_t sw = lg::sort_scratch_words(n);
    hipMalloc(&scratch, sw * 4);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, sum = 0.f;
    int side = 0;
    for (int r = 0; r < reps + 3; r++) {
        hipMemcpyAsync(ka, hk.data(), n * sizeof(KT), hipMemcpyHostToDevice, s); hipMemcpyAsync(va, hv.data(), n * 4, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        if constexpr (sizeof(KT) == 2) side = lg::launch_radix_sort_pairs16(ka, kb, va, vb, n, bits, scratch, s);
        else side = lg::launch_radix_sort_pairs(ka, kb, va, vb, n, bits, scratch, s);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r >= 3) { best = std::min(best, ms); sum += ms; }
    }
    std::vector<KT> ok(n); std::vector<uint32_t> ov(n);
    hipMemcpy(ok.data(), side ? kb : ka, n * sizeof(KT), hipMemcpyDeviceToHost); hipMemcpy(ov.data(), side ? vb : va, n * 4, hipMemcpyDeviceToHost);
    std::vector<uint32_t> idx(n); std::iota(idx.begin(), idx.end(), 0u);
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return hk[a] < hk[b]; });
    size_t bad = 0;
    for (size_t i = 0; i < n; i++) bad += (ov[i] != idx[i]) || (ok[i] != hk[idx[i]]);
    printf("{\"key_bytes\": %d, \"n\": %zu, \"bits\": %d, \"side\": %d, \"best_us\": %.1f, \"mean_us\": %.1f, \"mismatches\": %zu}\n", (int)sizeof(KT), n, bits, side, best * 1e3f,
           sum / reps * 1e3f, bad);
    hipFree(ka); hipFree(kb); hipFree(va); hipFree(vb); hipFree(scratch);
}

int main() {
    for (size_t n : {1000, 4000, 10000, 14087, 16384}) { run<uint16_t>(n, 7, 20); run<uint16_t>(n, 12, 20); run<uint32_t>(n, 8, 20); run<uint32_t>(n, 26, 20); }
    run<uint32_t>(20000, 26, 20);
    return 0;
}
