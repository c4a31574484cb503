// Where a radix-sort scatter launch spends its time: k_radix_scatter compiled with -DLG_PHASE_CLOCKS stores the 100-MHz wall clock at
// every phase boundary of every workgroup; this sorts n random pairs on one 8-bit digit (the range sort's first pass) and prints, per
// phase, the mean / p95 duration over the workgroups, and the launch's span from the first workgroup's start to the last one's end.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLG_PHASE_CLOCKS -I include -I lidar-gs_amd/csrc tools/micro/scatter_phases.cpp \
//         lidar-gs_amd/csrc/binning.hip -o /tmp/scp && /tmp/scp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>
#include "lidargs_common.h"
namespace lg { void phase_clocks_read(unsigned long long* out); }

static void run(size_t n, int bits, bool cold) {
    std::mt19937 rng(7);
    std::vector<uint32_t> hk(n), hv(n);
    for (size_t i = 0; i < n; i++) { hk[i] = rng(); hv[i] = (uint32_t)i; }
    uint32_t *ka, *kb, *va, *vb, *scratch;
    hipMalloc(&ka, n * 4 + 64); hipMalloc(&kb, n * 4 + 64); hipMalloc(&va, n * 4 + 64); hipMalloc(&vb, n * 4 + 64);
    hipMalloc(&scratch, lg::sort_scratch_words(n, lg::SORT_MAX_RADIX_BITS) * 4);
    hipStream_t s; hipStreamCreate(&s);
    hipMemcpy(ka, hk.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(va, hv.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    void* thrash = nullptr;
    if (cold) hipMalloc(&thrash, (size_t)1 << 30);
    for (int r = 0; r < 5; r++) {
        if (cold) { hipMemsetAsync(thrash, r, (size_t)1 << 30, s); }   // 1 GB through L2 and the 256-MB memory-side cache: the pairs come from HBM
        hipEventRecord(e0, s);
        lg::launch_radix_sort_pairs(ka, kb, va, vb, n, bits, scratch, s, bits, nullptr, lg::SORT_MAX_RADIX_BITS, false);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> c(8 * 4096);
    lg::phase_clocks_read(c.data());
    const size_t nb = std::min<size_t>(4096, (n + 2047) / 2048);
    const char* names[5] = {"load keys + digit bases", "rank (ballots, LDS counters)", "block-local digit starts", "park in LDS", "stream out + store drain"};
    unsigned long long t_min = ~0ull, t_max = 0;
    printf("{\"n\": %zu, \"digit_bits\": %d, \"input\": \"%s\", \"three_launches_ms\": %.4f, \"workgroups\": %zu, \"phases_us\": {", n, bits, cold ? "cold (1 GB written in between)" : "warm (the previous repetition's output side)", ms, nb);
    for (int p = 0; p < 5; p++) {
        std::vector<double> d;
        for (size_t b = 0; b < nb; b++) d.push_back((double)(c[b * 8 + p + 1] - c[b * 8 + p]) * 0.01);
        std::sort(d.begin(), d.end());
        double sum = 0; for (double x : d) sum += x;
        printf("%s\"%s\": {\"mean\": %.2f, \"p50\": %.2f, \"p95\": %.2f}", p ? ", " : "", names[p], sum / d.size(), d[d.size() / 2], d[d.size() * 95 / 100]);
    }
    std::vector<double> starts;
    for (size_t b = 0; b < nb; b++) { t_min = std::min(t_min, c[b * 8]); t_max = std::max(t_max, c[b * 8 + 5]); }
    for (size_t b = 0; b < nb; b++) starts.push_back((double)(c[b * 8] - t_min) * 0.01);
    std::sort(starts.begin(), starts.end());
    printf("}, \"scatter_span_us\": %.2f, \"workgroup_start_us\": {\"p50\": %.2f, \"p95\": %.2f, \"last\": %.2f}}\n", (double)(t_max - t_min) * 0.01,
           starts[starts.size() / 2], starts[starts.size() * 95 / 100], starts.back());
    if (thrash) hipFree(thrash);
    hipFree(ka); hipFree(kb); hipFree(va); hipFree(vb); hipFree(scratch);
}
int main() {
    run(2000000, 8, false);
    run(2000000, 8, true);
    run(2000000, 9, false);
    run(2000000, 9, true);
    run(500000, 8, false);
    return 0;
}
