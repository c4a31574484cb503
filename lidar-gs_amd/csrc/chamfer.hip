// chamfer.hip -- brute-force nearest neighbour between two point clouds (include/lidargs_chamfer.h).
//
// 170 k x 170 k points per evaluated frame = 2.9e10 point pairs per direction: pure VALU work (8 ops per pair as written, no
// contraction: the squared distance must round exactly like the reference's so that the argmin index is bit-identical), so the
// design is about issue efficiency: each thread owns CH_Q queries in registers, a block streams the other cloud through LDS in
// tiles, and every LDS read (a broadcast: all lanes read the same target) is amortised over CH_Q x 64 pairs.
// The reference (chamfer3D.cu:8-138) gives each thread one query and re-reads shared memory for every pair.
#include "lidargs_common.h"
#include "../../include/lidargs_rasterizer.h"
#include "../../include/lidargs_chamfer.h"

namespace lg {

#define CH_BLOCK 256
#define CH_Q 4
#define CH_TILE 1024
#define CH_SPLIT 8       // the target cloud is cut into CH_SPLIT slices per query block: 170 k queries alone are only 166 blocks

// One block = CH_BLOCK x CH_Q queries against one slice of the targets.  Partial results are merged with a 64-bit atomicMin on
// (distance bits << 32 | index): distances are >= 0, so their bit patterns order like the values, and among equal distances the
// lowest index wins -- exactly the reference's tie rule.
__global__ void __launch_bounds__(CH_BLOCK) k_chamfer_nn(int n, int m, const float* __restrict__ a, const float* __restrict__ b,
                                                         unsigned long long* __restrict__ keys) {
    __shared__ float s_b[CH_TILE * 3];
    const int batch = blockIdx.z;
    a += (size_t)batch * n * 3; b += (size_t)batch * m * 3; keys += (size_t)batch * n;
    const int q0 = (blockIdx.x * CH_BLOCK + threadIdx.x) * CH_Q;
    const int per = (m + CH_SPLIT - 1) / CH_SPLIT;
    const int m0 = blockIdx.y * per, m1 = min(m, m0 + per);
    float qx[CH_Q], qy[CH_Q], qz[CH_Q], best[CH_Q];
    int bi[CH_Q];
#pragma unroll
    for (int r = 0; r < CH_Q; r++) {
        const int q = min(q0 + r, n - 1);
        qx[r] = a[3 * (size_t)q]; qy[r] = a[3 * (size_t)q + 1]; qz[r] = a[3 * (size_t)q + 2];
        best[r] = __int_as_float(0x7f800000); bi[r] = 0;                // +inf: the first target always wins, as `k == 0 ||` does
    }
    for (int t0 = m0; t0 < m1; t0 += CH_TILE) {
        const int cnt = min(CH_TILE, m1 - t0);
        __syncthreads();
        for (int j = threadIdx.x; j < cnt * 3; j += CH_BLOCK) s_b[j] = b[(size_t)t0 * 3 + j];
        __syncthreads();
        for (int k = 0; k < cnt; k++) {
            const float bx = s_b[3 * k], by = s_b[3 * k + 1], bz = s_b[3 * k + 2];
#pragma unroll
            for (int r = 0; r < CH_Q; r++) {
                const float dx = bx - qx[r], dy = by - qy[r], dz = bz - qz[r];      // chamfer3D.cu:36-38
                const float d = dx * dx + dy * dy + dz * dz;                        // :39 (this file is built with -ffp-contract=off)
                const bool better = d < best[r];                                    // strict: the lowest index wins ties (:40, :130)
                best[r] = better ? d : best[r];
                bi[r] = better ? (t0 + k) : bi[r];
            }
        }
    }
    if (m1 > m0) {
#pragma unroll
        for (int r = 0; r < CH_Q; r++)
            if (q0 + r < n) atomicMin(keys + q0 + r, ((unsigned long long)__float_as_uint(best[r]) << 32) | (unsigned)bi[r]);
    }
}

__global__ void __launch_bounds__(256) k_chamfer_unpack(size_t count, const unsigned long long* __restrict__ keys, float* __restrict__ dist, int* __restrict__ idx) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const unsigned long long k = keys[i];
    dist[i] = __uint_as_float((unsigned)(k >> 32)); idx[i] = (int)(unsigned)(k & 0xFFFFFFFFull);
}

// chamfer3D.cu:167-195: g = 2 grad_dist; +g (p - q) to the point, -g (p - q) to its neighbour (float atomics: many points share one)
__global__ void __launch_bounds__(256) k_chamfer_grad(int n, int m, const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ g_dist, const int* __restrict__ idx, float* __restrict__ ga,
                                                      float* __restrict__ gb) {
    const int batch = blockIdx.y;
    a += (size_t)batch * n * 3; b += (size_t)batch * m * 3; g_dist += (size_t)batch * n; idx += (size_t)batch * n;
    ga += (size_t)batch * n * 3; gb += (size_t)batch * m * 3;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int j2 = idx[j];
    const float g = g_dist[j] * 2.f;
    for (int c = 0; c < 3; c++) {
        const float v = g * (a[3 * (size_t)j + c] - b[3 * (size_t)j2 + c]);
        atomicAdd(ga + 3 * (size_t)j + c, v);
        atomicAdd(gb + 3 * (size_t)j2 + c, -v);
    }
}

}  // namespace lg

extern "C" {

size_t lidargs_chamfer_scratch_bytes(int B, int n, int m) {
    return sizeof(unsigned long long) * ((size_t)(B > 0 ? B : 0) * ((size_t)(n > 0 ? n : 0) + (size_t)(m > 0 ? m : 0))) + 256;
}

int lidargs_chamfer_forward(int B, int n, int m, const float* xyz1, const float* xyz2, float* dist1, float* dist2, int* idx1, int* idx2,
                            char* scratch, size_t scratch_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (B < 0 || n < 0 || m < 0) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "chamfer_forward: negative size");
    if (B == 0) return 0;
    if ((n > 0 && m == 0) || (m > 0 && n == 0)) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "chamfer_forward: one of the clouds is empty");
    if (n == 0) return 0;
    if (!xyz1 || !xyz2 || !dist1 || !dist2 || !idx1 || !idx2 || !scratch) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "chamfer_forward: NULL pointer");
    if (scratch_bytes < lidargs_chamfer_scratch_bytes(B, n, m)) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "chamfer_forward: scratch too small");
    unsigned long long* k1 = reinterpret_cast<unsigned long long*>(((uintptr_t)scratch + 127) & ~(uintptr_t)127);
    unsigned long long* k2 = k1 + (size_t)B * n;
    hipError_t e = hipMemsetAsync(k1, 0xFF, sizeof(unsigned long long) * (size_t)B * ((size_t)n + m), stream);
    if (e != hipSuccess) return lg::api_fail(LIDARGS_ERR_HIP, hipGetErrorString(e));
    const int per = CH_BLOCK * CH_Q;
    hipLaunchKernelGGL(lg::k_chamfer_nn, dim3((n + per - 1) / per, CH_SPLIT, B), dim3(CH_BLOCK), 0, stream, n, m, xyz1, xyz2, k1);
    hipLaunchKernelGGL(lg::k_chamfer_nn, dim3((m + per - 1) / per, CH_SPLIT, B), dim3(CH_BLOCK), 0, stream, m, n, xyz2, xyz1, k2);
    hipLaunchKernelGGL(lg::k_chamfer_unpack, dim3((unsigned)(((size_t)B * n + 255) / 256)), dim3(256), 0, stream, (size_t)B * n, k1, dist1, idx1);
    hipLaunchKernelGGL(lg::k_chamfer_unpack, dim3((unsigned)(((size_t)B * m + 255) / 256)), dim3(256), 0, stream, (size_t)B * m, k2, dist2, idx2);
    e = hipGetLastError();
    if (e != hipSuccess) return lg::api_fail(LIDARGS_ERR_HIP, hipGetErrorString(e));
    return 0;
}

int lidargs_chamfer_backward(int B, int n, int m, const float* xyz1, const float* xyz2, const float* grad_dist1, const float* grad_dist2,
                             const int* idx1, const int* idx2, float* grad_xyz1, float* grad_xyz2, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (B < 0 || n < 0 || m < 0) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "chamfer_backward: negative size");
    if (B == 0 || n == 0 || m == 0) return 0;
    if (!xyz1 || !xyz2 || !grad_dist1 || !grad_dist2 || !idx1 || !idx2 || !grad_xyz1 || !grad_xyz2)
        return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "chamfer_backward: NULL pointer");
    hipLaunchKernelGGL(lg::k_chamfer_grad, dim3((n + 255) / 256, B), dim3(256), 0, stream, n, m, xyz1, xyz2, grad_dist1, idx1, grad_xyz1, grad_xyz2);
    hipLaunchKernelGGL(lg::k_chamfer_grad, dim3((m + 255) / 256, B), dim3(256), 0, stream, m, n, xyz2, xyz1, grad_dist2, idx2, grad_xyz2, grad_xyz1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lg::api_fail(LIDARGS_ERR_HIP, hipGetErrorString(e));
    return 0;
}

}  // extern "C"
