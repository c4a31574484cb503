// lidar_loss.hip -- fused per-frame image loss + gradient (C ABI in include/lidargs_loss.h).
//
// The reference evaluates train.py:150-203 as ~45 framework kernels over 64 x 2650 images (5 depthwise 11 x 11 convolutions
// for SSIM, a dozen masks / abs / means) plus their autograd mirror.  The images are tiny (0.7 MB a plane): the cost is
// launches, not bytes.  Here: 6 launches, one thread per pixel, no atomics (per-block partial sums folded in a fixed order):
//   k_loss_pointwise   L1 terms, ray-drop MSE, depth-difference term and their gradients
//   k_ssim_rows        horizontal 11-tap pass over (X, Y, X^2, Y^2, XY)                           X = intensity * mask
//   k_ssim_cols        vertical pass -> SSIM map, its sum, and dS/d(mu1, p11, p12) scaled by dL/dS
//   k_ssim_rows3 / k_ssim_cols3   the transposed convolution of those three maps (the window is symmetric) -> dL/dX
//   k_loss_finish      fold the partial sums into the six loss values
#include "lidargs_common.h"
#include "../../include/lidargs_rasterizer.h"
#include "../../include/lidargs_loss.h"

namespace lg {

#define LL_TAPS 11
#define LL_PAD 5
#define LL_BLOCK 256
struct LossWin { float w[LL_TAPS]; };

struct LossScratch { float* t5; float* m3; float* t3; float* gx; float* part; };
static size_t loss_carve(char* base, size_t N, size_t blocks, LossScratch* v) {
    Carver c(base);
    LossScratch s;
    s.t5 = c.take<float>(5 * N); s.m3 = c.take<float>(3 * N); s.t3 = c.take<float>(3 * N); s.gx = c.take<float>(N);
    s.part = c.take<float>(8 * blocks + 64);
    if (v) *v = s;
    return (size_t)(c.p - base) + 256;
}

__device__ __forceinline__ float ll_sign(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

// block sum of up to 5 values -> part[block*8 + k]
__device__ __forceinline__ void ll_block_sums(float (&v)[5], float* __restrict__ part) {
    __shared__ float s[5][LL_BLOCK / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        float x = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
        if (lane == 0) s[k][w] = x;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        float x = 0.f;
        for (int q = 0; q < LL_BLOCK / 64; q++) x += s[threadIdx.x][q];
        part[blockIdx.x * 8 + threadIdx.x] = x;
    }
}

__global__ void __launch_bounds__(LL_BLOCK) k_loss_pointwise(int H, int W, const float* __restrict__ image, const float* __restrict__ depth,
                                                             const float* __restrict__ gt, float lambda, float* __restrict__ g_image,
                                                             float* __restrict__ g_depth, float* __restrict__ gx_l1, float* __restrict__ part) {
    const int N = H * W;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < N) {
        const int x = i % W;
        const float invN = 1.f / (float)N, invNg = 1.f / (float)(H * (W - 1));
        const float rd = gt[i], gi = gt[N + i] * rd, gd = gt[2 * N + i] * rd;         // train.py:151-153
        const float X = image[i] * rd, dm = depth[i] * rd, rr = image[N + i];       // :161-162
        const float e1 = X - gi, e2 = dm - gd, e3 = rr - rd;
        v[0] = fabsf(e1); v[1] = fabsf(e2); v[2] = e3 * e3;
        g_image[N + i] = 10.0f * 2.0f * e3 * invN;                                   // :165
        gx_l1[i] = (1.f - lambda) * ll_sign(e1) * invN;                              // :171, :203 (times the mask later)
        float gdm = ll_sign(e2) * invN;                                              // :172
        // horizontal depth differences (:190-201): this pixel is the left element of pair (x, x+1) and the right one of (x-1, x)
        if (x + 1 < W) {
            const float rdn = gt[i + 1], dmn = depth[i + 1] * rdn, gdn = gt[2 * N + i + 1] * rdn;
            const float pg = fabsf(dm - dmn), gg = fabsf(gd - gdn);
            const float m = rd * ((gg < 0.01f) ? 1.f : 0.f);
            const float t = pg * m - gg * m;
            v[3] = fabsf(t);
            gdm += m * ll_sign(t) * invNg * ll_sign(dm - dmn);
        }
        if (x > 0) {
            const float rdp = gt[i - 1], dmp = depth[i - 1] * rdp, gdp = gt[2 * N + i - 1] * rdp;
            const float pg = fabsf(dmp - dm), gg = fabsf(gdp - gd);
            const float m = rdp * ((gg < 0.01f) ? 1.f : 0.f);
            const float t = pg * m - gg * m;
            gdm -= m * ll_sign(t) * invNg * ll_sign(dmp - dm);
        }
        g_depth[i] = gdm * rd;
    }
    ll_block_sums(v, part);
}

// horizontal pass over the five SSIM moments of (X, Y) = (intensity * mask, gt intensity * mask)
__global__ void __launch_bounds__(LL_BLOCK) k_ssim_rows(int H, int W, const float* __restrict__ image, const float* __restrict__ gt, LossWin win,
                                                        float* __restrict__ t5) {
    const int N = H * W;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int x = i % W;
    float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
    for (int k = 0; k < LL_TAPS; k++) {
        const int xx = x + k - LL_PAD;
        if (xx < 0 || xx >= W) continue;
        const int j = i + k - LL_PAD;
        const float rd = gt[j], X = image[j] * rd, Y = gt[N + j] * rd, w = win.w[k];
        a += w * X; b += w * Y; aa += w * X * X; bb += w * Y * Y; ab += w * X * Y;
    }
    t5[i] = a; t5[N + i] = b; t5[2 * N + i] = aa; t5[3 * N + i] = bb; t5[4 * N + i] = ab;
}

// vertical pass -> SSIM (loss_utils.py:42-57), its block sums, and the three gradient maps dL/dmu1, dL/dp11, dL/dp12
__global__ void __launch_bounds__(LL_BLOCK) k_ssim_cols(int H, int W, const float* __restrict__ t5, LossWin win, float lambda, float* __restrict__ m3,
                                                        float* __restrict__ part) {
    const int N = H * W;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < N) {
        const int y = i / W;
        float mu1 = 0.f, mu2 = 0.f, p11 = 0.f, p22 = 0.f, p12 = 0.f;
#pragma unroll
        for (int k = 0; k < LL_TAPS; k++) {
            const int yy = y + k - LL_PAD;
            if (yy < 0 || yy >= H) continue;
            const int j = i + (k - LL_PAD) * W;
            const float w = win.w[k];
            mu1 += w * t5[j]; mu2 += w * t5[N + j]; p11 += w * t5[2 * N + j]; p22 += w * t5[3 * N + j]; p12 += w * t5[4 * N + j];
        }
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float s11 = p11 - mu1 * mu1, s22 = p22 - mu2 * mu2, s12 = p12 - mu1 * mu2;
        const float A1 = 2.f * mu1 * mu2 + C1, A2 = 2.f * s12 + C2, B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s11 + s22 + C2;
        const float iB = 1.f / (B1 * B2);
        const float S = A1 * A2 * iB;
        v[0] = S;
        const float dS = -lambda / (float)N;                                        // d loss / d S (train.py:173, :203)
        const float d12 = 2.f * A1 * iB, d11 = -S / B2;
        const float dm1 = 2.f * mu2 * A2 * iB - S * 2.f * mu1 / B1 - d12 * mu2 - d11 * 2.f * mu1;
        m3[i] = dS * dm1; m3[N + i] = dS * d11; m3[2 * N + i] = dS * d12;
    }
    ll_block_sums(v, part);
}

__global__ void __launch_bounds__(LL_BLOCK) k_ssim_rows3(int H, int W, const float* __restrict__ m3, LossWin win, float* __restrict__ t3) {
    const int N = H * W;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int x = i % W;
    float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
    for (int k = 0; k < LL_TAPS; k++) {
        const int xx = x + k - LL_PAD;
        if (xx < 0 || xx >= W) continue;
        const int j = i + k - LL_PAD;
        const float w = win.w[k];
        a += w * m3[j]; b += w * m3[N + j]; c += w * m3[2 * N + j];
    }
    t3[i] = a; t3[N + i] = b; t3[2 * N + i] = c;
}

// dL/dX = G*(gm1) + 2 X G*(g11) + Y G*(g12) + the L1 part;  dL/dimage[0] = dL/dX * mask
__global__ void __launch_bounds__(LL_BLOCK) k_ssim_cols3(int H, int W, const float* __restrict__ t3, LossWin win, const float* __restrict__ image,
                                                         const float* __restrict__ gt, const float* __restrict__ gx_l1, float* __restrict__ g_image) {
    const int N = H * W;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int y = i / W;
    float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
    for (int k = 0; k < LL_TAPS; k++) {
        const int yy = y + k - LL_PAD;
        if (yy < 0 || yy >= H) continue;
        const int j = i + (k - LL_PAD) * W;
        const float w = win.w[k];
        a += w * t3[j]; b += w * t3[N + j]; c += w * t3[2 * N + j];
    }
    const float rd = gt[i], X = image[i] * rd, Y = gt[N + i] * rd;
    g_image[i] = (gx_l1[i] + a + 2.f * X * b + Y * c) * rd;
}

__global__ void __launch_bounds__(64) k_loss_finish(int H, int W, int blocks, const float* __restrict__ part_a, const float* __restrict__ part_s,
                                                    float lambda, float* __restrict__ losses) {
    const int lane = threadIdx.x;
    double s[5] = {0, 0, 0, 0, 0};
    for (int b = lane; b < blocks; b += 64) {
        s[0] += part_a[b * 8]; s[1] += part_a[b * 8 + 1]; s[2] += part_a[b * 8 + 2]; s[3] += part_a[b * 8 + 3]; s[4] += part_s[b * 8];
    }
#pragma unroll
    for (int k = 0; k < 5; k++)
        for (int o = 32; o > 0; o >>= 1) s[k] += __shfl_xor(s[k], o);
    if (lane == 0) {
        const double N = (double)H * W;
        const double Ll1 = s[0] / N, dl = s[1] / N, rdl = 10.0 * s[2] / N, gl = s[3] / ((double)H * (W - 1)), ssim_loss = 1.0 - s[4] / N;
        losses[0] = (float)(dl + (1.0 - lambda) * Ll1 + lambda * ssim_loss + rdl + gl);
        losses[1] = (float)Ll1; losses[2] = (float)dl; losses[3] = (float)ssim_loss; losses[4] = (float)rdl; losses[5] = (float)gl;
    }
}


// ---- scaling_reg = weight * mean(prod(scaling, dim=1))  (train.py:174) and its gradient, the per-Gaussian term of the frame loss -------
// One thread per Gaussian row: the product as torch.prod forms it ((s0 s1) s2 in fp32), the gradient row weight / M x (s1 s2, s0 s2, s0 s1)
// -- the products themselves, where the framework's prod backward divides the result by each input after a host-synchronising look for
// zeros -- and the block's sum of products in double; k_scaling_reg_finish folds the block sums in a fixed order (deterministic) and,
// if asked, adds the value onto the image loss.
#define SR_BLOCK 256
__global__ void __launch_bounds__(SR_BLOCK) k_scaling_reg(int M, const float* __restrict__ scaling, float wm, float* __restrict__ grad, double* __restrict__ part) {
    __shared__ double ws[SR_BLOCK / 64];
    const int i = blockIdx.x * SR_BLOCK + threadIdx.x;
    double p = 0.0;
    if (i < M) {
        const float s0 = scaling[3 * (size_t)i], s1 = scaling[3 * (size_t)i + 1], s2 = scaling[3 * (size_t)i + 2];
        p = (double)((s0 * s1) * s2);
        if (grad) { grad[3 * (size_t)i] = wm * (s1 * s2); grad[3 * (size_t)i + 1] = wm * (s0 * s2); grad[3 * (size_t)i + 2] = wm * (s0 * s1); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = p;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int k = 0; k < SR_BLOCK / 64; k++) t += ws[k]; part[blockIdx.x] = t; }
}
__global__ void __launch_bounds__(64) k_scaling_reg_finish(int M, int blocks, const double* __restrict__ part, float weight, float* __restrict__ value,
                                                           float* __restrict__ add_to) {
    const int lane = threadIdx.x;
    double s = 0.0;
    for (int b = lane; b < blocks; b += 64) s += part[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        const float v = weight * (float)(s / (double)M);              // M = 0: 0 / 0 = NaN, as the mean of an empty tensor is
        if (value) *value = v;
        if (add_to) *add_to += v;
    }
}

}  // namespace lg

extern "C" {

size_t lidargs_loss_scratch_bytes(int H, int W) {
    const size_t N = (size_t)(H > 0 ? H : 1) * (size_t)(W > 0 ? W : 1);
    return lg::loss_carve(nullptr, N, 2 * ((N + LL_BLOCK - 1) / LL_BLOCK), nullptr);
}

int lidargs_image_loss(int H, int W, const float* image, const float* depth, const float* gt, float lambda_dssim,
                       float* losses, float* dL_dimage, float* dL_ddepth, char* scratch, size_t scratch_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (H < 1 || W < 2) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "image_loss: needs H >= 1 and W >= 2");
    if (!image || !depth || !gt || !losses || !dL_dimage || !dL_ddepth || !scratch) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "image_loss: NULL pointer");
    if (scratch_bytes < lidargs_loss_scratch_bytes(H, W)) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "image_loss: scratch too small");
    const size_t N = (size_t)H * W;
    const unsigned blocks = (unsigned)((N + LL_BLOCK - 1) / LL_BLOCK);
    lg::LossScratch s; lg::loss_carve(scratch, N, 2 * (size_t)blocks, &s);
    lg::LossWin win;
    {   // utils/loss_utils.py:27-29: exp(-(x - 5)^2 / (2 * 1.5^2)) normalised, in float32 as torch.Tensor holds it
        float sum = 0.f;
        for (int k = 0; k < LL_TAPS; k++) { win.w[k] = (float)exp(-(double)((k - LL_PAD) * (k - LL_PAD)) / (2.0 * 1.5 * 1.5)); sum += win.w[k]; }
        for (int k = 0; k < LL_TAPS; k++) win.w[k] /= sum;
    }
    float* part_a = s.part; float* part_s = s.part + 8 * (size_t)blocks;
    hipLaunchKernelGGL(lg::k_loss_pointwise, dim3(blocks), dim3(LL_BLOCK), 0, stream, H, W, image, depth, gt, lambda_dssim, dL_dimage, dL_ddepth, s.gx, part_a);
    hipLaunchKernelGGL(lg::k_ssim_rows, dim3(blocks), dim3(LL_BLOCK), 0, stream, H, W, image, gt, win, s.t5);
    hipLaunchKernelGGL(lg::k_ssim_cols, dim3(blocks), dim3(LL_BLOCK), 0, stream, H, W, s.t5, win, lambda_dssim, s.m3, part_s);
    hipLaunchKernelGGL(lg::k_ssim_rows3, dim3(blocks), dim3(LL_BLOCK), 0, stream, H, W, s.m3, win, s.t3);
    hipLaunchKernelGGL(lg::k_ssim_cols3, dim3(blocks), dim3(LL_BLOCK), 0, stream, H, W, s.t3, win, image, gt, s.gx, dL_dimage);
    hipLaunchKernelGGL(lg::k_loss_finish, dim3(1), dim3(64), 0, stream, H, W, (int)blocks, part_a, part_s, lambda_dssim, losses);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lg::api_fail(LIDARGS_ERR_HIP, hipGetErrorString(e));
    return 0;
}

size_t lidargs_scaling_reg_scratch_bytes(int M) {
    const size_t blocks = ((size_t)(M > 0 ? M : 1) + SR_BLOCK - 1) / SR_BLOCK;
    return blocks * sizeof(double) + 256;
}

int lidargs_scaling_reg(int M, const float* scaling, float weight, float* value, float* add_to, float* dL_dscaling, char* scratch,
                        size_t scratch_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (M < 0) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "scaling_reg: M < 0");
    if ((M > 0 && !scaling) || !scratch || (!value && !add_to)) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "scaling_reg: NULL pointer");
    if (scratch_bytes < lidargs_scaling_reg_scratch_bytes(M)) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "scaling_reg: scratch too small");
    double* part = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(scratch) + 127) & ~uintptr_t(127));
    const unsigned blocks = (unsigned)(((size_t)M + SR_BLOCK - 1) / SR_BLOCK);
    if (blocks) hipLaunchKernelGGL(lg::k_scaling_reg, dim3(blocks), dim3(SR_BLOCK), 0, stream, M, scaling, M > 0 ? weight / (float)M : 0.f, dL_dscaling, part);
    hipLaunchKernelGGL(lg::k_scaling_reg_finish, dim3(1), dim3(64), 0, stream, M, (int)blocks, part, weight, value, add_to);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return lg::api_fail(LIDARGS_ERR_HIP, hipGetErrorString(e));
    return 0;
}

}  // extern "C"
