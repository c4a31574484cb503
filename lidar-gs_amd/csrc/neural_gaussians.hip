// neural_gaussians.hip -- fused anchor decode (C ABI in include/lidargs_neural_gaussians.h).
//
// The reference builds the rasterizer's inputs with ~40 PyTorch ops per frame (gaussian_renderer/__init__.py:17-119): boolean
// indexing of four tensors, four Linear-ReLU-Linear MLPs, three concatenations, an einops repeat, a masked gather of a
// [n*k, 6+3+2+7+3] temporary and the activations.  Here one thread owns one anchor: its 36-float input and the 32 hidden units
// live in registers, the weights are wave-uniform (scalar loads, no LDS), and the only thing written is the compact result.
//   k_ng_visflags   bool mask -> u32 flags                      } + two exclusive scans (binning.hip): compact anchor index and
//   k_ng_opacity    :22-68   view, distance, opacity MLP, mask  }   output row of every selected (anchor, offset) pair
//   k_ng_decode     :70-113  colour / ray-drop / covariance MLPs + post-processing, selected pairs only
//   k_ng_backward   the VJP of all of the above per anchor; weight gradients are left as plain GEMMs of what it writes
// Per anchor: 128 + 12 + 12k + 24 B read, 52 B written per selected pair; ~7 k FMA -> memory-bound by ~3x on MI355X.
#include "lidargs_common.h"
#include "../../include/lidargs_rasterizer.h"
#include "../../include/lidargs_neural_gaussians.h"

namespace lg {

#define NG_FEAT 32
#define NG_HID 32
#define NG_IN 36      // feature 32 + view 3 + distance 1

struct NgModel { int k; int din[4]; const float* W1[4]; const float* b1[4]; const float* W2[4]; const float* b2[4]; const float* W2T[4]; };
enum { NG_OPA = 0, NG_COV = 1, NG_COL = 2, NG_RD = 3 };

struct NgScratch { uint32_t* vis_flags; uint32_t* vis_idx; uint32_t* sel_flags; uint32_t* slot; uint32_t* totals; uint32_t* scan; };
static size_t ng_carve(char* base, size_t N, size_t k, NgScratch* v) {
    Carver c(base);
    NgScratch s;
    const size_t n = N ? N : 1;
    s.vis_flags = c.take<uint32_t>(n); s.vis_idx = c.take<uint32_t>(n);
    s.sel_flags = c.take<uint32_t>(n * k); s.slot = c.take<uint32_t>(n * k);
    s.totals = c.take<uint32_t>(64);
    s.scan = c.take<uint32_t>(scan_scratch_words(n * k) + 64);
    if (v) *v = s;
    return (size_t)(c.p - base) + 256;
}

__device__ __forceinline__ float ng_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// x = (feature, view, distance)  (:28-34, :50)
__device__ __forceinline__ void ng_input(const float* __restrict__ feat, const float* __restrict__ anchor, float3 cam, int i, float (&x)[NG_IN]) {
    const float4* f4 = reinterpret_cast<const float4*>(feat + (size_t)i * NG_FEAT);
#pragma unroll
    for (int q = 0; q < NG_FEAT / 4; q++) { const float4 v = f4[q]; x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w; }
    const float ox = anchor[3 * (size_t)i] - cam.x, oy = anchor[3 * (size_t)i + 1] - cam.y, oz = anchor[3 * (size_t)i + 2] - cam.z;
    const float dist = sqrtf(ox * ox + oy * oy + oz * oz);
    x[32] = ox / dist; x[33] = oy / dist; x[34] = oz / dist; x[35] = dist;
}

// hidden layer: h = relu(W1 x + b1); W1 is [32][din], din = 35 or 36 (wave-uniform -> scalar loads)
__device__ __forceinline__ void ng_hidden(const float* __restrict__ W1, const float* __restrict__ b1, int din, const float (&x)[NG_IN], float (&h)[NG_HID]) {
#pragma unroll
    for (int o = 0; o < NG_HID; o++) {
        const float* w = W1 + o * din;
        float acc = b1[o];
#pragma unroll
        for (int i = 0; i < NG_IN - 1; i++) acc += w[i] * x[i];
        if (din == NG_IN) acc += w[NG_IN - 1] * x[NG_IN - 1];
        h[o] = fmaxf(acc, 0.f);
    }
}
template <int DOUT>
__device__ __forceinline__ void ng_output(const float* __restrict__ W2, const float* __restrict__ b2, const float (&h)[NG_HID], float (&y)[DOUT]) {
#pragma unroll
    for (int o = 0; o < DOUT; o++) {
        float acc = b2[o];
#pragma unroll
        for (int t = 0; t < NG_HID; t++) acc += W2[o * NG_HID + t] * h[t];
        y[o] = acc;
    }
}

__global__ void __launch_bounds__(256) k_ng_visflags(int N, const uint8_t* __restrict__ mask, uint32_t* __restrict__ flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) flags[i] = (!mask || mask[i]) ? 1u : 0u;
}

template <int K>
__global__ void __launch_bounds__(64) k_ng_opacity(int N, NgModel m, float3 cam, const float* __restrict__ feat, const float* __restrict__ anchor,
                                                   const uint32_t* __restrict__ vis_flags, const uint32_t* __restrict__ vis_idx,
                                                   float* __restrict__ neural_opacity, uint8_t* __restrict__ mask, uint32_t* __restrict__ sel_flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (!vis_flags[i]) {
#pragma unroll
        for (int j = 0; j < K; j++) sel_flags[(size_t)i * K + j] = 0u;
        return;
    }
    float x[NG_IN], h[NG_HID], y[K];
    ng_input(feat, anchor, cam, i, x);
    ng_hidden(m.W1[NG_OPA], m.b1[NG_OPA], m.din[NG_OPA], x, h);
    ng_output<K>(m.W2[NG_OPA], m.b2[NG_OPA], h, y);
    const size_t c = vis_idx[i];
#pragma unroll
    for (int j = 0; j < K; j++) {
        const float o = tanhf(y[j]);                                   // nn.Tanh closes the opacity MLP (gaussian_model.py:118)
        neural_opacity[c * K + j] = o;
        const bool keep = o > 0.0f;                                    // :67
        mask[c * K + j] = keep ? 1 : 0;
        sel_flags[(size_t)i * K + j] = keep ? 1u : 0u;
    }
}

template <int K>
__global__ void __launch_bounds__(64) k_ng_decode(int N, NgModel m, float3 cam, const float* __restrict__ feat, const float* __restrict__ anchor,
                                                  const float* __restrict__ offset, const float* __restrict__ scaling,
                                                  const uint32_t* __restrict__ vis_flags, const uint32_t* __restrict__ vis_idx,
                                                  const uint32_t* __restrict__ sel_flags, const uint32_t* __restrict__ slot,
                                                  const float* __restrict__ neural_opacity, float* __restrict__ o_xyz, float* __restrict__ o_color,
                                                  float* __restrict__ o_opacity, float* __restrict__ o_scaling, float* __restrict__ o_rot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || !vis_flags[i]) return;
    bool any = false;
#pragma unroll
    for (int j = 0; j < K; j++) any = any || sel_flags[(size_t)i * K + j] != 0u;
    if (!any) return;
    float x[NG_IN], h[NG_HID];
    ng_input(feat, anchor, cam, i, x);
    float col[K], rd[K], sr[7 * K];
    ng_hidden(m.W1[NG_COL], m.b1[NG_COL], m.din[NG_COL], x, h);
    ng_output<K>(m.W2[NG_COL], m.b2[NG_COL], h, col);
    ng_hidden(m.W1[NG_RD], m.b1[NG_RD], m.din[NG_RD], x, h);
    ng_output<K>(m.W2[NG_RD], m.b2[NG_RD], h, rd);
    ng_hidden(m.W1[NG_COV], m.b1[NG_COV], m.din[NG_COV], x, h);
    ng_output<7 * K>(m.W2[NG_COV], m.b2[NG_COV], h, sr);
    const float* sc = scaling + 6 * (size_t)i;
    const float s0 = sc[0], s1 = sc[1], s2 = sc[2], s3 = sc[3], s4 = sc[4], s5 = sc[5];
    const float ax = anchor[3 * (size_t)i], ay = anchor[3 * (size_t)i + 1], az = anchor[3 * (size_t)i + 2];
    const size_t c = vis_idx[i];
#pragma unroll
    for (int j = 0; j < K; j++) {
        if (!sel_flags[(size_t)i * K + j]) continue;
        const size_t r = slot[(size_t)i * K + j];
        const float* of = offset + 3 * ((size_t)i * K + j);
        o_xyz[3 * r] = ax + of[0] * s0; o_xyz[3 * r + 1] = ay + of[1] * s1; o_xyz[3 * r + 2] = az + of[2] * s2;      // :111-112
        o_color[2 * r] = ng_sigmoid(col[j]); o_color[2 * r + 1] = ng_sigmoid(rd[j]);                                // :85-87
        o_opacity[r] = neural_opacity[c * K + j];                                                                    // :71
        o_scaling[3 * r] = s3 * ng_sigmoid(sr[7 * j]); o_scaling[3 * r + 1] = s4 * ng_sigmoid(sr[7 * j + 1]);       // :107
        o_scaling[3 * r + 2] = s5 * ng_sigmoid(sr[7 * j + 2]);
        const float q0 = sr[7 * j + 3], q1 = sr[7 * j + 4], q2 = sr[7 * j + 5], q3 = sr[7 * j + 6];
        const float qn = fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f);                               // F.normalize, :108
        o_rot[4 * r] = q0 / qn; o_rot[4 * r + 1] = q1 / qn; o_rot[4 * r + 2] = q2 / qn; o_rot[4 * r + 3] = q3 / qn;
    }
}

// ---- backward ---------------------------------------------------------------------------------------------------------------
// The hidden activations of the MLP being processed live in LDS (one column per lane), so the loops over the 32 hidden units
// are real loops: the fully unrolled formulation (32 x 36 + 32 x 42 FMAs, four times) overwhelmed the register allocator
// (512 VGPRs + 1600 spilled).  Inputs x, their gradient dx and the output-layer deltas stay in registers.
#define NG_BLOCK 64
#define NG_XS 40     // row stride of act_x: 36 inputs, a constant 1 (bias gradients fall out of the same GEMM), 3 zeros
#define NG_HS 132    // row stride of act_h: 4 x 32 hidden units, a constant 1, 3 zeros

// forward recompute of one MLP: hidden units -> LDS, outputs -> registers
template <int DOUT>
__device__ __forceinline__ void ng_recompute(const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ W2T,
                                             const float* __restrict__ b2, int din, const float (&x)[NG_IN], float* __restrict__ s_h, int lane,
                                             float (&y)[DOUT]) {
#pragma unroll 1
    for (int o = 0; o < NG_HID; o++) {
        const float* w = W1 + o * din;
        float acc = b1[o];
#pragma unroll
        for (int i = 0; i < NG_IN - 1; i++) acc += w[i] * x[i];
        if (din == NG_IN) acc += w[NG_IN - 1] * x[NG_IN - 1];
        s_h[o * NG_BLOCK + lane] = fmaxf(acc, 0.f);
    }
#pragma unroll
    for (int o = 0; o < DOUT; o++) y[o] = b2[o];
#pragma unroll 1
    for (int t = 0; t < NG_HID; t++) {
        const float ht = s_h[t * NG_BLOCK + lane];
        const float* w = W2T + t * DOUT;
#pragma unroll
        for (int o = 0; o < DOUT; o++) y[o] += w[o] * ht;
    }
}

// delta1 = relu'(h) * (W2^T delta2) -> global (with h);  dx += W1^T delta1
template <int DOUT>
__device__ __forceinline__ void ng_backprop(const float* __restrict__ W1, const float* __restrict__ W2T, int din, const float* __restrict__ s_h, int lane,
                                            const float (&d2)[DOUT], float (&dx)[NG_IN], float* __restrict__ act_h_row, float* __restrict__ delta1_row) {
#pragma unroll 1
    for (int t = 0; t < NG_HID; t++) {
        const float* w2 = W2T + t * DOUT;
        float acc = 0.f;
#pragma unroll
        for (int o = 0; o < DOUT; o++) acc += w2[o] * d2[o];
        const float ht = s_h[t * NG_BLOCK + lane];
        const float d = ht > 0.f ? acc : 0.f;
        act_h_row[t] = ht; delta1_row[t] = d;
        const float* w1 = W1 + t * din;
#pragma unroll
        for (int i = 0; i < NG_IN - 1; i++) dx[i] += w1[i] * d;
        if (din == NG_IN) dx[NG_IN - 1] += w1[NG_IN - 1] * d;
    }
}

// one of the three k-output MLPs: WHICH 0 = opacity (tanh), 1 = colour, 2 = ray-drop (sigmoid)
template <int K, int MM, int WHICH>
__device__ __forceinline__ void ng_bw_small(const NgModel& m, const float (&x)[NG_IN], float (&dx)[NG_IN], int i, size_t c, size_t nv, float* s_h, int lane,
                                            const uint32_t* __restrict__ sel_flags, const uint32_t* __restrict__ slot,
                                            const float* __restrict__ g_opacity, const float* __restrict__ g_color, const float* __restrict__ g_no,
                                            float* __restrict__ act_h, float* __restrict__ delta1, float* __restrict__ d2row) {
    float y[K], d2[K];
    ng_recompute<K>(m.W1[MM], m.b1[MM], m.W2T[MM], m.b2[MM], m.din[MM], x, s_h, lane, y);
#pragma unroll
    for (int j = 0; j < K; j++) {
        const bool sel = sel_flags[(size_t)i * K + j] != 0u;
        const size_t r = sel ? slot[(size_t)i * K + j] : 0;
        float g = 0.f;
        if (sel) g = WHICH == 0 ? g_opacity[r] : (WHICH == 1 ? g_color[2 * r] : g_color[2 * r + 1]);
        if (WHICH == 0 && g_no) g += g_no[c * K + j];                  // gradient of the un-masked neural_opacity output, if any
        if (WHICH == 0) { const float o = tanhf(y[j]); d2[j] = g * (1.f - o * o); }
        else { const float sg = ng_sigmoid(y[j]); d2[j] = g * sg * (1.f - sg); }
    }
    ng_backprop<K>(m.W1[MM], m.W2T[MM], m.din[MM], s_h, lane, d2, dx, act_h + c * NG_HS + MM * NG_HID, delta1 + c * (4 * NG_HID) + MM * NG_HID);
    constexpr int col0 = WHICH == 0 ? 0 : (WHICH == 1 ? 8 * K : 9 * K);          // layout [k | 7k | k | k] = opacity, cov, color, raydrop
#pragma unroll
    for (int j = 0; j < K; j++) d2row[col0 + j] = d2[j];
}

template <int K>
__global__ void __launch_bounds__(NG_BLOCK) k_ng_backward(int N, int n_vis, NgModel m, float3 cam, const float* __restrict__ feat, const float* __restrict__ anchor,
                                                    const float* __restrict__ offset, const float* __restrict__ scaling,
                                                    const uint32_t* __restrict__ vis_flags, const uint32_t* __restrict__ vis_idx,
                                                    const uint32_t* __restrict__ sel_flags, const uint32_t* __restrict__ slot,
                                                    const float* __restrict__ g_xyz, const float* __restrict__ g_color, const float* __restrict__ g_opacity,
                                                    const float* __restrict__ g_scaling, const float* __restrict__ g_rot, const float* __restrict__ g_no,
                                                    float* __restrict__ d_feat, float* __restrict__ d_anchor, float* __restrict__ d_offset,
                                                    float* __restrict__ d_scaling, float* __restrict__ act_x, float* __restrict__ act_h,
                                                    float* __restrict__ delta1, float* __restrict__ delta2) {
    __shared__ float s_h[NG_HID * NG_BLOCK];
    const int lane = threadIdx.x;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float* df = d_feat + (size_t)i * NG_FEAT;
    float* dofs = d_offset + 3 * (size_t)i * K;
    if (!vis_flags[i]) {                                               // every row of the dense outputs is written
        for (int q = 0; q < NG_FEAT; q++) df[q] = 0.f;
        for (int q = 0; q < 3 * K; q++) dofs[q] = 0.f;
        for (int q = 0; q < 3; q++) d_anchor[3 * (size_t)i + q] = 0.f;
        for (int q = 0; q < 6; q++) d_scaling[6 * (size_t)i + q] = 0.f;
        return;
    }
    const size_t c = vis_idx[i];
    if (!g_no) {
        // an anchor none of whose offsets survived the opacity mask receives no gradient at all: every output-layer delta is 0
        bool any = false;
#pragma unroll
        for (int j = 0; j < K; j++) any = any || sel_flags[(size_t)i * K + j] != 0u;
        if (!any) {
            for (int q = 0; q < NG_FEAT; q++) df[q] = 0.f;
            for (int q = 0; q < 3 * K; q++) dofs[q] = 0.f;
            for (int q = 0; q < 3; q++) d_anchor[3 * (size_t)i + q] = 0.f;
            for (int q = 0; q < 6; q++) d_scaling[6 * (size_t)i + q] = 0.f;
            for (int q = 0; q < NG_XS; q++) act_x[c * NG_XS + q] = 0.f;
            for (int q = 0; q < NG_HS; q++) act_h[c * NG_HS + q] = 0.f;
            for (int q = 0; q < 4 * NG_HID; q++) delta1[c * (4 * NG_HID) + q] = 0.f;
            for (int q = 0; q < 10 * K; q++) delta2[c * (size_t)(10 * K) + q] = 0.f;
            return;
        }
    }
    float x[NG_IN], dx[NG_IN];
    ng_input(feat, anchor, cam, i, x);
#pragma unroll
    for (int q = 0; q < NG_IN; q++) { dx[q] = 0.f; act_x[c * NG_XS + q] = x[q]; }
    act_x[c * NG_XS + 36] = 1.f; act_x[c * NG_XS + 37] = 0.f; act_x[c * NG_XS + 38] = 0.f; act_x[c * NG_XS + 39] = 0.f;
    act_h[c * NG_HS + 128] = 1.f; act_h[c * NG_HS + 129] = 0.f; act_h[c * NG_HS + 130] = 0.f; act_h[c * NG_HS + 131] = 0.f;
    const float* sc = scaling + 6 * (size_t)i;
    const float s0 = sc[0], s1 = sc[1], s2 = sc[2], s3 = sc[3], s4 = sc[4], s5 = sc[5];
    float ds[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, da[3] = {0.f, 0.f, 0.f};
    float* d2row = delta2 + c * (size_t)(10 * K);
    const size_t nv = (size_t)n_vis;

    // --- covariance MLP: scaling = s[3:6] * sigmoid(sr[0:3]), rot = normalize(sr[3:7]); and the direct paths of xyz / scaling
    {
        float sr[7 * K], d2[7 * K];
        ng_recompute<7 * K>(m.W1[NG_COV], m.b1[NG_COV], m.W2T[NG_COV], m.b2[NG_COV], m.din[NG_COV], x, s_h, lane, sr);
#pragma unroll
        for (int j = 0; j < K; j++) {
            const bool sel = sel_flags[(size_t)i * K + j] != 0u;
            const size_t r = sel ? slot[(size_t)i * K + j] : 0;
            float gx = 0.f, gy = 0.f, gz = 0.f, gs0 = 0.f, gs1 = 0.f, gs2 = 0.f, gr0 = 0.f, gr1 = 0.f, gr2 = 0.f, gr3 = 0.f;
            if (sel) {
                gx = g_xyz[3 * r]; gy = g_xyz[3 * r + 1]; gz = g_xyz[3 * r + 2];
                gs0 = g_scaling[3 * r]; gs1 = g_scaling[3 * r + 1]; gs2 = g_scaling[3 * r + 2];
                gr0 = g_rot[4 * r]; gr1 = g_rot[4 * r + 1]; gr2 = g_rot[4 * r + 2]; gr3 = g_rot[4 * r + 3];
            }
            const float* of = offset + 3 * ((size_t)i * K + j);
            const float o0 = of[0], o1 = of[1], o2 = of[2];
            dofs[3 * j] = gx * s0; dofs[3 * j + 1] = gy * s1; dofs[3 * j + 2] = gz * s2;
            ds[0] += gx * o0; ds[1] += gy * o1; ds[2] += gz * o2;
            da[0] += gx; da[1] += gy; da[2] += gz;
            const float g0 = ng_sigmoid(sr[7 * j]), g1 = ng_sigmoid(sr[7 * j + 1]), g2 = ng_sigmoid(sr[7 * j + 2]);
            ds[3] += gs0 * g0; ds[4] += gs1 * g1; ds[5] += gs2 * g2;
            d2[7 * j] = gs0 * s3 * g0 * (1.f - g0); d2[7 * j + 1] = gs1 * s4 * g1 * (1.f - g1); d2[7 * j + 2] = gs2 * s5 * g2 * (1.f - g2);
            const float q0 = sr[7 * j + 3], q1 = sr[7 * j + 4], q2 = sr[7 * j + 5], q3 = sr[7 * j + 6];
            const float qn = fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f);
            const float r0 = q0 / qn, r1 = q1 / qn, r2 = q2 / qn, r3 = q3 / qn;
            const float dotp = gr0 * r0 + gr1 * r1 + gr2 * r2 + gr3 * r3;
            d2[7 * j + 3] = (gr0 - r0 * dotp) / qn; d2[7 * j + 4] = (gr1 - r1 * dotp) / qn;
            d2[7 * j + 5] = (gr2 - r2 * dotp) / qn; d2[7 * j + 6] = (gr3 - r3 * dotp) / qn;
        }
        ng_backprop<7 * K>(m.W1[NG_COV], m.W2T[NG_COV], m.din[NG_COV], s_h, lane, d2, dx, act_h + c * NG_HS + NG_COV * NG_HID,
                           delta1 + c * (4 * NG_HID) + NG_COV * NG_HID);
#pragma unroll
        for (int q = 0; q < 7 * K; q++) d2row[K + q] = d2[q];
    }
    // --- opacity (tanh), colour and ray-drop (sigmoid) MLPs
    ng_bw_small<K, NG_OPA, 0>(m, x, dx, i, c, nv, s_h, lane, sel_flags, slot, g_opacity, g_color, g_no, act_h, delta1, d2row);
    ng_bw_small<K, NG_COL, 1>(m, x, dx, i, c, nv, s_h, lane, sel_flags, slot, g_opacity, g_color, g_no, act_h, delta1, d2row);
    ng_bw_small<K, NG_RD, 2>(m, x, dx, i, c, nv, s_h, lane, sel_flags, slot, g_opacity, g_color, g_no, act_h, delta1, d2row);
    // --- input: feature directly; view = ob/|ob|, dist = |ob| back to the anchor position
#pragma unroll
    for (int q = 0; q < NG_FEAT; q++) df[q] = dx[q];
    const float dist = x[35], vx = x[32], vy = x[33], vz = x[34];
    const float dv = dx[32] * vx + dx[33] * vy + dx[34] * vz;
    da[0] += dx[32] / dist - vx * (dv / dist) + dx[35] * vx;
    da[1] += dx[33] / dist - vy * (dv / dist) + dx[35] * vy;
    da[2] += dx[34] / dist - vz * (dv / dist) + dx[35] * vz;
#pragma unroll
    for (int q = 0; q < 3; q++) d_anchor[3 * (size_t)i + q] = da[q];
#pragma unroll
    for (int q = 0; q < 6; q++) d_scaling[6 * (size_t)i + q] = ds[q];
}

// ---- backward on the matrix pipe ---------------------------------------------------------------------------------------------------
// k_ng_backward above evaluates the MLPs one anchor per lane, one FMA per scalar-loaded weight, and hands the weight gradients to two
// library GEMMs through 1.4 KB of per-anchor rows (3 GB of write traffic per launch; VALU 22 % busy, 43 % of the wave cycles in
// s_waitcnt).  Here a wave owns 64 anchors as two 32-row tiles and everything matrix-shaped runs on v_mfma_f32_32x32x2_f32 (exact f32):
//   H = relu(X W1^T + b1), Y = H W2^T + b2            recompute     (A from LDS [unit][anchor], B = weights, L1-resident)
//   per-anchor activation derivatives                 lane = anchor (the only VALU stage; y and delta2 pass through LDS)
//   D1 = (D2 W2) * relu'(H), dX += D1 W1              back-propagation
//   G1_m += D1^T [X 1], G2_m += D2^T H                weight gradients: the reduction index is the anchor, two per instruction;
//                                                     13-15 accumulator tiles stay in registers for the life of the wave
// Waves are persistent (one per SIMD), walk the anchor tiles with a grid stride and write their accumulators once, as partial sums
// the caller adds up.  Nothing per-anchor is written except the dense input gradients.
typedef float ng_f16v __attribute__((ext_vector_type(16)));
#define NG_LS 65                                   // LDS stride of the [unit][anchor] arrays: transposed reads hit distinct banks
#define NG_COV_TILES(K) ((7 * (K) + 31) / 32)
#define NG_G2_TILES(K) (3 + NG_COV_TILES(K))       // opacity, covariance (1-3), colour, ray-drop
#define NG_PARTIAL_FLOATS(K) ((5 + NG_G2_TILES(K)) * 1024 + 128)

__device__ __forceinline__ int ng_crow(int e, int lane) { return (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5); }   // row of accumulator element e
// wave-wide sum on the VALU alone: four DPP steps leave every 16-lane row holding its total, four lane reads add the rows (the
// six-step bpermute butterfly went through the LDS crossbar, ~400 dependent cycles a sum, 60 sums per tile: 8 % of the kernel)
template <int CTRL>
__device__ __forceinline__ float ng_dpp(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true)); }
__device__ __forceinline__ float ng_wave_sum(float v) {
    v += ng_dpp<0xB1>(v);            // quad_perm [1,0,3,2]
    v += ng_dpp<0x4E>(v);            // quad_perm [2,3,0,1]
    v += ng_dpp<0x141>(v);           // row_half_mirror
    v += ng_dpp<0x140>(v);           // row_mirror
    const int b = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16)) +
           __int_as_float(__builtin_amdgcn_readlane(b, 32)) + __int_as_float(__builtin_amdgcn_readlane(b, 48));
}
// c0 / c1 [32 anchors of tile 0 / 1][32 columns from col0] += sum_{k < kdim} A[anchor][k] B[k][col];  A = s_a[k][anchor] (LDS),
// B[k][col] = Bm[k * sk + col * sc] (global, L1-resident; 0 for col >= ncols).  Both row tiles share the B operand, and the operands
// of NG_U steps are fetched before the first product: with one wave per SIMD nothing else hides their latency.
#define NG_U 4
// Inputs of a 64-anchor tile into s_x[unit][anchor].  The 32 features of 64 consecutive anchors are 8 KB of contiguous memory:
// eight 1-KB wave loads instead of every lane walking its own 128-byte row (64 lines per load instruction: 0.11 ms of the
// backward at 666 k anchors); view direction and distance are per lane.  Anchors without work read as zeros.  Returns the lane's
// (view, dist).
__device__ __forceinline__ float4 ng_stage_inputs(const float* __restrict__ feat, const float* __restrict__ anchor, float3 cam, int tile, int N,
                                                  bool active, const uint32_t* __restrict__ s_act, float* __restrict__ s_x, int lane) {
    const float4* f4 = reinterpret_cast<const float4*>(feat) + (size_t)tile * (NG_BLOCK * NG_FEAT / 4);
#pragma unroll
    for (int it = 0; it < NG_FEAT / 4; it++) {
        const int idx = it * NG_BLOCK + lane;
        const int a = idx >> 3, qb = (idx & 7) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tile * NG_BLOCK + a < N && s_act[a]) v = f4[idx];
        s_x[(qb + 0) * NG_LS + a] = v.x; s_x[(qb + 1) * NG_LS + a] = v.y; s_x[(qb + 2) * NG_LS + a] = v.z; s_x[(qb + 3) * NG_LS + a] = v.w;
    }
    float4 vd = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        const size_t i = (size_t)tile * NG_BLOCK + lane;
        const float ox = anchor[3 * i] - cam.x, oy = anchor[3 * i + 1] - cam.y, oz = anchor[3 * i + 2] - cam.z;
        const float dist = sqrtf(ox * ox + oy * oy + oz * oz);
        vd = make_float4(ox / dist, oy / dist, oz / dist, dist);
    }
    s_x[32 * NG_LS + lane] = vd.x; s_x[33 * NG_LS + lane] = vd.y; s_x[34 * NG_LS + lane] = vd.z; s_x[35 * NG_LS + lane] = vd.w;
    return vd;
}
// The LDS arrays are padded to a multiple of 2 NG_U rows (zero-initialised, always finite), so the A reads need no guard; only B is
// cut off at kdim (TAIL).  Pointers advance by one batch per trip and every operand sits at a fixed offset from them: no address
// arithmetic between the products (it was half of the kernel's VALU instructions).
template <bool TAIL>
__device__ __forceinline__ void ng_gemm_pair(const float* __restrict__ s_a, const float* __restrict__ Bm, int sk, int sc, int col0, int ncols,
                                             int kdim, ng_f16v& c0, ng_f16v& c1, int lane) {
    const int u = lane & 31, half = lane >> 5;
    const int col = col0 + u;
    const bool colok = col < ncols;
    const float* bp = Bm + (colok ? col * sc : 0) + half * sk;
    const float* ap = s_a + u + half * NG_LS;
    const int sk2 = 2 * sk;
#pragma unroll 1
    for (int k0 = 0; k0 < kdim; k0 += 2 * NG_U) {
        float a0[NG_U], a1[NG_U], b[NG_U];
#pragma unroll
        for (int j = 0; j < NG_U; j++) {
            a0[j] = ap[2 * j * NG_LS]; a1[j] = ap[2 * j * NG_LS + 32];
            const bool ok = colok && (!TAIL || k0 + 2 * j + half < kdim);
            b[j] = ok ? bp[j * sk2] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NG_U; j++) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b[j], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b[j], c1, 0, 0, 0);
        }
        ap += 2 * NG_U * NG_LS; bp += NG_U * sk2;
    }
}
__device__ __forceinline__ ng_f16v ng_splat(float v) {
    ng_f16v c;
#pragma unroll
    for (int e = 0; e < 16; e++) c[e] = v;
    return c;
}
// G1 / G2 of one MLP over the 64 anchors of the tile: g1a = G1 columns 0..31; g1n = the narrow columns 32..39 of ALL four MLPs in
// one tile (MLP MM owns its columns 8 MM .. 8 MM + 7); g2[] = G2 row tiles
template <int DOUT>
__device__ __forceinline__ void ng_accumulate(const float* __restrict__ s_x, const float* __restrict__ s_h, const float* __restrict__ s_d1,
                                              const float* __restrict__ s_d2, int MM, int lane, ng_f16v& g1a, ng_f16v& g1n, ng_f16v* __restrict__ g2) {
    constexpr int MT = (DOUT + 31) / 32;
    const int u = lane & 31, half = lane >> 5;
    const bool mine = (u >> 3) == MM;
    const float* pd1 = s_d1 + u * NG_LS + half;
    const float* px = s_x + u * NG_LS + half;
    const float* pxn = s_x + (32 + (u & 7)) * NG_LS + half;
    const float* ph = s_h + u * NG_LS + half;
    const float* pd2 = s_d2 + u * NG_LS + half;
#pragma unroll 1
    for (int a0 = 0; a0 < NG_BLOCK; a0 += 2 * NG_U) {
        float A1[NG_U], B0[NG_U], B1[NG_U], Bh[NG_U], A2[MT][NG_U];
#pragma unroll
        for (int j = 0; j < NG_U; j++) {
            A1[j] = pd1[2 * j];
            B0[j] = px[2 * j];
            B1[j] = mine ? pxn[2 * j] : 0.f;
            Bh[j] = ph[2 * j];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) A2[mt][j] = mt * 32 + u < DOUT ? pd2[mt * 32 * NG_LS + 2 * j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NG_U; j++) {
            g1a = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[j], B0[j], g1a, 0, 0, 0);
            g1n = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[j], B1[j], g1n, 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) g2[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(A2[mt][j], Bh[j], g2[mt], 0, 0, 0);
        }
        pd1 += 2 * NG_U; px += 2 * NG_U; pxn += 2 * NG_U; ph += 2 * NG_U; pd2 += 2 * NG_U;
    }
}
// recompute of one MLP for the tile: s_h = relu(X W1^T + b1), s_y = H W2^T + b2  (s_y shares its rows with delta2)
template <int DOUT>
__device__ __forceinline__ void ng_mfma_recompute(const NgModel& m, int MM, const float* s_x, float* s_h, float* s_y, int lane) {
    const int u = lane & 31;
    const int din = m.din[MM];
    const float b1 = m.b1[MM][u];
    {
        ng_f16v c0 = ng_splat(b1), c1 = c0;
        ng_gemm_pair<true>(s_x, m.W1[MM], 1, din, 0, NG_HID, din, c0, c1, lane);                              // B[k = q][t] = W1[t][q]
#pragma unroll
        for (int e = 0; e < 16; e++) { s_h[u * NG_LS + ng_crow(e, lane)] = fmaxf(c0[e], 0.f); s_h[u * NG_LS + 32 + ng_crow(e, lane)] = fmaxf(c1[e], 0.f); }
    }
    __builtin_amdgcn_wave_barrier();
    constexpr int NT = (DOUT + 31) / 32;
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int o = nt * 32 + u;
        const float b2 = o < DOUT ? m.b2[MM][o] : 0.f;
        ng_f16v c0 = ng_splat(b2), c1 = c0;
        ng_gemm_pair<false>(s_h, m.W2T[MM], DOUT, 1, nt * 32, DOUT, NG_HID, c0, c1, lane);                     // B[k = t][o] = W2T[t][o]
        if (o < DOUT) {
#pragma unroll
            for (int e = 0; e < 16; e++) { s_y[o * NG_LS + ng_crow(e, lane)] = c0[e]; s_y[o * NG_LS + 32 + ng_crow(e, lane)] = c1[e]; }
        }
    }
    __builtin_amdgcn_wave_barrier();
}
// back-propagation of one MLP for the tile (delta2 is in s_d2): s_d1 = (D2 W2) * relu'(h);  dx tiles += D1 W1;  then the weight gradients
template <int DOUT>
__device__ __forceinline__ void ng_mfma_backprop(const NgModel& m, int MM, const float* s_x, const float* s_h, float* s_d1, const float* s_d2, int lane,
                                                 ng_f16v (&dxa)[2][2], ng_f16v& g1a, ng_f16v& g1n, ng_f16v* g2) {
    const int u = lane & 31;
    const int din = m.din[MM];
    __builtin_amdgcn_wave_barrier();
    {
        ng_f16v c0 = ng_splat(0.f), c1 = c0;
        ng_gemm_pair<(DOUT % (2 * NG_U)) != 0>(s_d2, m.W2[MM], NG_HID, 1, 0, NG_HID, DOUT, c0, c1, lane);                         // B[k = o][t] = W2[o][t]
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int i0 = u * NG_LS + ng_crow(e, lane), i1 = i0 + 32;
            s_d1[i0] = s_h[i0] > 0.f ? c0[e] : 0.f;
            s_d1[i1] = s_h[i1] > 0.f ? c1[e] : 0.f;
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int nt = 0; nt < 2; nt++) ng_gemm_pair<false>(s_d1, m.W1[MM], din, 1, nt * 32, din, NG_HID, dxa[0][nt], dxa[1][nt], lane);   // B[k = t][q] = W1[t][q]
    ng_accumulate<DOUT>(s_x, s_h, s_d1, s_d2, MM, lane, g1a, g1n, g2);
    __builtin_amdgcn_wave_barrier();
}

// the per-anchor stage of one of the three k-output MLPs (lane = anchor): y -> delta2, both through s_d2.  WHICH 0 = opacity (tanh),
// 1 = colour, 2 = ray-drop (sigmoid)
template <int K, int WHICH>
__device__ __forceinline__ void ng_small_deltas(bool active, int i, size_t c, int lane, const uint32_t* __restrict__ sel_flags, const uint32_t* __restrict__ slot,
                                                const float* __restrict__ g_opacity, const float* __restrict__ g_color, const float* __restrict__ g_no,
                                                float* s_d2, float* s_db2) {
    constexpr int col0 = WHICH == 0 ? 0 : (WHICH == 1 ? 8 * K : 9 * K);          // layout [k | 7k | k | k] = opacity, cov, color, raydrop
#pragma unroll
    for (int j = 0; j < K; j++) {
        float d = 0.f;
        if (active) {
            const float y = s_d2[j * NG_LS + lane];
            const bool sel = sel_flags[(size_t)i * K + j] != 0u;
            const size_t r = sel ? slot[(size_t)i * K + j] : 0;
            float g = 0.f;
            if (sel) g = WHICH == 0 ? g_opacity[r] : (WHICH == 1 ? g_color[2 * r] : g_color[2 * r + 1]);
            if (WHICH == 0 && g_no) g += g_no[c * K + j];
            if (WHICH == 0) { const float o = tanhf(y); d = g * (1.f - o * o); }
            else { const float sg = ng_sigmoid(y); d = g * sg * (1.f - sg); }
        }
        s_d2[j * NG_LS + lane] = d;
        const float sum = ng_wave_sum(d);
        if (lane == 0) s_db2[col0 + j] += sum;
    }
}

template <int K>
__global__ void __launch_bounds__(NG_BLOCK) k_ng_backward_mfma(int N, NgModel m, float3 cam, const float* __restrict__ feat, const float* __restrict__ anchor,
                                                               const float* __restrict__ offset, const float* __restrict__ scaling,
                                                               const uint32_t* __restrict__ vis_flags, const uint32_t* __restrict__ vis_idx,
                                                               const uint32_t* __restrict__ sel_flags, const uint32_t* __restrict__ slot,
                                                               const float* __restrict__ g_xyz, const float* __restrict__ g_color, const float* __restrict__ g_opacity,
                                                               const float* __restrict__ g_scaling, const float* __restrict__ g_rot, const float* __restrict__ g_no,
                                                               float* __restrict__ d_feat, float* __restrict__ d_anchor, float* __restrict__ d_offset,
                                                               float* __restrict__ d_scaling, float* __restrict__ partial) {
    constexpr int NC = NG_COV_TILES(K), T2 = NG_G2_TILES(K);
    constexpr int D2ROWS = (7 * K + 2 * NG_U - 1) / (2 * NG_U) * (2 * NG_U);   // rows of y / delta2, padded to whole operand batches
    __shared__ float s_x[NG_XS * NG_LS], s_h[NG_HID * NG_LS], s_d1[NG_HID * NG_LS], s_d2[D2ROWS * NG_LS], s_db2[128];
    __shared__ uint32_t s_act[NG_BLOCK];
    const int lane = threadIdx.x, u = lane & 31;
    for (int q = lane; q < NG_XS * NG_LS; q += NG_BLOCK) s_x[q] = 0.f;         // nothing a product can meet is ever uninitialised
    for (int q = lane; q < NG_HID * NG_LS; q += NG_BLOCK) { s_h[q] = 0.f; s_d1[q] = 0.f; }
    for (int q = lane; q < D2ROWS * NG_LS; q += NG_BLOCK) s_d2[q] = 0.f;
    s_db2[lane] = 0.f; s_db2[64 + lane] = 0.f;
    ng_f16v g1[5], g2[T2];                                            // g1[0..3]: columns 0..31 of the four MLPs; g1[4]: their columns 32..39
#pragma unroll
    for (int t = 0; t < 5; t++) g1[t] = ng_splat(0.f);
#pragma unroll
    for (int t = 0; t < T2; t++) g2[t] = ng_splat(0.f);
    const int ntiles = (N + NG_BLOCK - 1) / NG_BLOCK;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int i = tile * NG_BLOCK + lane;
        bool active = false;
        size_t c = 0;
        if (i < N) {
            bool zero_dense = true;
            if (vis_flags[i]) {
                c = vis_idx[i];
                bool any = g_no != nullptr;                            // with a gradient on neural_opacity every visible anchor has work
#pragma unroll
                for (int j = 0; j < K; j++) any = any || sel_flags[(size_t)i * K + j] != 0u;
                active = any; zero_dense = !any;
            }
            if (zero_dense) {                                          // every row of the dense outputs is written
                float* df = d_feat + (size_t)i * NG_FEAT;
                float* dofs = d_offset + 3 * (size_t)i * K;
                for (int q = 0; q < NG_FEAT; q++) df[q] = 0.f;
                for (int q = 0; q < 3 * K; q++) dofs[q] = 0.f;
                for (int q = 0; q < 3; q++) d_anchor[3 * (size_t)i + q] = 0.f;
                for (int q = 0; q < 6; q++) d_scaling[6 * (size_t)i + q] = 0.f;
            }
        }
        if (__ballot(active) == 0ull) continue;                        // nothing to fold in this tile
        __builtin_amdgcn_wave_barrier();
        s_act[lane] = active ? 1u : 0u;
        float vx = 0.f, vy = 0.f, vz = 0.f, dist = 1.f;
        {
            __builtin_amdgcn_wave_barrier();
            const float4 vd = ng_stage_inputs(feat, anchor, cam, tile, N, active, s_act, s_x, lane);
            s_x[36 * NG_LS + lane] = active ? 1.f : 0.f;               // the constant input: bias gradients fall out of the same product
            if (active) { vx = vd.x; vy = vd.y; vz = vd.z; dist = vd.w; }
        }
        float ds[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, da[3] = {0.f, 0.f, 0.f};
        ng_f16v dxa[2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) dxa[a][b] = ng_splat(0.f);
        __builtin_amdgcn_wave_barrier();

        // --- covariance MLP: scaling = s[3:6] * sigmoid(sr[0:3]), rot = normalize(sr[3:7]); and the direct paths of xyz / scaling
        ng_mfma_recompute<7 * K>(m, NG_COV, s_x, s_h, s_d2, lane);
        {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f;
            if (active) { const float* sc = scaling + 6 * (size_t)i; s0 = sc[0]; s1 = sc[1]; s2 = sc[2]; s3 = sc[3]; s4 = sc[4]; s5 = sc[5]; }
            float* dofs = d_offset + 3 * (size_t)(active ? i : 0) * K;
#pragma unroll
            for (int j = 0; j < K; j++) {
                float d[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (active) {
                    float sr[7];
#pragma unroll
                    for (int q = 0; q < 7; q++) sr[q] = s_d2[(7 * j + q) * NG_LS + lane];
                    const bool sel = sel_flags[(size_t)i * K + j] != 0u;
                    const size_t r = sel ? slot[(size_t)i * K + j] : 0;
                    float gx = 0.f, gy = 0.f, gz = 0.f, gs0 = 0.f, gs1 = 0.f, gs2 = 0.f, gr0 = 0.f, gr1 = 0.f, gr2 = 0.f, gr3 = 0.f;
                    if (sel) {
                        gx = g_xyz[3 * r]; gy = g_xyz[3 * r + 1]; gz = g_xyz[3 * r + 2];
                        gs0 = g_scaling[3 * r]; gs1 = g_scaling[3 * r + 1]; gs2 = g_scaling[3 * r + 2];
                        gr0 = g_rot[4 * r]; gr1 = g_rot[4 * r + 1]; gr2 = g_rot[4 * r + 2]; gr3 = g_rot[4 * r + 3];
                    }
                    const float* of = offset + 3 * ((size_t)i * K + j);
                    const float o0 = of[0], o1 = of[1], o2 = of[2];
                    dofs[3 * j] = gx * s0; dofs[3 * j + 1] = gy * s1; dofs[3 * j + 2] = gz * s2;
                    ds[0] += gx * o0; ds[1] += gy * o1; ds[2] += gz * o2;
                    da[0] += gx; da[1] += gy; da[2] += gz;
                    const float g0 = ng_sigmoid(sr[0]), g1s = ng_sigmoid(sr[1]), g2s = ng_sigmoid(sr[2]);
                    ds[3] += gs0 * g0; ds[4] += gs1 * g1s; ds[5] += gs2 * g2s;
                    d[0] = gs0 * s3 * g0 * (1.f - g0); d[1] = gs1 * s4 * g1s * (1.f - g1s); d[2] = gs2 * s5 * g2s * (1.f - g2s);
                    const float q0 = sr[3], q1 = sr[4], q2 = sr[5], q3 = sr[6];
                    const float qn = fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f);
                    const float r0 = q0 / qn, r1 = q1 / qn, r2 = q2 / qn, r3 = q3 / qn;
                    const float dotp = gr0 * r0 + gr1 * r1 + gr2 * r2 + gr3 * r3;
                    d[3] = (gr0 - r0 * dotp) / qn; d[4] = (gr1 - r1 * dotp) / qn; d[5] = (gr2 - r2 * dotp) / qn; d[6] = (gr3 - r3 * dotp) / qn;
                }
#pragma unroll
                for (int q = 0; q < 7; q++) {
                    s_d2[(7 * j + q) * NG_LS + lane] = d[q];
                    const float sum = ng_wave_sum(d[q]);
                    if (lane == 0) s_db2[K + 7 * j + q] += sum;
                }
            }
        }
        ng_mfma_backprop<7 * K>(m, NG_COV, s_x, s_h, s_d1, s_d2, lane, dxa, g1[NG_COV], g1[4], &g2[1]);
        // --- opacity (tanh), colour and ray-drop (sigmoid) MLPs
        ng_mfma_recompute<K>(m, NG_OPA, s_x, s_h, s_d2, lane);
        ng_small_deltas<K, 0>(active, i, c, lane, sel_flags, slot, g_opacity, g_color, g_no, s_d2, s_db2);
        ng_mfma_backprop<K>(m, NG_OPA, s_x, s_h, s_d1, s_d2, lane, dxa, g1[NG_OPA], g1[4], &g2[0]);
        ng_mfma_recompute<K>(m, NG_COL, s_x, s_h, s_d2, lane);
        ng_small_deltas<K, 1>(active, i, c, lane, sel_flags, slot, g_opacity, g_color, g_no, s_d2, s_db2);
        ng_mfma_backprop<K>(m, NG_COL, s_x, s_h, s_d1, s_d2, lane, dxa, g1[NG_COL], g1[4], &g2[1 + NC]);
        ng_mfma_recompute<K>(m, NG_RD, s_x, s_h, s_d2, lane);
        ng_small_deltas<K, 2>(active, i, c, lane, sel_flags, slot, g_opacity, g_color, g_no, s_d2, s_db2);
        ng_mfma_backprop<K>(m, NG_RD, s_x, s_h, s_d1, s_d2, lane, dxa, g1[NG_RD], g1[4], &g2[2 + NC]);

        // --- input gradients: the feature part leaves in accumulator layout (a lane holds one column of 16 anchors: 128-byte
        //     row segments); the view / distance part goes through LDS back to its anchor's lane
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int a = mt * 32 + ng_crow(e, lane);
                if (s_act[a]) d_feat[(size_t)(tile * NG_BLOCK + a) * NG_FEAT + u] = dxa[mt][0][e];
                if (u < 4) s_d1[u * NG_LS + a] = dxa[mt][1][e];
            }
        __builtin_amdgcn_wave_barrier();
        if (active) {
            const float dx32 = s_d1[0 * NG_LS + lane], dx33 = s_d1[1 * NG_LS + lane], dx34 = s_d1[2 * NG_LS + lane], dx35 = s_d1[3 * NG_LS + lane];
            const float dv = dx32 * vx + dx33 * vy + dx34 * vz;
            da[0] += dx32 / dist - vx * (dv / dist) + dx35 * vx;
            da[1] += dx33 / dist - vy * (dv / dist) + dx35 * vy;
            da[2] += dx34 / dist - vz * (dv / dist) + dx35 * vz;
#pragma unroll
            for (int q = 0; q < 3; q++) d_anchor[3 * (size_t)i + q] = da[q];
#pragma unroll
            for (int q = 0; q < 6; q++) d_scaling[6 * (size_t)i + q] = ds[q];
        }
        __builtin_amdgcn_wave_barrier();
    }
    // --- this wave's partial sums: tiles [t or o][q or t] as 32 x 32 blocks, then the output-layer bias sums
    float* out = partial + (size_t)blockIdx.x * NG_PARTIAL_FLOATS(K);
#pragma unroll
    for (int t = 0; t < 5; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) out[t * 1024 + ng_crow(e, lane) * 32 + u] = g1[t][e];
#pragma unroll
    for (int t = 0; t < T2; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) out[(5 + t) * 1024 + ng_crow(e, lane) * 32 + u] = g2[t][e];
    __builtin_amdgcn_wave_barrier();
    out[(5 + T2) * 1024 + lane] = s_db2[lane]; out[(5 + T2) * 1024 + 64 + lane] = s_db2[64 + lane];
}

#include "neural_gaussians_t16.inc"

// ---- decode on the matrix pipe -----------------------------------------------------------------------------------------------------
// k_ng_decode with the three MLPs as tile products (ng_mfma_recompute, the backward's own recompute: the same f32 fma chains in
// the same order, so the backward re-derives exactly the activations the forward used).  A wave = 64 anchors; its outputs are
// read back per anchor from LDS for the post-processing, which is unchanged.
template <int K>
__global__ void __launch_bounds__(NG_BLOCK) k_ng_decode_mfma(int N, NgModel m, float3 cam, const float* __restrict__ feat, const float* __restrict__ anchor,
                                                             const float* __restrict__ offset, const float* __restrict__ scaling,
                                                             const uint32_t* __restrict__ vis_flags, const uint32_t* __restrict__ vis_idx,
                                                             const uint32_t* __restrict__ sel_flags, const uint32_t* __restrict__ slot,
                                                             const float* __restrict__ neural_opacity, float* __restrict__ o_xyz, float* __restrict__ o_color,
                                                             float* __restrict__ o_opacity, float* __restrict__ o_scaling, float* __restrict__ o_rot) {
    // LDS sets the occupancy here (one wave per workgroup): x (40 rows), h (32) and the covariance MLP's 7k outputs as three arrays were
    // 31 KB for k = 6, five waves to a CU.  x is dead once the LAST recompute has its hidden units, so that one writes its outputs over
    // x; the two k-output recomputes before it put theirs in the rows behind x: 20.8 KB, seven waves.
    constexpr int UROWS = (7 * K > NG_XS + K) ? 7 * K : NG_XS + K;
    __shared__ float s_u[UROWS * NG_LS], s_h[NG_HID * NG_LS];
    float* const s_x = s_u;
    float* const s_ys = s_u + NG_XS * NG_LS;                           // rows of the colour / ray-drop outputs (k each, one after the other)
    float* const s_y = s_u;                                            // rows of the covariance outputs (7k), over x
    const int lane = threadIdx.x;
    const int i = blockIdx.x * NG_BLOCK + lane;
    bool active = false;
    if (i < N && vis_flags[i]) {
#pragma unroll
        for (int j = 0; j < K; j++) active = active || sel_flags[(size_t)i * K + j] != 0u;
    }
    if (__ballot(active) == 0ull) return;
    {   // per-lane rows here: with five waves per CU to hide them behind, the cooperative tile load of the backward measured slower
        float x[NG_IN];
#pragma unroll
        for (int q = 0; q < NG_IN; q++) x[q] = 0.f;
        if (active) ng_input(feat, anchor, cam, i, x);
#pragma unroll
        for (int q = 0; q < NG_IN; q++) s_x[q * NG_LS + lane] = x[q];
#pragma unroll
        for (int q = NG_IN; q < NG_XS; q++) s_x[q * NG_LS + lane] = 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    float col[K], rd[K];
    ng_mfma_recompute<K>(m, NG_COL, s_x, s_h, s_ys, lane);
#pragma unroll
    for (int j = 0; j < K; j++) col[j] = s_ys[j * NG_LS + lane];
    __builtin_amdgcn_wave_barrier();
    ng_mfma_recompute<K>(m, NG_RD, s_x, s_h, s_ys, lane);
#pragma unroll
    for (int j = 0; j < K; j++) rd[j] = s_ys[j * NG_LS + lane];
    __builtin_amdgcn_wave_barrier();
    ng_mfma_recompute<7 * K>(m, NG_COV, s_x, s_h, s_y, lane);          // its outputs land on x (dead behind the hidden layer)
    if (!active) return;
    const float* sc = scaling + 6 * (size_t)i;
    const float s0 = sc[0], s1 = sc[1], s2 = sc[2], s3 = sc[3], s4 = sc[4], s5 = sc[5];
    const float ax = anchor[3 * (size_t)i], ay = anchor[3 * (size_t)i + 1], az = anchor[3 * (size_t)i + 2];
    const size_t c = vis_idx[i];
#pragma unroll
    for (int j = 0; j < K; j++) {
        if (!sel_flags[(size_t)i * K + j]) continue;
        const size_t r = slot[(size_t)i * K + j];
        const float* of = offset + 3 * ((size_t)i * K + j);
        o_xyz[3 * r] = ax + of[0] * s0; o_xyz[3 * r + 1] = ay + of[1] * s1; o_xyz[3 * r + 2] = az + of[2] * s2;      // :111-112
        o_color[2 * r] = ng_sigmoid(col[j]); o_color[2 * r + 1] = ng_sigmoid(rd[j]);                                // :85-87
        o_opacity[r] = neural_opacity[c * K + j];                                                                    // :71
        float sr[7];
#pragma unroll
        for (int q = 0; q < 7; q++) sr[q] = s_y[(7 * j + q) * NG_LS + lane];
        o_scaling[3 * r] = s3 * ng_sigmoid(sr[0]); o_scaling[3 * r + 1] = s4 * ng_sigmoid(sr[1]);                    // :107
        o_scaling[3 * r + 2] = s5 * ng_sigmoid(sr[2]);
        const float q0 = sr[3], q1 = sr[4], q2 = sr[5], q3 = sr[6];
        const float qn = fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f);                               // F.normalize, :108
        o_rot[4 * r] = q0 / qn; o_rot[4 * r + 1] = q1 / qn; o_rot[4 * r + 2] = q2 / qn; o_rot[4 * r + 3] = q3 / qn;
    }
}

// k_ng_opacity with its MLP as tile products (the same fma chains in the same order: the mask it decides is the per-lane kernel's)
template <int K>
__global__ void __launch_bounds__(NG_BLOCK) k_ng_opacity_mfma(int N, NgModel m, float3 cam, const float* __restrict__ feat, const float* __restrict__ anchor,
                                                              const uint32_t* __restrict__ vis_flags, const uint32_t* __restrict__ vis_idx,
                                                              float* __restrict__ neural_opacity, uint8_t* __restrict__ mask, uint32_t* __restrict__ sel_flags) {
    // one MLP, one wave: x is dead once the hidden layer's products are in the accumulators, so h is written over it (rows 0..31) and
    // the k outputs behind h -- 10.4 KB instead of x, h and y side by side (20.8 KB): the registers, not LDS, set the occupancy now
    constexpr int UROWS = NG_HID + K > NG_XS ? NG_HID + K : NG_XS;
    __shared__ float s_x[UROWS * NG_LS];
    float* const s_h = s_x;
    float* const s_y = s_x + NG_HID * NG_LS;
    const int lane = threadIdx.x;
    const int i = blockIdx.x * NG_BLOCK + lane;
    const bool vis = i < N && vis_flags[i] != 0u;
    if (i < N && !vis) {
#pragma unroll
        for (int j = 0; j < K; j++) sel_flags[(size_t)i * K + j] = 0u;
    }
    if (__ballot(vis) == 0ull) return;
    {
        float x[NG_IN];
#pragma unroll
        for (int q = 0; q < NG_IN; q++) x[q] = 0.f;
        if (vis) ng_input(feat, anchor, cam, i, x);
#pragma unroll
        for (int q = 0; q < NG_IN; q++) s_x[q * NG_LS + lane] = x[q];
#pragma unroll
        for (int q = NG_IN; q < NG_XS; q++) s_x[q * NG_LS + lane] = 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    ng_mfma_recompute<K>(m, NG_OPA, s_x, s_h, s_y, lane);
    if (!vis) return;
    const size_t c = vis_idx[i];
#pragma unroll
    for (int j = 0; j < K; j++) {
        const float o = tanhf(s_y[j * NG_LS + lane]);                 // nn.Tanh closes the opacity MLP (gaussian_model.py:118)
        neural_opacity[c * K + j] = o;
        const bool keep = o > 0.0f;                                    // :67
        mask[c * K + j] = keep ? 1 : 0;
        sel_flags[(size_t)i * K + j] = keep ? 1u : 0u;
    }
}

// ---- densification statistics (scene/gaussian_model.py:599-622) ------------------------------------------------------------
// flags of the selected pairs in GLOBAL (anchor, offset) order, from the compact mask the decode returned
__global__ void __launch_bounds__(256) k_ng_stats_flags(int N, int K, const uint32_t* __restrict__ vis_flags, const uint32_t* __restrict__ vis_idx,
                                                        const uint8_t* __restrict__ sel_mask, uint32_t* __restrict__ sel_flags) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)N * K) return;
    const size_t i = t / K, j = t - i * K;
    sel_flags[t] = (vis_flags[i] && sel_mask[(size_t)vis_idx[i] * K + j]) ? 1u : 0u;
}

__global__ void __launch_bounds__(256) k_ng_stats(int N, int K, const uint32_t* __restrict__ vis_flags, const uint32_t* __restrict__ vis_idx,
                                                  const uint32_t* __restrict__ sel_flags, const uint32_t* __restrict__ slot,
                                                  const uint8_t* __restrict__ update_filter, const float* __restrict__ opacity,
                                                  const float* __restrict__ grad, float* __restrict__ opacity_accum, float* __restrict__ anchor_demon,
                                                  float* __restrict__ grad_accum, float* __restrict__ denom) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || !vis_flags[i]) return;
    const size_t c = vis_idx[i];
    float osum = 0.f;
    for (int j = 0; j < K; j++) {
        osum += fmaxf(opacity[c * K + j], 0.f);                         // :601-604
        const size_t p = (size_t)i * K + j;
        if (sel_flags[p]) {
            const size_t r = slot[p];
            if (update_filter[r]) {                                     // :614-620
                const float gx = grad[4 * r + 2], gy = grad[4 * r + 3];
                grad_accum[p] += sqrtf(gx * gx + gy * gy);
                denom[p] += 1.f;
            }
        }
    }
    opacity_accum[i] += osum;                                           // :605
    anchor_demon[i] += 1.f;                                             // :608
}


// The weight gradients out of the backward's partial sums, in one launch: sum over the persistent waves' rows and unpack the tiles
// into the sixteen parameter gradients (the layout of include/lidargs_neural_gaussians.h).  Before: a framework reduction over
// [1024][10368] in 48 us, four concatenations and four strided copies (~90 us of a 0.55-ms backward).  One thread per OUTPUT element
// (neighbours in the output are neighbours in a tile row, so a wave's 64 loads of a row are a few lines); the four waves of a
// workgroup and the 32 row groups of the grid take 8 rows each, eight loads in flight, and meet in a fixed
// order (LDS, then a second tiny launch over the row groups): deterministic sums.  One workgroup per 64 outputs walking all
// 1024 rows took 42 us whether with 4 waves x 8 loads or 16 x 16 in flight (105 workgroups on 256 CUs); so did the framework's.
struct NgGradMap { int k, nc, per_wave, din[4], base[4]; int total; };   // base[m]: first output float of MLP m's block
__device__ __forceinline__ int ng_grad_source(const NgGradMap& g, int j) {
    int m = 3;
    if (j < g.base[1]) m = 0; else if (j < g.base[2]) m = 1; else if (j < g.base[3]) m = 2;
    const int dout = m == 1 ? 7 * g.k : g.k, din = g.din[m];
    int r = j - g.base[m];
    if (r < 32 * din) {                                                // dW1_m[t][q]
        const int t = r / din, q = r - t * din;
        return q < 32 ? m * 1024 + t * 32 + q : 4 * 1024 + t * 32 + 8 * m + (q - 32);
    }
    r -= 32 * din;
    if (r < 32) return 4 * 1024 + r * 32 + 8 * m + 4;                  // db1_m[t]: input 36 is the constant 1
    r -= 32;
    const int w2_tile = m == 0 ? 5 : (m == 1 ? 6 : (m == 2 ? 6 + g.nc : 7 + g.nc));
    if (r < dout * 32) return w2_tile * 1024 + r;                      // dW2_m[o][t] = row o of its tile(s)
    r -= dout * 32;
    const int col0 = m == 0 ? 0 : (m == 1 ? g.k : (m == 2 ? 8 * g.k : 9 * g.k));
    return (8 + g.nc) * 1024 + col0 + r;                               // db2_m[o]
}
constexpr int NG_REDUCE_SPLIT = 32;                                     // row groups of the first stage (its grid: column groups x this)
__global__ void __launch_bounds__(256) k_ng_reduce_weight_grads(const NgGradMap g, int waves, const float* __restrict__ partials, float* __restrict__ stage) {
    constexpr int W = 4, U = 8;                                        // waves per workgroup, loads in flight per wave
    __shared__ float part[W][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    const bool live = j < g.total;
    const int src = live ? ng_grad_source(g, j) : 0;
    const int rows = (waves + NG_REDUCE_SPLIT * W - 1) / (NG_REDUCE_SPLIT * W);
    const int r0 = min(waves, ((int)blockIdx.y * W + w) * rows), r1 = min(waves, r0 + rows);
    const float* p = partials + (size_t)r0 * g.per_wave + src;
    float acc = 0.f;
    int r = r0;
    for (; r + U <= r1; r += U) {                                      // a row every 41 KB: the loads are independent, only their sum is ordered
        float v[U];
#pragma unroll
        for (int q = 0; q < U; q++) v[q] = p[(size_t)q * g.per_wave];
#pragma unroll
        for (int q = 0; q < U; q++) acc += v[q];
        p += (size_t)U * g.per_wave;
    }
    for (; r < r1; r++) { acc += *p; p += g.per_wave; }
    part[w][lane] = acc;
    __syncthreads();
    if (w == 0 && live) stage[(size_t)blockIdx.y * g.total + j] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}
__global__ void __launch_bounds__(256) k_ng_reduce_weight_grads_fold(int total, const float* __restrict__ stage, float* __restrict__ out) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= total) return;
    float t = stage[j];
#pragma unroll
    for (int q = 1; q < NG_REDUCE_SPLIT; q++) t += stage[(size_t)q * total + j];
    out[j] = t;
}

// W2^T of the four MLPs in one launch (the B operand of Y = H W2^T): out = [32][k] | [32][7k] | [32][k] | [32][k], block m at
// 32 * (0, k, 8k, 9k).  Four framework transposes before, every forward.
struct NgW2 { const float* w[4]; int k; };
__global__ void __launch_bounds__(256) k_ng_transpose_w2(const NgW2 a, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 320 * a.k) return;
    const int k = a.k;
    const int m = i < 32 * k ? 0 : (i < 256 * k ? 1 : (i < 288 * k ? 2 : 3));
    const int dout = m == 1 ? 7 * k : k;
    const int r = i - 32 * (m == 0 ? 0 : (m == 1 ? k : (m == 2 ? 8 * k : 9 * k)));
    const int t = r / dout, o = r - t * dout;                           // out_m[t][o] = W2_m[o][t]
    out[i] = a.w[m][o * 32 + t];
}
}  // namespace lg

// ---------------------------------------------------------------------------------------------------------------------------
namespace {
int ng_model(const lidargs_ng_model* in, lg::NgModel* out) {
    if (!in) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "neural_gaussians: NULL model");
    const int k = in->n_offsets;
    if (!(k == 4 || k == 5 || k == 6 || k == 8 || k == 10)) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "neural_gaussians: n_offsets must be 4, 5, 6, 8 or 10");
    out->k = k;
    out->din[lg::NG_OPA] = 35 + (in->add_opacity_dist ? 1 : 0);
    out->din[lg::NG_COV] = 35 + (in->add_cov_dist ? 1 : 0);
    out->din[lg::NG_COL] = out->din[lg::NG_RD] = 35 + (in->add_color_dist ? 1 : 0);
    for (int m = 0; m < 4; m++) {
        if (!in->W1[m] || !in->b1[m] || !in->W2[m] || !in->b2[m]) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "neural_gaussians: NULL weight pointer");
        out->W1[m] = in->W1[m]; out->b1[m] = in->b1[m]; out->W2[m] = in->W2[m]; out->b2[m] = in->b2[m]; out->W2T[m] = in->W2T[m];
    }
    return 0;
}
// k <= 6: the forward's two MLP launches on 16x16x4 tiles (k_ng_opacity_t16 / k_ng_decode_t16: persistent workgroups of twelve waves, weights in
// LDS); k = 8, 10 and LIDARGS_NG_FORWARD_T16=0 (A/B, test variant): the 32x32x2 kernels, one wave per workgroup
static bool ng_forward_t16(int k) {
    if (k > 6) return false;
    const char* e = getenv("LIDARGS_NG_FORWARD_T16");
    return !(e && e[0] == '0');
}
static int ng_forward_grid(int N) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const int need = ((N + 31) / 32 + NGF_WAVES - 1) / NGF_WAVES;
    return need < cus ? (need > 0 ? need : 1) : cus;
}
#define NG_DISPATCH(K_, CALL) switch (K_) { case 4: { constexpr int K = 4; CALL; } break; case 5: { constexpr int K = 5; CALL; } break; \
    case 6: { constexpr int K = 6; CALL; } break; case 8: { constexpr int K = 8; CALL; } break; default: { constexpr int K = 10; CALL; } break; }
#define NG_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return lg::api_fail(LIDARGS_ERR_HIP, hipGetErrorString(e_)); } while (0)
}  // namespace

extern "C" {

size_t lidargs_ng_scratch_bytes(int N, int n_offsets) {
    return lg::ng_carve(nullptr, (size_t)(N > 0 ? N : 1), (size_t)(n_offsets > 0 ? n_offsets : 1), nullptr);
}

static int ng_forward_select(int N, const lidargs_ng_model* model, const uint8_t* visible_mask, const float* anchor_feat,
                             const float* anchor, const float* cam_center, float* neural_opacity, uint8_t* mask,
                             int* counts_host, char* scratch, size_t scratch_bytes, void* stream_, bool wait) {
    hipStream_t stream = (hipStream_t)stream_;
    lg::NgModel m;
    if (int rc = ng_model(model, &m)) return rc;
    if (N < 0 || !counts_host || !cam_center) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "ng_forward_select: bad argument");
    counts_host[0] = counts_host[1] = 0;
    if (N == 0) return 0;
    if (!anchor_feat || !anchor || !neural_opacity || !mask || !scratch) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "ng_forward_select: NULL pointer");
    if (scratch_bytes < lidargs_ng_scratch_bytes(N, m.k)) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "ng_forward_select: scratch too small");
    lg::NgScratch s; lg::ng_carve(scratch, (size_t)N, (size_t)m.k, &s);
    const float3 cam = make_float3(cam_center[0], cam_center[1], cam_center[2]);
    hipLaunchKernelGGL(lg::k_ng_visflags, dim3((N + 255) / 256), dim3(256), 0, stream, N, visible_mask, s.vis_flags);
    lg::launch_exclusive_scan(s.vis_flags, s.vis_idx, (size_t)N, s.totals, s.scan, stream);
    static const bool per_lane = [] { const char* e = getenv("LIDARGS_NG_PER_LANE_DECODE"); return e && atoi(e) != 0; }();
    if (per_lane || !m.W2T[0]) {
        NG_DISPATCH(m.k, hipLaunchKernelGGL(lg::k_ng_opacity<K>, dim3((N + 63) / 64), dim3(64), 0, stream, N, m, cam, anchor_feat, anchor, s.vis_flags,
                                            s.vis_idx, neural_opacity, mask, s.sel_flags));
    } else if (ng_forward_t16(m.k)) {
        const int grid = ng_forward_grid(N);
#define NG_OPA_T16(K_) hipLaunchKernelGGL(lg::k_ng_opacity_t16<K_>, dim3(grid), dim3(64 * NGF_WAVES), 0, stream, N, m, cam, anchor_feat, anchor, s.vis_flags, \
                                            s.vis_idx, neural_opacity, mask, s.sel_flags)
        if (m.k == 4) NG_OPA_T16(4); else if (m.k == 5) NG_OPA_T16(5); else NG_OPA_T16(6);
#undef NG_OPA_T16
    } else {
        NG_DISPATCH(m.k, hipLaunchKernelGGL(lg::k_ng_opacity_mfma<K>, dim3((N + 63) / 64), dim3(64), 0, stream, N, m, cam, anchor_feat, anchor, s.vis_flags,
                                            s.vis_idx, neural_opacity, mask, s.sel_flags));
    }
    lg::launch_exclusive_scan(s.sel_flags, s.slot, (size_t)N * m.k, s.totals + 1, s.scan, stream);
    if (!wait) {                                                       // counts_host is pinned memory of the caller's, who waits (an event behind this call)
        NG_HIP(hipMemcpyAsync(counts_host, s.totals, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        return 0;
    }
    uint32_t tot[2] = {0, 0};
    NG_HIP(hipMemcpyAsync(tot, s.totals, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    NG_HIP(hipStreamSynchronize(stream));
    counts_host[0] = (int)tot[0]; counts_host[1] = (int)tot[1];
    return (int)tot[1];
}
int lidargs_ng_forward_select(int N, const lidargs_ng_model* model, const uint8_t* visible_mask, const float* anchor_feat,
                              const float* anchor, const float* cam_center, float* neural_opacity, uint8_t* mask,
                              int* counts_host, char* scratch, size_t scratch_bytes, void* stream) {
    return ng_forward_select(N, model, visible_mask, anchor_feat, anchor, cam_center, neural_opacity, mask, counts_host, scratch, scratch_bytes, stream, true);
}
int lidargs_ng_forward_select_enqueue(int N, const lidargs_ng_model* model, const uint8_t* visible_mask, const float* anchor_feat,
                                      const float* anchor, const float* cam_center, float* neural_opacity, uint8_t* mask,
                                      int* counts_pinned, char* scratch, size_t scratch_bytes, void* stream) {
    return ng_forward_select(N, model, visible_mask, anchor_feat, anchor, cam_center, neural_opacity, mask, counts_pinned, scratch, scratch_bytes, stream, false);
}

int lidargs_ng_forward_decode(int N, const lidargs_ng_model* model, const float* anchor_feat, const float* anchor,
                              const float* offset, const float* scaling, const float* cam_center, const float* neural_opacity,
                              float* out_xyz, float* out_color, float* out_opacity, float* out_scaling, float* out_rot,
                              char* scratch, size_t scratch_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    lg::NgModel m;
    if (int rc = ng_model(model, &m)) return rc;
    if (N <= 0) return 0;
    if (!anchor_feat || !anchor || !offset || !scaling || !cam_center || !neural_opacity || !scratch)
        return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "ng_forward_decode: NULL pointer");
    if (scratch_bytes < lidargs_ng_scratch_bytes(N, m.k)) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "ng_forward_decode: scratch too small");
    lg::NgScratch s; lg::ng_carve(scratch, (size_t)N, (size_t)m.k, &s);
    const float3 cam = make_float3(cam_center[0], cam_center[1], cam_center[2]);
    static const bool per_lane = [] { const char* e = getenv("LIDARGS_NG_PER_LANE_DECODE"); return e && atoi(e) != 0; }();
    if (per_lane || !m.W2T[0] || !m.W2T[1] || !m.W2T[2] || !m.W2T[3]) {     // the matrix-pipe decode reads the transposed second-layer weights
    NG_DISPATCH(m.k, hipLaunchKernelGGL(lg::k_ng_decode<K>, dim3((N + 63) / 64), dim3(64), 0, stream, N, m, cam, anchor_feat, anchor, offset, scaling,
                                        s.vis_flags, s.vis_idx, s.sel_flags, s.slot, neural_opacity, out_xyz, out_color, out_opacity, out_scaling, out_rot));
    } else if (ng_forward_t16(m.k)) {
        const int grid = ng_forward_grid(N);
#define NG_DEC_T16(K_) hipLaunchKernelGGL(lg::k_ng_decode_t16<K_>, dim3(grid), dim3(64 * NGF_WAVES), 0, stream, N, m, cam, anchor_feat, anchor, offset, scaling, \
                                        s.vis_flags, s.vis_idx, s.sel_flags, s.slot, neural_opacity, out_xyz, out_color, out_opacity, out_scaling, out_rot)
        if (m.k == 4) NG_DEC_T16(4); else if (m.k == 5) NG_DEC_T16(5); else NG_DEC_T16(6);
#undef NG_DEC_T16
    } else {
    NG_DISPATCH(m.k, hipLaunchKernelGGL(lg::k_ng_decode_mfma<K>, dim3((N + 63) / 64), dim3(64), 0, stream, N, m, cam, anchor_feat, anchor, offset, scaling,
                                        s.vis_flags, s.vis_idx, s.sel_flags, s.slot, neural_opacity, out_xyz, out_color, out_opacity, out_scaling, out_rot));
    }

    NG_HIP(hipGetLastError());
    return 0;
}

int lidargs_ng_backward(int N, int n_visible, const lidargs_ng_model* model, const float* anchor_feat, const float* anchor,
                        const float* offset, const float* scaling, const float* cam_center,
                        const float* dL_dxyz, const float* dL_dcolor, const float* dL_dopacity, const float* dL_dscaling,
                        const float* dL_drot, const float* dL_dneural_opacity, float* dL_danchor_feat, float* dL_danchor, float* dL_doffset,
                        float* dL_dscaling_in, float* act_x, float* act_h, float* delta1, float* delta2,
                        char* scratch, size_t scratch_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    lg::NgModel m;
    if (int rc = ng_model(model, &m)) return rc;
    if (N <= 0) return 0;
    if (!anchor_feat || !anchor || !offset || !scaling || !cam_center || !dL_danchor_feat || !dL_danchor || !dL_doffset || !dL_dscaling_in || !scratch)
        return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "ng_backward: NULL pointer");
    for (int q = 0; q < 4; q++) if (!m.W2T[q]) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "ng_backward: model.W2T (transposed second-layer weights) is required");
    if (n_visible > 0 && (!act_x || !act_h || !delta1 || !delta2)) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "ng_backward: NULL activation buffer");
    if (scratch_bytes < lidargs_ng_scratch_bytes(N, m.k)) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "ng_backward: scratch too small");
    lg::NgScratch s; lg::ng_carve(scratch, (size_t)N, (size_t)m.k, &s);
    const float3 cam = make_float3(cam_center[0], cam_center[1], cam_center[2]);
    NG_DISPATCH(m.k, hipLaunchKernelGGL(lg::k_ng_backward<K>, dim3((N + 63) / 64), dim3(64), 0, stream, N, n_visible, m, cam, anchor_feat, anchor, offset,
                                        scaling, s.vis_flags, s.vis_idx, s.sel_flags, s.slot, dL_dxyz, dL_dcolor, dL_dopacity, dL_dscaling, dL_drot, dL_dneural_opacity,
                                        dL_danchor_feat, dL_danchor, dL_doffset, dL_dscaling_in, act_x, act_h, delta1, delta2));
    NG_HIP(hipGetLastError());
    return 0;
}

static int ng_cus() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return cus;
}
// k <= 6: k_ng_backward_t16 (16x16x4 tiles, one workgroup of eight waves per CU, one partial row per workgroup); k = 8, 10 (its LDS does
// not hold their delta2 rows for eight waves) and LIDARGS_NG_BACKWARD_T16=0 (A/B, test variant): k_ng_backward_mfma, one wave per SIMD
static bool ng_use_t16(int k) {
    if (k > 6) return false;
    const char* e = getenv("LIDARGS_NG_BACKWARD_T16");
    return !(e && e[0] == '0');
}
// LIDARGS_NG_T16_PASSES=1: the four MLPs in one launch (A/B); default two launches (see k_ng_backward_t16)
static bool ng_t16_two_pass() { const char* e = getenv("LIDARGS_NG_T16_PASSES"); return !(e && e[0] == '1'); }
// The variant is chosen by environment switches read per call (the suite selects variants in one process), and the backward takes no
// capacity argument: what lidargs_ng_backward_partials last told this thread for k is remembered, and a backward whose variant would now
// write a different number of partial rows refuses instead of overflowing the caller's buffer (round-5 advisor finding).
static thread_local int t_ng_rows_reported[16] = {0};
static int ng_partial_rows(int k) { return ng_use_t16(k) ? (ng_t16_two_pass() ? 2 : 1) * ng_cus() : 4 * ng_cus(); }
int lidargs_ng_backward_partials(int n_offsets, int* waves, int* floats_per_wave) {
    const int k = n_offsets;
    if (!(k == 4 || k == 5 || k == 6 || k == 8 || k == 10) || !waves || !floats_per_wave) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "ng_backward_partials: bad argument");
    *waves = ng_partial_rows(k);
    t_ng_rows_reported[k] = *waves;
    *floats_per_wave = (5 + 3 + (7 * k + 31) / 32) * 1024 + 128;
    return 0;
}
int lidargs_ng_backward_mfma(int N, const lidargs_ng_model* model, const float* anchor_feat, const float* anchor,
                             const float* offset, const float* scaling, const float* cam_center,
                             const float* dL_dxyz, const float* dL_dcolor, const float* dL_dopacity, const float* dL_dscaling,
                             const float* dL_drot, const float* dL_dneural_opacity, float* dL_danchor_feat, float* dL_danchor, float* dL_doffset,
                             float* dL_dscaling_in, float* partials, char* scratch, size_t scratch_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    lg::NgModel m;
    if (int rc = ng_model(model, &m)) return rc;
    if (N <= 0) return 0;
    if (!anchor_feat || !anchor || !offset || !scaling || !cam_center || !dL_danchor_feat || !dL_danchor || !dL_doffset || !dL_dscaling_in || !partials || !scratch)
        return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "ng_backward_mfma: NULL pointer");
    for (int q = 0; q < 4; q++) if (!m.W2T[q]) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "ng_backward_mfma: model.W2T (transposed second-layer weights) is required");
    if (scratch_bytes < lidargs_ng_scratch_bytes(N, m.k)) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "ng_backward_mfma: scratch too small");
    lg::NgScratch s; lg::ng_carve(scratch, (size_t)N, (size_t)m.k, &s);
    const float3 cam = make_float3(cam_center[0], cam_center[1], cam_center[2]);
    const int waves = ng_partial_rows(m.k);
    if (t_ng_rows_reported[m.k] && t_ng_rows_reported[m.k] != waves)
        return lg::api_fail(LIDARGS_ERR_STATE, "ng_backward_mfma: the backward variant (LIDARGS_NG_BACKWARD_T16 / LIDARGS_NG_T16_PASSES) changed since lidargs_ng_backward_partials sized the partial rows; call it again");
#define NG_T16_LAUNCH(K_, P_) hipLaunchKernelGGL((lg::k_ng_backward_t16<K_, P_>), dim3(ng_cus()), dim3(64 * NGT_WAVES), 0, stream, N, m, cam, anchor_feat, anchor, offset, \
                                        scaling, s.vis_flags, s.vis_idx, s.sel_flags, s.slot, dL_dxyz, dL_dcolor, dL_dopacity, dL_dscaling, dL_drot, dL_dneural_opacity, \
                                        dL_danchor_feat, dL_danchor, dL_doffset, dL_dscaling_in, partials, stagger)
    if (ng_use_t16(m.k)) {
        const char* sg = getenv("LIDARGS_NG_T16_STAGGER");
        const int stagger = sg ? atoi(sg) : 0;
        if (ng_t16_two_pass()) {
            if (m.k == 4) { NG_T16_LAUNCH(4, 2); NG_T16_LAUNCH(4, 1); } else if (m.k == 5) { NG_T16_LAUNCH(5, 2); NG_T16_LAUNCH(5, 1); } else { NG_T16_LAUNCH(6, 2); NG_T16_LAUNCH(6, 1); }
        } else {
            if (m.k == 4) NG_T16_LAUNCH(4, 0); else if (m.k == 5) NG_T16_LAUNCH(5, 0); else NG_T16_LAUNCH(6, 0);
        }
    } else {
        NG_DISPATCH(m.k, hipLaunchKernelGGL(lg::k_ng_backward_mfma<K>, dim3(waves), dim3(64), 0, stream, N, m, cam, anchor_feat, anchor, offset,
                                            scaling, s.vis_flags, s.vis_idx, s.sel_flags, s.slot, dL_dxyz, dL_dcolor, dL_dopacity, dL_dscaling, dL_drot, dL_dneural_opacity,
                                            dL_danchor_feat, dL_danchor, dL_doffset, dL_dscaling_in, partials));
    }
#undef NG_T16_LAUNCH
    NG_HIP(hipGetLastError());
    return 0;
}

int lidargs_ng_transpose_w2(int n_offsets, const float* const* W2, float* out, void* stream_) {
    const int k = n_offsets;
    if (!(k == 4 || k == 5 || k == 6 || k == 8 || k == 10) || !W2 || !out || !W2[0] || !W2[1] || !W2[2] || !W2[3])
        return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "ng_transpose_w2: bad argument");
    lg::NgW2 a; a.k = k;
    for (int m = 0; m < 4; m++) a.w[m] = W2[m];
    hipLaunchKernelGGL(lg::k_ng_transpose_w2, dim3((320 * k + 255) / 256), dim3(256), 0, (hipStream_t)stream_, a, out);
    NG_HIP(hipGetLastError());
    return 0;
}
static int ng_grad_map(int k, const int* din, lg::NgGradMap* g) {
    if (!(k == 4 || k == 5 || k == 6 || k == 8 || k == 10) || !din) return 1;
    g->k = k; g->nc = (7 * k + 31) / 32; g->per_wave = (5 + 3 + g->nc) * 1024 + 128;
    int off = 0;
    for (int m = 0; m < 4; m++) {
        if (din[m] < 32 || din[m] > 36) return 1;
        const int dout = m == 1 ? 7 * k : k;
        g->din[m] = din[m]; g->base[m] = off;
        off += 32 * din[m] + 32 + dout * 32 + dout;
    }
    g->total = off;
    return 0;
}
int lidargs_ng_weight_grad_floats(int n_offsets, const int* din) {
    lg::NgGradMap g;
    if (ng_grad_map(n_offsets, din, &g)) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "ng_weight_grad_floats: bad argument");
    return g.total;
}
int lidargs_ng_weight_grad_stage_floats(int n_offsets, const int* din) {
    lg::NgGradMap g;
    if (ng_grad_map(n_offsets, din, &g)) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "ng_weight_grad_stage_floats: bad argument");
    return lg::NG_REDUCE_SPLIT * g.total;
}
int lidargs_ng_reduce_weight_grads(int n_offsets, const int* din, int waves, const float* partials, float* grads, float* stage, void* stream_) {
    lg::NgGradMap g;
    if (ng_grad_map(n_offsets, din, &g) || waves <= 0 || !partials || !grads || !stage) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "ng_reduce_weight_grads: bad argument");
    hipLaunchKernelGGL(lg::k_ng_reduce_weight_grads, dim3((g.total + 63) / 64, lg::NG_REDUCE_SPLIT), dim3(256), 0, (hipStream_t)stream_, g, waves, partials, stage);
    hipLaunchKernelGGL(lg::k_ng_reduce_weight_grads_fold, dim3((g.total + 255) / 256), dim3(256), 0, (hipStream_t)stream_, g.total, stage, grads);
    NG_HIP(hipGetLastError());
    return 0;
}

int lidargs_ng_training_stats(int N, int n_offsets, const uint8_t* anchor_visible_mask, const uint8_t* offset_selection_mask,
                              const uint8_t* update_filter, const float* neural_opacity, const float* viewspace_grad,
                              float* opacity_accum, float* anchor_demon, float* offset_gradient_accum, float* offset_denom,
                              char* scratch, size_t scratch_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int k = n_offsets;
    if (N < 0 || k < 1) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "ng_training_stats: bad sizes");
    if (N == 0) return 0;
    if (!opacity_accum || !anchor_demon || !offset_gradient_accum || !offset_denom || !scratch)
        return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "ng_training_stats: NULL pointer");
    if (scratch_bytes < lidargs_ng_scratch_bytes(N, k)) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "ng_training_stats: scratch too small");
    lg::NgScratch s; lg::ng_carve(scratch, (size_t)N, (size_t)k, &s);
    hipLaunchKernelGGL(lg::k_ng_visflags, dim3((N + 255) / 256), dim3(256), 0, stream, N, anchor_visible_mask, s.vis_flags);
    lg::launch_exclusive_scan(s.vis_flags, s.vis_idx, (size_t)N, s.totals, s.scan, stream);
    const size_t NK = (size_t)N * k;
    hipLaunchKernelGGL(lg::k_ng_stats_flags, dim3((unsigned)((NK + 255) / 256)), dim3(256), 0, stream, N, k, s.vis_flags, s.vis_idx, offset_selection_mask, s.sel_flags);
    lg::launch_exclusive_scan(s.sel_flags, s.slot, NK, s.totals + 1, s.scan, stream);
    hipLaunchKernelGGL(lg::k_ng_stats, dim3((N + 255) / 256), dim3(256), 0, stream, N, k, s.vis_flags, s.vis_idx, s.sel_flags, s.slot, update_filter,
                       neural_opacity, viewspace_grad, opacity_accum, anchor_demon, offset_gradient_accum, offset_denom);
    NG_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
