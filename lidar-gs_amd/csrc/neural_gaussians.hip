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
#define NG_DISPATCH(K_, CALL) switch (K_) { case 4: { constexpr int K = 4; CALL; } break; case 5: { constexpr int K = 5; CALL; } break; \
    case 6: { constexpr int K = 6; CALL; } break; case 8: { constexpr int K = 8; CALL; } break; default: { constexpr int K = 10; CALL; } break; }
#define NG_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return lg::api_fail(LIDARGS_ERR_HIP, hipGetErrorString(e_)); } while (0)
}  // namespace

extern "C" {

size_t lidargs_ng_scratch_bytes(int N, int n_offsets) {
    return lg::ng_carve(nullptr, (size_t)(N > 0 ? N : 1), (size_t)(n_offsets > 0 ? n_offsets : 1), nullptr);
}

int lidargs_ng_forward_select(int N, const lidargs_ng_model* model, const uint8_t* visible_mask, const float* anchor_feat,
                              const float* anchor, const float* cam_center, float* neural_opacity, uint8_t* mask,
                              int* counts_host, char* scratch, size_t scratch_bytes, void* stream_) {
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
    NG_DISPATCH(m.k, hipLaunchKernelGGL(lg::k_ng_opacity<K>, dim3((N + 63) / 64), dim3(64), 0, stream, N, m, cam, anchor_feat, anchor, s.vis_flags,
                                        s.vis_idx, neural_opacity, mask, s.sel_flags));
    lg::launch_exclusive_scan(s.sel_flags, s.slot, (size_t)N * m.k, s.totals + 1, s.scan, stream);
    uint32_t tot[2] = {0, 0};
    NG_HIP(hipMemcpyAsync(tot, s.totals, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    NG_HIP(hipStreamSynchronize(stream));
    counts_host[0] = (int)tot[0]; counts_host[1] = (int)tot[1];
    return (int)tot[1];
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
    NG_DISPATCH(m.k, hipLaunchKernelGGL(lg::k_ng_decode<K>, dim3((N + 63) / 64), dim3(64), 0, stream, N, m, cam, anchor_feat, anchor, offset, scaling,
                                        s.vis_flags, s.vis_idx, s.sel_flags, s.slot, neural_opacity, out_xyz, out_color, out_opacity, out_scaling, out_rot));
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
