// render.hip -- forward and backward alpha-compositing of the range image (gfx950).
//
// Replaces renderCUDA fwd (R3/cr/forward.cu:502-641) and bwd (R3/cr/backward.cu:535-791).
//
// Mapping.  The reference runs 16-thread blocks (a quarter of a wave64) on 16x1 tiles and
// re-evaluates cos/sin of the pixel ray for every (pixel, Gaussian) pair.  Here one wave64 owns a
// 16-column x 4-row pixel patch; a list tile is 16 columns x TH rows (TH/4 waves, each an
// independent 64-thread workgroup, so there are no workgroup barriers across waves).  The 16-pixel
// tile WIDTH is kept equal to the reference's BLOCK_X so that "is this pixel inside the
// Gaussian's rect" -- which is observable, because the rect truncates the footprint at ~3 sigma
// where alpha can still exceed 1/255 -- reduces to a per-lane test on the pixel ROW only
// (ymin <= y < ymax): tile columns are shared with the reference grid.
//
// Per chunk of 64 list entries: lane l gathers entry l's 64-byte splat record (one aligned
// segment) + row span into registers while the previous chunk is being composited, then parks it
// in LDS component-major; the inner loop reads entry j with four broadcast ds_read_b128.
// The pixel's unit ray comes from two small tables (cos/sin per row and per column).
//
// Backward.  The reference issues 20 global float atomics per contributing (pixel, Gaussian)
// pair.  Here the 64 pixels of a wave handle the SAME Gaussian in the same step, so the 16 sums
// that are needed (dL/dsphere follows by linearity from dL/dmean2D, see preprocess.hip) go through
// a wave-level butterfly REDUCE-SCATTER (8+4+2+1+1+1 exchanges instead of 16x6), after which 16
// lanes each hold one finished sum and issue ONE atomic instruction into a packed 64-byte
// accumulator line of that Gaussian.  Entries no lane contributes to are skipped wholesale.
#include "lidargs_common.h"

namespace lg {

#define LG_CHUNK 64

struct PixelSetup {
    int x, y, pix;
    bool inside;
    float3 q;
};

__device__ __forceinline__ PixelSetup pixel_setup(const TileGrid& g, const float2* __restrict__ coltab, const float2* __restrict__ rowtab,
                                                  int tile, int sub, int lane) {
    PixelSetup p;
    const int tx = tile % g.tiles_x, ty = tile / g.tiles_x;
    p.x = tx * LG_TILE_W + (lane & 15);
    p.y = ty * g.TH + sub * LG_WAVE_ROWS + (lane >> 4);
    p.inside = (p.x < g.W) && (p.y < g.H);
    p.pix = p.y * g.W + p.x;
    p.q = make_float3(0.f, 0.f, 0.f);
    if (p.inside) {
        const float2 cb = coltab[p.x], ca = rowtab[p.y];
        p.q = make_float3(ca.x * cb.x, ca.x * cb.y, ca.y);           // (cos a cos b, cos a sin b, sin a)
    }
    return p;
}

struct Staged { float4 a0, a1, a2, a3; uint32_t span; };

__device__ __forceinline__ Staged gather_entry(const uint32_t* __restrict__ point_list, const float4* __restrict__ rec,
                                               const uint32_t* __restrict__ rowspan, uint32_t k, bool valid) {
    Staged s;
    s.a0 = s.a1 = s.a2 = s.a3 = make_float4(0.f, 0.f, 0.f, 0.f);
    s.span = 0;                                                        // empty row span: no pixel matches
    if (valid) {
        const uint32_t g = point_list[k];
        const float4* r = rec + 4 * (size_t)g;
        s.a0 = r[0]; s.a1 = r[1]; s.a2 = r[2]; s.a3 = r[3];
        s.span = rowspan[g];
    }
    return s;
}

// ------------------------------------------------------------------------------------------------
template <bool T_ONLY>
__global__ void __launch_bounds__(64) k_render_forward(const RenderFwdArgs a) {
    __shared__ float4 s_rec[4 * LG_CHUNK];
    __shared__ uint32_t s_span[LG_CHUNK];
    const int lane = threadIdx.x;
    const int wpt = a.grid.waves_per_tile;
    const int tile = blockIdx.x / wpt, sub = blockIdx.x - tile * wpt;
    const PixelSetup px = pixel_setup(a.grid, a.coltab, a.rowtab, tile, sub, lane);
    const uint2 range = a.ranges[tile];
    const uint32_t n = range.y - range.x;

    float T = 1.0f;
    if (a.T_in && px.inside) T = a.T_in[px.pix];
    float T_break = T;
    float C0 = 0.f, C1 = 0.f, D = 0.f;
    uint32_t last = 0;
    bool done = !px.inside;

    const uint32_t nchunks = (n + LG_CHUNK - 1) / LG_CHUNK;
    Staged st = gather_entry(a.point_list, a.rec, a.rowspan, range.x + lane, (uint32_t)lane < n);
    for (uint32_t c = 0; c < nchunks; c++) {
        __syncthreads();
        s_rec[lane] = st.a0; s_rec[LG_CHUNK + lane] = st.a1; s_rec[2 * LG_CHUNK + lane] = st.a2; s_rec[3 * LG_CHUNK + lane] = st.a3;
        s_span[lane] = st.span;
        __syncthreads();
        if (c + 1 < nchunks) {
            const uint32_t k = (c + 1) * LG_CHUNK + lane;
            st = gather_entry(a.point_list, a.rec, a.rowspan, range.x + k, k < n);
        }
        const uint32_t cnt = min((uint32_t)LG_CHUNK, n - c * LG_CHUNK);
        if (__ballot(!done) == 0ull) break;                           // R3/cr/forward.cu:559-561 early-out
        for (uint32_t j = 0; j < cnt; j++) {
            const float4 r0 = s_rec[j], r1 = s_rec[LG_CHUNK + j], r2 = s_rec[2 * LG_CHUNK + j], r3 = s_rec[3 * LG_CHUNK + j];
            const uint32_t span = s_span[j];
            const bool rows = ((uint32_t)px.y >= (span & 0xFFFFu)) && ((uint32_t)px.y < (span >> 16));
            if (!done && rows) {
                const float ex = r0.x - px.q.x, ey = r0.y - px.q.y, ez = r0.z - px.q.z;
                const float dx = ex * r1.x + ey * r1.y + ez * r1.z;
                const float dy = ex * r2.x + ey * r2.y + ez * r2.z;
                const float power = -0.5f * (r1.w * dx * dx + r3.x * dy * dy) - r2.w * dx * dy;   // :601
                if (power <= 0.0f) {
                    const float alpha = fminf(0.99f, r3.y * __expf(power));
                    if (alpha >= 1.0f / 255.0f) {
                        const float test_T = T * (1.f - alpha);
                        if (test_T < 0.0001f) { done = true; T_break = test_T; }
                        else {
                            if (!T_ONLY) {
                                const float w = alpha * T;
                                C0 += r3.z * w; C1 += r3.w * w; D += r0.w * w;
                            }
                            T = test_T; T_break = test_T;
                            last = c * LG_CHUNK + j + 1;
                        }
                    }
                }
            }
        }
    }

    if (px.inside) {
        if (a.T_pass) a.T_pass[px.pix] = T_break;
        if (!T_ONLY) {
            const size_t N = (size_t)a.grid.W * a.grid.H;
            a.final_T[px.pix] = T;
            a.n_contrib[px.pix] = last;
            const float b0 = a.bg ? a.bg[0] : 0.f, b1 = a.bg ? a.bg[1] : 0.f;
            a.out_color[px.pix] = C0 + T * b0;                         // :637
            a.out_color[N + px.pix] = C1 + T * b1;
            a.out_depth[px.pix] = D;
            a.out_occ[px.pix] = 1.f - T;
        }
    }
}

void launch_render_forward(const RenderFwdArgs& a, hipStream_t s) {
    const unsigned blocks = (unsigned)(a.grid.num_tiles() * a.grid.waves_per_tile);
    if (a.transmittance_only) hipLaunchKernelGGL(k_render_forward<true>, dim3(blocks), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(k_render_forward<false>, dim3(blocks), dim3(64), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// Butterfly reduce-scatter of 16 per-lane values over the 64 lanes of a wave.  On return lane L
// (L < 16; every 16-lane row holds the same) owns the wave-wide sum of value slot
//   id(L) = 8*(L&1) + 4*((L>>1)&1) + 2*((L>>2)&1) + ((L>>3)&1)
// in v[0].
__device__ __forceinline__ float reduce_scatter16(float (&v)[16], int lane) {
    {
        const bool hi = lane & 1;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float keep = hi ? v[k + 8] : v[k], send = hi ? v[k] : v[k + 8];
            v[k] = keep + __shfl_xor(send, 1);
        }
    }
    {
        const bool hi = lane & 2;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float keep = hi ? v[k + 4] : v[k], send = hi ? v[k] : v[k + 4];
            v[k] = keep + __shfl_xor(send, 2);
        }
    }
    {
        const bool hi = lane & 4;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const float keep = hi ? v[k + 2] : v[k], send = hi ? v[k] : v[k + 2];
            v[k] = keep + __shfl_xor(send, 4);
        }
    }
    {
        const bool hi = lane & 8;
        const float keep = hi ? v[1] : v[0], send = hi ? v[0] : v[1];
        v[0] = keep + __shfl_xor(send, 8);
    }
    v[0] += __shfl_xor(v[0], 16);
    v[0] += __shfl_xor(v[0], 32);
    return v[0];
}

__global__ void __launch_bounds__(64) k_render_backward(const RenderBwdArgs a) {
    __shared__ float4 s_rec[4 * LG_CHUNK];
    __shared__ uint32_t s_span[LG_CHUNK];
    __shared__ uint32_t s_gid[LG_CHUNK];
    const int lane = threadIdx.x;
    const int wpt = a.grid.waves_per_tile;
    const int tile = blockIdx.x / wpt, sub = blockIdx.x - tile * wpt;
    const PixelSetup px = pixel_setup(a.grid, a.coltab, a.rowtab, tile, sub, lane);
    const uint2 range = a.ranges[tile];
    const size_t N = (size_t)a.grid.W * a.grid.H;

    const uint32_t n_lane = px.inside ? a.n_contrib[px.pix] : 0u;
    uint32_t n_max = n_lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_max = max(n_max, (uint32_t)__shfl_xor((int)n_max, o));
    if (n_max == 0) return;

    // per-pixel state of the back-to-front walk (R3/cr/backward.cu:590-615)
    float T = px.inside ? a.final_T[px.pix] : 0.f;
    const float T_final = a.T_final_global ? (px.inside ? a.T_final_global[px.pix] : 0.f) : T;
    float g0 = 0.f, g1 = 0.f, gd = 0.f, go = 0.f;
    if (px.inside) { g0 = a.dL_dpix[px.pix]; g1 = a.dL_dpix[N + px.pix]; gd = a.dL_ddepth[px.pix]; go = a.dL_docc[px.pix]; }
    const float bgdot = a.bg ? (a.bg[0] * g0 + a.bg[1] * g1) : 0.f;
    float acc0 = 0.f, acc1 = 0.f, accd = 0.f, acco = 0.f;           // accum_rec[2], accum_red, accum_reo
    float last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, ld = 0.f;
    if (a.behind && px.inside && T > 0.f) {
        // what lies behind this range shell, as seen from its far boundary (multi-GPU only)
        const float inv = 1.f / T;
        acc0 = a.behind[px.pix] * inv; acc1 = a.behind[N + px.pix] * inv; accd = a.behind[2 * N + px.pix] * inv;
        acco = 1.f - T_final * inv;
    }

    const int c_last = (int)((n_max - 1) / LG_CHUNK);
    auto gather = [&](int c, Staged& st, uint32_t& gid) {
        const uint32_t k = (uint32_t)c * LG_CHUNK + lane;
        const bool valid = k < n_max;
        gid = valid ? a.point_list[range.x + k] : 0u;
        st = gather_entry(a.point_list, a.rec, a.rowspan, range.x + k, valid);
    };
    Staged st; uint32_t gid;
    gather(c_last, st, gid);
    for (int c = c_last; c >= 0; c--) {
        __syncthreads();
        s_rec[lane] = st.a0; s_rec[LG_CHUNK + lane] = st.a1; s_rec[2 * LG_CHUNK + lane] = st.a2; s_rec[3 * LG_CHUNK + lane] = st.a3;
        s_span[lane] = st.span; s_gid[lane] = gid;
        __syncthreads();
        if (c > 0) gather(c - 1, st, gid);
        const int hi = (int)min((uint32_t)LG_CHUNK, n_max - (uint32_t)c * LG_CHUNK) - 1;
        for (int j = hi; j >= 0; j--) {
            const uint32_t e = (uint32_t)c * LG_CHUNK + j;            // 0-based list position
            const float4 r0 = s_rec[j], r1 = s_rec[LG_CHUNK + j], r2 = s_rec[2 * LG_CHUNK + j], r3 = s_rec[3 * LG_CHUNK + j];
            const uint32_t span = s_span[j];
            const bool rows = ((uint32_t)px.y >= (span & 0xFFFFu)) && ((uint32_t)px.y < (span >> 16));
            bool contrib = false;
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = 0.f;
            if (rows && e < n_lane) {                                  // :650 skip entries behind the last contributor
                const float ex = r0.x - px.q.x, ey = r0.y - px.q.y, ez = r0.z - px.q.z;
                const float dx = ex * r1.x + ey * r1.y + ez * r1.z;
                const float dy = ex * r2.x + ey * r2.y + ez * r2.z;
                const float A = r1.w, B = r2.w, Cc = r3.x, op = r3.y;
                const float power = -0.5f * (A * dx * dx + Cc * dy * dy) - B * dx * dy;
                if (power <= 0.0f) {
                    const float G = __expf(power);
                    const float alpha = fminf(0.99f, op * G);
                    if (alpha >= 1.0f / 255.0f) {
                        contrib = true;
                        T = T / (1.f - alpha);                         // :681
                        const float w = alpha * T;
                        acc0 = last_alpha * lc0 + (1.f - last_alpha) * acc0; lc0 = r3.z;
                        acc1 = last_alpha * lc1 + (1.f - last_alpha) * acc1; lc1 = r3.w;
                        accd = last_alpha * ld + (1.f - last_alpha) * accd; ld = r0.w;
                        acco = last_alpha + (1.f - last_alpha) * acco;
                        float dL_dalpha = (r3.z - acc0) * g0 + (r3.w - acc1) * g1 + (r0.w - accd) * gd + (1.f - acco) * go;
                        dL_dalpha *= T;
                        last_alpha = alpha;
                        dL_dalpha += (-T_final / (1.f - alpha)) * bgdot;   // :727
                        const float dL_dG = op * dL_dalpha;
                        const float gdx = G * dx, gdy = G * dy;
                        const float gx = dL_dG * (-gdx * A - gdy * B);     // dL/dmean2D.x  (:734,:753)
                        const float gy = dL_dG * (-gdy * Cc - gdx * B);    // dL/dmean2D.y
                        // per-pixel sphere-gradient norm statistic (:759-779): |gx u1' + gy u2'|
                        const float sx = gx * r1.x + gy * r2.x, sy = gx * r1.y + gy * r2.y, sz = gx * r1.z + gy * r2.z;
                        // dL/du1 = gx (delta/uu1 - 2 dx u1'),  1/uu1 = |u1'|^2      (:738-750)
                        const float iu1 = r1.x * r1.x + r1.y * r1.y + r1.z * r1.z;
                        const float iu2 = r2.x * r2.x + r2.y * r2.y + r2.z * r2.z;
                        const float t1 = -2.f * dx, t2 = -2.f * dy;
                        v[0] = gx;
                        v[1] = gy;
                        v[2] = sqrtf(sx * sx + sy * sy + sz * sz);
                        v[3] = -0.5f * gdx * dx * dL_dG;               // conic A (:783)
                        v[4] = -0.5f * gdx * dy * dL_dG;               // conic B
                        v[5] = -0.5f * gdy * dy * dL_dG;               // conic C
                        v[6] = G * dL_dalpha;                          // opacity (:788)
                        v[7] = w * g0;                                 // colours (:702)
                        v[8] = w * g1;
                        v[9] = w * gd;                                 // range (:711)
                        v[10] = gx * (ex * iu1 + t1 * r1.x);
                        v[11] = gx * (ey * iu1 + t1 * r1.y);
                        v[12] = gx * (ez * iu1 + t1 * r1.z);
                        v[13] = gy * (ex * iu2 + t2 * r2.x);
                        v[14] = gy * (ey * iu2 + t2 * r2.y);
                        v[15] = gy * (ez * iu2 + t2 * r2.z);
                    }
                }
            }
            if (__ballot(contrib) == 0ull) continue;                   // wave-uniform
            const float mine = reduce_scatter16(v, lane);
            if (lane < 16) {
                const int slot = 8 * (lane & 1) + 4 * ((lane >> 1) & 1) + 2 * ((lane >> 2) & 1) + ((lane >> 3) & 1);
                atomicAdd(a.gacc + 16 * (size_t)s_gid[j] + slot, mine);
            }
        }
    }
}

void launch_render_backward(const RenderBwdArgs& a, hipStream_t s) {
    const unsigned blocks = (unsigned)(a.grid.num_tiles() * a.grid.waves_per_tile);
    hipLaunchKernelGGL(k_render_backward, dim3(blocks), dim3(64), 0, s, a);
}

}  // namespace lg
