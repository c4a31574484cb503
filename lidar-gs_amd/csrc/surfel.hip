// surfel.hip -- the 2DGS "laser-surfel" rasterizer (BASELINE config 5) on gfx950.
//
// Reference: /root/reference/submodules/diff_lidargs_surfel_rasterization ("R2/"):
//   K1'  preprocessCUDA_cylinder   R2/cr/forward.cu:217-325      -> k_sf_preprocess
//   K2'  filter_preprocessCUDA     R2/cr/forward.cu:551-631      -> k_sf_preprocess<FILTER>
//   K7'  renderCUDA (forward)      R2/cr/forward.cu:327-547      -> k_sf_render_forward
//   K8'  renderCUDA (backward)     R2/cr/backward.cu:143-605     -> k_sf_render_backward
//   K10' preprocessCUDA (backward) R2/cr/backward.cu:607-749     -> k_sf_gaussian_backward
// Binning (range sort, load-balanced instance emit, stable tile bin, ranges) is shared with the 3D variant
// (binning.hip); so are the pixel mapping (one wave64 = 16 columns x 4 rows, per-lane row test against the
// reference rect, tile width 16 = BLOCK_X) and the ray tables.
//
// Per-surfel record, 80 bytes, written once by k_sf_preprocess and gathered per (tile, surfel) instance:
//   r0 = (Tu'.x, Tv'.x, Tu'.y, Tv'.y)      Tu' = Tu / (Tu.Tu), Tu = view-space first axis (incl. scale), Tv likewise: interleaved
//   r1 = (Tu'.z, Tv'.z, opacity, colour0)  by component, so that s = (dp.Tu', dp.Tv') needs no division per pair
//                                          (R2/cr/forward.cu:463-468) and both halves run as one packed-fp32 operation
//   r2 = (Tw.xyz,  colour1)     Tw = view-space centre
//   r3 = (n.xyz,   lambda)      n = normal flipped towards the sensor, lambda = Tw.n (distance of the plane)
//   r4 = (p_c, p_r, |Tw|, Tw.n) projected centre in pixels (2-D filter, :469), range, and the raw dot Tw.n (backward)
//
// Backward sums.  The reference accumulates up to 27 atomics per pair.  Here 23 per-(wave, surfel) sums go
// through one 32-slot butterfly reduce-scatter and ONE 32-lane atomic instruction into a packed 128-byte line;
// several of the reference's accumulators are linear (or abs-linear) in others with per-surfel coefficients and
// are reconstructed in the per-surfel epilogue instead (the 3-D branch's dL/dmean2D statistics from the |dL/dTw|
// sums; the 2-D branch's dL/dTw from its dL/dmean2D sums and one dL/dz sum).
#include "lidargs_common.h"
#include <string.h>

namespace lg {

#define SF_CHUNK 64
typedef float sf2 __attribute__((ext_vector_type(2)));    // a (Tu-part, Tv-part) pair: one packed-fp32 operation
#define SF_NEAR_N 0.2f
#define SF_FAR_N 80.0f

__device__ __forceinline__ float3 sf3(float x, float y, float z) { return make_float3(x, y, z); }
__device__ __forceinline__ float sdot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// (column, row) of a view-space point (R2/cr/forward.cu:118-174); returns false when the beam-fan cull fires
__device__ __forceinline__ bool sf_pix(float3 p, int W, int H, const float* __restrict__ beams, bool with_cull, float col_step, float2& pix) {
    const float pi_f = 3.14159265358979323846f;
    // atan2 through double = the correctly rounded fp32 value, independent of the device libm: p_c ~ 1e3 px enters the
    // 2-D filter exponent with a gain of ~80 per pixel, so one ulp of atan2f is a 1e-3 change of a blend weight
    const float p_c = (pi_f - (float)atan2((double)p.y, (double)p.x)) / col_step;
    const float alpha = (float)atan2((double)p.z, (double)sqrtf(p.x * p.x + p.y * p.y));
    int bi;
    if (alpha >= beams[H - 1]) bi = H - 1;
    else if (alpha <= beams[0]) bi = 0;
    else {
        int lo = 0, hi = H;
        while (lo < hi) { const int md = (lo + hi) >> 1; if (beams[md] < alpha) lo = md + 1; else hi = md; }
        bi = lo;
    }
    float p_r;
    const float guard = 0.006f;                                        // Ray_Divergence_Angle, R2/cr/forward.cu:18
    if (bi > 0) {
        const float before = beams[bi - 1], after = beams[bi];
        p_r = (float)(bi - 1) + (alpha - before) / (after - before);
        if (with_cull && alpha > (after + guard)) return false;
    } else {
        const float before = beams[0], after = beams[1];
        p_r = (float)(bi + 1) + (alpha - after) / (after - before);
        if (with_cull && alpha < (before - guard)) return false;
    }
    pix = make_float2(p_c, (float)H - p_r - 1.f);
    return true;
}

// columns of the rotation matrix of the NORMALISED quaternion (R2/cr/auxiliary.h:249-271)
__device__ __forceinline__ void sf_quat_cols(float4 q, float3& c0, float3& c1, float3& c2) {
    const float s = 1.0f / sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    const float w = q.x * s, x = q.y * s, y = q.z * s, z = q.w * s;
    c0 = sf3(1.f - 2.f * (y * y + z * z), 2.f * (x * y + w * z), 2.f * (x * z - w * y));
    c1 = sf3(2.f * (x * y - w * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z + w * x));
    c2 = sf3(2.f * (x * z + w * y), 2.f * (y * z - w * x), 1.f - 2.f * (x * x + y * y));
}
// world -> view rotation of a direction: transformVec4x3
__device__ __forceinline__ float3 sf_rot_view(const float* vm, float3 v) {
    return sf3(vm[0] * v.x + vm[4] * v.y + vm[8] * v.z, vm[1] * v.x + vm[5] * v.y + vm[9] * v.z, vm[2] * v.x + vm[6] * v.y + vm[10] * v.z);
}
// view -> world: transformVec4x3Transpose
__device__ __forceinline__ float3 sf_rot_world(const float* vm, float3 v) {
    return sf3(vm[0] * v.x + vm[1] * v.y + vm[2] * v.z, vm[4] * v.x + vm[5] * v.y + vm[6] * v.z, vm[8] * v.x + vm[9] * v.y + vm[10] * v.z);
}

struct SfPreArgs {
    int P, W, H, TH, tiles_x;
    float scale_modifier, near_f, far_f, col_step;
    const float* view;
    const float* means3D; const float* scales; const float* rotations; const float* opacities; const float* colors; const float* beams;
    const float* transMat; // transMat_precomp or nullptr: the rows (Tu, Tv, Tw) the BLEND uses (the rect, the normal, the sort depth and the
                           // pixel centre still come from scales / rotations / means3D: R2/cr/rasterizer_impl.cu:332 vs forward.cu:271-325)
    int* radii; int* radii_xy;
    float4* rec; uint32_t* rowspan; uint4* spans; uint32_t* dkey; uint32_t* ids;
    unsigned long long* inst_slots;     // [LG_INST_SLOTS][4]: word 0 of each slot = part of the instance total (zeroed by the caller)
    unsigned long long* diag_slots;     // [LG_INST_SLOTS][2]: visible surfels, reference tiles_touched (diagnostics)
    uint32_t* key_span;                 // [LG_INST_SLOTS][2]: ~(smallest), largest range key of the visible surfels (zeroed by the caller), as k_preprocess
    int compact;                        // 4-byte span records (compact_spans)
    uint8_t* touched;      // [P] "some pixel took this surfel" marks of the blend (cleared here; see preprocess.hip k_zero_touched)
    int prune;             // conservative footprint pruning of the binned rect (below); 0 = bin the whole reference rect (LIDARGS_NO_PRUNE=1)
};

// ---- conservative footprint pruning of the surfel rect -------------------------------------------------------------------------
// The reference bins every 16-column tile and every row of the rect (+-3 sigma axis end points, at least one pixel each way).  Inside
// it a pixel still SKIPS the surfel unless alpha = min(0.99, o G) >= 1/255 (R2/cr/forward.cu:484-490), i.e. unless rho <= 2 tau with
// tau = ln(255 o), where rho is rho3d (hit point of the pixel's ray in the splat frame, :463-468, front hits only) or rho2d (the 2-D
// filter, :469).  So a taking pixel
//   (3-D)  looks along a ray that hits the disc  D = { Tw + s_u Tu + s_v Tv : s_u^2 + s_v^2 <= 2 tau }  in front of the sensor: its
//          (azimuth, elevation) is that of a point of D;
//   (2-D)  or lies within |dx| <= sqrt(2 tau / 80), |dy| <= sqrt(2 tau / 200) of the projected centre: < 0.373 columns, < 0.236 rows.
// Bounds on the disc's azimuths and elevations from the sensor, with c = Tw, a = k Tu, b = k Tv (k = sqrt(2 tau)), e = c.xy / |c.xy|:
//   horizontal distance of a disc point  >= rmin = |c.xy| - hypot(a.e, b.e),   <= rmax = |c.xy| + sqrt(|a.xy|^2 + |b.xy|^2)
//   offset across the centre's azimuth   <= m = hypot(a x e, b x e)            =>  |d azimuth| <= atan(m / rmin) <= m / rmin
//   height z in [c.z - dz, c.z + dz], dz = hypot(a.z, b.z)                     =>  elevation in [atan2(zlo, zlo < 0 ? rmin : rmax),
//                                                                                                atan2(zhi, zhi > 0 ? rmin : rmax)]
// Rows are looked up in the beam table (a pixel row's elevation is its beam's, R2/cr/forward.cu:386).  Tiles and rows outside the hull of
// the two parts are dropped from the binned span; the blend's own row test keeps the reference rect.  Margins: k is inflated by 1 % + 0.02
// (the computed s differs from the exact one by ~1e-7 |Tw| / |Tu|, the hardware exp and log by < 0.01 in tau), the angles by 1e-4 rad /
// 0.02 columns (float atan2f, the centre's own pixel coordinates, a hit plane displaced by the rounding of lambda).  Not applied (whole
// rect binned) with transMat_precomp (the blend's rows are then not the rect's), when an axis is shorter than 1e-4 of the distance (s is
// then rounding noise) and when the disc comes closer than a quarter of the centre's horizontal distance (azimuth / elevation bounds
// degenerate towards the vertical axis).
struct SfPruned { int tx0, tx1, y0, y1; };
__device__ __forceinline__ SfPruned sf_prune(float3 c, float3 Tu, float3 Tv, float dist, float op, float2 pim, int xmin, int xmax, int ymin, int ymax,
                                             int W, int H, int gx, float col_step, const float* __restrict__ beams) {
    SfPruned o = {xmin, xmax, ymin, ymax};
    if (op * 255.f < 1.f) { o.tx1 = o.tx0; return o; }               // can never reach 1/255 (a NaN opacity passes: min(0.99, NaN) is 0.99 in the blend)
    const float uu = sdot(Tu, Tu), vv = sdot(Tv, Tv), d2 = dist * dist;
    const float rc = sqrtf(c.x * c.x + c.y * c.y);
    if (!(uu >= 1e-8f * d2 && vv >= 1e-8f * d2 && rc > 0.f)) return o;
    const float k = sqrtf(2.f * (__logf(255.f * op) + 0.02f)) * 1.01f + 0.02f;
    const float ex = c.x / rc, ey = c.y / rc;
    const float a_par = Tu.x * ex + Tu.y * ey, a_prp = Tu.y * ex - Tu.x * ey, b_par = Tv.x * ex + Tv.y * ey, b_prp = Tv.y * ex - Tv.x * ey;
    const float rmin = (rc - k * sqrtf(a_par * a_par + b_par * b_par) * 1.0001f) * 0.9999f;
    if (!(rmin >= 0.25f * rc)) return o;
    const float rmax = (rc + k * sqrtf(Tu.x * Tu.x + Tu.y * Tu.y + Tv.x * Tv.x + Tv.y * Tv.y)) * 1.0001f;
    const float m = k * sqrtf(a_prp * a_prp + b_prp * b_prp) * 1.0001f;
    // columns: pixel x is reachable iff |x - p_c + j W| <= dcol for a j in {-1, 0, 1} (the azimuth difference is the column difference
    // up to whole turns: a surfel next to the seam of the panorama is seen from its first AND its last columns, and the reference's
    // rect -- whose end points then project to the other side -- covers every tile column between them)
    const float dcol = fmaxf(m / rmin / col_step, 0.373f) + 0.02f;
    if (2.f * dcol + 32.f < (float)W) {
        const float lo = pim.x - dcol, hi = pim.x + dcol;
        int n = 0, f0 = 0, f1 = 0, l0 = 0, l1 = 0;                    // first and last non-empty interval, in ascending tile order
#pragma unroll
        for (int j = -1; j <= 1; j++) {
            const float sh = (float)j * (float)W;
            const int t0 = max(xmin, (int)floorf((lo + sh) / 16.f)), t1 = min(xmax, (int)floorf((hi + sh) / 16.f) + 1);
            if (t1 > t0) { if (n == 0) { f0 = t0; f1 = t1; } l0 = t0; l1 = t1; n++; }
        }
        if (n == 0) { o.tx1 = o.tx0; return o; }
        if (n == 1) { o.tx0 = f0; o.tx1 = f1; }
        else if (n == 2 && f0 == 0 && l1 == gx && f1 < l0) { o.tx0 = l0; o.tx1 = gx + f1; }    // across the seam: tile columns l0 .. gx - 1, 0 .. f1 - 1 (the emit wraps)
        else { o.tx0 = f0; o.tx1 = l1; }
    }
    // rows
    const float dz = k * sqrtf(Tu.z * Tu.z + Tv.z * Tv.z) * 1.0001f;
    const float zlo = c.z - dz, zhi = c.z + dz;
    const float e_lo = atan2f(zlo, zlo < 0.f ? rmin : rmax) - 1e-4f, e_hi = atan2f(zhi, zhi > 0.f ? rmin : rmax) + 1e-4f;
    int lo = 0, hi = H;                                                // first beam >= e_lo
    while (lo < hi) { const int md = (lo + hi) >> 1; if (beams[md] < e_lo) lo = md + 1; else hi = md; }
    const int b_lo = lo;
    hi = H;                                                            // first beam > e_hi
    while (lo < hi) { const int md = (lo + hi) >> 1; if (beams[md] <= e_hi) lo = md + 1; else hi = md; }
    const int b_hi = lo;
    int y0 = (int)ceilf(pim.y - 0.24f), y1 = (int)floorf(pim.y + 0.24f) + 1;         // the 2-D filter's rows [y0, y1)
    if (b_hi > b_lo) { y0 = min(y0, H - b_hi); y1 = max(y1, H - b_lo); }             // beams [b_lo, b_hi) = pixel rows [H - b_hi, H - b_lo)
    o.y0 = max(ymin, y0); o.y1 = min(ymax, y1);
    return o;
}

template <bool FILTER>
__global__ void __launch_bounds__(256) k_sf_preprocess(const SfPreArgs a) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    // the beam table is binary-searched five times per surfel: keep it in LDS when it fits (as k_preprocess does)
    constexpr int BEAMS_LDS = 1024;
    __shared__ float s_beams[BEAMS_LDS];
    const bool lds_beams = a.H <= BEAMS_LDS;
    if (lds_beams) {
        for (int q = threadIdx.x; q < a.H; q += blockDim.x) s_beams[q] = a.beams[q];
        __syncthreads();
    }
    const float* __restrict__ beams_tab = lds_beams ? s_beams : a.beams;
    const bool in_range = idx < a.P;                                   // no early return: the whole wave takes part in the sum at the end
    int out_radius = 0, rx = 0, ry = 0;
    uint32_t key = 0xFFFFFFFFu, tiles = 0, reftiles = 0, rspan = 0, bspan = 0, xsp = 0;
    float4 r0, r1, r2, r3, r4;
    bool live = false;
    do {
        if (!in_range) break;
        const float* vm = a.view;
        const float3 pw = sf3(a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]);
        const float3 pv = sf3(vm[0] * pw.x + vm[4] * pw.y + vm[8] * pw.z + vm[12], vm[1] * pw.x + vm[5] * pw.y + vm[9] * pw.z + vm[13],
                              vm[2] * pw.x + vm[6] * pw.y + vm[10] * pw.z + vm[14]);
        const float dist = sqrtf(pv.x * pv.x + pv.y * pv.y + pv.z * pv.z);
        if (dist >= a.far_f || dist <= a.near_f) break;
        float2 pim;
        if (!sf_pix(pv, a.W, a.H, beams_tab, true, a.col_step, pim)) break;

        // every other input of a surviving surfel is requested here, together (the opacity and the colours are needed at the very end, where a
        // load issued on the spot would be waited for; K2 has neither)
        const float op_in = FILTER ? 0.f : a.opacities[idx], col0_in = FILTER ? 0.f : a.colors[2 * idx], col1_in = FILTER ? 0.f : a.colors[2 * idx + 1];
        const float4 q = make_float4(a.rotations[4 * idx], a.rotations[4 * idx + 1], a.rotations[4 * idx + 2], a.rotations[4 * idx + 3]);
        float3 c0, c1, c2;
        sf_quat_cols(q, c0, c1, c2);
        const float sx = a.scale_modifier * a.scales[2 * idx], sy = a.scale_modifier * a.scales[2 * idx + 1];
        // view-space axes of the surfel: rows of T = transpose(splat2world) * world2view (R2/cr/forward.cu:271-295)
        const float3 Tu = sf_rot_view(vm, sf3(c0.x * sx, c0.y * sx, c0.z * sx));
        const float3 Tv = sf_rot_view(vm, sf3(c1.x * sy, c1.y * sy, c1.z * sy));
        float3 n = sf_rot_view(vm, c2);
        if (!FILTER) {
            const float c = -(pv.x * n.x + pv.y * n.y + pv.z * n.z);  // DUAL_VISIABLE, :297-302
            if (c == 0.f) break;
            if (!(c > 0.f)) { n.x = -n.x; n.y = -n.y; n.z = -n.z; }
        }
        // extent: +-3 sigma axis end points through the beam model, at least one pixel (:177-215)
        float2 e0, e1, e2, e3;
        sf_pix(sf3(pv.x + 3.f * Tu.x, pv.y + 3.f * Tu.y, pv.z + 3.f * Tu.z), a.W, a.H, beams_tab, false, a.col_step, e0);
        sf_pix(sf3(pv.x - 3.f * Tu.x, pv.y - 3.f * Tu.y, pv.z - 3.f * Tu.z), a.W, a.H, beams_tab, false, a.col_step, e1);
        sf_pix(sf3(pv.x + 3.f * Tv.x, pv.y + 3.f * Tv.y, pv.z + 3.f * Tv.z), a.W, a.H, beams_tab, false, a.col_step, e2);
        sf_pix(sf3(pv.x - 3.f * Tv.x, pv.y - 3.f * Tv.y, pv.z - 3.f * Tv.z), a.W, a.H, beams_tab, false, a.col_step, e3);
        const float ax = fmaxf(fabsf(e0.x - pim.x), fabsf(e1.x - pim.x)), ay = fmaxf(fabsf(e0.y - pim.y), fabsf(e1.y - pim.y));
        const float bx = fmaxf(fabsf(e2.x - pim.x), fabsf(e3.x - pim.x)), by = fmaxf(fabsf(e2.y - pim.y), fabsf(e3.y - pim.y));
        rx = (int)ceilf(fmaxf(fmaxf(ax, bx), 1.0f));
        ry = (int)ceilf(fmaxf(fmaxf(ay, by), 1.0f));
        // R2/cr/auxiliary.h:99-112: x and ymin truncate, ymax = round(p.y + ry)
        const int gx = a.tiles_x, gy = a.H;
        int xmin, ymin, xmax, ymax;
        rect_surfel(pim.x, pim.y, rx, ry, gx, gy, xmin, ymin, xmax, ymax);
        if ((xmax - xmin) * (ymax - ymin) == 0) break;
        live = true;
        out_radius = max(rx, ry);
        if (FILTER) break;

        reftiles = (uint32_t)((xmax - xmin) * (ymax - ymin));
        rspan = (uint32_t)ymin | ((uint32_t)ymax << 16);              // the blend's row test: the reference rect
        SfPruned pr = {xmin, xmax, ymin, ymax};
        if (a.prune && !a.transMat) pr = sf_prune(pv, Tu, Tv, dist, op_in, pim, xmin, xmax, ymin, ymax, a.W, a.H, gx, a.col_step, beams_tab);
        if (pr.tx1 > pr.tx0 && pr.y1 > pr.y0) {
            const int ty0 = pr.y0 / a.TH, ty1 = (pr.y1 - 1) / a.TH;
            tiles = (uint32_t)((pr.tx1 - pr.tx0) * (ty1 - ty0 + 1));
            bspan = (uint32_t)pr.y0 | ((uint32_t)pr.y1 << 16);
            xsp = (uint32_t)pr.tx0 | ((uint32_t)pr.tx1 << 16);
        }
        if (tiles) key = __float_as_uint(dist);                       // binned nowhere: sorted with the culled ones (no instances either way)
        float3 Bu = Tu, Bv = Tv, Bw = pv;                                 // the blend's rows
        float bw_len = dist;
        if (a.transMat) {
            const float* t = a.transMat + 9 * (size_t)idx;
            Bu = sf3(t[0], t[1], t[2]); Bv = sf3(t[3], t[4], t[5]); Bw = sf3(t[6], t[7], t[8]);
            bw_len = sqrtf(Bw.x * Bw.x + Bw.y * Bw.y + Bw.z * Bw.z);
        }
        const float uu = sdot(Bu, Bu), vv = sdot(Bv, Bv);
        const float iu = uu > 0.f ? 1.f / uu : 0.f, iv = vv > 0.f ? 1.f / vv : 0.f;
        // (Tu', Tv') interleaved by component: the blend evaluates s = (dp.Tu', dp.Tv') and its gradients as packed-fp32 pairs
        r0 = make_float4(Bu.x * iu, Bv.x * iv, Bu.y * iu, Bv.y * iv);
        r1 = make_float4(Bu.z * iu, Bv.z * iv, op_in, col0_in);
        r2 = make_float4(Bw.x, Bw.y, Bw.z, col1_in);
        // lambda = |Tw| * cos(phi1) with cos(phi1) = (Tw.n)/|Tw|, rounded in the reference's order (:449-452): the hit
        // point lam2 * p - Tw cancels ~3 digits, so a 1-ulp change of lambda is a 1e-4 change of the Gaussian weight
        const float wn = Bw.x * n.x + Bw.y * n.y + Bw.z * n.z;
        r3 = make_float4(n.x, n.y, n.z, bw_len * (wn / bw_len));
        r4 = make_float4(pim.x, pim.y, bw_len, wn);
    } while (false);

    if (in_range) {
        a.radii[idx] = out_radius;
        a.radii_xy[2 * idx] = live ? rx : 0; a.radii_xy[2 * idx + 1] = live ? ry : 0;
    }
    if (FILTER) return;
    {   // the instance total (what the host sizes the binning buffer with), known before anything is sorted: one sum per wave, added
        // to one of LG_INST_SLOTS slots (k_preprocess has the same, per tile height); the host adds the slots up after its one read
        uint32_t sum = tiles, sv = reftiles ? 1u : 0u, sr = reftiles;       // (+ the diagnostics of lidargs_last_counters)
        // ~(smallest) and largest range key of the wave's visible surfels (0xFFFFFFFF: culled -> neutral for both maxima): the range sort
        // runs on key - kmin and skips the passes the frame's key span does not need (api.hip, surfel_api.inc)
        uint32_t kinv = ~key, kmx = (key == 0xFFFFFFFFu) ? 0u : key;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sum += __shfl_xor(sum, o); sv += __shfl_xor(sv, o); sr += __shfl_xor(sr, o);
            kinv = max(kinv, (uint32_t)__shfl_xor((int)kinv, o)); kmx = max(kmx, (uint32_t)__shfl_xor((int)kmx, o));
        }
        // one set of atomics per BLOCK, not per wave (k_preprocess does the same): 31 k waves x 5 atomics on a few cache lines serialise
        // at ~11 ns each on a line -- two more per wave for the key span made this launch 141 -> 205 us before they met in LDS first
        __shared__ uint32_t s_part[4][5];
        if ((threadIdx.x & 63) == 0) {
            uint32_t* q = s_part[threadIdx.x >> 6];
            q[0] = sum; q[1] = sv; q[2] = sr; q[3] = kinv; q[4] = kmx;
        }
        __syncthreads();
        if (threadIdx.x < 5) {
            const int c = threadIdx.x;
            const size_t slot = (size_t)(blockIdx.x % LG_INST_SLOTS);
            if (c < 3) {
                const uint32_t v = s_part[0][c] + s_part[1][c] + s_part[2][c] + s_part[3][c];
                if (v) {
                    if (c == 0) atomicAdd(a.inst_slots + slot * 4, (unsigned long long)v);
                    else atomicAdd(a.diag_slots + slot * 2 + (c - 1), (unsigned long long)v);
                }
            } else if (a.key_span) {
                const uint32_t m = max(max(s_part[0][c], s_part[1][c]), max(s_part[2][c], s_part[3][c]));
                uint32_t* ks = a.key_span + 2 * slot + (c - 3);
                if (m) atomicMax(ks, m);
            }
        }
    }
    // The packed 128-byte gradient lines the backward adds into: rounds 2-4 zeroed all P of them here (256 MB at 2 M surfels).  Only the
    // surfels some pixel's walk takes are ever added to: the blend marks them (one byte each, cleared below), the backward's first
    // launch clears exactly their lines (preprocess.hip k_zero_touched).
    if (!in_range) return;
    a.touched[idx] = 0;
    a.dkey[idx] = key;                                                 // (the ids of the range sort are the positions: not written)
    // lidargs_common.h: one gather per Gaussian when the lists are built (4-byte records when the image allows)
    if (a.compact) reinterpret_cast<uint32_t*>(a.spans)[idx] = span_pack(bspan, tiles ? xsp : 0u);
    else a.spans[idx] = make_uint4(bspan, tiles ? xsp : 0u, 0u, 0u);
    if (live && tiles) {                                               // (only binned surfels' records are ever gathered: 39 % of cfg5's visible ones are hit by no beam)
        a.rowspan[idx] = rspan;
        float4* r = a.rec + 5 * (size_t)idx;
        r[0] = r0; r[1] = r1; r[2] = r2; r[3] = r3; r[4] = r4;
    }
}

void launch_sf_preprocess(const SfPreArgs& a, bool filter_only, hipStream_t s) {
    const dim3 grid((a.P + 255) / 256), block(256);
    if (filter_only) hipLaunchKernelGGL(k_sf_preprocess<true>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(k_sf_preprocess<false>, grid, block, 0, s, a);
}

// ------------------------------------------------------------------------------------------------
struct SfPixel { int x, y, pix; bool inside; float3 p; };

__device__ __forceinline__ SfPixel sf_pixel(const TileGrid& g, const float2* __restrict__ coltab, const float2* __restrict__ rowtab, int patch, int lane) {
    SfPixel o;
    const int wpt = g.waves_per_tile;
    const int tile = patch / wpt, sub = patch - tile * wpt;
    o.x = (tile % g.tiles_x) * LG_TILE_W + (lane & 15);
    o.y = (tile / g.tiles_x) * g.TH + sub * LG_WAVE_ROWS + (lane >> 4);
    o.inside = o.x < g.W && o.y < g.H;
    o.pix = o.y * g.W + o.x;
    o.p = sf3(0.f, 0.f, 1.f);
    if (o.inside) { const float2 cb = coltab[o.x], ca = rowtab[o.y]; o.p = sf3(ca.x * cb.x, ca.x * cb.y, ca.y); }
    return o;
}

struct SfStaged { float4 a0, a1, a2, a3, a4; uint32_t span, gid; };

__device__ __forceinline__ SfStaged sf_gather(const uint32_t* __restrict__ point_list, const float4* __restrict__ rec,
                                              const uint32_t* __restrict__ rowspan, uint32_t k, bool valid) {
    SfStaged s;
    s.a0 = s.a1 = s.a2 = s.a3 = s.a4 = make_float4(0.f, 0.f, 0.f, 0.f);
    s.span = 0; s.gid = 0;
    if (valid) {
        const uint32_t g = point_list[k];
        const float4* r = rec + 5 * (size_t)g;
        s.a0 = r[0]; s.a1 = r[1]; s.a2 = r[2]; s.a3 = r[3]; s.a4 = r[4];
        s.span = rowspan[g]; s.gid = g;
    }
    return s;
}

// Everything up to alpha for one (pixel, surfel) pair: R2/cr/forward.cu:426-490 == R2/cr/backward.cu:283-340.
// `op` is the surfel's opacity on this pixel's row, 0 outside its row span (rows_opacity): alpha = 0 there fails the 1/255 test,
// which is the row test of the rect without two compares per pair.  `gate` (the lane still walks | the entry is not behind the
// pixel's last contributor) and the reference's other skips (:459 cos2 == 0, :476 depth < near, :483 power > 0) go into the
// exponent the same way -- exp(-inf) = 0 -- so that `ok` is ONE compare whose lane mask is the ballot itself.
struct SfPair { bool ok, in3d; float sx, sy, dxp, dyp, lam2, cos2, depth, G, alpha; float3 dp; };

// ALPHA_ONLY (the T-only walk, which uses nothing but ok / alpha): no finite stand-in for a zero cos2 -- the pair fails the cos2 test
// whatever its other fields become.
template <bool ALPHA_ONLY = false>
__device__ __forceinline__ SfPair sf_pair(const SfPixel& px, float4 r0, float4 r1, float4 r2, float4 r3, float4 r4, float op, bool gate) {
    SfPair o;
    const float3 n = sf3(r3.x, r3.y, r3.z);
    o.cos2 = sdot(px.p, n);
    const float safe = (ALPHA_ONLY || o.cos2 != 0.f) ? o.cos2 : 1.f;
    // ray / plane hit distance (:449-457): one division per (pixel, surfel) pair of every walk.  The IEEE sequence is ten instructions;
    // reciprocal + one residual correction is four and differs from it in the last bit only on a small fraction of operands
    {
        const float rc = __builtin_amdgcn_rcpf(safe);
        const float q0 = r3.w * rc;
        o.lam2 = __builtin_fmaf(__builtin_fmaf(-safe, q0, r3.w), rc, q0);
    }
    o.dp = sf3(o.lam2 * px.p.x - r2.x, o.lam2 * px.p.y - r2.y, o.lam2 * px.p.z - r2.z);
    sf2 sxy;
    {   // the hit point in the splat's frame: three products of like size per component, summed as fused multiply-adds (the offset dp
        // itself, where three digits cancel, is formed above exactly as the reference writes it)
#pragma clang fp contract(fast)
        sxy = o.dp.x * sf2{r0.x, r0.y} + o.dp.y * sf2{r0.z, r0.w} + o.dp.z * sf2{r1.x, r1.y};
    }
    o.sx = sxy.x; o.sy = sxy.y;
    // (the two squared distances as fused multiply-adds: a last-bit difference in a quantity that is compared and exponentiated, not
    //  differenced -- three instructions less per pair in walks that are bound by the vector pipe)
    const float rho3d = __builtin_fmaf(o.sx, o.sx, o.sy * o.sy);
    o.dxp = r4.x - (float)px.x; o.dyp = r4.y - (float)px.y;
    const float rho2d = __builtin_fmaf(80.f * o.dxp, o.dxp, (200.f * o.dyp) * o.dyp);   // FilterInvSquare * (40 dx^2 + 100 dy^2), :469
    const bool front = o.lam2 > 0.f;
    o.in3d = front && (rho3d <= rho2d);
    const float rho = o.in3d ? rho3d : rho2d;                          // = front ? fminf(rho3d, rho2d) : rho2d, on the compare in3d needs anyway
    o.depth = o.in3d ? o.lam2 : r4.z;
    const float power = -0.5f * rho;
    // (:483's `power > 0` skip cannot fire: rho is a sum of squares, and a NaN power is not > 0 either -- the test is dropped)
    const bool pass = gate && (o.cos2 != 0.f) && !(o.depth < SF_NEAR_N);
    o.G = __expf(pass ? power : -INFINITY);
    o.alpha = fminf(0.99f, op * o.G);
    o.ok = o.alpha >= 1.0f / 255.0f;
    return o;
}

// Per-(patch, segment) planes of the surfel blend (each 64 floats).  Lists are cut into segments exactly as in the 3-D
// variant (render.hip): pass 1 walks every segment from T = 1 for its transmittance product and the contribution flags,
// pass 2 walks the flagged entries from the true T_in, the combine folds the segments in order.
enum { SF_SEG_TPASS = 0, SF_SEG_C0, SF_SEG_C1, SF_SEG_D, SF_SEG_N0, SF_SEG_N1, SF_SEG_N2, SF_SEG_M1, SF_SEG_M2, SF_SEG_DIST,
       SF_SEG_TEND, SF_SEG_TBREAK, SF_SEG_LAST, SF_SEG_MEDC, SF_SEG_MED, SF_SEG_PLANES = 16 };

struct SfFwdArgs {
    TileGrid grid;
    const uint2* ranges; const uint32_t* point_list; const float4* rec; const uint32_t* rowspan;
    const float2* coltab; const float2* rowtab; const float* bg;
    float* accum;          // [3N] final_T, M1, M2
    uint32_t* n_contrib;   // [2N] last contributor, median contributor (1-based positions in the tile list)
    float* out_color; float* out_others;
    float* seg; int S; int seg_len;     // [patches][S][SF_SEG_PLANES][64]
    uint8_t* flags; size_t R;           // [waves_per_tile][R]
    uint8_t* touched;                   // [P] set to 1 for the surfel of every entry whose flag is set
    uint8_t* alive;                     // [patches] segments pass 1 walked (255 = all); nullptr = no gating
    int seg_lo, seg_hi, front;
};

// The T-only walk over entries [0, cnt) of a chunk, second form (render.hip walk_T_only_v2, which this follows): alpha of an entry does
// not depend on the entries before it, only T does, through one multiply; entries in groups of four at constant LDS offsets (the lanes
// behind the chunk's count parked zero records: opacity 0, alpha 0); the lanes that are out only gate the contribution flags, refreshed
// once per group; a tripped lane keeps multiplying (any hand-over value below 1e-4 is as good as another).  `took` comes back with bit
// e set if some live pixel takes entry e.
template <int G>
__device__ __forceinline__ void sf_walk_T_only_v2(const int cnt, const float4* s_rec, const float* oprow, const SfPixel& px, float& T_io, bool& done_io,
                                                  const unsigned long long dead0, unsigned long long& took) {
    struct Rec { float4 r0, r1, r2, r3, r4; float op; };
    auto read = [&](int jj) {
        Rec r;
        const float* f1 = reinterpret_cast<const float*>(&s_rec[SF_CHUNK + jj]);
        const float* f2 = reinterpret_cast<const float*>(&s_rec[2 * SF_CHUNK + jj]);
        const float* f4 = reinterpret_cast<const float*>(&s_rec[4 * SF_CHUNK + jj]);
        r.r0 = lds_ahead(&s_rec[jj]);
        const sf2 tz = *(LG_LDS_VOLATILE(sf2))f1;
        r.r3 = lds_ahead(&s_rec[3 * SF_CHUNK + jj]);
        const v3f c4 = *(LG_LDS_VOLATILE(v3f))f4;
        r.r4 = make_float4(c4.x, c4.y, c4.z, 0.f);
        const v3f tw = *(LG_LDS_VOLATILE(v3f))f2;
        r.r1 = make_float4(tz.x, tz.y, 0.f, 0.f);
        r.r2 = make_float4(tw.x, tw.y, tw.z, 0.f);
        r.op = lds_ahead(&oprow[4 * jj]);
        return r;
    };
    auto factor = [&](const Rec& r, unsigned long long& hitmask) {
        const SfPair q = sf_pair<true>(px, r.r0, r.r1, r.r2, r.r3, r.r4, r.op, true);
        hitmask = __ballot(q.ok);
        return q.ok ? 1.f - q.alpha : 1.f;
    };
    float T = T_io;
    unsigned long long dead = dead0 | __ballot(T < 0.0001f);
    uint32_t acc_lo = 0u, acc_hi = 0u;                                 // took bits, newest entry at bit 0 (render.hip walk_T_only_v2)
    auto note = [&](unsigned long long hitmask) {
        unsigned long long tmp;
        asm volatile("s_andn2_b64 %2, %3, %4\n\ts_addc_u32 %0, %0, %0\n\ts_addc_u32 %1, %1, %1"
                     : "+s"(acc_lo), "+s"(acc_hi), "=&s"(tmp) : "s"(hitmask), "s"(dead) : "scc");
    };
    static_assert(G == 2 || G == 4, "group of 2 or 4 entries");
    const int ng = (cnt + G - 1) / G;
    Rec ra = read(0), rb = read(1);
    for (int g = 0; g < ng; g++) {
        const int j = G * g;
        unsigned long long h[G]; float f[G];
#pragma unroll
        for (int i = 0; i < G; i += 2) {
            f[i] = factor(ra, h[i]); ra = read(j + i + 2);
            f[i + 1] = factor(rb, h[i + 1]); rb = read(j + i + 3);
        }
#pragma unroll
        for (int i = 0; i < G; i++) note(h[i]);
#pragma unroll
        for (int i = 0; i < G; i++) T = T * f[i];
        dead = dead0 | __ballot(T < 0.0001f);
    }
    T_io = T;
    done_io = done_io || (T < 0.0001f);
    const unsigned long long acc = ((unsigned long long)acc_hi << 32) | acc_lo;
    took = __brevll(acc) >> (64 - G * ng);
}

// One workgroup = (patch, segment).  T_ONLY: pass 1 (transmittance product + flags).  Otherwise pass 2 (all sums).
template <bool T_ONLY, bool V2 = false>
#ifdef LG_SF_FWD_WAVES   /* experiment (tools/waves_ab.sh) */
__attribute__((amdgpu_waves_per_eu(LG_SF_FWD_WAVES, LG_SF_FWD_WAVES)))
#endif
__global__ void __launch_bounds__(64) k_sf_render_forward(const SfFwdArgs a) {
    __shared__ float4 s_rec[5 * SF_CHUNK + 2];                         // (+2: the second form's look-ahead reads)
    __shared__ float4 s_oprow[SF_CHUNK + 2];                           // opacity per pixel row of the patch, 0 outside the surfel's row span
    const int lane = threadIdx.x;
    const int S = a.S;
    const int wpt = a.grid.waves_per_tile;
    int patch, seg;
    if (!block_patch_segment(blockIdx.x, a.grid.num_tiles() * wpt, a.seg_hi - a.seg_lo, patch, seg)) return;
    seg += a.seg_lo;
    const int tile = patch / wpt, sub = patch - tile * wpt;
    const uint2 tr = a.ranges[tile];
    const int St = segment_count(tr, S, a.seg_len);
    if (seg >= St) return;
    if (a.alive && seg >= (int)a.alive[patch]) return;
    const SfPixel px = sf_pixel(a.grid, a.coltab, a.rowtab, patch, lane);
    const uint2 sr = segment_range(tr, St, seg);
    const uint32_t n = sr.y - sr.x;
    float* segbase = a.seg + ((size_t)patch * S + seg) * (SF_SEG_PLANES * 64);

    float T = 1.f;
    if (!T_ONLY) {
        const float* tp = a.seg + (size_t)patch * S * (SF_SEG_PLANES * 64) + SF_SEG_TPASS * 64 + lane;
        T = plane_product(T, tp, SF_SEG_PLANES * 64, seg);
    }
    float T_break = T;
    float C0 = 0.f, C1 = 0.f, D = 0.f, M1 = 0.f, M2 = 0.f, dist = 0.f, med = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f;
    uint32_t last = 0, med_c = 0;
    bool done = !px.inside || (!T_ONLY && T < 0.0001f);
    uint8_t* fl = a.flags + (size_t)sub * a.R + sr.x;
    const int y0 = (tile / a.grid.tiles_x) * a.grid.TH + sub * LG_WAVE_ROWS;       // first pixel row of the patch
    const float* oprow = reinterpret_cast<const float*>(s_oprow) + (lane >> 4);
    const uint32_t nchunks = (n + SF_CHUNK - 1) / SF_CHUNK;
    uint32_t c_done = 0;
    if (__ballot(!done) != 0ull && n > 0) {
        auto entry_valid = [&](uint32_t k) { return k < n && (T_ONLY || fl[k] != 0); };
        bool have = entry_valid(lane);
        SfStaged st = sf_gather(a.point_list, a.rec, a.rowspan, sr.x + lane, have);
        for (uint32_t c = 0; c < nchunks; c++) {
            const uint32_t g_mine = st.gid;                            // entry c * 64 + lane's surfel (pass 1 marks it as touched with its flag)
            __syncthreads();
            s_rec[lane] = st.a0; s_rec[SF_CHUNK + lane] = st.a1; s_rec[2 * SF_CHUNK + lane] = st.a2; s_rec[3 * SF_CHUNK + lane] = st.a3;
            s_rec[4 * SF_CHUNK + lane] = st.a4; s_oprow[lane] = rows_opacity(st.span, st.a1.z, y0);
            unsigned long long todo = __ballot(have);
            __syncthreads();
            if (c + 1 < nchunks) { const uint32_t k = (c + 1) * SF_CHUNK + lane; have = entry_valid(k); st = sf_gather(a.point_list, a.rec, a.rowspan, sr.x + k, have); }
            if (__ballot(!done) == 0ull) break;
            unsigned long long took = 0ull;
            if (todo && T_ONLY && V2) {
                sf_walk_T_only_v2<4>(__builtin_popcountll(todo), s_rec, oprow, px, T, done, __ballot(!px.inside), took);
            } else if (todo) {
                // Two register sets used in turn, look-ahead reads that stay where they are written, only the fields the pass uses:
                // see walk_flagged (render.hip), which this follows.
                struct Rec { float4 r0, r1, r2, r3, r4; float op; };
                auto read = [&](int jj) {
                    Rec r;
                    const float* f1 = reinterpret_cast<const float*>(&s_rec[SF_CHUNK + jj]);
                    const float* f2 = reinterpret_cast<const float*>(&s_rec[2 * SF_CHUNK + jj]);
                    const float* f4 = reinterpret_cast<const float*>(&s_rec[4 * SF_CHUNK + jj]);
                    r.r0 = lds_ahead(&s_rec[jj]);
                    const sf2 tz = *(LG_LDS_VOLATILE(sf2))f1;
                    r.r3 = lds_ahead(&s_rec[3 * SF_CHUNK + jj]);
                    const v3f c4 = *(LG_LDS_VOLATILE(v3f))f4;
                    r.r4 = make_float4(c4.x, c4.y, c4.z, 0.f);
                    if (T_ONLY) {
                        const v3f tw = *(LG_LDS_VOLATILE(v3f))f2;
                        r.r1 = make_float4(tz.x, tz.y, 0.f, 0.f);
                        r.r2 = make_float4(tw.x, tw.y, tw.z, 0.f);
                    } else {
                        r.r1 = make_float4(tz.x, tz.y, 0.f, lds_ahead(f1 + 3));
                        r.r2 = lds_ahead(&s_rec[2 * SF_CHUNK + jj]);
                    }
                    r.op = lds_ahead(&oprow[4 * jj]);
                    return r;
                };
                auto evaluate = [&](const Rec& r, int j) {
                    const SfPair q = sf_pair(px, r.r0, r.r1, r.r2, r.r3, r.r4, r.op, !done);
                    const bool hit = q.ok;
                    const float test_T = T * (1.f - q.alpha);
                    const bool trip = hit && test_T < 0.0001f;
                    if (T_ONLY) {
                        // only the hand-over value is kept: T takes the tripping value too, and the lane is done from there on
                        T = hit ? test_T : T;
                        took |= (__ballot(hit) != 0ull) ? (1ull << j) : 0ull;
                    } else {
                        const bool blend = hit != trip;
                        const float w = blend ? q.alpha * T : 0.f;
                        // (this walk is bound by the vector pipe: the mapped depth's division as reciprocal + residual correction, like
                        //  sf_pair's; m forced to 0 where nothing blends -- a depth of 0 would make it infinite --, after which every sum
                        //  carries w = 0 there and needs no select of its own; the weighted sums as fused multiply-adds)
                        const float rcd = __builtin_amdgcn_rcpf(q.depth), qd = SF_NEAR_N * rcd;
                        const float nd = __builtin_fmaf(__builtin_fmaf(-q.depth, qd, SF_NEAR_N), rcd, qd);     // SF_NEAR_N / depth
                        const float m_ = SF_FAR_N / (SF_FAR_N - SF_NEAR_N) * (1.f - nd);
                        const float m = blend ? m_ : 0.f;
                        // distortion with the segment-local prefix sums; the combine adds the terms in the sums of the segments
                        // in front (the expression is linear in them), R2/cr/forward.cu:497-499
                        dist += (m * m * (1.f - T) + M2 - 2.f * m * M1) * w;
                        D = __builtin_fmaf(q.depth, w, D);
                        M1 = __builtin_fmaf(m, w, M1);
                        M2 = __builtin_fmaf(m * m, w, M2);
                        const bool is_med = blend && T > 0.5f;                             // :503-507
                        med = is_med ? q.depth : med;
                        med_c = is_med ? (sr.x - tr.x + c * SF_CHUNK + (uint32_t)j + 1u) : med_c;
                        N0 = __builtin_fmaf(r.r3.x, w, N0); N1 = __builtin_fmaf(r.r3.y, w, N1); N2 = __builtin_fmaf(r.r3.z, w, N2);
                        C0 = __builtin_fmaf(r.r1.w, w, C0); C1 = __builtin_fmaf(r.r2.w, w, C1);
                        T = blend ? test_T : T;
                        T_break = hit ? test_T : T_break;
                        last = blend ? (c * SF_CHUNK + (uint32_t)j + 1u) : last;
                    }
                    done = done || trip;
                };
                int ja = __builtin_ctzll(todo);
                todo &= todo - 1ull;
                Rec ra = read(ja), rb;
                while (true) {
                    const bool more_b = todo != 0ull;
                    const int jb = more_b ? __builtin_ctzll(todo) : ja;
                    todo &= todo - 1ull;
                    rb = read(jb);
                    evaluate(ra, ja);
                    if (!more_b) break;
                    const bool more_a = todo != 0ull;
                    ja = more_a ? __builtin_ctzll(todo) : jb;
                    todo &= todo - 1ull;
                    ra = read(ja);
                    evaluate(rb, jb);
                    if (!more_a) break;
                }
            }
            if (T_ONLY) {
                const uint32_t k = c * SF_CHUNK + lane;
                const bool tk = (took >> lane) & 1ull;
                if (k < n) fl[k] = (uint8_t)tk;
                if (tk && k < n) a.touched[g_mine] = 1;                 // (same value from every patch that takes it: a plain byte store)
                c_done = c + 1;
            }
        }
    }
    if (T_ONLY) {
        for (uint32_t c = c_done; c < nchunks; c++) { const uint32_t k = c * SF_CHUNK + lane; if (k < n) fl[k] = 0; }
        segbase[SF_SEG_TPASS * 64 + lane] = T;
    } else {
        segbase[SF_SEG_C0 * 64 + lane] = C0; segbase[SF_SEG_C1 * 64 + lane] = C1; segbase[SF_SEG_D * 64 + lane] = D;
        segbase[SF_SEG_N0 * 64 + lane] = N0; segbase[SF_SEG_N1 * 64 + lane] = N1; segbase[SF_SEG_N2 * 64 + lane] = N2;
        segbase[SF_SEG_M1 * 64 + lane] = M1; segbase[SF_SEG_M2 * 64 + lane] = M2; segbase[SF_SEG_DIST * 64 + lane] = dist;
        segbase[SF_SEG_TEND * 64 + lane] = T; segbase[SF_SEG_TBREAK * 64 + lane] = T_break;
        reinterpret_cast<uint32_t*>(segbase)[SF_SEG_LAST * 64 + lane] = last;
        reinterpret_cast<uint32_t*>(segbase)[SF_SEG_MEDC * 64 + lane] = med_c;
        segbase[SF_SEG_MED * 64 + lane] = med;
    }
}

// After a pass-1 round ending at segment `front`: patches with an unsaturated pixel and more list stay open (255).
__global__ void __launch_bounds__(64) k_sf_alive(const SfFwdArgs a) {
    const int lane = threadIdx.x;
    const int patch = blockIdx.x;
    if (a.seg_lo > 0 && a.alive[patch] != 255) return;
    const int wpt = a.grid.waves_per_tile;
    const int tile = patch / wpt;
    const SfPixel px = sf_pixel(a.grid, a.coltab, a.rowtab, patch, lane);
    const int St = segment_count(a.ranges[tile], a.S, a.seg_len);
    const float* sb = a.seg + (size_t)patch * a.S * (SF_SEG_PLANES * 64) + lane;
    float T = 1.f;
    const int kf = min(a.front, St);
    for (int k = 0; k < kf && T >= 0.0001f; k++) T *= sb[(size_t)k * (SF_SEG_PLANES * 64) + SF_SEG_TPASS * 64];
    const unsigned long long open = __ballot(px.inside && T >= 0.0001f);
    if (lane == 0) a.alive[patch] = (St > a.front && open != 0ull) ? 255 : (uint8_t)min(a.front, 254);
}

// Per patch: fold the segments in order into the image planes.
__global__ void __launch_bounds__(64) k_sf_combine(const SfFwdArgs a) {
    const int lane = threadIdx.x;
    const int patch = blockIdx.x;
    const int wpt = a.grid.waves_per_tile;
    const int tile = patch / wpt;
    const SfPixel px = sf_pixel(a.grid, a.coltab, a.rowtab, patch, lane);
    if (!px.inside) return;
    const uint2 tr = a.ranges[tile];
    int St = segment_count(tr, a.S, a.seg_len);
    const int Sfull = St;
    if (a.alive) St = min(St, (int)a.alive[patch]);
    const float* sb = a.seg + (size_t)patch * a.S * (SF_SEG_PLANES * 64) + lane;
    const uint32_t* su = reinterpret_cast<const uint32_t*>(sb);
    const size_t stride = SF_SEG_PLANES * 64;
    float C0 = 0.f, C1 = 0.f, D = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f, M1 = 0.f, M2 = 0.f, dist = 0.f, med = 0.f;
    float T_final = 1.f, T_start = 1.f;
    uint32_t last = 0, med_c = 0;
    bool stopped = false;
    // four segments per step (two until round 5: 34 -> see EXPERIMENTS): their sixty loads are issued together (every plane below St was written by pass 2), and a pixel whose
    // walk has ended simply stops taking them; the patch leaves once all of its pixels have
    struct Seg { float tend, m1, m2, dist, c0, c1, d, n0, n1, n2, med, tpass, tbreak; uint32_t l, mc; };
    auto load = [&](int k) {
        Seg g;
        const float* p = sb + (size_t)k * stride;
        g.tend = p[SF_SEG_TEND * 64]; g.m1 = p[SF_SEG_M1 * 64]; g.m2 = p[SF_SEG_M2 * 64]; g.dist = p[SF_SEG_DIST * 64];
        g.c0 = p[SF_SEG_C0 * 64]; g.c1 = p[SF_SEG_C1 * 64]; g.d = p[SF_SEG_D * 64];
        g.n0 = p[SF_SEG_N0 * 64]; g.n1 = p[SF_SEG_N1 * 64]; g.n2 = p[SF_SEG_N2 * 64];
        g.med = p[SF_SEG_MED * 64]; g.tpass = p[SF_SEG_TPASS * 64]; g.tbreak = p[SF_SEG_TBREAK * 64];
        g.l = su[(size_t)k * stride + SF_SEG_LAST * 64]; g.mc = su[(size_t)k * stride + SF_SEG_MEDC * 64];
        return g;
    };
    auto fold = [&](const Seg& g, int k, bool in) {
        const bool use = in && !stopped;
        const float wsum = T_start - g.tend;                               // sum of the blend weights of this segment
        dist += use ? g.dist + M2 * wsum - 2.f * M1 * g.m1 : 0.f;          // cross terms with the segments in front
        C0 += use ? g.c0 : 0.f; C1 += use ? g.c1 : 0.f; D += use ? g.d : 0.f;
        N0 += use ? g.n0 : 0.f; N1 += use ? g.n1 : 0.f; N2 += use ? g.n2 : 0.f;
        M1 += use ? g.m1 : 0.f; M2 += use ? g.m2 : 0.f;
        if (use && g.l) last = segment_range(tr, Sfull, k).x - tr.x + g.l;
        if (use && g.mc) { med_c = g.mc; med = g.med; }
        T_final = use ? g.tend : T_final;
        T_start = use ? T_start * g.tpass : T_start;                       // what pass 2 started the next segment from
        stopped = stopped || (use && g.tbreak < 0.0001f);
    };
    for (int k0 = 0; k0 < St; k0 += 4) {
        const Seg g0 = load(k0), g1 = load(min(k0 + 1, St - 1)), g2 = load(min(k0 + 2, St - 1)), g3 = load(min(k0 + 3, St - 1));
        fold(g0, k0, true);
        fold(g1, k0 + 1, k0 + 1 < St);
        fold(g2, k0 + 2, k0 + 2 < St);
        fold(g3, k0 + 3, k0 + 3 < St);
        if (__ballot(!stopped) == 0ull) break;
    }
    const size_t N = (size_t)a.grid.W * a.grid.H;
    a.accum[px.pix] = T_final; a.accum[N + px.pix] = M1; a.accum[2 * N + px.pix] = M2;
    a.n_contrib[px.pix] = last; a.n_contrib[N + px.pix] = med_c;
    a.out_color[px.pix] = C0 + T_final * a.bg[0];
    a.out_color[N + px.pix] = C1 + T_final * a.bg[1];
    a.out_others[0 * N + px.pix] = D;
    a.out_others[1 * N + px.pix] = 1.f - T_final;
    a.out_others[2 * N + px.pix] = N0; a.out_others[3 * N + px.pix] = N1; a.out_others[4 * N + px.pix] = N2;
    a.out_others[5 * N + px.pix] = med;
    a.out_others[6 * N + px.pix] = dist;
}

void launch_sf_render_pass1(const SfFwdArgs& a, hipStream_t s) {
    static const bool walk2 = [] { const char* e = getenv("LIDARGS_WALK2"); return !e || (atoi(e) & 1) != 0; }();   // render.hip walk2()
    if (walk2) hipLaunchKernelGGL((k_sf_render_forward<true, true>), dim3(segment_grid(a.grid.num_tiles() * a.grid.waves_per_tile, a.seg_hi - a.seg_lo)), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((k_sf_render_forward<true, false>), dim3(segment_grid(a.grid.num_tiles() * a.grid.waves_per_tile, a.seg_hi - a.seg_lo)), dim3(64), 0, s, a);
}
void launch_sf_render_pass2(const SfFwdArgs& a, hipStream_t s) {
    hipLaunchKernelGGL((k_sf_render_forward<false, false>), dim3(segment_grid(a.grid.num_tiles() * a.grid.waves_per_tile, a.seg_hi - a.seg_lo)), dim3(64), 0, s, a);
}
void launch_sf_alive(const SfFwdArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_sf_alive, dim3((unsigned)(a.grid.num_tiles() * a.grid.waves_per_tile)), dim3(64), 0, s, a);
}
void launch_sf_combine(const SfFwdArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_sf_combine, dim3((unsigned)(a.grid.num_tiles() * a.grid.waves_per_tile)), dim3(64), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// 32-slot reduce-scatter over the 64 lanes of a wave.
__device__ __forceinline__ float sf_x1(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float sf_x2(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true)); }
template <int M> __device__ __forceinline__ float sf_xs(float x) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), (M << 10) | 0x1F)); }

// The two wide steps (most values to fold) are a gfx950 lane swap + an add each (fold_halves32 / fold_halves16: no selects); the
// lane ^ 1, ^ 2, ^ 4 steps exchange through DPP quad permutes / the swizzle crossbar; lanes that differ in bit 3 end with the same
// sum (one row rotation).  On return lane L holds, in v[0], the wave-wide sum of slot
//   16*bit5(L) + 8*bit4(L) + 4*bit0(L) + 2*bit1(L) + bit2(L).
// Only 23 of the 32 slots carry a sum.  The nine unused ones (SFA_DEAD) sit where the folds pair them with each other -- a pair of
// unused slots is not folded at all: 12 + 6 + 3 + 2 + 1 folds instead of 16 + 8 + 4 + 2 + 1.
constexpr uint32_t SFA_DEAD = (1u << 3) | (1u << 7) | (1u << 11) | (1u << 14) | (1u << 15) | (1u << 19) | (1u << 23) | (1u << 27) | (1u << 31);
__device__ __forceinline__ constexpr bool sfa_dead(uint32_t mask, int k) { return (mask >> k) & 1u; }
// slots still unused after folding pairs (k, k + half): both halves unused
__device__ __forceinline__ constexpr uint32_t sfa_fold_mask(uint32_t mask, int half) { return mask & (mask >> half) & ((1u << half) - 1u); }

__device__ __forceinline__ float sf_reduce_scatter32(float (&v)[32], int lane) {
    constexpr uint32_t D32 = SFA_DEAD, D16 = sfa_fold_mask(D32, 16), D8 = sfa_fold_mask(D16, 8), D4 = sfa_fold_mask(D8, 4), D2 = sfa_fold_mask(D4, 2);
#pragma unroll
    for (int k = 0; k < 16; k++) if (!sfa_dead(D16, k)) v[k] = fold_halves32(v[k], v[k + 16]);
#pragma unroll
    for (int k = 0; k < 8; k++) if (!sfa_dead(D8, k)) v[k] = fold_halves16(v[k], v[k + 8]);
    { const bool hi = lane & 1;
#pragma unroll
      for (int k = 0; k < 4; k++) if (!sfa_dead(D4, k)) { const float keep = hi ? v[k + 4] : v[k], send = hi ? v[k] : v[k + 4]; v[k] = keep + sf_x1(send); } }
    { const bool hi = lane & 2;
#pragma unroll
      for (int k = 0; k < 2; k++) if (!sfa_dead(D2, k)) { const float keep = hi ? v[k + 2] : v[k], send = hi ? v[k] : v[k + 2]; v[k] = keep + sf_x2(send); } }
    { const bool hi = lane & 4;
      const float keep = hi ? v[1] : v[0], send = hi ? v[0] : v[1]; v[0] = keep + sf_xs<4>(send); }
    v[0] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[0]), 0x128, 0xF, 0xF, true));   // row_ror:8 = lane ^ 8
    return v[0];
}

// packed accumulator slots (32 floats = 128 B per surfel; the indices skip SFA_DEAD)
enum { SFA_COL0 = 0, SFA_COL1 = 1, SFA_OPA = 2, SFA_N0 = 4, SFA_N1 = 5, SFA_N2 = 6, SFA_TU0 = 8, SFA_TU1 = 9, SFA_TU2 = 10,
       SFA_TV0 = 12, SFA_TV1 = 13, SFA_TV2 = 16, SFA_TW0 = 17, SFA_TW1 = 18, SFA_TW2 = 20, SFA_AW0 = 21, SFA_AW1 = 22, SFA_AW2 = 24,
       SFA_M2X = 25, SFA_M2Y = 26, SFA_M2AX = 28, SFA_M2AY = 29, SFA_Z2D = 30 };

struct SfBwdArgs {
    TileGrid grid;
    const uint2* ranges; const uint32_t* point_list; const float4* rec; const uint32_t* rowspan;
    const float2* coltab; const float2* rowtab; const float* bg;
    const float* accum; const uint32_t* n_contrib;
    const float* dL_dpix; const float* dL_dothers;
    float* gacc;           // [32 P]
    const float* seg; int S; int seg_len;
    const uint8_t* flags; size_t R;
    const uint8_t* alive;
};

// One workgroup = (patch, segment): back-to-front walk of the segment's flagged entries.  T starts at the segment's own end
// value; the "what lies behind" recurrences are seeded with the partial sums of the segments behind it, as seen from there.
#ifdef LG_SF_BWD_WAVES   /* experiment (tools/waves_ab.sh) */
__attribute__((amdgpu_waves_per_eu(LG_SF_BWD_WAVES, LG_SF_BWD_WAVES)))
#endif
__global__ void __launch_bounds__(64) k_sf_render_backward(const SfBwdArgs a) {
    __shared__ float4 s_rec[5 * SF_CHUNK];
    __shared__ float4 s_oprow[SF_CHUNK];                               // opacity per pixel row of the patch, 0 outside the surfel's row span
    __shared__ uint32_t s_gid[SF_CHUNK];
    const int lane = threadIdx.x;
    const int S = a.S;
    const int wpt = a.grid.waves_per_tile;
    int patch, seg;
    if (!block_patch_segment(blockIdx.x, a.grid.num_tiles() * wpt, S, patch, seg)) return;
    const int tile = patch / wpt, sub = patch - tile * wpt;
    const uint2 tr = a.ranges[tile];
    const int St = segment_count(tr, S, a.seg_len);
    if (seg >= St) return;
    if (a.alive && seg >= (int)a.alive[patch]) return;
    const size_t stride = SF_SEG_PLANES * 64;
    const float* sb = a.seg + (size_t)patch * S * stride + lane;
    const uint32_t n_lane = reinterpret_cast<const uint32_t*>(sb)[(size_t)seg * stride + SF_SEG_LAST * 64];
    uint32_t n_max = n_lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_max = max(n_max, (uint32_t)__shfl_xor((int)n_max, o));
    if (n_max == 0) return;

    const SfPixel px = sf_pixel(a.grid, a.coltab, a.rowtab, patch, lane);
    const uint2 sr = segment_range(tr, St, seg);
    const uint32_t seg_off = sr.x - tr.x;                               // list position of the segment's first entry
    const size_t N = (size_t)a.grid.W * a.grid.H;
    const float T_final = px.inside ? a.accum[px.pix] : 0.f;
    const float final_D = px.inside ? a.accum[N + px.pix] : 0.f;
    const float final_A = 1.f - T_final;
    const uint32_t med_c = px.inside ? a.n_contrib[N + px.pix] : 0u;
    float T = sb[(size_t)seg * stride + SF_SEG_TEND * 64];
    float g0 = 0.f, g1 = 0.f, g_depth = 0.f, g_alpha = 0.f, gn0 = 0.f, gn1 = 0.f, gn2 = 0.f, g_med = 0.f, g_reg = 0.f;
    if (px.inside) {
        g0 = a.dL_dpix[px.pix]; g1 = a.dL_dpix[N + px.pix];
        g_depth = a.dL_dothers[0 * N + px.pix]; g_alpha = a.dL_dothers[1 * N + px.pix];
        gn0 = a.dL_dothers[2 * N + px.pix]; gn1 = a.dL_dothers[3 * N + px.pix]; gn2 = a.dL_dothers[4 * N + px.pix];
        g_med = a.dL_dothers[5 * N + px.pix]; g_reg = a.dL_dothers[6 * N + px.pix];
    }
    const float bgdot = a.bg[0] * g0 + a.bg[1] * g1;
    float acc_c0 = 0.f, acc_d = 0.f, acc_a = 0.f, acc_n0 = 0.f, acc_n1 = 0.f, acc_n2 = 0.f;
    {
        float b0 = 0.f, bd = 0.f, bn0 = 0.f, bn1 = 0.f, bn2 = 0.f;
        const int Send = a.alive ? min(St, (int)a.alive[patch]) : St;
        {
            float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            const int planes[5] = {SF_SEG_C0, SF_SEG_D, SF_SEG_N0, SF_SEG_N1, SF_SEG_N2};
            plane_sums<5>(acc, sb + (size_t)(seg + 1) * stride, stride, planes, Send - (seg + 1));
            b0 = acc[0]; bd = acc[1]; bn0 = acc[2]; bn1 = acc[3]; bn2 = acc[4];
        }
        if (T > 0.f) {
            const float inv = 1.f / T;
            acc_c0 = b0 * inv; acc_d = bd * inv; acc_n0 = bn0 * inv; acc_n1 = bn1 * inv; acc_n2 = bn2 * inv;
            acc_a = 1.f - T_final * inv;
        }
    }
    float last_alpha = 0.f, l_c0 = 0.f, l_d = 0.f, l_n0 = 0.f, l_n1 = 0.f, l_n2 = 0.f;

    const int c_last = (int)((n_max - 1) / SF_CHUNK);
    const int y0 = (tile / a.grid.tiles_x) * a.grid.TH + sub * LG_WAVE_ROWS;       // first pixel row of the patch
    const float* oprow = reinterpret_cast<const float*>(s_oprow) + (lane >> 4);
    const uint8_t* fl = a.flags + (size_t)sub * a.R + sr.x;
    bool have;
    auto gather = [&](int c) {
        const uint32_t k = (uint32_t)c * SF_CHUNK + lane;
        have = k < n_max && fl[k] != 0;
        return sf_gather(a.point_list, a.rec, a.rowspan, sr.x + k, have);
    };
    SfStaged st = gather(c_last);
    for (int c = c_last; c >= 0; c--) {
        __syncthreads();
        s_rec[lane] = st.a0; s_rec[SF_CHUNK + lane] = st.a1; s_rec[2 * SF_CHUNK + lane] = st.a2; s_rec[3 * SF_CHUNK + lane] = st.a3;
        s_rec[4 * SF_CHUNK + lane] = st.a4; s_oprow[lane] = rows_opacity(st.span, st.a1.z, y0); s_gid[lane] = st.gid;
        unsigned long long todo = __ballot(have);
        __syncthreads();
        if (c > 0) st = gather(c - 1);
        while (todo) {
            const int j = 63 - __builtin_clzll(todo);
            todo &= ~(1ull << j);
            const uint32_t e_loc = (uint32_t)c * SF_CHUNK + j;       // position inside the segment
            const uint32_t e = seg_off + e_loc;                       // 0-based list position == the reference's `contributor`
            const float4 r0 = s_rec[j], r1 = s_rec[SF_CHUNK + j], r2 = s_rec[2 * SF_CHUNK + j], r3 = s_rec[3 * SF_CHUNK + j], r4 = s_rec[4 * SF_CHUNK + j];
            const float op = oprow[4 * j];                              // the surfel's opacity on this pixel's row, 0 outside its row span
            const SfPair q = sf_pair(px, r0, r1, r2, r3, r4, op, e_loc < n_lane);
            const bool contrib = q.ok;
            if (__ballot(contrib) == 0ull) continue;
            // A pixel that does not blend this entry treats it as an alpha = 0 entry (as in render.hip): T / (1 - 0) = T, the
            // recurrences commit the previous entry and then carry (alpha 0, this entry), which the next step folds away
            // exactly -- so no select on the state registers; and every sum carries dL_dalpha, w or dL_dz as a factor, so
            // zeroing those three zeroes the 23 sums.  (Selects, not products: a skipped pair's depth may be 0 or huge.)
            const float alpha = contrib ? q.alpha : 0.f;
            const float G = q.G;
            const float c_d = contrib ? q.depth : 1.f;
            const float inv = __builtin_amdgcn_rcpf(1.f - alpha);       // 1 ulp; (1 - alpha) >= 0.01
            const float Tn = T * inv;
            const float w = alpha * Tn;
            // recurrences of "what lies behind" (R2/cr/backward.cu:349-410) and the sums over the channels: multiply-adds of terms of
            // one sign or of differences taken first -- fused here (the launch is bound by the vector pipe; the file is built without
            // contraction for the hit point's and the normal's cancelling expressions, which stay as written below)
            float a_c0, a_d, a_a, a_n0, a_n1, a_n2, dL_dalpha, dL_dz;
            const float icd = __builtin_amdgcn_rcpf(c_d);
            {
#pragma clang fp contract(fast)
                const float keep = 1.f - last_alpha;
                a_c0 = last_alpha * l_c0 + keep * acc_c0;
                a_d = last_alpha * l_d + keep * acc_d;
                a_a = last_alpha + keep * acc_a;
                a_n0 = last_alpha * l_n0 + keep * acc_n0;
                a_n1 = last_alpha * l_n1 + keep * acc_n1;
                a_n2 = last_alpha * l_n2 + keep * acc_n2;
                dL_dalpha = (r1.w - a_c0) * g0;                        // only channel 0 (:358-359)
                const float m_d = SF_FAR_N / (SF_FAR_N - SF_NEAR_N) * (1.f - SF_NEAR_N * icd);
                const float dmd_dd = (SF_FAR_N * SF_NEAR_N) / (SF_FAR_N - SF_NEAR_N) * icd * icd;
                dL_dz = (contrib && e + 1u == med_c) ? g_med : 0.f;     // contributor == median_contributor - 1 (:371)
                dL_dz += 2.0f * w * (m_d * final_A - final_D) * g_reg * dmd_dd;     // DETACH_WEIGHT: only the m_d path (:375-388)
                dL_dalpha += (c_d - a_d) * g_depth + (1.f - a_a) * g_alpha;
                dL_dalpha += (r3.x - a_n0) * gn0 + (r3.y - a_n1) * gn1 + (r3.z - a_n2) * gn2;
                dL_dalpha *= Tn;
                dL_dalpha -= T_final * inv * bgdot;
                dL_dalpha = contrib ? dL_dalpha : 0.f;
                dL_dz += w * g_depth;                                   // :420
            }
            const float dL_dG = op * dL_dalpha;
            // 3-D branch: gradient through s = (dp.Tu', dp.Tv'), dp = lam2 p - Tw, lam2 = (Tw.n)/(p.n)   (:427-563); its three
            // roots are zeroed for a 2-D-branch pair, the 2-D branch's two for a 3-D one
            const bool in3d = q.in3d;
            const float ga = in3d ? -dL_dG * G * q.sx : 0.f, gb = in3d ? -dL_dG * G * q.sy : 0.f;   // dL/ds
            const sf2 tx = sf2{r0.x, r0.y}, ty = sf2{r0.z, r0.w}, tz = sf2{r1.x, r1.y};   // (Tu', Tv') by component
            const float icos = __builtin_amdgcn_rcpf(q.cos2 != 0.f ? q.cos2 : 1.f);
            float3 gTu, gTv, gdp, gTw;
            float g_lam;
            {   // (fused multiply-adds here too; the normal's gradient below stays as written)
#pragma clang fp contract(fast)
                const sf2 iuv = tx * tx + ty * ty + tz * tz;              // (1/(Tu.Tu), 1/(Tv.Tv))
                const sf2 gab = sf2{ga, gb}, s2 = 2.f * sf2{q.sx, q.sy};
                // dL/dTu = ga (dp - 2 sx Tu)/(Tu.Tu) = ga (dp * iu - 2 sx Tu')
                const sf2 gTx = gab * (q.dp.x * iuv - s2 * tx), gTy = gab * (q.dp.y * iuv - s2 * ty), gTz = gab * (q.dp.z * iuv - s2 * tz);
                gTu = sf3(gTx.x, gTy.x, gTz.x); gTv = sf3(gTx.y, gTy.y, gTz.y);
                const sf2 px2 = gab * tx, py2 = gab * ty, pz2 = gab * tz;
                gdp = sf3(px2.x + px2.y, py2.x + py2.y, pz2.x + pz2.y);
                g_lam = in3d ? gdp.x * px.p.x + gdp.y * px.p.y + gdp.z * px.p.z + dL_dz : 0.f;
                const float gl = g_lam * icos;
                gTw = sf3(-gdp.x + gl * r3.x, -gdp.y + gl * r3.y, -gdp.z + gl * r3.z);
            }
            // d lam2 / d n = (Tw (p.n) - (Tw.n) p) / (p.n)^2 = -dp / (p.n): evaluated as the reference writes it (:452-461), the
            // difference of two ~range-sized vectors, so that its rounding (3 digits of cancellation) is the reference's
            const float icos2 = icos * icos;
            const float3 gN = sf3(g_lam * ((r2.x * q.cos2 - r4.w * px.p.x) * icos2), g_lam * ((r2.y * q.cos2 - r4.w * px.p.y) * icos2),
                                  g_lam * ((r2.z * q.cos2 - r4.w * px.p.z) * icos2));
            // 2-D branch (:578-599)
            const float dL_dG2 = in3d ? 0.f : dL_dG;
            const float m2x = dL_dG2 * (-G * 2.0f * 40.f * q.dxp), m2y = dL_dG2 * (-G * 2.0f * 100.f * q.dyp);
            float v[32];
#pragma unroll
            for (int k = 0; k < 32; k++) v[k] = 0.f;
            v[SFA_COL0] = w * g0; v[SFA_COL1] = w * g1;
            v[SFA_OPA] = G * dL_dalpha;
            v[SFA_N0] = __builtin_fmaf(w, gn0, gN.x); v[SFA_N1] = __builtin_fmaf(w, gn1, gN.y); v[SFA_N2] = __builtin_fmaf(w, gn2, gN.z);
            v[SFA_TU0] = gTu.x; v[SFA_TU1] = gTu.y; v[SFA_TU2] = gTu.z;
            v[SFA_TV0] = gTv.x; v[SFA_TV1] = gTv.y; v[SFA_TV2] = gTv.z;
            v[SFA_TW0] = gTw.x; v[SFA_TW1] = gTw.y; v[SFA_TW2] = gTw.z;
            v[SFA_AW0] = fabsf(gTw.x); v[SFA_AW1] = fabsf(gTw.y); v[SFA_AW2] = fabsf(gTw.z);
            v[SFA_M2X] = m2x; v[SFA_M2Y] = m2y; v[SFA_M2AX] = fabsf(m2x); v[SFA_M2AY] = fabsf(m2y);
            v[SFA_Z2D] = in3d ? 0.f : dL_dz;
            T = Tn;
            acc_c0 = a_c0; acc_d = a_d; acc_a = a_a; acc_n0 = a_n0; acc_n1 = a_n1; acc_n2 = a_n2;
            l_c0 = r1.w; l_d = c_d; l_n0 = r3.x; l_n1 = r3.y; l_n2 = r3.z;
            last_alpha = alpha;
            const float mine = sf_reduce_scatter32(v, lane);
            if ((lane & 8) == 0) {                                      // one owner per slot
                const int slot = 16 * ((lane >> 5) & 1) + 8 * ((lane >> 4) & 1) + 4 * (lane & 1) + 2 * ((lane >> 1) & 1) + ((lane >> 2) & 1);
                if (!((SFA_DEAD >> slot) & 1u)) atomicAdd(a.gacc + 32 * (size_t)s_gid[j] + slot, mine);
            }
        }
    }
}

void launch_sf_render_backward(const SfBwdArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_sf_render_backward, dim3(segment_grid(a.grid.num_tiles() * a.grid.waves_per_tile, a.S)), dim3(64), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
struct SfGaussBwdArgs {
    int P, W, H;
    const float* view; const float* means3D; const float* scales; const float* rotations; const float* beams; const int* radii;
    const float* transMat;             // transMat_precomp or nullptr: the Tw the blend's backward saw (the per-pair terms reconstructed below)
    const float* gacc; const uint8_t* tlist; const uint16_t* tcount;   // packed sums (lines of touched surfels only) | the touched surfels, region by region
    float* dL_dmean2D; float* dL_dnormal; float* dL_dopacity; float* dL_dcolor; float* dL_dmean3D; float* dL_dtransMat;
    float* dL_dtransMat_2dtemp; float* dL_dscale; float* dL_drot; float* depth;
};

// Sparse (round 5, as k_gaussian_backward in preprocess.hip): only the surfels the blend marked as touched have a gradient.  The
// backward's first launch (k_zero_touched) zeroed every gradient row and listed the touched surfels region by region; here a block
// covers four regions: every thread first writes the planar depth of four surfels (an output, not a gradient: R2/cr/backward.cu:670,
// 0 for a culled one), then each wave runs the chain on the listed surfels of one region and overwrites their rows.
__device__ __forceinline__ void sf_gb_row(const SfGaussBwdArgs& a, const int idx);
__global__ void __launch_bounds__(256) k_sf_gaussian_backward(const SfGaussBwdArgs a) {
    const int lane = threadIdx.x & 63;
    {
        const float* vm = a.view;
        const int b0 = blockIdx.x * 4 * LG_REGION;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = b0 + k * 256 + (int)threadIdx.x;
            if (i >= a.P) break;
            float dep = 0.f;
            if (a.radii[i] > 0) {
                const float3 pw = get3(a.means3D, i);
                const float px = vm[0] * pw.x + vm[4] * pw.y + vm[8] * pw.z + vm[12], pz = vm[2] * pw.x + vm[6] * pw.y + vm[10] * pw.z + vm[14];
                dep = sqrtf(px * px + pz * pz);                        // :670 (x, z only, as the reference)
            }
            a.depth[i] = dep;
        }
    }
    const int region = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int base = region * LG_REGION;
    if (base >= a.P) return;
    int off = (int)a.tlist[base + lane];                               // with the count, not behind it (the list is padded); through an empty asm
    int cnt = (int)a.tcount[region];                                   // statement, or the compiler sinks the slot's load below the test on the count
    asm volatile("" : "+v"(off), "+v"(cnt));
    for (int j = lane; j < cnt; j += 64) {
        sf_gb_row(a, base + off);
        if (j + 64 < cnt) off = (int)a.tlist[base + j + 64];
    }
}

__device__ __forceinline__ void sf_gb_row(const SfGaussBwdArgs& a, const int idx) {
    const float* vm = a.view;
    // the surfel's 128-byte line as eight 16-byte loads issued together (the slots are read all over the function: left to the
    // compiler they became a dozen 4-/8-/12-byte loads, each a pass over 64 different lines for the wave)
    float gl[32];
    {
        const float4* g4 = reinterpret_cast<const float4*>(a.gacc) + 8 * (size_t)idx;
#pragma unroll
        for (int k = 0; k < 8; k++) { const float4 v = g4[k]; gl[4 * k] = v.x; gl[4 * k + 1] = v.y; gl[4 * k + 2] = v.z; gl[4 * k + 3] = v.w; }
    }
    const float* g = gl;
    const float3 pw = get3(a.means3D, idx);
    const float3 pv = sf3(vm[0] * pw.x + vm[4] * pw.y + vm[8] * pw.z + vm[12], vm[1] * pw.x + vm[5] * pw.y + vm[9] * pw.z + vm[13],
                          vm[2] * pw.x + vm[6] * pw.y + vm[10] * pw.z + vm[14]);
    // the per-pair terms folded into per-surfel coefficients below were evaluated with the blend's Tw (R2/cr/backward.cu:267); the
    // chain through scales / rotations / means3D that follows uses p_view whatever the blend used (:625-690, Ts_precomp == nullptr)
    const float3 Tw = a.transMat ? sf3(a.transMat[9 * (size_t)idx + 6], a.transMat[9 * (size_t)idx + 7], a.transMat[9 * (size_t)idx + 8]) : pv;
    const float rho_r = sqrtf(sdot(Tw, Tw));
    const float rxy = sqrtf(Tw.x * Tw.x + Tw.y * Tw.y);
    const float pi_f = 3.14159265358979323846f;
    const float ga = fabsf(a.beams[a.H - 1] - a.beams[0]) / ((float)a.H - 1.f);   // grad_alpha (:425)

    put2(a.dL_dcolor, idx, g[SFA_COL0], g[SFA_COL1]);
    a.dL_dopacity[idx] = g[SFA_OPA];
    const float3 gn = sf3(g[SFA_N0], g[SFA_N1], g[SFA_N2]);
    if (a.dL_dnormal) put3(a.dL_dnormal, idx, gn.x, gn.y, gn.z);   // the three intermediates are optional
    const float3 aw = sf3(g[SFA_AW0], g[SFA_AW1], g[SFA_AW2]);
    if (a.dL_dtransMat_2dtemp) put3(a.dL_dtransMat_2dtemp, idx, aw.x, aw.y, aw.z);

    // dL/dmean2D: 3-D branch statistics are abs-linear in |dL/dTw| with per-surfel coefficients (:564-577);
    // |sin(beta_t) cos(alpha_t)| = |Tw.y|/rho_r, |cos(beta_t) cos(alpha_t)| = |Tw.x|/rho_r, ...
    const float irho = rho_r > 0.f ? 1.f / rho_r : 0.f, irxy = rxy > 0.f ? 1.f / rxy : 0.f;
    const float mx3 = pi_f * (fabsf(Tw.y) * aw.x + fabsf(Tw.x) * aw.y);                        // (|..|2pi/W) * rho_r * 0.5 W
    const float my3 = 0.5f * (float)a.H * ga * (fabsf(Tw.z) * fabsf(Tw.x) * irxy * aw.x + fabsf(Tw.z) * fabsf(Tw.y) * irxy * aw.y + rxy * aw.z);
    const float m2x = g[SFA_M2X], m2y = g[SFA_M2Y];
    put4(a.dL_dmean2D, idx, mx3 + m2x * 0.5f * (float)a.W, my3 + m2y * 0.5f * (float)a.H, mx3 + g[SFA_M2AX] * 0.5f * (float)a.W,
         my3 + g[SFA_M2AY] * 0.5f * (float)a.H);

    // dL/dT rows: 3-D sums + the 2-D branch's Tw terms (:590-598)
    float3 gTw = sf3(g[SFA_TW0], g[SFA_TW1], g[SFA_TW2]);
    {
        const float z2 = g[SFA_Z2D];
        const float kx = (float)a.W / (2.f * pi_f) * irxy * irxy;
        const float3 ddelx = sf3(kx * Tw.y, -kx * Tw.x, 0.f);
        const float ky = ga * irho * irho;
        const float3 ddely = sf3(-ky * Tw.z * Tw.x * irxy, -ky * Tw.z * Tw.y * irxy, ky * rxy);
        gTw.x += z2 * Tw.x * irho + m2x * ddelx.x + m2y * ddely.x;
        gTw.y += z2 * Tw.y * irho + m2x * ddelx.y + m2y * ddely.y;
        gTw.z += z2 * Tw.z * irho + m2y * ddely.z;
    }
    const float3 gTu = sf3(g[SFA_TU0], g[SFA_TU1], g[SFA_TU2]), gTv = sf3(g[SFA_TV0], g[SFA_TV1], g[SFA_TV2]);
    if (a.dL_dtransMat) {
        put3(a.dL_dtransMat, 3 * (size_t)idx, gTu.x, gTu.y, gTu.z); put3(a.dL_dtransMat, 3 * (size_t)idx + 1, gTv.x, gTv.y, gTv.z);
        put3(a.dL_dtransMat, 3 * (size_t)idx + 2, gTw.x, gTw.y, gTw.z);
    }

    // K10': T rows are (Rv L0, Rv L1, p_view)  =>  dL/dL0 = Rv^T dL/dTu, dL/dL1 = Rv^T dL/dTv, dL/dp = Rv^T dL/dTw
    const float4 q = get4(a.rotations, idx);
    float3 c0, c1, c2;
    sf_quat_cols(q, c0, c1, c2);
    const float3 dL0 = sf_rot_world(vm, gTu), dL1 = sf_rot_world(vm, gTv), dLp = sf_rot_world(vm, gTw);
    float3 dtn = sf_rot_world(vm, gn);
    const float3 nv = sf_rot_view(vm, c2);
    const float cs = -(pv.x * nv.x + pv.y * nv.y + pv.z * nv.z);
    if (!(cs > 0.f)) { dtn.x = -dtn.x; dtn.y = -dtn.y; dtn.z = -dtn.z; }
    const float s0 = a.scales[2 * idx], s1 = a.scales[2 * idx + 1];   // the backward ignores scale_modifier (:632)
    put2(a.dL_dscale, idx, sdot(dL0, c0), sdot(dL1, c1));
    put3(a.dL_dmean3D, idx, dLp.x, dLp.y, dLp.z);
    // quat_to_rotmat_vjp with v_R columns (dL0*s0, dL1*s1, dtn)  (R2/cr/auxiliary.h:274-316)
    const float3 v0 = sf3(dL0.x * s0, dL0.y * s0, dL0.z * s0), v1 = sf3(dL1.x * s1, dL1.y * s1, dL1.z * s1), v2 = dtn;
    const float sn = 1.0f / sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    const float w = q.x * sn, x = q.y * sn, y = q.z * sn, z = q.w * sn;
    // v_R[c][r]: column c = v_c, row r = component
    const float R01 = v0.y, R02 = v0.z, R10 = v1.x, R12 = v1.z, R20 = v2.x, R21 = v2.y, R00 = v0.x, R11 = v1.y, R22 = v2.z;
    put4(a.dL_drot, idx, 2.f * (x * (R12 - R21) + y * (R20 - R02) + z * (R01 - R10)),
         2.f * (-2.f * x * (R11 + R22) + y * (R01 + R10) + z * (R02 + R20) + w * (R12 - R21)),
         2.f * (x * (R01 + R10) - 2.f * y * (R00 + R22) + z * (R12 + R21) + w * (R20 - R02)),
         2.f * (x * (R02 + R20) + y * (R12 + R21) - 2.f * z * (R00 + R11) + w * (R01 - R10)));
}

void launch_sf_gaussian_backward(const SfGaussBwdArgs& a, hipStream_t s) {
    const unsigned regions = (unsigned)((a.P + LG_REGION - 1) / LG_REGION);
    hipLaunchKernelGGL(k_sf_gaussian_backward, dim3((regions + 3) / 4), dim3(256), 0, s, a);
}

}  // namespace lg

#include "surfel_api.inc"
