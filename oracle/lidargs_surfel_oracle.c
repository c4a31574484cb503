/*
 * lidargs_surfel_oracle.c -- CPU restatement of the 2DGS "laser-surfel" rasterizer (BASELINE config 5).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as lidargs_oracle.c).  "Parity unpinned" for the CUDA
 * arithmetic: the reference ships no tests / golden vectors for this path and cannot be built here.
 *
 * Reference: /root/reference/submodules/diff_lidargs_surfel_rasterization/ ("R2/"), cr/ = cuda_rasterizer/.
 * Compile-time configuration restated: NUM_CHANNELS 2, BLOCK_X 16, BLOCK_Y 1 (R2/cr/config.h:15-17);
 * RENDER_AXUTILITY 1, DUAL_VISIABLE 1, DETACH_WEIGHT 1, near_n 0.2, far_n 80, FilterInvSquare 2
 * (R2/cr/auxiliary.h:21-39); Ray_Divergence_Angle 0.006 (R2/cr/forward.cu:18).
 * Plain C, -ffp-contract=off: every expression rounds as written.  atan2 and the ray-table cos/sin are evaluated in
 * double and rounded to float (sf_atan2, sf_cosf, sf_sinf), i.e. the correctly rounded fp32 result, so the oracle does not depend on the host libm's
 * atan2f error (CUDA's atan2f is 2-ulp, glibc's 1-ulp: the projected centre p_c ~ 1e3 px enters the 2-D filter
 * exponent with a gain of ~80/px, so one ulp of atan2f is a 1e-3 change of a blend weight).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define SF_CHANNELS 2
#define SF_BLOCK_X 16
#define SF_BLOCK_Y 1
#define SF_OTHERS 7            /* depth, alpha, normal x3, median depth, distortion (R2/cr/auxiliary.h:23-27) */
#define SF_DEPTH_OFFSET 0
#define SF_ALPHA_OFFSET 1
#define SF_NORMAL_OFFSET 2
#define SF_MIDDEPTH_OFFSET 5
#define SF_DISTORTION_OFFSET 6

static const float SF_PI = 3.14159265358979323846f;
static const float SF_RAY_DIV = 0.006f;
static const float SF_NEAR_N = 0.2f, SF_FAR_N = 80.0f, SF_FILTER_INV_SQ = 2.0f;

typedef struct { float x, y, z; } sf3;
typedef struct { float x, y; } sf2;

/* Test knob, as lgo_set_ulp_perturbation of the 3-D oracle (lidargs_oracle.c): every cos / sin / atan2 / exp result the reference takes
 * from CUDA libdevice (R2/setup.py builds without -use_fast_math: cosf 1 ulp, sinf 1 ulp, atan2f 2 ulp, expf 2 ulp) is moved by an
 * integer number of ulps inside that bound -- mode 1: pseudo-random in [-amp, +amp], a pure function of (input bits, seed), so that the
 * forward and the backward re-evaluate a pair identically; mode 2: always +amp; mode 3: always -amp.  The spread between such runs is
 * what this restatement can promise about the CUDA reference's outputs (tools/parity_sweep.py judges its residue against it). */
static int sfo_ulp_mode = 0;
static uint32_t sfo_ulp_seed = 0;
void sfo_set_ulp_perturbation(int mode, unsigned seed) { sfo_ulp_mode = mode; sfo_ulp_seed = seed; }
static float sf_perturb(float r, float in1, float in2, int amp) {
    if (!sfo_ulp_mode || !(r == r) || r == 0.0f || isinf(r)) return r;
    int k;
    if (sfo_ulp_mode == 2) k = amp;
    else if (sfo_ulp_mode == 3) k = -amp;
    else {
        uint32_t a, b;
        memcpy(&a, &in1, 4); memcpy(&b, &in2, 4);
        uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (sfo_ulp_seed + (uint32_t)amp) * 0xC2B2AE3Du;
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
        k = (int)(h % (uint32_t)(2 * amp + 1)) - amp;
    }
    int32_t bits;
    memcpy(&bits, &r, 4);
    bits += (bits < 0) ? -k : k;
    float out;
    memcpy(&out, &bits, 4);
    return (out == out && !isinf(out)) ? out : r;
}
static float sf_cosf(float x) { return sf_perturb((float)cos((double)x), x, 1.0f, 1); }
static float sf_sinf(float x) { return sf_perturb((float)sin((double)x), x, 2.0f, 1); }
static float sf_atan2(float y, float x) { return sf_perturb((float)atan2((double)y, (double)x), y, x, 2); }
static float sf_expf(float x) { return sf_perturb(expf(x), x, 4.0f, 2); }

static int sfo_reverse_pixel_order = 0;
void sfo_set_reverse_pixel_order(int on) { sfo_reverse_pixel_order = on; }
/* Test knob (round 6), as lgo_set_accumulate_double: the per-surfel sums of the backward blend in float64 (every term stays the float32
 * value R2/cr/backward.cu computes; the reference sums them with float atomics in scheduling order). */
static int sfo_accumulate_double = 0;
void sfo_set_accumulate_double(int on) { sfo_accumulate_double = on; }

static char sfo_err[256] = "";
const char* sfo_last_error(void) { return sfo_err; }

/* R2/cr/auxiliary.h:58-80 find_closest_label (same as R3) */
static int sf_find_closest_label(const float* beams, float angle, int length) {
    if (angle >= beams[length - 1]) return length - 1;
    else if (angle <= beams[0]) return 0;
    int lo = 0, hi = length;
    while (lo < hi) { int mid = (lo + hi) / 2; if (beams[mid] < angle) lo = mid + 1; else hi = mid; }
    return lo;
}

static unsigned sf_umin(unsigned a, unsigned b) { return a < b ? a : b; }
static int sf_imax(int a, int b) { return a > b ? a : b; }

/* R2/cr/auxiliary.h:99-112 getRect_lidar: x and ymin truncate, ymax = round(p.y + ry) */
static void sf_get_rect(float px, float py, int rx, int ry, unsigned gx, unsigned gy,
                        unsigned* xmin, unsigned* ymin, unsigned* xmax, unsigned* ymax) {
    *xmin = sf_umin(gx, (unsigned)sf_imax(0, (int)((px - rx) / SF_BLOCK_X)));
    *ymin = sf_umin(gy, (unsigned)sf_imax(0, (int)((py - ry) / SF_BLOCK_Y)));
    *xmax = sf_umin(gx, (unsigned)sf_imax(0, (int)((px + rx + SF_BLOCK_X - 1) / SF_BLOCK_X)));
    *ymax = sf_umin(gy, (unsigned)sf_imax(0, (int)(roundf((py + ry)))));
}

/* sf_get_rect on n caller-supplied inputs (tests/: against lidargs_debug_rects with surfel = 1) */
void sfo_rects(int n, const float* p_cr, const int* r_xy, int gx, int gy, int* rects) {
    for (int i = 0; i < n; i++) {
        unsigned xmin, ymin, xmax, ymax;
        sf_get_rect(p_cr[2 * i], p_cr[2 * i + 1], r_xy[2 * i], r_xy[2 * i + 1], (unsigned)gx, (unsigned)gy, &xmin, &ymin, &xmax, &ymax);
        rects[4 * i] = (int)xmin; rects[4 * i + 1] = (int)ymin; rects[4 * i + 2] = (int)xmax; rects[4 * i + 3] = (int)ymax;
    }
}

static sf3 sf_point4x3(sf3 p, const float* m) {
    sf3 t = { m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
              m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14] };
    return t;
}
static sf3 sf_vec4x3(sf3 p, const float* m) {          /* transformVec4x3 */
    sf3 t = { m[0] * p.x + m[4] * p.y + m[8] * p.z, m[1] * p.x + m[5] * p.y + m[9] * p.z, m[2] * p.x + m[6] * p.y + m[10] * p.z };
    return t;
}
static sf3 sf_vec4x3_t(sf3 p, const float* m) {        /* transformVec4x3Transpose */
    sf3 t = { m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z, m[8] * p.x + m[9] * p.y + m[10] * p.z };
    return t;
}

/* R2/cr/forward.cu:118-174 cpmpute_pix_f / cpmpute_pix: (column, row) of a view-space point.
 * with_cull: the Ray_Divergence_Angle beam-fan cull (single, not doubled as in R3). */
static int sf_compute_pix(sf3 p, int W, int H, const float* beams, int with_cull, sf2* pix) {
    float beta = SF_PI - sf_atan2(p.y, p.x);
    float p_c = beta / (2 * SF_PI / (float)W);
    float alpha = sf_atan2(p.z, sqrtf(p.x * p.x + p.y * p.y));
    int i = sf_find_closest_label(beams, alpha, H);
    float before, after, p_r;
    if (i > 0) {
        before = beams[i - 1]; after = beams[i];
        p_r = i - 1 + (alpha - before) / (after - before);
        if (with_cull && alpha > (after + SF_RAY_DIV)) return 0;
    } else {
        before = beams[i]; after = beams[i + 1];
        p_r = i + 1 + (alpha - after) / (after - before);
        if (with_cull && alpha < (before - SF_RAY_DIV)) return 0;
    }
    p_r = (float)H - p_r - 1;
    pix->x = p_c; pix->y = p_r;
    return 1;
}

/* R2/cr/auxiliary.h:249-271 quat_to_rotmat (normalises; glm column-major: R[c][r]) */
static void sf_quat_to_rotmat(const float* q, float R[3][3]) {
    float s = 1.0f / sqrtf(q[3] * q[3] + q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);   /* rsqrtf */
    float w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y + w * z); R[0][2] = 2.f * (x * z - w * y);
    R[1][0] = 2.f * (x * y - w * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z + w * x);
    R[2][0] = 2.f * (x * z + w * y); R[2][1] = 2.f * (y * z - w * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
}

typedef struct {
    int P, W, H, R;
    unsigned gx, gy;
    float* depths;          /* P  range */
    float* means2D;         /* 2P */
    float* transMat;        /* 9P rows Tu, Tv, Tw (view-space axes and centre) */
    float* normal_opacity;  /* 4P */
    uint32_t* tiles_touched; uint32_t* point_offsets;
    int* radii_xy;
    uint64_t* keys; uint32_t* point_list; uint32_t* ranges;
    float* accum;           /* 3N: final_T, M1, M2 (R2/cr/forward.cu:533-535) */
    uint32_t* n_contrib;    /* 2N: last contributor, median contributor */
} sfo_state;

void sfo_free(void* h) {
    sfo_state* s = (sfo_state*)h;
    if (!s) return;
    free(s->depths); free(s->means2D); free(s->transMat); free(s->normal_opacity); free(s->tiles_touched);
    free(s->point_offsets); free(s->radii_xy); free(s->keys); free(s->point_list); free(s->ranges); free(s->accum); free(s->n_contrib);
    free(s);
}
int sfo_num_rendered(const void* h) { return ((const sfo_state*)h)->R; }
const void* sfo_state_array(const void* h, int which, long long* count) {
    const sfo_state* s = (const sfo_state*)h;
    long long P = s->P, N = (long long)s->W * s->H, T = (long long)s->gx * s->gy, R = s->R;
    switch (which) {
        case 0: *count = P; return s->depths;
        case 1: *count = 2 * P; return s->means2D;
        case 2: *count = 9 * P; return s->transMat;
        case 3: *count = 4 * P; return s->normal_opacity;
        case 4: *count = P; return s->tiles_touched;
        case 5: *count = 2 * P; return s->radii_xy;
        case 6: *count = R; return s->point_list;
        case 7: *count = 2 * T; return s->ranges;
        case 8: *count = 3 * N; return s->accum;
        case 9: *count = 2 * N; return s->n_contrib;
    }
    *count = 0; return NULL;
}

/* T = transpose(splat2world) * world2view restated element-wise (R2/cr/forward.cu:271-295):
 * rows of the stored transMat are  Tu = Rv*L0,  Tv = Rv*L1,  Tw = p_view  with L = R*S. */
static void sf_trans_mat(const float* p_orig, const float* scale, float mod, const float* rot, const float* vm,
                         float L[3][3], float T9[9]) {
    float R[3][3];
    sf_quat_to_rotmat(rot, R);
    /* L = R * S (glm): column c of L = column c of R times S[c][c]; S[2][2] = 1 (scale_to_mat, auxiliary.h:319-326) */
    float sx = mod * scale[0], sy = mod * scale[1];
    for (int r = 0; r < 3; r++) {
        /* glm mat*mat with a diagonal right factor: L[c][r] = R[0][r]*S[c][0] + R[1][r]*S[c][1] + R[2][r]*S[c][2] */
        L[0][r] = R[0][r] * sx + R[1][r] * 0.f + R[2][r] * 0.f;
        L[1][r] = R[0][r] * 0.f + R[1][r] * sy + R[2][r] * 0.f;
        L[2][r] = R[0][r] * 0.f + R[1][r] * 0.f + R[2][r] * 1.f;
    }
    /* glm::transpose(mat3x4 splat2world) is mat4x3; times mat3x4 world2view -> mat3.
     * T[c][r] = sum_k M^T[k][r] * world2view[c][k],  M^T[k][r] = splat2world[r][k]:
     * splat2world columns: (L0,0), (L1,0), (p,1); world2view column c = (vm[c], vm[4+c], vm[8+c], vm[12+c]). */
    float Mcol[3][4] = { { L[0][0], L[0][1], L[0][2], 0.f }, { L[1][0], L[1][1], L[1][2], 0.f }, { p_orig[0], p_orig[1], p_orig[2], 1.f } };
    for (int c = 0; c < 3; c++) {
        float wv[4] = { vm[c], vm[4 + c], vm[8 + c], vm[12 + c] };
        for (int r = 0; r < 3; r++) {
            float t = Mcol[r][0] * wv[0] + Mcol[r][1] * wv[1] + Mcol[r][2] * wv[2] + Mcol[r][3] * wv[3];
            /* stored as T_ptr[idx*3 + r] = {T[0][r], T[1][r], T[2][r]} */
            T9[3 * r + c] = t;
        }
    }
}

/* R2/cr/forward.cu:177-215 compute_aabb_cylinder: extent from projecting the +-3 sigma axis end points */
static void sf_aabb(const float T9[9], float cutoff, int W, int H, float cx, float cy, const float* beams, sf2* extent) {
    sf3 T0 = { T9[0] * cutoff, T9[1] * cutoff, T9[2] * cutoff };
    sf3 T1 = { T9[3] * cutoff, T9[4] * cutoff, T9[5] * cutoff };
    sf3 T3 = { T9[6], T9[7], T9[8] };
    sf3 La = { T0.x + T3.x, T0.y + T3.y, T0.z + T3.z }, Lb = { T1.x + T3.x, T1.y + T3.y, T1.z + T3.z };
    sf3 La2 = { -T0.x + T3.x, -T0.y + T3.y, -T0.z + T3.z }, Lb2 = { -T1.x + T3.x, -T1.y + T3.y, -T1.z + T3.z };
    sf2 a, a2, b, b2;
    sf_compute_pix(La, W, H, beams, 0, &a); sf_compute_pix(La2, W, H, beams, 0, &a2);
    sf_compute_pix(Lb, W, H, beams, 0, &b); sf_compute_pix(Lb2, W, H, beams, 0, &b2);
    float ax = fmaxf(fabsf(a.x - cx), fabsf(a2.x - cx)), ay = fmaxf(fabsf(a.y - cy), fabsf(a2.y - cy));
    float bx = fmaxf(fabsf(b.x - cx), fabsf(b2.x - cx)), by = fmaxf(fabsf(b.y - cy), fabsf(b2.y - cy));
    extent->x = ceilf(fmaxf(fmaxf(ax, bx), 1.0f));
    extent->y = ceilf(fmaxf(fmaxf(ay, by), 1.0f));
}

/* K1': R2/cr/forward.cu:217-325 preprocessCUDA_cylinder; filter=1 -> R2/cr/forward.cu:551-631 */
static void sf_preprocess_one(int idx, int filter, sfo_state* s, const float* means3D, const float* scales, float mod,
                              const float* rotations, const float* opacities, const float* vm, const float* beams,
                              int far_, int near_, int* radii) {
    const int W = s->W, H = s->H;
    radii[idx] = 0; s->radii_xy[2 * idx] = 0; s->radii_xy[2 * idx + 1] = 0;
    if (!filter) s->tiles_touched[idx] = 0;
    sf3 p_orig = { means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
    sf3 pv = sf_point4x3(p_orig, vm);
    float dist = sqrtf(pv.x * pv.x + pv.y * pv.y + pv.z * pv.z);
    if (dist >= far_ || dist <= near_) return;
    sf2 pim;
    if (!sf_compute_pix(pv, W, H, beams, 1, &pim)) return;

    float L[3][3], T9[9];
    sf_trans_mat(means3D + 3 * idx, scales + 2 * idx, mod, rotations + 4 * idx, vm, L, T9);
    memcpy(s->transMat + 9 * idx, T9, sizeof T9);

    sf3 normal = { 0, 0, 0 };
    if (!filter) {
        sf3 l2 = { L[2][0], L[2][1], L[2][2] };
        normal = sf_vec4x3(l2, vm);
        /* DUAL_VISIABLE (R2/cr/forward.cu:297-302): flip the normal to face the sensor */
        float c = -(pv.x * normal.x + pv.y * normal.y + pv.z * normal.z);
        if (c == 0) return;
        float mult = c > 0 ? 1.f : -1.f;
        normal.x *= mult; normal.y *= mult; normal.z *= mult;
    }
    sf2 extent;
    sf_aabb(T9, 3.0f, W, H, pim.x, pim.y, beams, &extent);
    unsigned xmin, ymin, xmax, ymax;
    sf_get_rect(pim.x, pim.y, (int)extent.x, (int)extent.y, s->gx, s->gy, &xmin, &ymin, &xmax, &ymax);
    if ((xmax - xmin) * (ymax - ymin) == 0) return;

    radii[idx] = (int)fmaxf(extent.x, extent.y);
    s->radii_xy[2 * idx] = (int)extent.x; s->radii_xy[2 * idx + 1] = (int)extent.y;
    if (filter) return;
    s->depths[idx] = dist;
    s->means2D[2 * idx] = pim.x; s->means2D[2 * idx + 1] = pim.y;
    s->normal_opacity[4 * idx] = normal.x; s->normal_opacity[4 * idx + 1] = normal.y; s->normal_opacity[4 * idx + 2] = normal.z;
    s->normal_opacity[4 * idx + 3] = opacities[idx];
    s->tiles_touched[idx] = (ymax - ymin) * (xmax - xmin);
}

static uint32_t sf_higher_msb(uint32_t n) {
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) { step /= 2; if (n >> msb) msb += step; else msb -= step; }
    if (n >> msb) msb++;
    return msb;
}

static void sf_stable_sort_pairs(uint64_t* keys, uint32_t* vals, long long n, int end_bit) {
    if (n <= 1) return;
    uint64_t* k2 = (uint64_t*)malloc(sizeof(uint64_t) * n); uint32_t* v2 = (uint32_t*)malloc(sizeof(uint32_t) * n);
    uint64_t *ka = keys, *kb = k2; uint32_t *va = vals, *vb = v2;
    for (int shift = 0; shift < end_bit; shift += 16) {
        int bits = end_bit - shift < 16 ? end_bit - shift : 16;
        uint32_t nb = 1u << bits;
        long long* cnt = (long long*)calloc((size_t)nb + 1, sizeof(long long));
        for (long long i = 0; i < n; i++) cnt[((ka[i] >> shift) & (nb - 1)) + 1]++;
        for (uint32_t b = 0; b < nb; b++) cnt[b + 1] += cnt[b];
        for (long long i = 0; i < n; i++) { long long d = cnt[(ka[i] >> shift) & (nb - 1)]++; kb[d] = ka[i]; vb[d] = va[i]; }
        free(cnt);
        uint64_t* tk = ka; ka = kb; kb = tk; uint32_t* tv = va; va = vb; vb = tv;
    }
    if (ka != keys) { memcpy(keys, ka, sizeof(uint64_t) * n); memcpy(vals, va, sizeof(uint32_t) * n); }
    free(k2); free(v2);
}

/* pixel ray (R2/cr/forward.cu:444-456): beta in double then float, float cos/sin */
static sf3 sf_pixel_dir(int x, int y, int W, int H, const float* beams) {
    float pixfx = (float)x;
    float beta = (float)(-((double)pixfx - (double)(float)W / 2.0) / (double)(float)W * 2.0 * (double)SF_PI);
    float alp = beams[H - 1 - y];
    sf3 p = { sf_cosf(alp) * sf_cosf(beta), sf_cosf(alp) * sf_sinf(beta), sf_sinf(alp) };
    return p;
}

/* Per-pair geometry shared by forward and backward (R2/cr/forward.cu:426-476, R2/cr/backward.cu:283-330).
 * Returns 0 if the pair is skipped before alpha. */
typedef struct { sf3 p, Tu, Tv, Tw, dp; float normal[3], opa, rho_r, real_depth, rho3d, rho2d, rho, depth; sf2 s, d; float Tu_Tu, Tv_Tv, dp_Tu, dp_Tv; } sf_pair;
static int sf_pair_geom(const sfo_state* s, uint32_t g, int x, int y, sf3 p, sf_pair* o) {
    const float* T9 = s->transMat + 9 * g;
    const float* no = s->normal_opacity + 4 * g;
    o->p = p;
    o->Tu.x = T9[0]; o->Tu.y = T9[1]; o->Tu.z = T9[2]; o->Tv.x = T9[3]; o->Tv.y = T9[4]; o->Tv.z = T9[5];
    o->Tw.x = T9[6]; o->Tw.y = T9[7]; o->Tw.z = T9[8];
    o->normal[0] = no[0]; o->normal[1] = no[1]; o->normal[2] = no[2]; o->opa = no[3];
    sf3 Tw = o->Tw;
    o->rho_r = sqrtf(Tw.x * Tw.x + Tw.y * Tw.y + Tw.z * Tw.z);
    float L2_normal = 1.0f, L2_Tw = o->rho_r, L2_p = 1.0f;
    float cos_phi1 = (Tw.x * no[0] + Tw.y * no[1] + Tw.z * no[2]) / (L2_Tw * L2_normal);
    float lambda = L2_Tw * cos_phi1;
    float cos_phi2 = (p.x * no[0] + p.y * no[1] + p.z * no[2]) / (L2_p * L2_normal);
    if (cos_phi2 == 0) return 0;
    float lambda2 = lambda / cos_phi2;
    o->real_depth = lambda2;
    sf3 real_p = { lambda2 * p.x, lambda2 * p.y, lambda2 * p.z };
    o->dp.x = real_p.x - Tw.x; o->dp.y = real_p.y - Tw.y; o->dp.z = real_p.z - Tw.z;
    o->Tu_Tu = o->Tu.x * o->Tu.x + o->Tu.y * o->Tu.y + o->Tu.z * o->Tu.z;
    o->Tv_Tv = o->Tv.x * o->Tv.x + o->Tv.y * o->Tv.y + o->Tv.z * o->Tv.z;
    o->dp_Tu = o->dp.x * o->Tu.x + o->dp.y * o->Tu.y + o->dp.z * o->Tu.z;
    o->dp_Tv = o->dp.x * o->Tv.x + o->dp.y * o->Tv.y + o->dp.z * o->Tv.z;
    o->s.x = o->dp_Tu / o->Tu_Tu; o->s.y = o->dp_Tv / o->Tv_Tv;
    o->rho3d = (o->s.x * o->s.x + o->s.y * o->s.y);
    o->d.x = s->means2D[2 * g] - (float)x; o->d.y = s->means2D[2 * g + 1] - (float)y;
    o->rho2d = SF_FILTER_INV_SQ * (40 * o->d.x * o->d.x + 100 * o->d.y * o->d.y);
    o->rho = (o->real_depth > 0) ? (o->rho3d < o->rho2d ? o->rho3d : o->rho2d) : o->rho2d;
    o->depth = (o->rho3d <= o->rho2d && o->real_depth > 0) ? o->real_depth : o->rho_r;
    if (o->depth < SF_NEAR_N) return 0;
    return 1;
}

/* Forward: R2/cr/rasterizer_impl.cu:200-360.  Outputs: out_color[2,H,W], out_others[7,H,W], radii[P]
 * (pixels[P,1] is allocated by the binding and never written: the atomicAdd is commented out, forward.cu:522). */
/* transMat_precomp (may be NULL): the preprocess builds T from scales and rotations whatever is passed (rect, normal, depth, centre:
 * R2/cr/forward.cu:271-325); the blend -- forward AND backward -- then reads the rows from transMat_precomp when it is there
 * (R2/cr/rasterizer_impl.cu:332, :408).  Restated by overwriting the state's rows after the preprocess: only the blends read them. */
void* sfo_forward_tm(int P, const float* background, int width, int height, const float* means3D, const float* colors_precomp,
                     const float* opacities, const float* scales, float scale_modifier, const float* rotations, const float* transMat_precomp,
                     const float* viewmatrix, const float* beams, int far_, int near_,
                     float* out_color, float* out_others, int* radii);

void* sfo_forward(int P, const float* background, int width, int height, const float* means3D, const float* colors_precomp,
                  const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                  const float* viewmatrix, const float* beams, int far_, int near_,
                  float* out_color, float* out_others, int* radii) {
    return sfo_forward_tm(P, background, width, height, means3D, colors_precomp, opacities, scales, scale_modifier, rotations, NULL,
                          viewmatrix, beams, far_, near_, out_color, out_others, radii);
}

void* sfo_forward_tm(int P, const float* background, int width, int height, const float* means3D, const float* colors_precomp,
                     const float* opacities, const float* scales, float scale_modifier, const float* rotations, const float* transMat_precomp,
                     const float* viewmatrix, const float* beams, int far_, int near_,
                     float* out_color, float* out_others, int* radii) {
    if (colors_precomp == NULL) { snprintf(sfo_err, sizeof sfo_err, "For non-RGB, provide precomputed Gaussian colors!"); return NULL; }
    const int W = width, H = height;
    const long long N = (long long)W * H;
    sfo_state* s = (sfo_state*)calloc(1, sizeof(sfo_state));
    s->P = P; s->W = W; s->H = H;
    s->gx = (W + SF_BLOCK_X - 1) / SF_BLOCK_X; s->gy = (H + SF_BLOCK_Y - 1) / SF_BLOCK_Y;
    size_t Pz = P > 0 ? (size_t)P : 1;
    s->depths = (float*)calloc(Pz, 4); s->means2D = (float*)calloc(Pz * 2, 4); s->transMat = (float*)calloc(Pz * 9, 4);
    s->normal_opacity = (float*)calloc(Pz * 4, 4); s->tiles_touched = (uint32_t*)calloc(Pz, 4);
    s->point_offsets = (uint32_t*)calloc(Pz, 4); s->radii_xy = (int*)calloc(Pz * 2, 4);
    s->ranges = (uint32_t*)calloc((size_t)s->gx * s->gy * 2, 4);
    s->accum = (float*)calloc((size_t)N * 3, 4); s->n_contrib = (uint32_t*)calloc((size_t)N * 2, 4);
    for (int i = 0; i < P; i++)
        sf_preprocess_one(i, 0, s, means3D, scales, scale_modifier, rotations, opacities, viewmatrix, beams, far_, near_, radii);
    if (transMat_precomp && P > 0) memcpy(s->transMat, transMat_precomp, sizeof(float) * 9 * (size_t)P);
    uint32_t run = 0;
    for (int i = 0; i < P; i++) { run += s->tiles_touched[i]; s->point_offsets[i] = run; }
    const long long R = P > 0 ? (long long)s->point_offsets[P - 1] : 0;
    s->R = (int)R;
    s->keys = (uint64_t*)malloc(sizeof(uint64_t) * (R > 0 ? R : 1));
    s->point_list = (uint32_t*)malloc(sizeof(uint32_t) * (R > 0 ? R : 1));
    for (int idx = 0; idx < P; idx++) {              /* R2/cr/rasterizer_impl.cu:70-112 */
        if (radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : s->point_offsets[idx - 1];
            unsigned xmin, ymin, xmax, ymax;
            sf_get_rect(s->means2D[2 * idx], s->means2D[2 * idx + 1], s->radii_xy[2 * idx], s->radii_xy[2 * idx + 1], s->gx, s->gy,
                        &xmin, &ymin, &xmax, &ymax);
            uint32_t dbits; memcpy(&dbits, &s->depths[idx], 4);
            for (int y = (int)ymin; y < (int)ymax; y++)
                for (int x = (int)xmin; x < (int)xmax; x++) {
                    uint64_t key = (uint64_t)(y * s->gx + x); key <<= 32; key |= dbits;
                    s->keys[off] = key; s->point_list[off] = (uint32_t)idx; off++;
                }
        }
    }
    sf_stable_sort_pairs(s->keys, s->point_list, R, 32 + (int)sf_higher_msb(s->gx * s->gy));
    for (long long i = 0; i < R; i++) {
        uint32_t cur = (uint32_t)(s->keys[i] >> 32);
        if (i == 0) s->ranges[2 * cur] = 0;
        else { uint32_t prev = (uint32_t)(s->keys[i - 1] >> 32); if (cur != prev) { s->ranges[2 * prev + 1] = (uint32_t)i; s->ranges[2 * cur] = (uint32_t)i; } }
        if (i == R - 1) s->ranges[2 * cur + 1] = (uint32_t)R;
    }

    /* K7': R2/cr/forward.cu:327-547 renderCUDA, per pixel */
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const uint32_t tile = (uint32_t)(y / SF_BLOCK_Y) * s->gx + (uint32_t)(x / SF_BLOCK_X);
            const uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
            const sf3 p = sf_pixel_dir(x, y, W, H, beams);
            float T = 1.0f;
            uint32_t contributor = 0, last_contributor = 0;
            float C[SF_CHANNELS] = { 0 }, Nn[3] = { 0 }, D = 0, M1 = 0, M2 = 0, distortion = 0, median_depth = 0;
            float median_contributor = -1;
            int done = 0;
            for (uint32_t k = r0; k < r1 && !done; k++) {
                contributor++;
                const uint32_t g = s->point_list[k];
                sf_pair q;
                if (!sf_pair_geom(s, g, x, y, p, &q)) continue;
                float power = -0.5f * q.rho;
                if (power > 0.0f) continue;
                float a = q.opa * sf_expf(power);
                float alpha = 0.99f < a ? 0.99f : a;
                if (alpha < 1.0f / 255.0f) continue;
                float test_T = T * (1 - alpha);
                if (test_T < 0.0001f) { done = 1; continue; }
                float w = alpha * T;
                float A = 1 - T;
                float m = SF_FAR_N / (SF_FAR_N - SF_NEAR_N) * (1 - SF_NEAR_N / q.depth);
                distortion += (m * m * A + M2 - 2 * m * M1) * w;
                D += q.depth * w;
                M1 += m * w;
                M2 += m * m * w;
                if (T > 0.5) { median_depth = q.depth; median_contributor = (float)contributor; }   /* T > 0.5 compares in double */
                for (int ch = 0; ch < 3; ch++) Nn[ch] += q.normal[ch] * w;
                for (int ch = 0; ch < SF_CHANNELS; ch++) C[ch] += colors_precomp[g * SF_CHANNELS + ch] * w;
                T = test_T;
                last_contributor = contributor;
            }
            const long long pix = (long long)W * y + x;
            s->accum[pix] = T; s->accum[pix + N] = M1; s->accum[pix + 2 * N] = M2;
            s->n_contrib[pix] = last_contributor;
            /* the reference stores the float (-1 = none) into a uint32; CUDA's conversion saturates negatives to 0 */
            s->n_contrib[pix + N] = median_contributor < 0 ? 0u : (uint32_t)median_contributor;
            for (int ch = 0; ch < SF_CHANNELS; ch++) out_color[ch * N + pix] = C[ch] + T * background[ch];
            out_others[pix + SF_DEPTH_OFFSET * N] = D;
            out_others[pix + SF_ALPHA_OFFSET * N] = 1 - T;
            for (int ch = 0; ch < 3; ch++) out_others[pix + (SF_NORMAL_OFFSET + ch) * N] = Nn[ch];
            out_others[pix + SF_MIDDEPTH_OFFSET * N] = median_depth;
            out_others[pix + SF_DISTORTION_OFFSET * N] = distortion;
        }
    return s;
}

/* R2/cr/auxiliary.h:274-316 quat_to_rotmat_vjp; v_R[c][r] column-major */
static void sf_quat_vjp(const float* q, float vR[3][3], float* out) {
    float s = 1.0f / sqrtf(q[3] * q[3] + q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    float w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
    out[0] = 2.f * (x * (vR[1][2] - vR[2][1]) + y * (vR[2][0] - vR[0][2]) + z * (vR[0][1] - vR[1][0]));
    out[1] = 2.f * (-2.f * x * (vR[1][1] + vR[2][2]) + y * (vR[0][1] + vR[1][0]) + z * (vR[0][2] + vR[2][0]) + w * (vR[1][2] - vR[2][1]));
    out[2] = 2.f * (x * (vR[0][1] + vR[1][0]) - 2.f * y * (vR[0][0] + vR[2][2]) + z * (vR[1][2] + vR[2][1]) + w * (vR[2][0] - vR[0][2]));
    out[3] = 2.f * (x * (vR[0][2] + vR[2][0]) + y * (vR[1][2] + vR[2][1]) - 2.f * z * (vR[0][0] + vR[1][1]) + w * (vR[0][1] - vR[1][0]));
}

/* Backward: R2/cr/rasterizer_impl.cu:362-462.  dL_dout_others = gradient of all 7 auxiliary planes.
 * All outputs zero-initialised by the caller (R2/rasterize_points.cu:190-201). */
int sfo_backward(const void* h, int P, int R, const float* background, int width, int height, const float* means3D,
                 const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
                 const float* viewmatrix, const float* beams, const int* radii,
                 const float* dL_dpix, const float* dL_dothers,
                 float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
                 float* dL_dtransMat, float* dL_dtransMat_2dtemp, float* dL_dscale, float* dL_drot, float* gs_depth) {
    (void)scale_modifier;
    const sfo_state* s = (const sfo_state*)h;
    if (s->P != P || s->W != width || s->H != height || s->R != R) { snprintf(sfo_err, sizeof sfo_err, "backward: state mismatch"); return -1; }
    const int W = width, H = height, C = SF_CHANNELS;
    const long long N = (long long)W * H;
    const float pi = SF_PI;

    double* acc64 = NULL;
    if (sfo_accumulate_double) {
        acc64 = (double*)calloc((size_t)P * 22 + 1, sizeof(double));
        if (!acc64) { snprintf(sfo_err, sizeof sfo_err, "backward: out of memory (float64 sums)"); return -1; }
    }
    double* a_col = acc64;                                       /* [P][2] */
    double* a_nrm = acc64 ? acc64 + (size_t)P * 2 : NULL;        /* [P][3] */
    double* a_tm = acc64 ? acc64 + (size_t)P * 5 : NULL;         /* [P][9] */
    double* a_t2 = acc64 ? acc64 + (size_t)P * 14 : NULL;        /* [P][3] */
    double* a_m2 = acc64 ? acc64 + (size_t)P * 17 : NULL;        /* [P][4] */
    double* a_op = acc64 ? acc64 + (size_t)P * 21 : NULL;        /* [P]    */
#define SFO_ACC(arr, sh, idx, val) do { if (acc64) (sh)[idx] += (double)(val); else (arr)[idx] += (val); } while (0)

    /* K8': R2/cr/backward.cu:143-605 */
    for (int yy = 0; yy < H; yy++)
        for (int xx = 0; xx < W; xx++) {
            /* the reference accumulates with float atomics in scheduling order; the knob replays the pixels in reverse
             * so that tests can measure how wide that summation-order band is */
            const int y = sfo_reverse_pixel_order ? H - 1 - yy : yy, x = sfo_reverse_pixel_order ? W - 1 - xx : xx;
            const uint32_t tile = (uint32_t)(y / SF_BLOCK_Y) * s->gx + (uint32_t)(x / SF_BLOCK_X);
            const uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
            const long long pix = (long long)W * y + x;
            const sf3 p = sf_pixel_dir(x, y, W, H, beams);
            const float T_final = s->accum[pix];
            float T = T_final;
            double Td = (double)T;          /* sfo_set_accumulate_double(2): the chain below in float64 (see lgo_set_accumulate_double) */
            uint32_t contributor = r1 - r0;
            const int last_contributor = (int)s->n_contrib[pix];
            float accum_rec[SF_CHANNELS] = { 0 }, dL_dpixel[SF_CHANNELS];
            const int median_contributor = (int)s->n_contrib[pix + N];
            float dL_ddepth = dL_dothers[SF_DEPTH_OFFSET * N + pix], dL_daccum = dL_dothers[SF_ALPHA_OFFSET * N + pix];
            float dL_dreg = dL_dothers[SF_DISTORTION_OFFSET * N + pix];
            float dL_dnormal2D[3];
            for (int i = 0; i < 3; i++) dL_dnormal2D[i] = dL_dothers[(SF_NORMAL_OFFSET + i) * N + pix];
            float dL_dmedian_depth = dL_dothers[SF_MIDDEPTH_OFFSET * N + pix];
            float last_depth = 0, last_normal[3] = { 0 }, accum_depth_rec = 0, accum_alpha_rec = 0, accum_normal_rec[3] = { 0 };
            const float final_D = s->accum[pix + N], final_A = 1 - T_final;
            float last_dL_dT = 0;
            for (int i = 0; i < C; i++) dL_dpixel[i] = dL_dpix[i * N + pix];
            float last_alpha = 0, last_color[SF_CHANNELS] = { 0 };
            for (uint32_t kk = r1; kk > r0; kk--) {
                const uint32_t g = s->point_list[kk - 1];
                contributor--;
                if ((int)contributor >= last_contributor) continue;
                sf_pair q;
                if (!sf_pair_geom(s, g, x, y, p, &q)) continue;
                const float c_d = q.depth;
                float power = -0.5f * q.rho;
                if (power > 0.0f) continue;
                const float G = sf_expf(power);
                const float aa = q.opa * G;
                const float alpha = 0.99f < aa ? 0.99f : aa;
                if (alpha < 1.0f / 255.0f) continue;
                if (sfo_accumulate_double >= 2) { Td = Td / (double)(1.f - alpha); T = (float)Td; } else T = T / (1.f - alpha);
                const float dchannel_dcolor = alpha * T;
                float dL_dalpha = 0.0f;
                for (int ch = 0; ch < C; ch++) {
                    const float c = colors_precomp[g * C + ch];
                    accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                    last_color[ch] = c;
                    const float dL_dchannel = dL_dpixel[ch];
                    if (ch == 0) dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;       /* :358-359 ray-drop channel detached */
                    SFO_ACC(dL_dcolor, a_col, g * C + ch, dchannel_dcolor * dL_dchannel);
                }
                float dL_dz = 0.0f, dL_dweight = 0;
                const float m_d = SF_FAR_N / (SF_FAR_N - SF_NEAR_N) * (1 - SF_NEAR_N / c_d);
                const float dmd_dd = (SF_FAR_N * SF_NEAR_N) / ((SF_FAR_N - SF_NEAR_N) * c_d * c_d);
                if ((int)contributor == median_contributor - 1) dL_dz += dL_dmedian_depth;
                dL_dweight += 0;                                                          /* DETACH_WEIGHT */
                dL_dalpha += dL_dweight - last_dL_dT;
                last_dL_dT = dL_dweight * alpha + (1 - alpha) * last_dL_dT;
                const float dL_dmd = 2.0f * (T * alpha) * (m_d * final_A - final_D) * dL_dreg;
                dL_dz += dL_dmd * dmd_dd;
                accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
                last_depth = c_d;
                dL_dalpha += (c_d - accum_depth_rec) * dL_ddepth;
                accum_alpha_rec = (float)((double)last_alpha * 1.0 + (double)((1.f - last_alpha) * accum_alpha_rec));
                dL_dalpha += (1 - accum_alpha_rec) * dL_daccum;
                for (int ch = 0; ch < 3; ch++) {
                    accum_normal_rec[ch] = last_alpha * last_normal[ch] + (1.f - last_alpha) * accum_normal_rec[ch];
                    last_normal[ch] = q.normal[ch];
                    dL_dalpha += (q.normal[ch] - accum_normal_rec[ch]) * dL_dnormal2D[ch];
                    SFO_ACC(dL_dnormal, a_nrm, g * 3 + ch, alpha * T * dL_dnormal2D[ch]);
                }
                dL_dalpha *= T;
                last_alpha = alpha;
                float bg_dot_dpixel = 0;
                for (int i = 0; i < C; i++) bg_dot_dpixel += background[i] * dL_dpixel[i];
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                const float dL_dG = q.opa * dL_dalpha;
                dL_dz += alpha * T * dL_ddepth;
                const sf3 Tw = q.Tw, Tu = q.Tu, Tv = q.Tv, dp = q.dp;
                float beta_temp = pi - sf_atan2(Tw.y, Tw.x);
                float alpha_temp = sf_atan2(Tw.z, sqrtf(Tw.x * Tw.x + Tw.y * Tw.y));
                float grad_alpha = fabsf(beams[H - 1] - beams[0]) / ((float)H - 1);
                if (q.rho3d <= q.rho2d && q.real_depth > 0) {
                    float dL_dD = dL_dz;
                    float dD_dlambda2 = 1.0f, L2_p = 1.0f;
                    float sum_p_normal = p.x * q.normal[0] + p.y * q.normal[1] + p.z * q.normal[2];
                    float sum_Tw_normal = Tw.x * q.normal[0] + Tw.y * q.normal[1] + Tw.z * q.normal[2];
                    sf3 dl_dTw = { q.normal[0] * L2_p / sum_p_normal, q.normal[1] * L2_p / sum_p_normal, q.normal[2] * L2_p / sum_p_normal };
                    sf3 dl_dn = { (Tw.x * L2_p * sum_p_normal - sum_Tw_normal * L2_p * p.x) / (sum_p_normal * sum_p_normal),
                                  (Tw.y * L2_p * sum_p_normal - sum_Tw_normal * L2_p * p.y) / (sum_p_normal * sum_p_normal),
                                  (Tw.z * L2_p * sum_p_normal - sum_Tw_normal * L2_p * p.z) / (sum_p_normal * sum_p_normal) };
                    sf2 dL_ds = { dL_dG * -G * q.s.x, dL_dG * -G * q.s.y };
                    const float TuTu = q.Tu_Tu, TvTv = q.Tv_Tv;
                    sf3 dsx_dTu = { (dp.x * TuTu - q.dp_Tu * 2 * Tu.x) / (TuTu * TuTu), (dp.y * TuTu - q.dp_Tu * 2 * Tu.y) / (TuTu * TuTu),
                                    (dp.z * TuTu - q.dp_Tu * 2 * Tu.z) / (TuTu * TuTu) };
                    sf3 dsx_ddp = { Tu.x / TuTu, Tu.y / TuTu, Tu.z / TuTu };
                    sf3 dsy_dTv = { (dp.x * TvTv - q.dp_Tv * 2 * Tv.x) / (TvTv * TvTv), (dp.y * TvTv - q.dp_Tv * 2 * Tv.y) / (TvTv * TvTv),
                                    (dp.z * TvTv - q.dp_Tv * 2 * Tv.z) / (TvTv * TvTv) };
                    sf3 dsy_ddp = { Tv.x / TvTv, Tv.y / TvTv, Tv.z / TvTv };
                    /* "- 1.0" is a double literal: these three entries are evaluated in double and rounded (:484-498) */
                    sf3 ddpx_dTw = { (float)((double)(p.x * dD_dlambda2 * dl_dTw.x) - 1.0), p.x * dD_dlambda2 * dl_dTw.y, p.x * dD_dlambda2 * dl_dTw.z };
                    sf3 ddpy_dTw = { p.y * dD_dlambda2 * dl_dTw.x, (float)((double)(p.y * dD_dlambda2 * dl_dTw.y) - 1.0), p.y * dD_dlambda2 * dl_dTw.z };
                    sf3 ddpz_dTw = { p.z * dD_dlambda2 * dl_dTw.x, p.z * dD_dlambda2 * dl_dTw.y, (float)((double)(p.z * dD_dlambda2 * dl_dTw.z) - 1.0) };
                    sf3 ddpx_dn = { p.x * dD_dlambda2 * dl_dn.x, p.x * dD_dlambda2 * dl_dn.y, p.x * dD_dlambda2 * dl_dn.z };
                    sf3 ddpy_dn = { p.y * dD_dlambda2 * dl_dn.x, p.y * dD_dlambda2 * dl_dn.y, p.y * dD_dlambda2 * dl_dn.z };
                    sf3 ddpz_dn = { p.z * dD_dlambda2 * dl_dn.x, p.z * dD_dlambda2 * dl_dn.y, p.z * dD_dlambda2 * dl_dn.z };
#define SF_MIX(a, bx, by, bz, comp) (a.x * bx.comp + a.y * by.comp + a.z * bz.comp)
                    sf3 dsx_dTw = { SF_MIX(dsx_ddp, ddpx_dTw, ddpy_dTw, ddpz_dTw, x), SF_MIX(dsx_ddp, ddpx_dTw, ddpy_dTw, ddpz_dTw, y), SF_MIX(dsx_ddp, ddpx_dTw, ddpy_dTw, ddpz_dTw, z) };
                    sf3 dsy_dTw = { SF_MIX(dsy_ddp, ddpx_dTw, ddpy_dTw, ddpz_dTw, x), SF_MIX(dsy_ddp, ddpx_dTw, ddpy_dTw, ddpz_dTw, y), SF_MIX(dsy_ddp, ddpx_dTw, ddpy_dTw, ddpz_dTw, z) };
                    sf3 dsx_dn = { SF_MIX(dsx_ddp, ddpx_dn, ddpy_dn, ddpz_dn, x), SF_MIX(dsx_ddp, ddpx_dn, ddpy_dn, ddpz_dn, y), SF_MIX(dsx_ddp, ddpx_dn, ddpy_dn, ddpz_dn, z) };
                    sf3 dsy_dn = { SF_MIX(dsy_ddp, ddpx_dn, ddpy_dn, ddpz_dn, x), SF_MIX(dsy_ddp, ddpx_dn, ddpy_dn, ddpz_dn, y), SF_MIX(dsy_ddp, ddpx_dn, ddpy_dn, ddpz_dn, z) };
#undef SF_MIX
                    sf3 gTu = { dL_ds.x * dsx_dTu.x, dL_ds.x * dsx_dTu.y, dL_ds.x * dsx_dTu.z };
                    sf3 gTv = { dL_ds.y * dsy_dTv.x, dL_ds.y * dsy_dTv.y, dL_ds.y * dsy_dTv.z };
                    sf3 gTw = { dL_ds.x * dsx_dTw.x + dL_ds.y * dsy_dTw.x + dL_dD * dD_dlambda2 * dl_dTw.x,
                                dL_ds.x * dsx_dTw.y + dL_ds.y * dsy_dTw.y + dL_dD * dD_dlambda2 * dl_dTw.y,
                                dL_ds.x * dsx_dTw.z + dL_ds.y * dsy_dTw.z + dL_dD * dD_dlambda2 * dl_dTw.z };
                    sf3 gN = { dL_ds.x * dsx_dn.x + dL_ds.y * dsy_dn.x + dL_dD * dD_dlambda2 * dl_dn.x,
                               dL_ds.x * dsx_dn.y + dL_ds.y * dsy_dn.y + dL_dD * dD_dlambda2 * dl_dn.y,
                               dL_ds.x * dsx_dn.z + dL_ds.y * dsy_dn.z + dL_dD * dD_dlambda2 * dl_dn.z };
                    SFO_ACC(dL_dtransMat, a_tm, 9 * g + 0, gTu.x); SFO_ACC(dL_dtransMat, a_tm, 9 * g + 1, gTu.y); SFO_ACC(dL_dtransMat, a_tm, 9 * g + 2, gTu.z); SFO_ACC(dL_dtransMat, a_tm, 9 * g + 3, gTv.x); SFO_ACC(dL_dtransMat, a_tm, 9 * g + 4, gTv.y); SFO_ACC(dL_dtransMat, a_tm, 9 * g + 5, gTv.z);
                    SFO_ACC(dL_dtransMat, a_tm, 9 * g + 6, gTw.x); SFO_ACC(dL_dtransMat, a_tm, 9 * g + 7, gTw.y); SFO_ACC(dL_dtransMat, a_tm, 9 * g + 8, gTw.z);
                    SFO_ACC(dL_dtransMat_2dtemp, a_t2, 3 * g, fabsf(gTw.x)); SFO_ACC(dL_dtransMat_2dtemp, a_t2, 3 * g + 1, fabsf(gTw.y)); SFO_ACC(dL_dtransMat_2dtemp, a_t2, 3 * g + 2, fabsf(gTw.z));
                    SFO_ACC(dL_dnormal, a_nrm, 3 * g, gN.x); SFO_ACC(dL_dnormal, a_nrm, 3 * g + 1, gN.y); SFO_ACC(dL_dnormal, a_nrm, 3 * g + 2, gN.z);
                    /* :564-577 heuristic mean2D statistics; "/float(W)*2.0*pi" evaluates in double */
                    float mx = (float)(fabs((double)(gTw.x * sinf(beta_temp) * cosf(alpha_temp) / (float)W) * 2.0 * (double)pi) +
                                       fabs((double)(gTw.y * cosf(beta_temp) * cosf(alpha_temp) / (float)W) * 2.0 * (double)pi));
                    mx = (float)((double)(mx * q.rho_r) * 0.5 * (double)(float)W);
                    float my = fabsf(gTw.x * sinf(alpha_temp) * cosf(beta_temp) * grad_alpha) + fabsf(gTw.y * sinf(alpha_temp) * sinf(beta_temp) * grad_alpha) +
                               fabsf(gTw.z * cosf(alpha_temp) * grad_alpha);
                    my = (float)((double)(my * q.rho_r) * 0.5 * (double)(float)H);
                    SFO_ACC(dL_dmean2D, a_m2, 4 * g, mx); SFO_ACC(dL_dmean2D, a_m2, 4 * g + 1, my); SFO_ACC(dL_dmean2D, a_m2, 4 * g + 2, fabsf(mx)); SFO_ACC(dL_dmean2D, a_m2, 4 * g + 3, fabsf(my));
                } else {
                    const float dG_ddelx = -G * SF_FILTER_INV_SQ * 40 * q.d.x;
                    const float dG_ddely = -G * SF_FILTER_INV_SQ * 100 * q.d.y;
                    /* "* 0.5 * W": double 0.5 -> double product, int W (:582-585) */
                    float m0 = (float)((double)(dL_dG * dG_ddelx) * 0.5 * (double)W), m1 = (float)((double)(dL_dG * dG_ddely) * 0.5 * (double)H);
                    SFO_ACC(dL_dmean2D, a_m2, 4 * g, m0); SFO_ACC(dL_dmean2D, a_m2, 4 * g + 1, m1);
                    SFO_ACC(dL_dmean2D, a_m2, 4 * g + 2, (float)fabs((double)(dL_dG * dG_ddelx) * 0.5 * (double)W));
                    SFO_ACC(dL_dmean2D, a_m2, 4 * g + 3, (float)fabs((double)(dL_dG * dG_ddely) * 0.5 * (double)H));
                    float rho_xy2 = sqrtf(Tw.x * Tw.x + Tw.y * Tw.y);
                    float ddelx_dpx = (float)W / (2 * pi) * Tw.y / (rho_xy2 * rho_xy2);
                    float ddelx_dpy = (float)(-1.0 * (double)(float)W / (double)(2 * pi) * (double)Tw.x / (double)(rho_xy2 * rho_xy2));
                    float ddely_dpx = (float)((double)grad_alpha * (-1.0) * (double)Tw.z * (double)Tw.x / (double)(q.rho_r * q.rho_r * rho_xy2));
                    float ddely_dpy = (float)((double)grad_alpha * (-1.0) * (double)Tw.z * (double)Tw.y / (double)(q.rho_r * q.rho_r * rho_xy2));
                    float ddely_dpz = grad_alpha * rho_xy2 / (q.rho_r * q.rho_r);
                    SFO_ACC(dL_dtransMat, a_tm, 9 * g + 6, dL_dz * (Tw.x / q.rho_r) + dL_dG * (dG_ddelx * ddelx_dpx + dG_ddely * ddely_dpx));
                    SFO_ACC(dL_dtransMat, a_tm, 9 * g + 7, dL_dz * (Tw.y / q.rho_r) + dL_dG * (dG_ddelx * ddelx_dpy + dG_ddely * ddely_dpy));
                    SFO_ACC(dL_dtransMat, a_tm, 9 * g + 8, dL_dz * (Tw.z / q.rho_r) + dL_dG * (dG_ddely * ddely_dpz));
                }
                SFO_ACC(dL_dopacity, a_op, g, G * dL_dalpha);
            }
        }

#undef SFO_ACC
    if (acc64) {
        for (long long i = 0; i < (long long)P * 2; i++) dL_dcolor[i] = (float)((double)dL_dcolor[i] + a_col[i]);
        for (long long i = 0; i < (long long)P * 3; i++) dL_dnormal[i] = (float)((double)dL_dnormal[i] + a_nrm[i]);
        for (long long i = 0; i < (long long)P * 9; i++) dL_dtransMat[i] = (float)((double)dL_dtransMat[i] + a_tm[i]);
        for (long long i = 0; i < (long long)P * 3; i++) dL_dtransMat_2dtemp[i] = (float)((double)dL_dtransMat_2dtemp[i] + a_t2[i]);
        for (long long i = 0; i < (long long)P * 4; i++) dL_dmean2D[i] = (float)((double)dL_dmean2D[i] + a_m2[i]);
        for (long long i = 0; i < (long long)P; i++) dL_dopacity[i] = (float)((double)dL_dopacity[i] + a_op[i]);
        free(acc64);
    }

    /* K10': R2/cr/backward.cu:607-749 compute_cylinder_transmat_aabb */
    for (int idx = 0; idx < P; idx++) {
        if (!(radii[idx] > 0)) continue;
        const float* rot = rotations + 4 * idx; const float* scale = scales + 2 * idx;
        float Rm[3][3];
        sf_quat_to_rotmat(rot, Rm);
        sf3 p_orig = { means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
        sf3 pv = sf_point4x3(p_orig, viewmatrix);
        /* L = R * scale_to_mat(scale, 1.0f): the backward ignores scale_modifier (:632) */
        sf3 l2 = { Rm[2][0], Rm[2][1], Rm[2][2] };
        sf3 normal = sf_vec4x3(l2, viewmatrix);
        /* dL_dT (glm) = mat3(g0,g3,g6, g1,g4,g7, g2,g5,g8): column c = (g[c], g[3+c], g[6+c]); dL_dM = world2view * transpose(dL_dT) */
        const float* g = dL_dtransMat + 9 * idx;
        /* transpose(dL_dT)[c][r] = dL_dT[r][c] = g[3*c + r]  -> column c of the transpose = (g[3c], g[3c+1], g[3c+2]) */
        float dL_dM[3][4];
        for (int c = 0; c < 3; c++)
            for (int r = 0; r < 4; r++) {
                /* world2view (mat3x4) column k = (vm[k], vm[4+k], vm[8+k], vm[12+k]); result[c][r] = sum_k w2v[k][r] * tr[c][k] */
                dL_dM[c][r] = viewmatrix[4 * r + 0] * g[3 * c + 0] + viewmatrix[4 * r + 1] * g[3 * c + 1] + viewmatrix[4 * r + 2] * g[3 * c + 2];
            }
        sf3 gn = { dL_dnormal[3 * idx], dL_dnormal[3 * idx + 1], dL_dnormal[3 * idx + 2] };
        sf3 dL_dtn = sf_vec4x3_t(gn, viewmatrix);
        gs_depth[idx] = sqrtf(pv.x * pv.x + pv.z * pv.z);                       /* :670 (x,z only) */
        float cs = -(pv.x * normal.x + pv.y * normal.y + pv.z * normal.z);
        float mult = cs > 0 ? 1.f : -1.f;
        dL_dtn.x *= mult; dL_dtn.y *= mult; dL_dtn.z *= mult;
        float dL_dRS[3][3] = { { dL_dM[0][0], dL_dM[0][1], dL_dM[0][2] }, { dL_dM[1][0], dL_dM[1][1], dL_dM[1][2] }, { dL_dtn.x, dL_dtn.y, dL_dtn.z } };
        float dL_dR[3][3];
        for (int r = 0; r < 3; r++) { dL_dR[0][r] = dL_dRS[0][r] * scale[0]; dL_dR[1][r] = dL_dRS[1][r] * scale[1]; dL_dR[2][r] = dL_dRS[2][r]; }
        sf_quat_vjp(rot, dL_dR, dL_drot + 4 * idx);
        dL_dscale[2 * idx] = dL_dRS[0][0] * Rm[0][0] + dL_dRS[0][1] * Rm[0][1] + dL_dRS[0][2] * Rm[0][2];
        dL_dscale[2 * idx + 1] = dL_dRS[1][0] * Rm[1][0] + dL_dRS[1][1] * Rm[1][1] + dL_dRS[1][2] * Rm[1][2];
        dL_dmean3D[3 * idx] = dL_dM[2][0]; dL_dmean3D[3 * idx + 1] = dL_dM[2][1]; dL_dmean3D[3 * idx + 2] = dL_dM[2][2];
    }
    return 0;
}

/* R2/cr/rasterizer_impl.cu:465-518 visible_filter */
int sfo_visible_filter(int P, int width, int height, const float* means3D, const float* scales, float scale_modifier,
                       const float* rotations, const float* viewmatrix, const float* beams, int far_, int near_, int* radii) {
    sfo_state s; memset(&s, 0, sizeof s);
    s.P = P; s.W = width; s.H = height;
    s.gx = (width + SF_BLOCK_X - 1) / SF_BLOCK_X; s.gy = (height + SF_BLOCK_Y - 1) / SF_BLOCK_Y;
    size_t Pz = P > 0 ? (size_t)P : 1;
    s.transMat = (float*)calloc(Pz * 9, 4); s.radii_xy = (int*)calloc(Pz * 2, 4);
    for (int i = 0; i < P; i++) sf_preprocess_one(i, 1, &s, means3D, scales, scale_modifier, rotations, NULL, viewmatrix, beams, far_, near_, radii);
    free(s.transMat); free(s.radii_xy);
    return 0;
}
