/*
 * lidargs_oracle.c -- CPU restatement of the LiDAR-GS "laser-beam splatting" rasterizer.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product path (lidar-gs_amd/) never does.
 *
 * PARITY STATUS: "parity unpinned" for the CUDA arithmetic.  The reference ships no tests,
 * golden vectors or fixtures for this path, and its kernels need nvcc,
 * the CUDA runtime headers and CUB, none of which exist in this image, so the reference
 * cannot be built here without writing stand-ins for them.  What IS pinned: the range-view
 * geometry conventions (azimuth->column, beam->row, pixel->ray) against fixtures generated
 * by executing the reference's own numpy projector (tests/golden/make_rangeview_golden.py).
 *
 * Every function states the reference file:line it restates.  Paths are relative to
 * /root/reference/submodules/diff_lidargs_rasterization/ ("R3/"); "cr/" = cuda_rasterizer/.
 *
 * Arithmetic: plain C, single thread, compiled with -ffp-contract=off so every expression
 * rounds exactly as written in the reference source (float ops in float, the handful of
 * double promotions of SURVEY.md Appendix A.5 in double).  GLM's column-major mat3 product
 * order (third_party/glm/glm/detail/type_mat3x3.inl:486-520) is reproduced by m3_mul().
 * The cos/sin of the per-pixel ray table are evaluated in double and rounded to float (lgo_cosf / lgo_sinf):
 * the correctly rounded fp32 value, so the oracle does not inherit the host libm's last-ulp choices (glibc's
 * cosf/sinf differ from the correctly rounded result on ~1.5 % of the 2650 azimuths of a 64x2650 frame, CUDA's
 * on others; one ulp of a ray component moves a blend weight by up to 1e-3 at 50 m range).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define LGO_CHANNELS 2   /* cr/config.h:15 NUM_CHANNELS */
#define LGO_BLOCK_X 16   /* cr/config.h:16 */
#define LGO_BLOCK_Y 1    /* cr/config.h:17 */

static const float LGO_PI = 3.14159265358979323846f; /* cr/forward.cu:21, cr/backward.cu:18 */

/* Test knob: the error bar between ANY two conforming evaluations of the reference source.  The reference calls the float
 * overloads of cos, sin (cr/forward.cu:589-591, cr/backward.cu:659-661), atan2 (cr/forward.cu:333,:336), tan (:361-362) and
 * exp (cr/forward.cu:604, cr/backward.cu:676) and is built without -use_fast_math (R3/setup.py:29), i.e. CUDA libdevice, whose
 * documented maximum errors are cosf 1 ulp, sinf 1 ulp, atan2f 2 ulp, tanf 4 ulp, expf 2 ulp -- a different last-bit choice
 * from glibc's and from this file's correctly rounded cos/sin.  (sqrtf, division, fma-free +,-,* are IEEE-exact on both.)
 * With the knob on, every such result is moved by an integer number of ulps within the documented bound of its function:
 *   mode 1: pseudo-random in [-amp, +amp], a pure function of (input bits, seed) so that the forward and the backward
 *           re-evaluate a pair identically, as they do on any one platform;   mode 2: always +amp;   mode 3: always -amp.
 * tests/test_ulp_band_cpu.py runs the oracle against itself under this knob: that spread is what "the oracle" can promise
 * about the CUDA reference's outputs, and what the 1e-4 parity bar has to be read against. */
static int lgo_ulp_mode = 0;
static uint32_t lgo_ulp_seed = 0;
void lgo_set_ulp_perturbation(int mode, unsigned seed) { lgo_ulp_mode = mode; lgo_ulp_seed = seed; }
static float lgo_perturb(float r, float in1, float in2, int amp) {
    if (!lgo_ulp_mode || !(r == r) || r == 0.0f || isinf(r)) return r;
    int k;
    if (lgo_ulp_mode == 2) k = amp;
    else if (lgo_ulp_mode == 3) k = -amp;
    else {
        uint32_t a, b;
        memcpy(&a, &in1, 4); memcpy(&b, &in2, 4);
        uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (lgo_ulp_seed + (uint32_t)amp) * 0xC2B2AE3Du;
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
        k = (int)(h % (uint32_t)(2 * amp + 1)) - amp;
    }
    int32_t bits;
    memcpy(&bits, &r, 4);
    bits += (bits < 0) ? -k : k;          /* sign-magnitude: one step of the integer = one ulp away from / towards zero */
    float out;
    memcpy(&out, &bits, 4);
    return (out == out && !isinf(out)) ? out : r;
}
static float lgo_cosf(float x) { return lgo_perturb((float)cos((double)x), x, 1.0f, 1); }
static float lgo_sinf(float x) { return lgo_perturb((float)sin((double)x), x, 2.0f, 1); }
static float lgo_atan2f(float y, float x) { return lgo_perturb(atan2f(y, x), y, x, 2); }
static float lgo_tanf(float x) { return lgo_perturb(tanf(x), x, 3.0f, 4); }
static float lgo_expf(float x) { return lgo_perturb(expf(x), x, 4.0f, 2); }
static const float LGO_RAY_DIV = 0.002f;             /* cr/forward.cu:22 Ray_Divergence_Angle */

typedef struct { float x, y, z; } f3;

/* GLM-style column-major 3x3: c[i][j] = column i, row j (what glm writes as M[i][j]). */
typedef struct { float c[3][3]; } m3;

/* glm::mat3(a,b,c, d,e,f, g,h,i): consecutive triples are COLUMNS. */
static m3 m3_make(float a, float b, float c, float d, float e, float f, float g, float h, float i) {
    m3 r;
    r.c[0][0] = a; r.c[0][1] = b; r.c[0][2] = c;
    r.c[1][0] = d; r.c[1][1] = e; r.c[1][2] = f;
    r.c[2][0] = g; r.c[2][1] = h; r.c[2][2] = i;
    return r;
}

/* glm mat3*mat3, evaluation order of type_mat3x3.inl:486-520:
 * R[c][r] = A[0][r]*B[c][0] + A[1][r]*B[c][1] + A[2][r]*B[c][2], summed left to right. */
static m3 m3_mul(m3 a, m3 b) {
    m3 r;
    for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++)
            r.c[c][rr] = a.c[0][rr] * b.c[c][0] + a.c[1][rr] * b.c[c][1] + a.c[2][rr] * b.c[c][2];
    return r;
}

static m3 m3_transpose(m3 a) {
    m3 r;
    for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++) r.c[c][rr] = a.c[rr][c];
    return r;
}

/* cr/auxiliary.h:94-102 transformPoint4x3 */
static f3 transform_point_4x3(f3 p, const float* m) {
    f3 t;
    t.x = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];
    t.y = m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13];
    t.z = m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14];
    return t;
}

/* cr/auxiliary.h:125-133 transformVec4x3Transpose */
static f3 transform_vec_4x3_transpose(f3 p, const float* m) {
    f3 t;
    t.x = m[0] * p.x + m[1] * p.y + m[2] * p.z;
    t.y = m[4] * p.x + m[5] * p.y + m[6] * p.z;
    t.z = m[8] * p.x + m[9] * p.y + m[10] * p.z;
    return t;
}

/* cr/forward.cu:80-88 normalize_f3 (zero vector stays zero) */
static f3 normalize_fwd(f3 v) {
    float length = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    if (length > 0.0f) { v.x /= length; v.y /= length; v.z /= length; }
    return v;
}

/* cr/backward.cu:20-29 norm_f3 */
static f3 normalize_bwd(f3 v) {
    if (v.x * v.x + v.y * v.y + v.z * v.z == 0) return v;
    float length = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    if (length > 0.0f) { v.x /= length; v.y /= length; v.z /= length; }
    return v;
}

/* cr/auxiliary.h:41-63 find_closest_label: clamp at the ends, else bisect-left. */
static int find_closest_label(const float* beams, float angle, int length) {
    if (angle >= beams[length - 1]) return length - 1;
    else if (angle <= beams[0]) return 0;
    int lo = 0, hi = length;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (beams[mid] < angle) lo = mid + 1; else hi = mid;
    }
    return lo;
}

static unsigned umin_u(unsigned a, unsigned b) { return a < b ? a : b; }
static int imax_i(int a, int b) { return a > b ? a : b; }

/* cr/auxiliary.h:80-92 getRect_lidar.  BLOCK_Y == 1, so "/ BLOCK_Y" are no-ops; x truncates,
 * y rounds half away from zero; no azimuth wrap. */
static void get_rect_lidar(float px, float py, int rx, int ry, unsigned gx, unsigned gy,
                           unsigned* xmin, unsigned* ymin, unsigned* xmax, unsigned* ymax) {
    *xmin = umin_u(gx, (unsigned)imax_i(0, (int)((px - rx) / LGO_BLOCK_X)));
    *ymin = umin_u(gy, (unsigned)imax_i(0, (int)(roundf((py - ry) / LGO_BLOCK_Y))));
    *xmax = umin_u(gx, (unsigned)imax_i(0, (int)((px + rx + LGO_BLOCK_X - 1) / LGO_BLOCK_X)));
    float a = roundf(py + ry / LGO_BLOCK_Y);
    float b = roundf(py / LGO_BLOCK_Y) + 1;
    *ymax = umin_u(gy, (unsigned)imax_i(0, (int)(a > b ? a : b)));
}

/* get_rect_lidar on n caller-supplied inputs: what tests/ hold lidargs_debug_rects (the device function) against, bit for bit */
void lgo_rects(int n, const float* p_cr, const int* r_xy, int gx, int gy, int* rects) {
    for (int i = 0; i < n; i++) {
        unsigned xmin, ymin, xmax, ymax;
        get_rect_lidar(p_cr[2 * i], p_cr[2 * i + 1], r_xy[2 * i], r_xy[2 * i + 1], (unsigned)gx, (unsigned)gy, &xmin, &ymin, &xmax, &ymax);
        rects[4 * i] = (int)xmin; rects[4 * i + 1] = (int)ymin; rects[4 * i + 2] = (int)xmax; rects[4 * i + 3] = (int)ymax;
    }
}

/* cr/forward.cu:216-253 computeCov3D (quaternion NOT normalised, :228) */
static void compute_cov3d(const float* scale, float mod, const float* rot, float* cov3D) {
    m3 S = m3_make(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.c[0][0] = mod * scale[0];
    S.c[1][1] = mod * scale[1];
    S.c[2][2] = mod * scale[2];
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    m3 R = m3_make(
        1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
        2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
        2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    m3 M = m3_mul(S, R);
    m3 Sigma = m3_mul(m3_transpose(M), M);
    cov3D[0] = Sigma.c[0][0]; cov3D[1] = Sigma.c[0][1]; cov3D[2] = Sigma.c[0][2];
    cov3D[3] = Sigma.c[1][1]; cov3D[4] = Sigma.c[1][2]; cov3D[5] = Sigma.c[2][2];
}

/* cr/forward.cu:95-119 _proj_2basis: tangent basis (u1,u2) at dir = p/|p| */
static void proj_2basis(f3 mean, f3* u1o, f3* u2o) {
    f3 dir = normalize_fwd(mean);
    f3 u1 = { dir.y, -dir.x, 0 };
    u1 = normalize_fwd(u1);
    f3 u2 = { dir.y * u1.z - dir.z * u1.y, dir.z * u1.x - dir.x * u1.z, dir.x * u1.y - dir.y * u1.x };
    *u1o = u1; *u2o = u2;
}

/* cr/forward.cu:146-169 computeCov2D_lidar; returns (cov00, cov01, cov11) with +0.01 low-pass */
static f3 compute_cov2d_lidar(f3 u1, f3 u2, const float* cov3D, const float* vm) {
    m3 P = m3_make(u1.x, u1.y, u1.z, u2.x, u2.y, u2.z, 0, 0, 0);
    m3 W = m3_make(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
    m3 T = m3_mul(W, P);
    m3 Vrk = m3_make(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    m3 cov = m3_mul(m3_mul(m3_transpose(T), m3_transpose(Vrk)), T);
    cov.c[0][0] += 0.01f;
    cov.c[1][1] += 0.01f;
    f3 r = { cov.c[0][0], cov.c[0][1], cov.c[1][1] };
    return r;
}

/* ------------------------------------------------------------------------------------------
 * State kept between forward and backward (the reference's geom/binning/image chunks,
 * cr/rasterizer_impl.h:21-76).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int P, W, H, R;
    unsigned gx, gy;
    float* depths;          /* P   */
    float* means2D;         /* 2P  (p_c, p_r) */
    float* cov3D;           /* 6P  */
    float* conic_opacity;   /* 4P  */
    float* basis_u1;        /* 3P  */
    float* basis_u2;        /* 3P  */
    float* sphere;          /* 3P  */
    uint32_t* tiles_touched;/* P   */
    uint32_t* point_offsets;/* P   */
    int* radii_xy;          /* 2P  */
    uint64_t* keys;         /* R sorted */
    uint32_t* point_list;   /* R sorted */
    uint32_t* ranges;       /* 2*gx*gy */
    float* final_T;         /* W*H */
    uint32_t* n_contrib;    /* W*H */
} lgo_state;

enum {
    LGO_A_DEPTHS = 0, LGO_A_MEANS2D, LGO_A_COV3D, LGO_A_CONIC_OPACITY, LGO_A_BASIS_U1, LGO_A_BASIS_U2,
    LGO_A_SPHERE, LGO_A_TILES_TOUCHED, LGO_A_POINT_OFFSETS, LGO_A_RADII_XY, LGO_A_KEYS, LGO_A_POINT_LIST,
    LGO_A_RANGES, LGO_A_FINAL_T, LGO_A_N_CONTRIB
};

static char lgo_err[256] = "";
/* Test knob: traverse pixels in reverse raster order in the backward blend.  The reference sums
 * per-Gaussian gradients with float atomicAdd in scheduling order (R3/cr/backward.cu:702-788), so
 * its own result is only defined up to summation order; tests use this knob to measure that band. */
static int lgo_reverse_pixel_order = 0;
void lgo_set_reverse_pixel_order(int on) { lgo_reverse_pixel_order = on; }
/* Test knob (round 6): the per-Gaussian sums of the backward blend in float64.  Every TERM stays the float32 value the reference
 * computes per (pixel, Gaussian); only the sum -- which the reference takes with float atomicAdd in whatever order the hardware
 * schedules (R3/cr/backward.cu:702-788) and this restatement in raster order -- is made exact.  It is the centre every conforming
 * summation order scatters around: an implementation whose result is closer to it than the raster-order float32 sum is, is not
 * "off" where the two float32 sums disagree. */
static int lgo_accumulate_double = 0;
void lgo_set_accumulate_double(int on) { lgo_accumulate_double = on; }
/* mode 2 (diagnostic): additionally the backward's transmittance chain T = T / (1 - alpha) (R3/cr/backward.cu:676) is carried in float64
 * and rounded per entry: the reference walks back from T_final with one float32 division per entry, so its T drifts from the forward's
 * by ~sqrt(entries) half-ulps -- 1300-entry lists (scale modifier 30): ~2e-6, which a cancelling gradient row amplifies.  An
 * implementation that restarts the chain at stored per-segment values does not share that drift. */
const char* lgo_last_error(void) { return lgo_err; }

void lgo_free(void* h) {
    lgo_state* s = (lgo_state*)h;
    if (!s) return;
    free(s->depths); free(s->means2D); free(s->cov3D); free(s->conic_opacity); free(s->basis_u1);
    free(s->basis_u2); free(s->sphere); free(s->tiles_touched); free(s->point_offsets); free(s->radii_xy);
    free(s->keys); free(s->point_list); free(s->ranges); free(s->final_T); free(s->n_contrib);
    free(s);
}

int lgo_num_rendered(const void* h) { return ((const lgo_state*)h)->R; }

/* Element counts are in units of the array's scalar type. */
const void* lgo_state_array(const void* h, int which, long long* count) {
    const lgo_state* s = (const lgo_state*)h;
    long long P = s->P, N = (long long)s->W * s->H, T = (long long)s->gx * s->gy, R = s->R;
    switch (which) {
        case LGO_A_DEPTHS: *count = P; return s->depths;
        case LGO_A_MEANS2D: *count = 2 * P; return s->means2D;
        case LGO_A_COV3D: *count = 6 * P; return s->cov3D;
        case LGO_A_CONIC_OPACITY: *count = 4 * P; return s->conic_opacity;
        case LGO_A_BASIS_U1: *count = 3 * P; return s->basis_u1;
        case LGO_A_BASIS_U2: *count = 3 * P; return s->basis_u2;
        case LGO_A_SPHERE: *count = 3 * P; return s->sphere;
        case LGO_A_TILES_TOUCHED: *count = P; return s->tiles_touched;
        case LGO_A_POINT_OFFSETS: *count = P; return s->point_offsets;
        case LGO_A_RADII_XY: *count = 2 * P; return s->radii_xy;
        case LGO_A_KEYS: *count = R; return s->keys;
        case LGO_A_POINT_LIST: *count = R; return s->point_list;
        case LGO_A_RANGES: *count = 2 * T; return s->ranges;
        case LGO_A_FINAL_T: *count = N; return s->final_T;
        case LGO_A_N_CONTRIB: *count = N; return s->n_contrib;
    }
    *count = 0;
    return NULL;
}

/* ------------------------------------------------------------------------------------------
 * K1: cr/forward.cu:256-384 preprocessCUDA (one Gaussian).  `filter` selects the K2 variant
 * cr/forward.cu:388-497 filter_preprocessCUDA, which differs only in the elevation formula
 * (:456 vs :336) and in what it stores.
 * ---------------------------------------------------------------------------------------- */
static void preprocess_one(int idx, int filter, lgo_state* s,
                           const float* means3D, const float* scales, float scale_modifier,
                           const float* rotations, const float* opacities, const float* cov3D_precomp,
                           const float* vm, const float* beams, int far_, int near_, float shell_lo, float shell_hi, int* radii) {
    const int W = s->W, H = s->H;
    radii[idx] = 0;
    if (!filter) s->tiles_touched[idx] = 0;

    f3 p_orig = { means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2] };
    f3 pv = transform_point_4x3(p_orig, vm);
    float dist = sqrtf((pv.x) * (pv.x) + (pv.y) * (pv.y) + (pv.z) * (pv.z));
    if (dist >= far_ || dist <= near_) return;                      /* :304 int -> float compare */
    if (!(dist >= shell_lo && dist < shell_hi)) return;             /* multi-GPU range shell (no reference counterpart) */

    const float* cov3D;
    if (cov3D_precomp != NULL) cov3D = cov3D_precomp + idx * 6;
    else { compute_cov3d(scales + 3 * idx, scale_modifier, rotations + 4 * idx, s->cov3D + idx * 6); cov3D = s->cov3D + idx * 6; }

    f3 u1, u2;
    proj_2basis(pv, &u1, &u2);
    f3 cov = compute_cov2d_lidar(u1, u2, cov3D, vm);
    cov.x = cov.x / (dist * dist);
    cov.y = cov.y / (dist * dist);
    cov.z = cov.z / (dist * dist);
    float det = (cov.x * cov.z - cov.y * cov.y);
    if (det == 0.0f) return;
    float det_inv = 1.f / det;
    f3 conic = { cov.z * det_inv, -cov.y * det_inv, cov.x * det_inv };
    float mid = 0.5f * (cov.x + cov.z);
    /* :328-330 max(1e-9, float) and sqrt are DOUBLE; the sum rounds back to float. */
    float lambda1 = (float)((double)mid + sqrt(fmax(1e-9, (double)(mid * mid - det))));
    float lambda2 = (float)((double)mid - sqrt(fmax(1e-9, (double)(mid * mid - det))));
    float my_radius = (float)sqrt(fmax(1e-9, (double)(lambda1 > lambda2 ? lambda1 : lambda2)));

    float beta = LGO_PI - lgo_atan2f(pv.y, pv.x);
    float p_c = beta / (2 * LGO_PI / W);

    float alpha;
    if (!filter) alpha = lgo_atan2f(pv.z, sqrtf(pv.x * pv.x + pv.y * pv.y));                                /* :336 */
    else alpha = (float)atan2((double)pv.z, sqrt(fmax(1e-9, (double)(pv.x * pv.x + pv.y * pv.y))));      /* :456 */
    int p_r_int = find_closest_label(beams, alpha, H);
    float before = 0, after = 0, p_r = 0;
    if (p_r_int > 0) {
        before = beams[p_r_int - 1];
        after = beams[p_r_int];
        p_r = p_r_int - 1 + (alpha - before) / (after - before);
        if (alpha > (after + LGO_RAY_DIV * 2)) return;
    } else {
        before = beams[p_r_int];
        after = beams[p_r_int + 1];
        p_r = p_r_int + 1 + (alpha - after) / (after - before);
        if (alpha < (before - LGO_RAY_DIV * 2)) return;
    }
    p_r = H - p_r - 1;

    int my_radius_y = (int)ceilf(3.f * my_radius / lgo_tanf(fabsf(after - before)));
    int my_radius_x = (int)ceilf(3.f * my_radius / lgo_tanf(2 * LGO_PI / W));

    unsigned xmin, ymin, xmax, ymax;
    get_rect_lidar(p_c, p_r, my_radius_x, my_radius_y, s->gx, s->gy, &xmin, &ymin, &xmax, &ymax);
    if ((xmax - xmin) * (ymax - ymin) == 0) return;

    radii[idx] = my_radius_x > my_radius_y ? my_radius_x : my_radius_y;
    s->radii_xy[2 * idx + 0] = my_radius_x;
    s->radii_xy[2 * idx + 1] = my_radius_y;
    s->means2D[2 * idx + 0] = p_c;
    s->means2D[2 * idx + 1] = p_r;
    if (filter) return;

    s->conic_opacity[4 * idx + 0] = conic.x; s->conic_opacity[4 * idx + 1] = conic.y;
    s->conic_opacity[4 * idx + 2] = conic.z; s->conic_opacity[4 * idx + 3] = opacities[idx];
    s->depths[idx] = dist;
    s->basis_u1[3 * idx + 0] = u1.x; s->basis_u1[3 * idx + 1] = u1.y; s->basis_u1[3 * idx + 2] = u1.z;
    s->basis_u2[3 * idx + 0] = u2.x; s->basis_u2[3 * idx + 1] = u2.y; s->basis_u2[3 * idx + 2] = u2.z;
    s->sphere[3 * idx + 0] = pv.x / dist; s->sphere[3 * idx + 1] = pv.y / dist; s->sphere[3 * idx + 2] = pv.z / dist;
    s->tiles_touched[idx] = (ymax - ymin) * (xmax - xmin);
}

/* cr/rasterizer_impl.cu:35-50 getHigherMsb */
static uint32_t get_higher_msb(uint32_t n) {
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) { step /= 2; if (n >> msb) msb += step; else msb -= step; }
    if (n >> msb) msb++;
    return msb;
}

/* Stable LSD radix sort of (key,value) on key bits [0,end_bit): what
 * cub::DeviceRadixSort::SortPairs(..., 0, 32+bit) guarantees (cr/rasterizer_impl.cu:317-322). */
static void stable_sort_pairs(uint64_t* keys, uint32_t* vals, long long n, int end_bit) {
    if (n <= 1) return;
    uint64_t* k2 = (uint64_t*)malloc(sizeof(uint64_t) * n);
    uint32_t* v2 = (uint32_t*)malloc(sizeof(uint32_t) * n);
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

/* Unit direction of pixel (x,y): cr/forward.cu:589-591 == cr/backward.cu:659-661.
 * beta is evaluated in double and rounded to float; cos/sin are the float overloads (correctly rounded here). */
static f3 pixel_dir(int x, int y, int W, int H, const float* beams) {
    float pixfx = (float)x;
    float alp = beams[H - 1 - y];
    float beta = (float)(-((double)pixfx - (double)(float)W / 2.0) / (double)(float)W * 2.0 * (double)LGO_PI);
    f3 q = { lgo_cosf(alp) * lgo_cosf(beta), lgo_cosf(alp) * lgo_sinf(beta), lgo_sinf(alp) };
    return q;
}

/* test accessor for pixel_dir() */
void lgo_pixel_dir(int x, int y, int W, int H, const float* beams, float* out3) {
    f3 q = pixel_dir(x, y, W, H, beams);
    out3[0] = q.x; out3[1] = q.y; out3[2] = q.z;
}

/* K7: cr/forward.cu:502-641 renderCUDA, per pixel.  The block-cooperative staging and the
 * __syncthreads_count early-out only change scheduling: each pixel walks its tile's list in order
 * until `done`.  T_in / t_only / T_pass are the multi-GPU range-shell extension (T_in == NULL,
 * t_only == 0 is exactly the reference): the walk starts from T_in instead of 1, and T_pass
 * receives the transmittance handed to the next shell (the value that tripped T < 1e-4, if any). */
static void render_pixels(lgo_state* s, const float* colors_precomp, const float* background, const float* beams,
                          const float* T_in, int t_only, float* out_color, float* out_depth, float* out_occ, float* T_pass) {
    const int W = s->W, H = s->H;
    const long long N = (long long)W * H;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const uint32_t tile = (uint32_t)(y / LGO_BLOCK_Y) * s->gx + (uint32_t)(x / LGO_BLOCK_X);
            const uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
            const f3 q = pixel_dir(x, y, W, H, beams);
            const long long pix = (long long)W * y + x;
            float T = T_in ? T_in[pix] : 1.0f;
            float T_break = T;
            uint32_t contributor = 0, last_contributor = 0;
            float C[LGO_CHANNELS] = { 0 };
            float Dp = 0.0f;
            int done = 0;
            for (uint32_t k = r0; k < r1 && !done; k++) {
                contributor++;
                const uint32_t g = s->point_list[k];
                const float* sp = s->sphere + 3 * g; const float* u1 = s->basis_u1 + 3 * g; const float* u2 = s->basis_u2 + 3 * g;
                float dx_ = sp[0] - q.x, dy_ = sp[1] - q.y, dz_ = sp[2] - q.z;
                float u1_u1 = u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2];
                float u2_u2 = u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2];
                float d_u1 = dx_ * u1[0] + dy_ * u1[1] + dz_ * u1[2];
                float d_u2 = dx_ * u2[0] + dy_ * u2[1] + dz_ * u2[2];
                float ddx = d_u1 / u1_u1, ddy = d_u2 / u2_u2;
                const float* co = s->conic_opacity + 4 * g;
                float power = -0.5f * (co[0] * ddx * ddx + co[2] * ddy * ddy) - co[1] * ddx * ddy;
                if (power > 0.0f) continue;
                float a = co[3] * lgo_expf(power);
                float alpha = 0.99f < a ? 0.99f : a;        /* min(0.99f, .) */
                if (alpha < 1.0f / 255.0f) continue;
                float test_T = T * (1 - alpha);
                if (test_T < 0.0001f) { done = 1; T_break = test_T; continue; }
                if (!t_only) {
                    for (int ch = 0; ch < LGO_CHANNELS; ch++) C[ch] += colors_precomp[g * LGO_CHANNELS + ch] * alpha * T;
                    Dp += s->depths[g] * alpha * T;
                }
                T = test_T; T_break = test_T;
                last_contributor = contributor;
            }
            if (T_pass) T_pass[pix] = T_break;
            if (t_only) continue;
            s->final_T[pix] = T;
            s->n_contrib[pix] = last_contributor;
            for (int ch = 0; ch < LGO_CHANNELS; ch++) out_color[ch * N + pix] = C[ch] + T * (background ? background[ch] : 0.0f);
            out_depth[pix] = Dp;
            out_occ[pix] = 1 - T;
        }
}

/* ------------------------------------------------------------------------------------------
 * Forward: cr/rasterizer_impl.cu:202-359 (K1 -> scan -> duplicateWithKeys -> sort -> ranges -> K7)
 * Arguments mirror CudaRasterizer::Rasterizer::forward (cr/rasterizer.h:31-58); D, M, shs,
 * projmatrix, cam_pos and prefiltered are accepted and unused, exactly like the LiDAR path.
 * Returns a state handle (free with lgo_free) or NULL with lgo_last_error() set.
 * ---------------------------------------------------------------------------------------- */
void* lgo_forward_ex(int P, int D, int M, const float* background, int width, int height,
                     const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                     const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                     const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                     const float* beams, int prefiltered, int far_, int near_,
                     float shell_lo, float shell_hi, const float* T_in, int t_only, float* T_pass,
                     float* out_color, float* out_depth, float* out_occ, int* radii) {
    (void)D; (void)M; (void)shs; (void)projmatrix; (void)cam_pos; (void)prefiltered;
    if (colors_precomp == NULL) {  /* cr/rasterizer_impl.cu:249-252 */
        snprintf(lgo_err, sizeof lgo_err, "For non-RGB, provide precomputed Gaussian colors!");
        return NULL;
    }
    const int W = width, H = height;
    const long long N = (long long)W * H;
    lgo_state* s = (lgo_state*)calloc(1, sizeof(lgo_state));
    s->P = P; s->W = W; s->H = H;
    s->gx = (W + LGO_BLOCK_X - 1) / LGO_BLOCK_X;
    s->gy = (H + LGO_BLOCK_Y - 1) / LGO_BLOCK_Y;
    size_t Pz = P > 0 ? (size_t)P : 1;
    s->depths = (float*)calloc(Pz, 4); s->means2D = (float*)calloc(Pz * 2, 4); s->cov3D = (float*)calloc(Pz * 6, 4);
    s->conic_opacity = (float*)calloc(Pz * 4, 4); s->basis_u1 = (float*)calloc(Pz * 3, 4);
    s->basis_u2 = (float*)calloc(Pz * 3, 4); s->sphere = (float*)calloc(Pz * 3, 4);
    s->tiles_touched = (uint32_t*)calloc(Pz, 4); s->point_offsets = (uint32_t*)calloc(Pz, 4);
    s->radii_xy = (int*)calloc(Pz * 2, 4);
    s->ranges = (uint32_t*)calloc((size_t)s->gx * s->gy * 2, 4);
    s->final_T = (float*)calloc((size_t)N, 4); s->n_contrib = (uint32_t*)calloc((size_t)N, 4);

    for (int i = 0; i < P; i++)
        preprocess_one(i, 0, s, means3D, scales, scale_modifier, rotations, opacities, cov3D_precomp,
                       viewmatrix, beams, far_, near_, shell_lo, shell_hi, radii);

    /* cr/rasterizer_impl.cu:288 InclusiveSum */
    uint32_t run = 0;
    for (int i = 0; i < P; i++) { run += s->tiles_touched[i]; s->point_offsets[i] = run; }
    const long long R = P > 0 ? (long long)s->point_offsets[P - 1] : 0;
    s->R = (int)R;
    s->keys = (uint64_t*)malloc(sizeof(uint64_t) * (R > 0 ? R : 1));
    s->point_list = (uint32_t*)malloc(sizeof(uint32_t) * (R > 0 ? R : 1));

    /* cr/rasterizer_impl.cu:70-112 duplicateWithKeys */
    for (int idx = 0; idx < P; idx++) {
        if (radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : s->point_offsets[idx - 1];
            unsigned xmin, ymin, xmax, ymax;
            get_rect_lidar(s->means2D[2 * idx], s->means2D[2 * idx + 1], s->radii_xy[2 * idx], s->radii_xy[2 * idx + 1],
                           s->gx, s->gy, &xmin, &ymin, &xmax, &ymax);
            uint32_t dbits; memcpy(&dbits, &s->depths[idx], 4);
            for (int y = (int)ymin; y < (int)ymax; y++)
                for (int x = (int)xmin; x < (int)xmax; x++) {
                    uint64_t key = (uint64_t)(y * s->gx + x);
                    key <<= 32; key |= dbits;
                    s->keys[off] = key; s->point_list[off] = (uint32_t)idx; off++;
                }
        }
    }
    int bit = (int)get_higher_msb(s->gx * s->gy);
    stable_sort_pairs(s->keys, s->point_list, R, 32 + bit);

    /* cr/rasterizer_impl.cu:117-139 identifyTileRanges (ranges pre-zeroed, :324) */
    for (long long i = 0; i < R; i++) {
        uint32_t cur = (uint32_t)(s->keys[i] >> 32);
        if (i == 0) s->ranges[2 * cur] = 0;
        else {
            uint32_t prev = (uint32_t)(s->keys[i - 1] >> 32);
            if (cur != prev) { s->ranges[2 * prev + 1] = (uint32_t)i; s->ranges[2 * cur] = (uint32_t)i; }
        }
        if (i == R - 1) s->ranges[2 * cur + 1] = (uint32_t)R;
    }

    render_pixels(s, colors_precomp, background, beams, T_in, t_only, out_color, out_depth, out_occ, T_pass);
    return s;
}

/* The reference entry point: no shell, T starts at 1. */
void* lgo_forward(int P, int D, int M, const float* background, int width, int height,
                  const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                  const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                  const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                  const float* beams, int prefiltered, int far_, int near_,
                  float* out_color, float* out_depth, float* out_occ, int* radii) {
    return lgo_forward_ex(P, D, M, background, width, height, means3D, shs, colors_precomp, opacities, scales, scale_modifier,
                          rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos, beams, prefiltered, far_, near_,
                          -INFINITY, INFINITY, NULL, 0, NULL, out_color, out_depth, out_occ, radii);
}

/* Phase 2 of the two-phase shell render: composite the already-binned state again from T_in. */
void lgo_render_ex(void* h, const float* colors_precomp, const float* background, const float* beams, const float* T_in,
                   int t_only, float* out_color, float* out_depth, float* out_occ, float* T_pass) {
    render_pixels((lgo_state*)h, colors_precomp, background, beams, T_in, t_only, out_color, out_depth, out_occ, T_pass);
}

/* cr/backward.cu:385-448 computeCov3D VJP */
static void compute_cov3d_bwd(int idx, const float* scale, float mod, const float* rot, const float* dL_dcov3Ds,
                              float* dL_dscales, float* dL_drots) {
    float r = rot[0], x = rot[1], y = rot[2], z = rot[3];
    m3 R = m3_make(
        1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
        2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
        2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    m3 S = m3_make(1, 0, 0, 0, 1, 0, 0, 0, 1);
    float sx = mod * scale[0], sy = mod * scale[1], sz = mod * scale[2];
    S.c[0][0] = sx; S.c[1][1] = sy; S.c[2][2] = sz;
    m3 M = m3_mul(S, R);
    const float* g = dL_dcov3Ds + 6 * idx;
    m3 dL_dSigma = m3_make(g[0], 0.5f * g[1], 0.5f * g[2], 0.5f * g[1], g[3], 0.5f * g[4], 0.5f * g[2], 0.5f * g[4], g[5]);
    m3 M2 = M;                                  /* 2.0f * M  (scalar*mat first, :423) */
    for (int c = 0; c < 3; c++) for (int rr = 0; rr < 3; rr++) M2.c[c][rr] = M.c[c][rr] * 2.0f;
    m3 dL_dM = m3_mul(M2, dL_dSigma);
    m3 Rt = m3_transpose(R);
    m3 dL_dMt = m3_transpose(dL_dM);
    float* ds = dL_dscales + 3 * idx;
    /* glm::dot(a,b) = a.x*b.x + a.y*b.y + a.z*b.z (func_geometric.inl compute_dot<vec3>) */
    ds[0] = Rt.c[0][0] * dL_dMt.c[0][0] + Rt.c[0][1] * dL_dMt.c[0][1] + Rt.c[0][2] * dL_dMt.c[0][2];
    ds[1] = Rt.c[1][0] * dL_dMt.c[1][0] + Rt.c[1][1] * dL_dMt.c[1][1] + Rt.c[1][2] * dL_dMt.c[1][2];
    ds[2] = Rt.c[2][0] * dL_dMt.c[2][0] + Rt.c[2][1] * dL_dMt.c[2][1] + Rt.c[2][2] * dL_dMt.c[2][2];
    for (int k = 0; k < 3; k++) { dL_dMt.c[0][k] *= sx; dL_dMt.c[1][k] *= sy; dL_dMt.c[2][k] *= sz; }
#define MT(i, j) dL_dMt.c[i][j]
    float qx = 2 * z * (MT(0, 1) - MT(1, 0)) + 2 * y * (MT(2, 0) - MT(0, 2)) + 2 * x * (MT(1, 2) - MT(2, 1));
    float qy = 2 * y * (MT(1, 0) + MT(0, 1)) + 2 * z * (MT(2, 0) + MT(0, 2)) + 2 * r * (MT(1, 2) - MT(2, 1)) - 4 * x * (MT(2, 2) + MT(1, 1));
    float qz = 2 * x * (MT(1, 0) + MT(0, 1)) + 2 * r * (MT(2, 0) - MT(0, 2)) + 2 * z * (MT(1, 2) + MT(2, 1)) - 4 * y * (MT(2, 2) + MT(0, 0));
    float qw = 2 * r * (MT(0, 1) - MT(1, 0)) + 2 * x * (MT(2, 0) + MT(0, 2)) + 2 * y * (MT(1, 2) + MT(2, 1)) - 4 * z * (MT(1, 1) + MT(0, 0));
#undef MT
    float* dr = dL_drots + 4 * idx;      /* :447 no normalisation Jacobian */
    dr[0] = qx; dr[1] = qy; dr[2] = qz; dr[3] = qw;
}

/* K9: cr/backward.cu:157-382 computeCov2DCUDA (one Gaussian) */
static void cov2d_bwd_one(int idx, const float* means, const int* radii, const float* cov3Ds, const float* vm,
                          const float* dL_dbasis_u1, const float* dL_dbasis_u2, const float* dL_dconics,
                          float* dL_dmeans, float* dL_dcov) {
    if (!(radii[idx] > 0)) return;
    const float* cov3D = cov3Ds + 6 * idx;
    f3 mean = { means[3 * idx], means[3 * idx + 1], means[3 * idx + 2] };
    f3 dLc = { dL_dconics[4 * idx], dL_dconics[4 * idx + 1], dL_dconics[4 * idx + 3] };
    f3 d = transform_point_4x3(mean, vm);
    float dist = sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
    f3 dir = normalize_bwd(d);
    f3 u1 = { dir.y, -dir.x, 0 };
    u1 = normalize_bwd(u1);
    f3 u2 = { dir.y * u1.z - dir.z * u1.y, dir.z * u1.x - dir.x * u1.z, dir.x * u1.y - dir.y * u1.x };
    m3 J = m3_make(u1.x, u1.y, u1.z, u2.x, u2.y, u2.z, 0, 0, 0);
    m3 W = m3_make(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
    m3 Vrk = m3_make(cov3D[0], cov3D[1], cov3D[2], cov3D[1], cov3D[3], cov3D[4], cov3D[2], cov3D[4], cov3D[5]);
    m3 T = m3_mul(W, J);
    m3 cov2D = m3_mul(m3_mul(m3_transpose(T), m3_transpose(Vrk)), T);
    float _a = cov2D.c[0][0] += 0.01f;
    float _b = cov2D.c[0][1];
    float _c = cov2D.c[1][1] += 0.01f;
    float a = 1 / (dist * dist) * _a;
    float b = 1 / (dist * dist) * _b;
    float c = 1 / (dist * dist) * _c;

    float denom = a * c - b * b;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);   /* :237 the damping of SURVEY 0.2 */
    f3 dL_dcov_mean = { 0, 0, 0 };   /* reference leaves this uninitialised on the (unreachable) else branch */
    if (denom2inv != 0) {
        dL_da = denom2inv * (-1 * c * c * dLc.x + 2 * b * c * dLc.y + (denom - a * c) * dLc.z);
        dL_dc = denom2inv * (-1 * a * a * dLc.z + 2 * a * b * dLc.y + (denom - a * c) * dLc.x);
        dL_db = denom2inv * 2 * (b * c * dLc.x - (denom + 2 * b * b) * dLc.y + a * b * dLc.z);
        float dist4 = dist * dist * dist * dist;
        dL_dcov_mean.x = dL_da * (-2 * d.x * _a) / dist4 + dL_db * (-2 * d.x * _b) / dist4 + dL_dc * (-2 * d.x * _c) / dist4;
        dL_dcov_mean.y = dL_da * (-2 * d.y * _a) / dist4 + dL_db * (-2 * d.y * _b) / dist4 + dL_dc * (-2 * d.y * _c) / dist4;
        dL_dcov_mean.z = dL_da * (-2 * d.z * _a) / dist4 + dL_db * (-2 * d.z * _b) / dist4 + dL_dc * (-2 * d.z * _c) / dist4;
        dL_da = 1 / (dist * dist) * dL_da;
        dL_dc = 1 / (dist * dist) * dL_dc;
        dL_db = 1 / (dist * dist) * dL_db;
#define TT(i, j) T.c[i][j]
        dL_dcov[6 * idx + 0] = (TT(0, 0) * TT(0, 0) * dL_da + TT(0, 0) * TT(1, 0) * dL_db + TT(1, 0) * TT(1, 0) * dL_dc);
        dL_dcov[6 * idx + 3] = (TT(0, 1) * TT(0, 1) * dL_da + TT(0, 1) * TT(1, 1) * dL_db + TT(1, 1) * TT(1, 1) * dL_dc);
        dL_dcov[6 * idx + 5] = (TT(0, 2) * TT(0, 2) * dL_da + TT(0, 2) * TT(1, 2) * dL_db + TT(1, 2) * TT(1, 2) * dL_dc);
        dL_dcov[6 * idx + 1] = 2 * TT(0, 0) * TT(0, 1) * dL_da + (TT(0, 0) * TT(1, 1) + TT(0, 1) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 1) * dL_dc;
        dL_dcov[6 * idx + 2] = 2 * TT(0, 0) * TT(0, 2) * dL_da + (TT(0, 0) * TT(1, 2) + TT(0, 2) * TT(1, 0)) * dL_db + 2 * TT(1, 0) * TT(1, 2) * dL_dc;
        dL_dcov[6 * idx + 4] = 2 * TT(0, 2) * TT(0, 1) * dL_da + (TT(0, 1) * TT(1, 2) + TT(0, 2) * TT(1, 1)) * dL_db + 2 * TT(1, 1) * TT(1, 2) * dL_dc;
    } else {
        for (int i = 0; i < 6; i++) dL_dcov[6 * idx + i] = 0;
    }
#define VV(i, j) Vrk.c[i][j]
    float dL_dT00 = 2 * (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_da + (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_db;
    float dL_dT01 = 2 * (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_da + (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_db;
    float dL_dT02 = 2 * (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_da + (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_db;
    float dL_dT10 = 2 * (TT(1, 0) * VV(0, 0) + TT(1, 1) * VV(0, 1) + TT(1, 2) * VV(0, 2)) * dL_dc + (TT(0, 0) * VV(0, 0) + TT(0, 1) * VV(0, 1) + TT(0, 2) * VV(0, 2)) * dL_db;
    float dL_dT11 = 2 * (TT(1, 0) * VV(1, 0) + TT(1, 1) * VV(1, 1) + TT(1, 2) * VV(1, 2)) * dL_dc + (TT(0, 0) * VV(1, 0) + TT(0, 1) * VV(1, 1) + TT(0, 2) * VV(1, 2)) * dL_db;
    float dL_dT12 = 2 * (TT(1, 0) * VV(2, 0) + TT(1, 1) * VV(2, 1) + TT(1, 2) * VV(2, 2)) * dL_dc + (TT(0, 0) * VV(2, 0) + TT(0, 1) * VV(2, 1) + TT(0, 2) * VV(2, 2)) * dL_db;
#undef VV
#undef TT
#define WW(i, j) W.c[i][j]
    float dL_dJ00 = WW(0, 0) * dL_dT00 + WW(0, 1) * dL_dT01 + WW(0, 2) * dL_dT02;
    float dL_dJ01 = WW(1, 0) * dL_dT00 + WW(1, 1) * dL_dT01 + WW(1, 2) * dL_dT02;
    float dL_dJ02 = WW(2, 0) * dL_dT00 + WW(2, 1) * dL_dT01 + WW(2, 2) * dL_dT02;
    float dL_dJ10 = WW(0, 0) * dL_dT10 + WW(0, 1) * dL_dT11 + WW(0, 2) * dL_dT12;
    float dL_dJ11 = WW(1, 0) * dL_dT10 + WW(1, 1) * dL_dT11 + WW(1, 2) * dL_dT12;
    float dL_dJ12 = WW(2, 0) * dL_dT10 + WW(2, 1) * dL_dT11 + WW(2, 2) * dL_dT12;
#undef WW
    dL_dJ00 = dL_dJ00 + dL_dbasis_u1[3 * idx + 0];
    dL_dJ01 = dL_dJ01 + dL_dbasis_u1[3 * idx + 1];
    dL_dJ02 = dL_dJ02 + dL_dbasis_u1[3 * idx + 2];
    dL_dJ10 = dL_dJ10 + dL_dbasis_u2[3 * idx + 0];
    dL_dJ11 = dL_dJ11 + dL_dbasis_u2[3 * idx + 1];
    dL_dJ12 = dL_dJ12 + dL_dbasis_u2[3 * idx + 2];
    (void)dL_dJ02;   /* J02 == 0: computed by the reference, never used (:297,:304) */

    float d_sum2 = d.x * d.x + d.y * d.y + d.z * d.z;
    float inv_d_sum32 = (float)(1.0f / ((double)sqrtf(d_sum2 * d_sum2 * d_sum2) + 1e-9));   /* :313 double eps */
    float ddirx_dmeanx = (d_sum2 - d.x * d.x) * inv_d_sum32;
    float ddirx_dmeany = (-d.x * d.y) * inv_d_sum32;
    float ddirx_dmeanz = (-d.x * d.z) * inv_d_sum32;
    float ddiry_dmeanx = (-d.x * d.y) * inv_d_sum32;
    float ddiry_dmeany = (d_sum2 - d.y * d.y) * inv_d_sum32;
    float ddiry_dmeanz = (-d.y * d.z) * inv_d_sum32;
    float ddirz_dmeanx = (-d.x * d.z) * inv_d_sum32;
    float ddirz_dmeany = (-d.y * d.z) * inv_d_sum32;
    float ddirz_dmeanz = (d_sum2 - d.z * d.z) * inv_d_sum32;

    float dir_sum2 = dir.x * dir.x + dir.y * dir.y;
    float inv_dir_sum32 = (float)(1.0f / ((double)sqrtf(dir_sum2 * dir_sum2 * dir_sum2) + 1e-9));   /* :337 */
    float dJ00_ddiry = (dir.x * dir.x) * inv_dir_sum32;
    float dJ00_ddirx = (-dir.y * dir.x) * inv_dir_sum32;
    float dJ01_ddirx = (-dir.y * dir.y) * inv_dir_sum32;
    float dJ01_ddiry = (dir.x * dir.y) * inv_dir_sum32;
    float dJ10_ddirx = dir.z * dir.y * dir.y * inv_dir_sum32;
    float dJ10_ddiry = -dir.x * dir.y * dir.z * inv_dir_sum32;
    float dJ10_ddirz = (float)(dir.x / ((double)sqrtf(dir_sum2) + 1e-9));                          /* :347 */
    float dJ11_ddirx = -dir.x * dir.y * dir.z * inv_dir_sum32;
    float dJ11_ddiry = dir.z * dir.x * dir.x * inv_dir_sum32;
    float dJ11_ddirz = (float)(dir.y / ((double)sqrtf(dir_sum2) + 1e-9));                          /* :351 */
    float dJ12_ddirx = (float)(-dir.x / ((double)sqrtf(dir_sum2) + 1e-9));                         /* :353 */
    float dJ12_ddiry = (float)(-dir.y / ((double)sqrtf(dir_sum2) + 1e-9));                         /* :354 */

    float dJ00_dmeanx = dJ00_ddirx * ddirx_dmeanx + dJ00_ddiry * ddiry_dmeanx;
    float dJ01_dmeanx = dJ01_ddirx * ddirx_dmeanx + dJ01_ddiry * ddiry_dmeanx;
    float dJ10_dmeanx = dJ10_ddirx * ddirx_dmeanx + dJ10_ddiry * ddiry_dmeanx + dJ10_ddirz * ddirz_dmeanx;
    float dJ11_dmeanx = dJ11_ddirx * ddirx_dmeanx + dJ11_ddiry * ddiry_dmeanx + dJ11_ddirz * ddirz_dmeanx;
    float dJ12_dmeanx = dJ12_ddirx * ddirx_dmeanx + dJ12_ddiry * ddiry_dmeanx;
    float dL_dmeanx = dL_dcov_mean.x + dL_dJ00 * dJ00_dmeanx + dL_dJ01 * dJ01_dmeanx + dL_dJ10 * dJ10_dmeanx + dL_dJ11 * dJ11_dmeanx + dL_dJ12 * dJ12_dmeanx;

    float dJ00_dmeany = dJ00_ddirx * ddirx_dmeany + dJ00_ddiry * ddiry_dmeany;
    float dJ01_dmeany = dJ01_ddirx * ddirx_dmeany + dJ01_ddiry * ddiry_dmeany;
    float dJ10_dmeany = dJ10_ddirx * ddirx_dmeany + dJ10_ddiry * ddiry_dmeany + dJ10_ddirz * ddirz_dmeany;
    float dJ11_dmeany = dJ11_ddirx * ddirx_dmeany + dJ11_ddiry * ddiry_dmeany + dJ11_ddirz * ddirz_dmeany;
    float dJ12_dmeany = dJ12_ddirx * ddirx_dmeany + dJ12_ddiry * ddiry_dmeany;
    float dL_dmeany = dL_dcov_mean.y + dL_dJ00 * dJ00_dmeany + dL_dJ01 * dJ01_dmeany + dL_dJ10 * dJ10_dmeany + dL_dJ11 * dJ11_dmeany + dL_dJ12 * dJ12_dmeany;

    float dJ00_dmeanz = dJ00_ddirx * ddirx_dmeanz + dJ00_ddiry * ddiry_dmeanz;
    float dJ01_dmeanz = dJ01_ddirx * ddirx_dmeanz + dJ01_ddiry * ddiry_dmeanz;
    float dJ10_dmeanz = dJ10_ddirx * ddirx_dmeanz + dJ10_ddiry * ddiry_dmeanz + dJ10_ddirz * ddirz_dmeanz;
    float dJ11_dmeanz = dJ11_ddirx * ddirx_dmeanz + dJ11_ddiry * ddiry_dmeanz + dJ11_ddirz * ddirz_dmeanz;
    float dJ12_dmeanz = dJ12_ddirx * ddirx_dmeanz + dJ12_ddiry * ddiry_dmeanz;
    float dL_dmeanz = dL_dcov_mean.z + dL_dJ00 * dJ00_dmeanz + dL_dJ01 * dJ01_dmeanz + dL_dJ10 * dJ10_dmeanz + dL_dJ11 * dJ11_dmeanz + dL_dJ12 * dJ12_dmeanz;

    dL_dmeans[3 * idx + 0] = dL_dmeanx;   /* still in view space (:380-381) */
    dL_dmeans[3 * idx + 1] = dL_dmeany;
    dL_dmeans[3 * idx + 2] = dL_dmeanz;
}

/* K10: cr/backward.cu:453-532 preprocessCUDA (one Gaussian) */
static void preprocess_bwd_one(int idx, const float* means, const int* radii, const float* scales, const float* rotations,
                               float scale_modifier, const float* vm, const float* dL_dsphere, float* dL_dmeans,
                               const float* dL_ddepths, const float* dL_dcov3D, float* dL_dscale, float* dL_drot) {
    if (!(radii[idx] > 0)) return;
    f3 m = { means[3 * idx], means[3 * idx + 1], means[3 * idx + 2] };
    f3 p = transform_point_4x3(m, vm);
    float dist = sqrtf(p.x * p.x + p.y * p.y + p.z * p.z);
    if (dist <= 0) return;
    float ddist_dmeanx = p.x / dist, ddist_dmeany = p.y / dist, ddist_dmeanz = p.z / dist;
    float p_sum2 = p.x * p.x + p.y * p.y + p.z * p.z;
    float inv_p_sum32 = 1.0f / sqrtf(p_sum2 * p_sum2 * p_sum2);
    float dspx_dmeanx = (p_sum2 - p.x * p.x) * inv_p_sum32;
    float dspx_dmeany = (-p.x * p.y) * inv_p_sum32;
    float dspx_dmeanz = (-p.x * p.z) * inv_p_sum32;
    float dspy_dmeanx = (-p.x * p.y) * inv_p_sum32;
    float dspy_dmeany = (p_sum2 - p.y * p.y) * inv_p_sum32;
    float dspy_dmeanz = (-p.y * p.z) * inv_p_sum32;
    float dspz_dmeanx = (-p.x * p.z) * inv_p_sum32;
    float dspz_dmeany = (-p.y * p.z) * inv_p_sum32;
    float dspz_dmeanz = (p_sum2 - p.z * p.z) * inv_p_sum32;
    const float* gs = dL_dsphere + 3 * idx;
    f3 v;
    v.x = dL_dmeans[3 * idx + 0] + gs[0] * dspx_dmeanx + gs[1] * dspy_dmeanx + gs[2] * dspz_dmeanx + dL_ddepths[idx] * ddist_dmeanx;
    v.y = dL_dmeans[3 * idx + 1] + gs[0] * dspx_dmeany + gs[1] * dspy_dmeany + gs[2] * dspz_dmeany + dL_ddepths[idx] * ddist_dmeany;
    v.z = dL_dmeans[3 * idx + 2] + gs[0] * dspx_dmeanz + gs[1] * dspy_dmeanz + gs[2] * dspz_dmeanz + dL_ddepths[idx] * ddist_dmeanz;
    f3 w = transform_vec_4x3_transpose(v, vm);
    dL_dmeans[3 * idx + 0] = w.x; dL_dmeans[3 * idx + 1] = w.y; dL_dmeans[3 * idx + 2] = w.z;
    if (scales) compute_cov3d_bwd(idx, scales + 3 * idx, scale_modifier, rotations + 4 * idx, dL_dcov3D, dL_dscale, dL_drot);
}

/* ------------------------------------------------------------------------------------------
 * Backward: cr/rasterizer_impl.cu:431-549 (K8 render-bwd -> K9 -> K10).  Arguments mirror
 * CudaRasterizer::Rasterizer::backward (cr/rasterizer.h:86-122).  All dL_d* outputs must be
 * zero-initialised by the caller, as R3/rasterize_points.cu:163-175 does.
 * ---------------------------------------------------------------------------------------- */
int lgo_backward_ex(const void* h, int P, int D, int M, int R, const float* background, int width, int height,
                 const float* means3D, const float* shs, const float* colors_precomp,
                 const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                 const float* viewmatrix, const float* projmatrix, const float* campos, const float* beams,
                 float tan_fovx, float tan_fovy, const int* radii,
                 const float* dL_dpix, const float* dL_dout_depth, const float* dL_dout_occ,
                 float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_ddepths,
                 float* dL_dmean3D, float* dL_dsphere, float* dL_dbasis_u1, float* dL_dbasis_u2,
                 float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                 const float* behind, const float* T_final_global) {
    (void)D; (void)M; (void)shs; (void)projmatrix; (void)campos; (void)tan_fovx; (void)tan_fovy; (void)dL_dsh;
    const lgo_state* s = (const lgo_state*)h;
    if (s->P != P || s->W != width || s->H != height || s->R != R) {
        snprintf(lgo_err, sizeof lgo_err, "backward: state does not match (P,R,W,H)");
        return -1;
    }
    const int W = width, H = height, C = LGO_CHANNELS;
    const long long N = (long long)W * H;
    /* lgo_set_accumulate_double: float64 shadows of the eight per-Gaussian arrays K8 adds to (29 values per Gaussian) */
    double* acc64 = NULL;
    if (lgo_accumulate_double) {
        acc64 = (double*)calloc((size_t)P * 24 + 1, sizeof(double));
        if (!acc64) { snprintf(lgo_err, sizeof lgo_err, "backward: out of memory (float64 sums)"); return -1; }
    }
    double* a_color = acc64;                         /* [P][2] */
    double* a_depth = acc64 ? acc64 + (size_t)P * 2 : NULL;    /* [P]    */
    double* a_u1 = acc64 ? acc64 + (size_t)P * 3 : NULL;       /* [P][3] */
    double* a_u2 = acc64 ? acc64 + (size_t)P * 6 : NULL;       /* [P][3] */
    double* a_m2 = acc64 ? acc64 + (size_t)P * 9 : NULL;       /* [P][4] */
    double* a_sp = acc64 ? acc64 + (size_t)P * 13 : NULL;      /* [P][3] */
    double* a_con = acc64 ? acc64 + (size_t)P * 16 : NULL;     /* [P][4] */
    double* a_op = acc64 ? acc64 + (size_t)P * 20 : NULL;      /* [P]    */
#define LGO_ACC(arr, sh, idx, val) do { if (acc64) (sh)[idx] += (double)(val); else (arr)[idx] += (val); } while (0)

    /* K8: cr/backward.cu:535-791 renderCUDA, per pixel, back to front */
    for (int yy = 0; yy < H; yy++)
        for (int xx = 0; xx < W; xx++) {
            const int y = lgo_reverse_pixel_order ? H - 1 - yy : yy, x = lgo_reverse_pixel_order ? W - 1 - xx : xx;
            const uint32_t tile = (uint32_t)(y / LGO_BLOCK_Y) * s->gx + (uint32_t)(x / LGO_BLOCK_X);
            const uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
            const long long pix = (long long)W * y + x;
            const f3 q = pixel_dir(x, y, W, H, beams);
            /* shell extension: T starts at this shell's own end value, T_final is the global one and the
             * "colour behind" recurrences are seeded with what the farther shells composited */
            float T = s->final_T[pix];
            double Td = (double)T;
            const float T_final = T_final_global ? T_final_global[pix] : T;
            uint32_t contributor = r1 - r0;
            const int last_contributor = (int)s->n_contrib[pix];
            float accum_rec[LGO_CHANNELS] = { 0 };
            float accum_red = 0, accum_reo = 0;
            if (behind && T > 0.f) {
                const float inv = 1.f / T;
                accum_rec[0] = behind[pix] * inv; accum_rec[1] = behind[N + pix] * inv; accum_red = behind[2 * N + pix] * inv;
                accum_reo = 1.f - T_final * inv;
            }
            float dL_dpixel[LGO_CHANNELS];
            for (int i = 0; i < C; i++) dL_dpixel[i] = dL_dpix[i * N + pix];
            float dL_dod = dL_dout_depth[pix], dL_doo = dL_dout_occ[pix];
            float last_alpha = 0, last_color[LGO_CHANNELS] = { 0 }, last_depth = 0;
            for (uint32_t kk = r1; kk > r0; kk--) {
                const uint32_t g = s->point_list[kk - 1];
                contributor--;
                if ((int)contributor >= last_contributor) continue;
                const float* sp = s->sphere + 3 * g; const float* u1 = s->basis_u1 + 3 * g; const float* u2 = s->basis_u2 + 3 * g;
                const float sdx = sp[0] - q.x, sdy = sp[1] - q.y, sdz = sp[2] - q.z;
                const float u1_u1 = u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2];
                const float u2_u2 = u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2];
                const float _d_u1 = sdx * u1[0] + sdy * u1[1] + sdz * u1[2];
                const float _d_u2 = sdx * u2[0] + sdy * u2[1] + sdz * u2[2];
                const float dx = _d_u1 / u1_u1, dy = _d_u2 / u2_u2;
                const float* co = s->conic_opacity + 4 * g;
                const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0.0f) continue;
                const float G = lgo_expf(power);
                const float aa = co[3] * G;
                const float alpha = 0.99f < aa ? 0.99f : aa;
                if (alpha < 1.0f / 255.0f) continue;

                if (lgo_accumulate_double >= 2) { Td = Td / (double)(1.f - alpha); T = (float)Td; } else T = T / (1.f - alpha);
                const float dchannel_dcolor = alpha * T;
                float dL_dalpha = 0.0f;
                for (int ch = 0; ch < C; ch++) {
                    const float c = colors_precomp[g * C + ch];
                    accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                    last_color[ch] = c;
                    const float dL_dchannel = dL_dpixel[ch];
                    dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
                    LGO_ACC(dL_dcolor, a_color, g * C + ch, dchannel_dcolor * dL_dchannel);
                }
                const float dep = s->depths[g];
                accum_red = last_alpha * last_depth + (1.f - last_alpha) * accum_red;
                last_depth = dep;
                dL_dalpha += (dep - accum_red) * dL_dod;
                LGO_ACC(dL_ddepths, a_depth, g, dchannel_dcolor * dL_dod);
                accum_reo = (float)((double)last_alpha * 1.0 + (double)((1.f - last_alpha) * accum_reo));   /* :714 */
                dL_dalpha += (1 - accum_reo) * dL_doo;
                dL_dalpha *= T;
                last_alpha = alpha;
                float bg_dot_dpixel = 0;
                for (int i = 0; i < C; i++) bg_dot_dpixel += background[i] * dL_dpixel[i];
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;

                const float dL_dG = co[3] * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                const float dG_ddely = -gdy * co[2] - gdx * co[1];
                const float ddx_du1x = (sdx * u1_u1 - _d_u1 * 2 * u1[0]) / (u1_u1 * u1_u1);
                const float ddx_du1y = (sdy * u1_u1 - _d_u1 * 2 * u1[1]) / (u1_u1 * u1_u1);
                const float ddx_du1z = (sdz * u1_u1 - _d_u1 * 2 * u1[2]) / (u1_u1 * u1_u1);
                const float ddy_du2x = (sdx * u2_u2 - _d_u2 * 2 * u2[0]) / (u2_u2 * u2_u2);
                const float ddy_du2y = (sdy * u2_u2 - _d_u2 * 2 * u2[1]) / (u2_u2 * u2_u2);
                const float ddy_du2z = (sdz * u2_u2 - _d_u2 * 2 * u2[2]) / (u2_u2 * u2_u2);
                LGO_ACC(dL_dbasis_u1, a_u1, 3 * g + 0, dL_dG * dG_ddelx * ddx_du1x);
                LGO_ACC(dL_dbasis_u1, a_u1, 3 * g + 1, dL_dG * dG_ddelx * ddx_du1y);
                LGO_ACC(dL_dbasis_u1, a_u1, 3 * g + 2, dL_dG * dG_ddelx * ddx_du1z);
                LGO_ACC(dL_dbasis_u2, a_u2, 3 * g + 0, dL_dG * dG_ddely * ddy_du2x);
                LGO_ACC(dL_dbasis_u2, a_u2, 3 * g + 1, dL_dG * dG_ddely * ddy_du2y);
                LGO_ACC(dL_dbasis_u2, a_u2, 3 * g + 2, dL_dG * dG_ddely * ddy_du2z);
                LGO_ACC(dL_dmean2D, a_m2, 4 * g + 0, dL_dG * dG_ddelx);
                LGO_ACC(dL_dmean2D, a_m2, 4 * g + 1, dL_dG * dG_ddely);
                const float ddx_dsx = u1[0] / u1_u1, ddx_dsy = u1[1] / u1_u1, ddx_dsz = u1[2] / u1_u1;
                const float ddy_dsx = u2[0] / u2_u2, ddy_dsy = u2[1] / u2_u2, ddy_dsz = u2[2] / u2_u2;
                const float dG_dsx = dG_ddelx * ddx_dsx + dG_ddely * ddy_dsx;
                const float dG_dsy = dG_ddelx * ddx_dsy + dG_ddely * ddy_dsy;
                const float dG_dsz = dG_ddelx * ddx_dsz + dG_ddely * ddy_dsz;
                const float gsx = dL_dG * dG_dsx, gsy = dL_dG * dG_dsy, gsz = dL_dG * dG_dsz;
                LGO_ACC(dL_dsphere, a_sp, 3 * g + 0, gsx);
                LGO_ACC(dL_dsphere, a_sp, 3 * g + 1, gsy);
                LGO_ACC(dL_dsphere, a_sp, 3 * g + 2, gsz);
                LGO_ACC(dL_dmean2D, a_m2, 4 * g + 2, sqrtf(gsx * gsx + gsy * gsy + gsz * gsz));   /* :779 a statistic, not a gradient */
                LGO_ACC(dL_dmean2D, a_m2, 4 * g + 3, 0.0f);
                LGO_ACC(dL_dconic, a_con, 4 * g + 0, -0.5f * gdx * dx * dL_dG);
                LGO_ACC(dL_dconic, a_con, 4 * g + 1, -0.5f * gdx * dy * dL_dG);
                LGO_ACC(dL_dconic, a_con, 4 * g + 3, -0.5f * gdy * dy * dL_dG);
                LGO_ACC(dL_dopacity, a_op, g, G * dL_dalpha);
            }
        }

#undef LGO_ACC
    if (acc64) {   /* the exact sums, rounded once (added to whatever the caller's arrays held: zeros) */
        for (long long i = 0; i < (long long)P * 2; i++) dL_dcolor[i] = (float)((double)dL_dcolor[i] + a_color[i]);
        for (long long i = 0; i < (long long)P; i++) dL_ddepths[i] = (float)((double)dL_ddepths[i] + a_depth[i]);
        for (long long i = 0; i < (long long)P * 3; i++) dL_dbasis_u1[i] = (float)((double)dL_dbasis_u1[i] + a_u1[i]);
        for (long long i = 0; i < (long long)P * 3; i++) dL_dbasis_u2[i] = (float)((double)dL_dbasis_u2[i] + a_u2[i]);
        for (long long i = 0; i < (long long)P * 4; i++) dL_dmean2D[i] = (float)((double)dL_dmean2D[i] + a_m2[i]);
        for (long long i = 0; i < (long long)P * 3; i++) dL_dsphere[i] = (float)((double)dL_dsphere[i] + a_sp[i]);
        for (long long i = 0; i < (long long)P * 4; i++) dL_dconic[i] = (float)((double)dL_dconic[i] + a_con[i]);
        for (long long i = 0; i < (long long)P; i++) dL_dopacity[i] = (float)((double)dL_dopacity[i] + a_op[i]);
        free(acc64);
    }
    const float* cov3D_ptr = (cov3D_precomp != NULL) ? cov3D_precomp : s->cov3D;
    for (int i = 0; i < P; i++)
        cov2d_bwd_one(i, means3D, radii, cov3D_ptr, viewmatrix, dL_dbasis_u1, dL_dbasis_u2, dL_dconic, dL_dmean3D, dL_dcov3D);
    for (int i = 0; i < P; i++)
        preprocess_bwd_one(i, means3D, radii, scales, rotations, scale_modifier, viewmatrix, dL_dsphere, dL_dmean3D,
                           dL_ddepths, dL_dcov3D, dL_dscale, dL_drot);
    return 0;
}

int lgo_backward(const void* h, int P, int D, int M, int R, const float* background, int width, int height,
                 const float* means3D, const float* shs, const float* colors_precomp,
                 const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                 const float* viewmatrix, const float* projmatrix, const float* campos, const float* beams,
                 float tan_fovx, float tan_fovy, const int* radii,
                 const float* dL_dpix, const float* dL_dout_depth, const float* dL_dout_occ,
                 float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_ddepths,
                 float* dL_dmean3D, float* dL_dsphere, float* dL_dbasis_u1, float* dL_dbasis_u2,
                 float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot) {
    return lgo_backward_ex(h, P, D, M, R, background, width, height, means3D, shs, colors_precomp, scales, scale_modifier, rotations,
                           cov3D_precomp, viewmatrix, projmatrix, campos, beams, tan_fovx, tan_fovy, radii, dL_dpix, dL_dout_depth,
                           dL_dout_occ, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_ddepths, dL_dmean3D, dL_dsphere,
                           dL_dbasis_u1, dL_dbasis_u2, dL_dcov3D, dL_dsh, dL_dscale, dL_drot, NULL, NULL);
}

/* cr/rasterizer_impl.cu:362-426 visible_filter -> K2 only; radii out (R3/rasterize_points.cu:243-318) */
int lgo_visible_filter(int P, int M, int width, int height, const float* means3D, const float* scales,
                       float scale_modifier, const float* rotations, const float* cov3D_precomp,
                       const float* viewmatrix, const float* projmatrix, const float* cam_pos, const float* beams,
                       float tan_fovx, float tan_fovy, int prefiltered, int far_, int near_, int* radii) {
    (void)M; (void)projmatrix; (void)cam_pos; (void)tan_fovx; (void)tan_fovy; (void)prefiltered;
    lgo_state s; memset(&s, 0, sizeof s);
    s.P = P; s.W = width; s.H = height;
    s.gx = (width + LGO_BLOCK_X - 1) / LGO_BLOCK_X; s.gy = (height + LGO_BLOCK_Y - 1) / LGO_BLOCK_Y;
    size_t Pz = P > 0 ? (size_t)P : 1;
    s.cov3D = (float*)calloc(Pz * 6, 4); s.means2D = (float*)calloc(Pz * 2, 4); s.radii_xy = (int*)calloc(Pz * 2, 4);
    for (int i = 0; i < P; i++)
        preprocess_one(i, 1, &s, means3D, scales, scale_modifier, rotations, NULL, cov3D_precomp, viewmatrix, beams, far_, near_,
                       -INFINITY, INFINITY, radii);
    free(s.cov3D); free(s.means2D); free(s.radii_xy);
    return 0;
}

/* cr/rasterizer_impl.cu:54-66 checkFrustum + cr/auxiliary.h:175-200 in_frustum: present = !(z_view <= 0.2) */
void lgo_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix, unsigned char* present) {
    (void)projmatrix;
    for (int i = 0; i < P; i++) {
        f3 p = { means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2] };
        f3 v = transform_point_4x3(p, viewmatrix);
        present[i] = (v.z <= 0.2f) ? 0 : 1;
    }
}
