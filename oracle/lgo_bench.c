/*
 * lgo_bench.c -- all-core timing driver for the CPU oracles (bench.py `cpu_baseline.all_cores`).
 *
 * TEST / BASELINE INFRASTRUCTURE, like the oracles themselves.  The C restatements are single-threaded (frames are what a CPU
 * box would parallelise over: they are independent), so "all host cores" = T POSIX threads, each rendering its own frames of
 * the same scene into its own output buffers -- no Python, no GIL, no shared state except the read-only inputs.
 * Returns the wall-clock seconds for threads x frames_each frames (forward, or forward + backward).
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

void* lgo_forward(int P, int D, int M, const float* background, int width, int height, const float* means3D, const float* shs,
                  const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                  const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                  const float* cam_pos, const float* beams, int prefiltered, int far_, int near_, float* out_color, float* out_depth,
                  float* out_occ, int* radii);
int lgo_backward(const void* h, int P, int D, int M, int R, const float* background, int width, int height, const float* means3D,
                 const float* shs, const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
                 const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* campos, const float* beams,
                 float tan_fovx, float tan_fovy, const int* radii, const float* dL_dpix, const float* dL_dout_depth,
                 const float* dL_dout_occ, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_ddepths,
                 float* dL_dmean3D, float* dL_dsphere, float* dL_dbasis_u1, float* dL_dbasis_u2, float* dL_dcov3D, float* dL_dsh,
                 float* dL_dscale, float* dL_drot);
int lgo_num_rendered(const void* h);
void lgo_free(void* h);
void* sfo_forward(int P, const float* background, int width, int height, const float* means3D, const float* colors_precomp,
                  const float* opacities, const float* scales, float scale_modifier, const float* rotations, const float* viewmatrix,
                  const float* beams, int far_, int near_, float* out_color, float* out_others, int* radii);
int sfo_backward(const void* h, int P, int R, const float* background, int width, int height, const float* means3D,
                 const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations, const float* viewmatrix,
                 const float* beams, const int* radii, const float* dL_dpix, const float* dL_dothers, float* dL_dmean2D,
                 float* dL_dnormal, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dtransMat,
                 float* dL_dtransMat_2dtemp, float* dL_dscale, float* dL_drot, float* gs_depth);
int sfo_num_rendered(const void* h);
void sfo_free(void* h);

typedef struct {
    int surfel, frames, fwd_only, P, W, H, far_, near_;
    const float *bg, *means, *colors, *opac, *scales, *rots, *vm, *beams, *g0, *g1, *g2;
    int failed;
} job_t;

static void* worker(void* arg) {
    job_t* j = (job_t*)arg;
    const size_t P = (size_t)j->P, N = (size_t)j->W * j->H;
    float zero16[16] = { 0 };
    float* img = (float*)malloc(sizeof(float) * 9 * N);              /* colour 2 + (depth, occ | 7 surfel planes) */
    int* radii = (int*)malloc(sizeof(int) * (P ? P : 1));
    float* g = (float*)malloc(sizeof(float) * 48 * (P ? P : 1));     /* every per-Gaussian gradient array, 3-D (37) or surfel (33) */
    for (int f = 0; f < j->frames; f++) {
        if (!j->surfel) {
            void* h = lgo_forward(j->P, 1, 0, j->bg, j->W, j->H, j->means, NULL, j->colors, j->opac, j->scales, 1.0f, j->rots, NULL, j->vm,
                                  zero16, zero16, j->beams, 0, j->far_, j->near_, img, img + 2 * N, img + 3 * N, radii);
            if (!h) { j->failed = 1; break; }
            if (!j->fwd_only) {
                memset(g, 0, sizeof(float) * 48 * P);               /* the binding's zero-filled gradient tensors (R3/rasterize_points.cu:163-175) */
                float* q = g;
                float* m2 = q; q += 4 * P; float* con = q; q += 4 * P; float* op = q; q += P; float* col = q; q += 2 * P;
                float* dep = q; q += P; float* m3 = q; q += 3 * P; float* sph = q; q += 3 * P; float* u1 = q; q += 3 * P;
                float* u2 = q; q += 3 * P; float* cov = q; q += 6 * P; float* sc = q; q += 3 * P; float* rot = q; q += 4 * P;
                if (lgo_backward(h, j->P, 1, 0, lgo_num_rendered(h), j->bg, j->W, j->H, j->means, NULL, j->colors, j->scales, 1.0f, j->rots,
                                 NULL, j->vm, zero16, zero16, j->beams, 1.0f, 1.0f, radii, j->g0, j->g1, j->g2, m2, con, op, col, dep, m3, sph,
                                 u1, u2, cov, NULL, sc, rot) != 0) j->failed = 1;
            }
            lgo_free(h);
        } else {
            void* h = sfo_forward(j->P, j->bg, j->W, j->H, j->means, j->colors, j->opac, j->scales, 1.0f, j->rots, j->vm, j->beams, j->far_,
                                  j->near_, img, img + 2 * N, radii);
            if (!h) { j->failed = 1; break; }
            if (!j->fwd_only) {
                memset(g, 0, sizeof(float) * 48 * P);
                float* q = g;
                float* m2 = q; q += 4 * P; float* nrm = q; q += 3 * P; float* op = q; q += P; float* col = q; q += 2 * P;
                float* m3 = q; q += 3 * P; float* tm = q; q += 9 * P; float* t2 = q; q += 3 * P; float* sc = q; q += 2 * P;
                float* rot = q; q += 4 * P; float* dep = q; q += P;
                if (sfo_backward(h, j->P, sfo_num_rendered(h), j->bg, j->W, j->H, j->means, j->colors, j->scales, 1.0f, j->rots, j->vm,
                                 j->beams, radii, j->g0, j->g1, m2, nrm, op, col, m3, tm, t2, sc, rot, dep) != 0) j->failed = 1;
            }
            sfo_free(h);
        }
    }
    free(img); free(radii); free(g);
    return NULL;
}

/* g0/g1/g2: upstream gradients (3-D: colour[2N], depth[N], occ[N]; surfel: colour[2N], others[7N], unused). */
double lgo_bench_frames(int surfel, int threads, int frames_each, int fwd_only, int P, int W, int H, int far_, int near_,
                        const float* bg, const float* means, const float* colors, const float* opac, const float* scales,
                        const float* rots, const float* vm, const float* beams, const float* g0, const float* g1, const float* g2) {
    if (threads < 1) threads = 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    job_t* jobs = (job_t*)malloc(sizeof(job_t) * (size_t)threads);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; t++) {
        job_t j = { surfel, frames_each, fwd_only, P, W, H, far_, near_, bg, means, colors, opac, scales, rots, vm, beams, g0, g1, g2, 0 };
        jobs[t] = j;
        if (pthread_create(&th[t], NULL, worker, &jobs[t]) != 0) { jobs[t].failed = 1; th[t] = 0; worker(&jobs[t]); }
    }
    int failed = 0;
    for (int t = 0; t < threads; t++) { if (th[t]) pthread_join(th[t], NULL); failed |= jobs[t].failed; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th); free(jobs);
    if (failed) return -1.0;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
