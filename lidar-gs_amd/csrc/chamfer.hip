// chamfer.hip -- exact nearest neighbour between two point clouds (include/lidargs_chamfer.h).
//
// Round 4: a uniform-grid search in front of the brute force.  The reference compares every pair (2.9e10 per direction at 170 k points);
// the answer -- the smallest fp32 `dx^2 + dy^2 + dz^2`, lowest index on ties -- only needs the pairs that can be nearest.  Per direction:
// bounding box of both clouds -> grid description (about eight cells per target point, at most 1024 per axis) -> counting sort of the
// targets into cells -> one thread per query walks the cell cube around its own cell ring by ring, evaluating the reference's
// expression on the candidates, and stops once its best distance is below the distance to everything outside the cube (with margins
// for the roundings of the cell assignment and of the distance).  Queries that are not settled within CH_RINGS rings raise a flag,
// and the brute-force launch behind -- which otherwise leaves on one load -- then recomputes that direction completely: pathological
// inputs (a query cloud far from the targets) cost what round 3 cost, everything else ~50x less.  Same distances and indices bit for
// bit (tests/test_chamfer.py).
//
// The brute force:
// 170 k x 170 k points per evaluated frame = 2.9e10 point pairs per direction: pure VALU work (8 ops per pair as written, no
// contraction: the squared distance must round exactly like the reference's so that the argmin index is bit-identical), so the
// design is about issue efficiency: each thread owns CH_Q queries in registers, a block streams the other cloud through LDS in
// tiles, and every LDS read (a broadcast: all lanes read the same target) is amortised over CH_Q x 64 pairs.
// The reference (chamfer3D.cu:8-138) gives each thread one query and re-reads shared memory for every pair.
#include "lidargs_common.h"
#include "../../include/lidargs_rasterizer.h"
#include "../../include/lidargs_chamfer.h"
#include <algorithm>
#include <stdlib.h>

namespace lg {

#define CH_BLOCK 256
#define CH_Q 4
#define CH_TILE 1024
#define CH_SPLIT 8       // the target cloud is cut into CH_SPLIT slices per query block: 170 k queries alone are only 166 blocks

// One block = CH_BLOCK x CH_Q queries against one slice of the targets.  Partial results are merged with a 64-bit atomicMin on
// (distance bits << 32 | index): distances are >= 0, so their bit patterns order like the values, and among equal distances the
// lowest index wins -- exactly the reference's tie rule.
__global__ void __launch_bounds__(CH_BLOCK) k_chamfer_nn(int n, int m, const float* __restrict__ a, const float* __restrict__ b,
                                                         unsigned long long* __restrict__ keys, const uint32_t* __restrict__ need, uint32_t few) {
    __shared__ float s_b[CH_TILE * 3];
    if (need && *need <= few) return;                                  // the grid search settled (nearly) every query of this direction: k_chamfer_nn_listed's turn, or nobody's
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

__global__ void __launch_bounds__(256) k_chamfer_unpack(size_t count, const unsigned long long* __restrict__ keys, float* __restrict__ dist, int* __restrict__ idx,
                                                        const uint32_t* __restrict__ need, uint32_t few) {
    if (need && *need <= few) return;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const unsigned long long k = keys[i];
    dist[i] = __uint_as_float((unsigned)(k >> 32)); idx[i] = (int)(unsigned)(k & 0xFFFFFFFFull);
}


// The few queries the grid search could not settle, one workgroup per listed query (block b takes items b, b + gridDim, ...): 256
// threads stride over the targets with the reference's expression (each sees ascending indices, so the strict compare keeps the lowest
// index of its share), the shares are merged by the minimum of (distance bits << 32 | index) -- the tie rule again.
__global__ void __launch_bounds__(256) k_chamfer_nn_listed(int m, const float* __restrict__ a, const float* __restrict__ b, const uint32_t* __restrict__ count,
                                                           uint32_t few, const uint32_t* __restrict__ list, float* __restrict__ dist, int* __restrict__ idx) {
    __shared__ unsigned long long s_k[4];
    const uint32_t cnt = *count;
    if (cnt == 0u || cnt > few) return;
    for (uint32_t it = blockIdx.x; it < cnt; it += gridDim.x) {
        const uint32_t q = list[it];
        const float qx = a[3 * (size_t)q], qy = a[3 * (size_t)q + 1], qz = a[3 * (size_t)q + 2];
        float best = __int_as_float(0x7f800000); int bi = 0;
        for (int k = threadIdx.x; k < m; k += 256) {
            const float dx = b[3 * (size_t)k] - qx, dy = b[3 * (size_t)k + 1] - qy, dz = b[3 * (size_t)k + 2] - qz;   // chamfer3D.cu:36-38
            const float d = dx * dx + dy * dy + dz * dz;                                                               // :39
            const bool better = d < best;
            best = better ? d : best; bi = better ? k : bi;
        }
        unsigned long long key = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)bi;
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long other = __shfl_xor(key, o); key = other < key ? other : key; }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_k[threadIdx.x >> 6] = key;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long k0 = s_k[0];
            for (int w = 1; w < 4; w++) k0 = s_k[w] < k0 ? s_k[w] : k0;
            dist[q] = __uint_as_float((unsigned)(k0 >> 32)); idx[q] = (int)(unsigned)(k0 & 0xFFFFFFFFull);
        }
    }
}

// ---- uniform-grid search ------------------------------------------------------------------------------------------------------------
#define CH_RINGS 6                       // rings of cells a query may walk before it gives up (13^3 cells)
#define CH_MAX_CELLS (1u << 22)
struct ChGrid { float ox, oy, oz, h, inv_h, eps; int dx, dy, dz; uint32_t cells; };
// monotone uint encoding of a float: atomicMin / atomicMax on it order like the floats
__device__ __forceinline__ uint32_t ch_enc(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ch_dec(uint32_t e) { return __uint_as_float((e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e); }

// box[0..2] = min, box[3..5] = max (encoded) over both clouds: every query lies inside the grid
__global__ void __launch_bounds__(256) k_ch_bbox(int n, const float* __restrict__ a, int m, const float* __restrict__ b, uint32_t* __restrict__ box) {
    __shared__ uint32_t s[6][4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    if (i < n + m) {
        const float* p = i < n ? a + 3 * (size_t)i : b + 3 * (size_t)(i - n);
        for (int k = 0; k < 3; k++) { const uint32_t e = ch_enc(p[k]); lo[k] = e; hi[k] = e; }
    }
    for (int k = 0; k < 3; k++)
        for (int o = 32; o > 0; o >>= 1) { lo[k] = min(lo[k], (uint32_t)__shfl_xor((int)lo[k], o)); hi[k] = max(hi[k], (uint32_t)__shfl_xor((int)hi[k], o)); }
    if ((threadIdx.x & 63) == 0) for (int k = 0; k < 3; k++) { s[k][threadIdx.x >> 6] = lo[k]; s[3 + k][threadIdx.x >> 6] = hi[k]; }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(box + threadIdx.x, min(min(s[threadIdx.x][0], s[threadIdx.x][1]), min(s[threadIdx.x][2], s[threadIdx.x][3])));
    else if (threadIdx.x < 6) atomicMax(box + threadIdx.x, max(max(s[threadIdx.x][0], s[threadIdx.x][1]), max(s[threadIdx.x][2], s[threadIdx.x][3])));
}
// one thread: about eight cells per target, at most 1024 per axis and CH_MAX_CELLS in all
__global__ void k_ch_grid_desc(int m, uint32_t max_cells, const uint32_t* __restrict__ box, ChGrid* __restrict__ g) {
    float lo[3], ex[3], maxabs = 0.f, emax = 0.f;
    for (int k = 0; k < 3; k++) {
        lo[k] = ch_dec(box[k]); const float hi = ch_dec(box[3 + k]);
        ex[k] = fmaxf(hi - lo[k], 0.f); emax = fmaxf(emax, ex[k]); maxabs = fmaxf(maxabs, fmaxf(fabsf(lo[k]), fabsf(hi)));
    }
    const float emin = fmaxf(emax * 1e-3f, 1e-12f);
    const float want = fminf(fmaxf(8.f * (float)m, 64.f), 0.5f * (float)max_cells);
    float h = cbrtf(fmaxf(ex[0], emin) * fmaxf(ex[1], emin) * fmaxf(ex[2], emin) / want);
    h = fmaxf(h, fmaxf(emax / 1024.f, 1e-12f));
    int d[3];
    for (int it = 0; it < 40; it++) {
        for (int k = 0; k < 3; k++) d[k] = min(1024, max(1, (int)(ex[k] / h) + 1));
        if ((unsigned long long)d[0] * d[1] * d[2] <= (unsigned long long)max_cells && (float)d[0] * h >= ex[0] && (float)d[1] * h >= ex[1] && (float)d[2] * h >= ex[2]) break;
        h *= 1.26f;
    }
    // (the loop always ends within its 40 steps for finite extents -- h grows 10^4-fold --; a non-finite bounding box, NaN or inf
    //  coordinates, must still leave a grid the arrays were sized for: one cell, every query scans every target)
    if (!((unsigned long long)d[0] * d[1] * d[2] <= (unsigned long long)max_cells) || !(h > 0.f) || !(h < 3.0e38f)) { d[0] = d[1] = d[2] = 1; h = fmaxf(emax, 1e-12f) * 1.01f; if (!(h < 3.0e38f)) h = 3.0e38f; }
    g->ox = lo[0]; g->oy = lo[1]; g->oz = lo[2]; g->h = h; g->inv_h = 1.f / h;
    g->eps = 2e-3f * h + 1e-6f * maxabs;                               // what the cell assignment's rounding can move a point by, with room
    g->dx = d[0]; g->dy = d[1]; g->dz = d[2]; g->cells = (uint32_t)d[0] * d[1] * d[2];
}
__device__ __forceinline__ int3 ch_cell(const ChGrid& g, float x, float y, float z) {
    return make_int3(min(g.dx - 1, max(0, (int)floorf((x - g.ox) * g.inv_h))), min(g.dy - 1, max(0, (int)floorf((y - g.oy) * g.inv_h))),
                     min(g.dz - 1, max(0, (int)floorf((z - g.oz) * g.inv_h))));
}
__global__ void __launch_bounds__(256) k_ch_count(int m, const float* __restrict__ b, const ChGrid* __restrict__ gp, uint32_t* __restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const ChGrid g = *gp;
    const int3 c = ch_cell(g, b[3 * (size_t)i], b[3 * (size_t)i + 1], b[3 * (size_t)i + 2]);
    atomicAdd(cnt + ((size_t)c.z * g.dy + c.y) * g.dx + c.x, 1u);
}
__global__ void __launch_bounds__(256) k_ch_fill(int m, const float* __restrict__ b, const ChGrid* __restrict__ gp, const uint32_t* __restrict__ start,
                                                 uint32_t* __restrict__ cursor, float4* __restrict__ sorted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const ChGrid g = *gp;
    const float x = b[3 * (size_t)i], y = b[3 * (size_t)i + 1], z = b[3 * (size_t)i + 2];
    const int3 c = ch_cell(g, x, y, z);
    const size_t cell = ((size_t)c.z * g.dy + c.y) * g.dx + c.x;
    sorted[start[cell] + atomicAdd(cursor + cell, 1u)] = make_float4(x, y, z, __int_as_float(i));   // (order inside a cell: whatever; the search takes the minimum of (distance, index))
}
// One thread per query.  Ring r adds the cells at Chebyshev distance r from the query's own cell.  Behind ring r every unscanned point
// lies beyond one of the cube's faces that the grid does not clip: at least `bound` away along that axis.
__global__ void __launch_bounds__(256) k_ch_query(int n, const float* __restrict__ a, const ChGrid* __restrict__ gp, const uint32_t* __restrict__ start,
                                                  const uint32_t* __restrict__ cnt, const float4* __restrict__ sorted, float* __restrict__ dist,
                                                  int* __restrict__ idx, uint32_t* __restrict__ unsettled, uint32_t* __restrict__ list) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ChGrid g = *gp;
    const float qx = a[3 * (size_t)i], qy = a[3 * (size_t)i + 1], qz = a[3 * (size_t)i + 2];
    const int3 c = ch_cell(g, qx, qy, qz);
    float best = __int_as_float(0x7f800000); int bi = 0x7fffffff;
    bool settled = false;
    for (int r = 0; r <= CH_RINGS && !settled; r++) {
        const int z0 = max(0, c.z - r), z1 = min(g.dz - 1, c.z + r), y0 = max(0, c.y - r), y1 = min(g.dy - 1, c.y + r), x0 = max(0, c.x - r), x1 = min(g.dx - 1, c.x + r);
        for (int z = z0; z <= z1; z++)
            for (int y = y0; y <= y1; y++) {
                const bool shell_row = (abs(z - c.z) == r) || (abs(y - c.y) == r);
                for (int x = x0; x <= x1; x += (shell_row || r == 0) ? 1 : max(1, x1 - x0)) {     // inside the cube only the two end cells of a row are new
                    if (!shell_row && abs(x - c.x) != r) continue;
                    const size_t cell = ((size_t)z * g.dy + y) * g.dx + x;
                    const uint32_t s0 = start[cell], s1 = s0 + cnt[cell];
                    for (uint32_t k = s0; k < s1; k++) {
                        const float4 p = sorted[k];
                        const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;        // chamfer3D.cu:36-38
                        const float d = dx * dx + dy * dy + dz * dz;                    // :39
                        const int pi = __float_as_int(p.w);
                        if (d < best || (d == best && pi < bi)) { best = d; bi = pi; }  // lowest index on ties (:40, :130)
                    }
                }
            }
        // distance to the nearest face of the scanned cube that has cells behind it
        float bound = 3.0e38f;
        if (c.x - r > 0) bound = fminf(bound, qx - (g.ox + (float)(c.x - r) * g.h));
        if (c.x + r < g.dx - 1) bound = fminf(bound, (g.ox + (float)(c.x + r + 1) * g.h) - qx);
        if (c.y - r > 0) bound = fminf(bound, qy - (g.oy + (float)(c.y - r) * g.h));
        if (c.y + r < g.dy - 1) bound = fminf(bound, (g.oy + (float)(c.y + r + 1) * g.h) - qy);
        if (c.z - r > 0) bound = fminf(bound, qz - (g.oz + (float)(c.z - r) * g.h));
        if (c.z + r < g.dz - 1) bound = fminf(bound, (g.oz + (float)(c.z + r + 1) * g.h) - qz);
        if (bound > 1.0e38f) settled = bi != 0x7fffffff;               // the cube covers the whole grid
        else {
            bound -= g.eps;
            settled = bi != 0x7fffffff && bound > 0.f && best < bound * bound * 0.99999f;
        }
    }
    if (settled) { dist[i] = best; idx[i] = bi; }
    else list[atomicAdd(unsettled, 1u)] = (uint32_t)i;                 // (order: whatever -- every listed query is searched on its own)
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

namespace {
struct ChWork { unsigned long long* keys; uint32_t* list; uint32_t* box; lg::ChGrid* grid; uint32_t* flags; uint32_t* cnt; uint32_t* start; uint32_t* cursor; uint32_t* scan; float4* sorted; size_t cells; };
size_t ch_cells(int n, int m) { const size_t t = (size_t)(n > m ? n : m); return std::min<size_t>(CH_MAX_CELLS, std::max<size_t>(4096, 16 * t)); }
size_t ch_carve(char* base, int n, int m, ChWork* w) {
    lg::Carver c(base);
    ChWork k;
    k.cells = ch_cells(n, m);
    k.keys = c.take<unsigned long long>((size_t)(n > m ? n : m) + 1);
    k.list = c.take<uint32_t>((size_t)(n > m ? n : m) + 1);
    k.box = c.take<uint32_t>(8); k.grid = c.take<lg::ChGrid>(1); k.flags = c.take<uint32_t>(2);
    k.cnt = c.take<uint32_t>(k.cells); k.start = c.take<uint32_t>(k.cells); k.cursor = c.take<uint32_t>(k.cells);
    k.scan = c.take<uint32_t>(lg::scan_scratch_words(k.cells));
    k.sorted = c.take<float4>((size_t)(n > m ? n : m) + 1);
    if (w) *w = k;
    return (size_t)(c.p - base) + 256;
}
}  // namespace

size_t lidargs_chamfer_scratch_bytes(int B, int n, int m) {
    (void)B;                                                           // the batches run one after the other through the same work area
    return ch_carve(nullptr, n > 0 ? n : 0, m > 0 ? m : 0, nullptr);
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
    static const bool brute_only = [] { const char* e = getenv("LIDARGS_CHAMFER_BRUTE"); return e && atoi(e) != 0; }();   // A/B, tests: the round-3 path
    ChWork w; ch_carve(scratch, n, m, &w);
    const int per = CH_BLOCK * CH_Q;
    hipError_t e = hipSuccess;
    auto direction = [&](int nq, const float* q, int nt, const float* t, float* dist, int* idx, uint32_t* flag) {
        // grid search of the nq queries among the nt targets; *flag afterwards: the number of queries it did not settle (their indices in w.list)
        const uint32_t few = (uint32_t)nq / 16u;
        if (!brute_only) {
            (void)hipMemsetAsync(w.cnt, 0, sizeof(uint32_t) * w.cells, stream);
            (void)hipMemsetAsync(w.cursor, 0, sizeof(uint32_t) * w.cells, stream);
            hipLaunchKernelGGL(lg::k_ch_grid_desc, dim3(1), dim3(1), 0, stream, nt, (uint32_t)w.cells, w.box, w.grid);
            hipLaunchKernelGGL(lg::k_ch_count, dim3((nt + 255) / 256), dim3(256), 0, stream, nt, t, w.grid, w.cnt);
            lg::launch_exclusive_scan(w.cnt, w.start, w.cells, nullptr, w.scan, stream);
            hipLaunchKernelGGL(lg::k_ch_fill, dim3((nt + 255) / 256), dim3(256), 0, stream, nt, t, w.grid, w.start, w.cursor, w.sorted);
            hipLaunchKernelGGL(lg::k_ch_query, dim3((nq + 255) / 256), dim3(256), 0, stream, nq, q, w.grid, w.start, w.cnt, w.sorted, dist, idx, flag, w.list);
            // a few unsettled queries: each searched exhaustively on its own (leaves on one load when there are none, or too many)
            hipLaunchKernelGGL(lg::k_chamfer_nn_listed, dim3((unsigned)std::min<uint32_t>(std::max<uint32_t>(few, 1u), 2048u)), dim3(256), 0, stream, nt, q, t, flag, few, w.list, dist, idx);
        }
        // the brute force behind it: leaves on one load unless more than `few` queries are unsettled (or it is all there is)
        (void)hipMemsetAsync(w.keys, 0xFF, sizeof(unsigned long long) * (size_t)nq, stream);
        hipLaunchKernelGGL(lg::k_chamfer_nn, dim3((nq + per - 1) / per, CH_SPLIT, 1), dim3(CH_BLOCK), 0, stream, nq, nt, q, t, w.keys, brute_only ? nullptr : flag, few);
        hipLaunchKernelGGL(lg::k_chamfer_unpack, dim3((unsigned)(((size_t)nq + 255) / 256)), dim3(256), 0, stream, (size_t)nq, w.keys, dist, idx, brute_only ? nullptr : flag, few);
    };
    for (int b = 0; b < B; b++) {
        const float* a1 = xyz1 + (size_t)b * n * 3; const float* a2 = xyz2 + (size_t)b * m * 3;
        e = hipMemsetAsync(w.box, 0xFF, 3 * sizeof(uint32_t), stream);                   // minima start at the top of the encoding
        if (e == hipSuccess) e = hipMemsetAsync(w.box + 3, 0, 5 * sizeof(uint32_t), stream);   // maxima at the bottom
        if (e == hipSuccess) e = hipMemsetAsync(w.flags, 0, 2 * sizeof(uint32_t), stream);
        if (e != hipSuccess) return lg::api_fail(LIDARGS_ERR_HIP, hipGetErrorString(e));
        if (!brute_only) hipLaunchKernelGGL(lg::k_ch_bbox, dim3((n + m + 255) / 256), dim3(256), 0, stream, n, a1, m, a2, w.box);
        direction(n, a1, m, a2, dist1 + (size_t)b * n, idx1 + (size_t)b * n, w.flags);
        direction(m, a2, n, a1, dist2 + (size_t)b * m, idx2 + (size_t)b * m, w.flags + 1);
    }
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
