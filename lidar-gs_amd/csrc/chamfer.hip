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
                                                         unsigned long long* __restrict__ keys, const uint32_t* __restrict__ need, uint32_t few,
                                                         const uint32_t* __restrict__ n_dev, const uint32_t* __restrict__ m_dev) {
    __shared__ float s_b[CH_TILE * 3];
    if (n_dev) { n = min(n, (int)*n_dev); m = min(m, (int)*m_dev); }   // the clouds' sizes live on the device (lidargs_points_meter): n, m are capacities
    if (need && *need <= (n_dev ? (uint32_t)n / 16u : few)) return;                                  // the grid search settled (nearly) every query of this direction: k_chamfer_nn_listed's turn, or nobody's
    const int batch = blockIdx.z;
    a += (size_t)batch * n * 3; b += (size_t)batch * m * 3; keys += (size_t)batch * n;
    const int q0 = (blockIdx.x * CH_BLOCK + threadIdx.x) * CH_Q;
    if (blockIdx.x * CH_BLOCK * CH_Q >= n) return;                      // (block-uniform)
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
                                                        const uint32_t* __restrict__ need, uint32_t few, const uint32_t* __restrict__ n_dev) {
    if (n_dev) count = min(count, (size_t)*n_dev);
    if (need && *need <= (n_dev ? (uint32_t)count / 16u : few)) return;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const unsigned long long k = keys[i];
    dist[i] = __uint_as_float((unsigned)(k >> 32)); idx[i] = (int)(unsigned)(k & 0xFFFFFFFFull);
}


// The few queries the grid search could not settle, one workgroup per listed query (block b takes items b, b + gridDim, ...): 256
// threads stride over the targets with the reference's expression (each sees ascending indices, so the strict compare keeps the lowest
// index of its share), the shares are merged by the minimum of (distance bits << 32 | index) -- the tie rule again.
__global__ void __launch_bounds__(256) k_chamfer_nn_listed(int m, const float* __restrict__ a, const float* __restrict__ b, const uint32_t* __restrict__ count,
                                                           uint32_t few, const uint32_t* __restrict__ list, float* __restrict__ dist, int* __restrict__ idx,
                                                           const uint32_t* __restrict__ n_dev, const uint32_t* __restrict__ m_dev) {
    __shared__ unsigned long long s_k[4];
    if (n_dev) { m = min(m, (int)*m_dev); few = *n_dev / 16u; }
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
__global__ void __launch_bounds__(256) k_ch_bbox(int n, const float* __restrict__ a, int m, const float* __restrict__ b, uint32_t* __restrict__ box,
                                                 const uint32_t* __restrict__ nm_dev) {
    __shared__ uint32_t s[6][4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    const int cap_n = n;                                               // thread i < cap_n looks at cloud 1, the others at cloud 2
    if (nm_dev) { n = min(n, (int)nm_dev[0]); m = min(m, (int)nm_dev[1]); }
    if (i < cap_n ? i < n : i - cap_n < m) {
        const float* p = i < cap_n ? a + 3 * (size_t)i : b + 3 * (size_t)(i - cap_n);
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
__global__ void k_ch_grid_desc(int m, uint32_t max_cells, const uint32_t* __restrict__ box, ChGrid* __restrict__ g, const uint32_t* __restrict__ m_dev) {
    if (m_dev) m = min(m, (int)*m_dev);
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
__global__ void __launch_bounds__(256) k_ch_count(int m, const float* __restrict__ b, const ChGrid* __restrict__ gp, uint32_t* __restrict__ cnt,
                                                  const uint32_t* __restrict__ m_dev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (m_dev) m = min(m, (int)*m_dev);
    if (i >= m) return;
    const ChGrid g = *gp;
    const int3 c = ch_cell(g, b[3 * (size_t)i], b[3 * (size_t)i + 1], b[3 * (size_t)i + 2]);
    atomicAdd(cnt + ((size_t)c.z * g.dy + c.y) * g.dx + c.x, 1u);
}
__global__ void __launch_bounds__(256) k_ch_fill(int m, const float* __restrict__ b, const ChGrid* __restrict__ gp, const uint32_t* __restrict__ start,
                                                 uint32_t* __restrict__ cursor, float4* __restrict__ sorted, const uint32_t* __restrict__ m_dev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (m_dev) m = min(m, (int)*m_dev);
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
                                                  int* __restrict__ idx, uint32_t* __restrict__ unsettled, uint32_t* __restrict__ list, const uint32_t* __restrict__ n_dev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) n = min(n, (int)*n_dev);
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

namespace {
// nm_dev (nullable): device words (n, m) -- the clouds' sizes when they are only known on the device (lidargs_points_meter: the launches
// then cover the capacities n, m and every kernel clamps to the real sizes)
int chamfer_forward_impl(int B, int n, int m, const float* xyz1, const float* xyz2, float* dist1, float* dist2, int* idx1, int* idx2,
                         char* scratch, const uint32_t* nm_dev, hipStream_t stream) {
    static const bool brute_only = [] { const char* e = getenv("LIDARGS_CHAMFER_BRUTE"); return e && atoi(e) != 0; }();   // A/B, tests: the round-3 path
    ChWork w; ch_carve(scratch, n, m, &w);
    const int per = CH_BLOCK * CH_Q;
    hipError_t e = hipSuccess;
    auto direction = [&](int nq, const float* q, int nt, const float* t, float* dist, int* idx, uint32_t* flag, const uint32_t* nq_dev, const uint32_t* nt_dev) {
        // grid search of the nq queries among the nt targets; *flag afterwards: the number of queries it did not settle (their indices in w.list)
        const uint32_t few = (uint32_t)nq / 16u;
        if (!brute_only) {
            (void)hipMemsetAsync(w.cnt, 0, sizeof(uint32_t) * w.cells, stream);
            (void)hipMemsetAsync(w.cursor, 0, sizeof(uint32_t) * w.cells, stream);
            hipLaunchKernelGGL(lg::k_ch_grid_desc, dim3(1), dim3(1), 0, stream, nt, (uint32_t)w.cells, w.box, w.grid, nt_dev);
            hipLaunchKernelGGL(lg::k_ch_count, dim3((nt + 255) / 256), dim3(256), 0, stream, nt, t, w.grid, w.cnt, nt_dev);
            lg::launch_exclusive_scan(w.cnt, w.start, w.cells, nullptr, w.scan, stream);
            hipLaunchKernelGGL(lg::k_ch_fill, dim3((nt + 255) / 256), dim3(256), 0, stream, nt, t, w.grid, w.start, w.cursor, w.sorted, nt_dev);
            hipLaunchKernelGGL(lg::k_ch_query, dim3((nq + 255) / 256), dim3(256), 0, stream, nq, q, w.grid, w.start, w.cnt, w.sorted, dist, idx, flag, w.list, nq_dev);
            // a few unsettled queries: each searched exhaustively on its own (leaves on one load when there are none, or too many)
            hipLaunchKernelGGL(lg::k_chamfer_nn_listed, dim3((unsigned)std::min<uint32_t>(std::max<uint32_t>(few, 1u), 2048u)), dim3(256), 0, stream, nt, q, t, flag, few, w.list, dist, idx,
                               nq_dev, nt_dev);
        }
        // the brute force behind it: leaves on one load unless more than `few` queries are unsettled (or it is all there is)
        (void)hipMemsetAsync(w.keys, 0xFF, sizeof(unsigned long long) * (size_t)nq, stream);
        hipLaunchKernelGGL(lg::k_chamfer_nn, dim3((nq + per - 1) / per, CH_SPLIT, 1), dim3(CH_BLOCK), 0, stream, nq, nt, q, t, w.keys, brute_only ? nullptr : flag, few, nq_dev, nt_dev);
        hipLaunchKernelGGL(lg::k_chamfer_unpack, dim3((unsigned)(((size_t)nq + 255) / 256)), dim3(256), 0, stream, (size_t)nq, w.keys, dist, idx, brute_only ? nullptr : flag, few, nq_dev);
    };
    for (int b = 0; b < B; b++) {
        const float* a1 = xyz1 + (size_t)b * n * 3; const float* a2 = xyz2 + (size_t)b * m * 3;
        e = hipMemsetAsync(w.box, 0xFF, 3 * sizeof(uint32_t), stream);                   // minima start at the top of the encoding
        if (e == hipSuccess) e = hipMemsetAsync(w.box + 3, 0, 5 * sizeof(uint32_t), stream);   // maxima at the bottom
        if (e == hipSuccess) e = hipMemsetAsync(w.flags, 0, 2 * sizeof(uint32_t), stream);
        if (e != hipSuccess) return lg::api_fail(LIDARGS_ERR_HIP, hipGetErrorString(e));
        if (!brute_only) hipLaunchKernelGGL(lg::k_ch_bbox, dim3((n + m + 255) / 256), dim3(256), 0, stream, n, a1, m, a2, w.box, nm_dev);
        direction(n, a1, m, a2, dist1 + (size_t)b * n, idx1 + (size_t)b * n, w.flags, nm_dev, nm_dev ? nm_dev + 1 : nullptr);
        direction(m, a2, n, a1, dist2 + (size_t)b * m, idx2 + (size_t)b * m, w.flags + 1, nm_dev ? nm_dev + 1 : nullptr, nm_dev);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return lg::api_fail(LIDARGS_ERR_HIP, hipGetErrorString(e));
    return 0;
}
}  // namespace

int lidargs_chamfer_forward(int B, int n, int m, const float* xyz1, const float* xyz2, float* dist1, float* dist2, int* idx1, int* idx2,
                            char* scratch, size_t scratch_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (B < 0 || n < 0 || m < 0) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "chamfer_forward: negative size");
    if (B == 0) return 0;
    if ((n > 0 && m == 0) || (m > 0 && n == 0)) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "chamfer_forward: one of the clouds is empty");
    if (n == 0) return 0;
    if (!xyz1 || !xyz2 || !dist1 || !dist2 || !idx1 || !idx2 || !scratch) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "chamfer_forward: NULL pointer");
    if (scratch_bytes < lidargs_chamfer_scratch_bytes(B, n, m)) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "chamfer_forward: scratch too small");
    return chamfer_forward_impl(B, n, m, xyz1, xyz2, dist1, dist2, idx1, idx2, scratch, nullptr, stream);
}

// ---- PointsMeter on the device (utils/lidar_utils.py:234-292) -----------------------------------------------------------------------
// The reference's evaluation metric takes the predicted and the ground-truth range image to the host, back-projects their non-empty pixels
// with numpy (pano_to_lidar, :171-232), sends the two clouds back to the GPU for the chamfer kernel and reads means and the F-score
// off the result.  Here the whole metric is one call on device-resident images:
//   flags (pixel != 0) of both images -> one exclusive scan -> the two compacted clouds in the reference's row-major order, each point =
//   pixel ray x range with the reference's float32 sequence of operations (beta = -(i - W/2)/W * 2 * pi, the beam of row j from the top,
//   cos / sin correctly rounded) -> nearest neighbours both ways (the grid search above, sizes on the device) -> mean squared distances
//   and the share below the threshold -> chamfer distance, F-score, precision, recall, the two point counts.  No host read.
struct PmWork { uint32_t* flags; uint32_t* offs; uint32_t* total; uint32_t* scan; uint32_t* nm; float* pts1; float* pts2; float* dist1; float* dist2;
                int* idx1; int* idx2; double* part; char* ch; size_t ch_bytes; };
static size_t pm_carve(char* base, int N, PmWork* w) {
    lg::Carver c(base);
    PmWork k;
    const size_t n = (size_t)N;
    k.flags = c.take<uint32_t>(2 * n + 1); k.offs = c.take<uint32_t>(2 * n + 1); k.total = c.take<uint32_t>(64);
    k.scan = c.take<uint32_t>(lg::scan_scratch_words(2 * n + 1)); k.nm = c.take<uint32_t>(4);
    k.pts1 = c.take<float>(3 * n + 4); k.pts2 = c.take<float>(3 * n + 4); k.dist1 = c.take<float>(n + 1); k.dist2 = c.take<float>(n + 1);
    k.idx1 = c.take<int>(n + 1); k.idx2 = c.take<int>(n + 1); k.part = c.take<double>(4 * 256);
    k.ch_bytes = lidargs_chamfer_scratch_bytes(1, N, N);
    k.ch = c.take<char>(k.ch_bytes);
    if (w) *w = k;
    return (size_t)(c.p - base) + 256;
}
}  // extern "C"  (the kernels below are C++)

namespace lg {
__global__ void __launch_bounds__(256) k_pm_flags(int N, const float* __restrict__ pred, const float* __restrict__ truth, float scale, uint32_t* __restrict__ flags) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > 2 * N) return;
    if (i == 2 * N) { flags[i] = 0u; return; }                         // (one word behind: the scan's last exclusive prefix is the total)
    const float v = (i < N ? pred[i] : truth[i - N]) / scale;          // preds / self.scale (:255-256), then pano != 0.0 (:209)
    flags[i] = v != 0.0f ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k_pm_points(int H, int W, const float* __restrict__ pred, const float* __restrict__ truth, float scale,
                                                   const float* __restrict__ beams, float fov_up, float fov, const uint32_t* __restrict__ flags,
                                                   const uint32_t* __restrict__ offs, float* __restrict__ pts1, float* __restrict__ pts2, uint32_t* __restrict__ nm) {
    const int N = H * W;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { nm[0] = offs[N]; nm[1] = offs[2 * N] - offs[N]; }
    if (i >= 2 * N || !flags[i]) return;
    const int cloud = i >= N, pix = i - cloud * N;
    const int row = pix / W, col = pix - row * W;
    const float pano = (cloud ? truth[pix] : pred[pix]) / scale;
    // utils/lidar_utils.py:186-199, float32 operation by operation (this file is built without contraction)
    const float pi_f = 3.14159265358979323846f;
    const float beta = ((-((float)col - (float)W / 2.0f)) / (float)W) * 2.0f * pi_f;
    float alpha;
    if (beams) alpha = beams[H - 1 - row];                             // beam_inclinations[::-1][j]
    else alpha = ((fov_up - (float)row / (float)H * fov) / 180.0f) * pi_f;
    const float ca = (float)cos((double)alpha), sa = (float)sin((double)alpha), cb = (float)cos((double)beta), sb = (float)sin((double)beta);
    float* q = (cloud ? pts2 + 3 * (size_t)(offs[i] - offs[N]) : pts1 + 3 * (size_t)offs[i]);
    q[0] = (ca * cb) * pano; q[1] = (ca * sb) * pano; q[2] = sa * pano;
}
// partial sums of (distance, distance < threshold) over both directions: block b -> part[b][0..3] (fixed order: deterministic)
__global__ void __launch_bounds__(256) k_pm_partial(const float* __restrict__ d1, const float* __restrict__ d2, const uint32_t* __restrict__ nm, float thr,
                                                    double* __restrict__ part) {
    __shared__ double s[4][4];
    const uint32_t n = nm[0], m = nm[1];
    double a = 0, ac = 0, b = 0, bc = 0;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) { const float v = d1[i]; a += (double)v; ac += v < thr ? 1.0 : 0.0; }
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256) { const float v = d2[i]; b += (double)v; bc += v < thr ? 1.0 : 0.0; }
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); ac += __shfl_xor(ac, o); b += __shfl_xor(b, o); bc += __shfl_xor(bc, o); }
    if ((threadIdx.x & 63) == 0) { double* q = s[threadIdx.x >> 6]; q[0] = a; q[1] = ac; q[2] = b; q[3] = bc; }
    __syncthreads();
    if (threadIdx.x < 4) part[4 * blockIdx.x + threadIdx.x] = s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x];
}
// out[0] = dist1.mean() + dist2.mean() (:274), [1] = F-score, [2] = precision, [3] = recall (extern/fscore.py:14-17; NaN -> 0), [4], [5] = points
__global__ void k_pm_final(const double* __restrict__ part, int nparts, const uint32_t* __restrict__ nm, float* __restrict__ out) {
    double v[4] = {0, 0, 0, 0};
    for (int b = 0; b < nparts; b++) for (int k = 0; k < 4; k++) v[k] += part[4 * b + k];
    const double n = (double)nm[0], m = (double)nm[1];
    const float mean1 = (float)(v[0] / n), mean2 = (float)(v[2] / m);     // an empty cloud: 0 / 0 = NaN, as torch's mean of an empty tensor
    const float p1 = (float)(v[1] / n), p2 = (float)(v[3] / m);
    float f = 2.f * p1 * p2 / (p1 + p2);
    if (f != f) f = 0.f;                                                 // fscore[torch.isnan(fscore)] = 0
    out[0] = mean1 + mean2; out[1] = f; out[2] = p1; out[3] = p2; out[4] = (float)nm[0]; out[5] = (float)nm[1];
}
}  // namespace lg

extern "C" {

size_t lidargs_points_meter_scratch_bytes(int H, int W) {
    if (H <= 0 || W <= 0) return 0;
    return pm_carve(nullptr, H * W, nullptr);
}

int lidargs_points_meter(int H, int W, const float* pred, const float* truth, float scale, const float* beam_inclinations, float fov_up, float fov,
                         float threshold, float* out, char* scratch, size_t scratch_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (H <= 0 || W <= 0 || (long long)H * W > (1ll << 28)) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "points_meter: bad image size");
    if (!pred || !truth || !out || !scratch) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "points_meter: NULL pointer");
    if (!(scale != 0.f)) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "points_meter: scale must not be 0");
    if (scratch_bytes < lidargs_points_meter_scratch_bytes(H, W)) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "points_meter: scratch too small");
    const int N = H * W;
    PmWork w; pm_carve(scratch, N, &w);
    hipLaunchKernelGGL(lg::k_pm_flags, dim3((2 * N + 1 + 255) / 256), dim3(256), 0, stream, N, pred, truth, scale, w.flags);
    lg::launch_exclusive_scan(w.flags, w.offs, (size_t)2 * N + 1, w.total, w.scan, stream);
    hipLaunchKernelGGL(lg::k_pm_points, dim3((2 * N + 255) / 256), dim3(256), 0, stream, H, W, pred, truth, scale, beam_inclinations, fov_up, fov, w.flags, w.offs,
                       w.pts1, w.pts2, w.nm);
    const int rc = chamfer_forward_impl(1, N, N, w.pts1, w.pts2, w.dist1, w.dist2, w.idx1, w.idx2, w.ch, w.nm, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(lg::k_pm_partial, dim3(256), dim3(256), 0, stream, w.dist1, w.dist2, w.nm, threshold, w.part);
    hipLaunchKernelGGL(lg::k_pm_final, dim3(1), dim3(1), 0, stream, w.part, 256, w.nm, out);
    const hipError_t e = hipGetLastError();
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
