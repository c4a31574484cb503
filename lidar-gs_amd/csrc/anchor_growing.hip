// anchor_growing.hip -- one level of GaussianModel.anchor_growing (include/lidargs_anchor_growing.h).
//
// The reference (/root/reference/scene/gaussian_model.py:677-775) builds the level's candidate mask with framework ops, quantises
// anchors and candidate offsets to voxels, takes torch.unique of the candidates' voxel rows, and then asks for every (candidate voxel,
// existing anchor) PAIR whether the rows are equal, 4096 anchors at a time (:714-727): O(candidates x anchors) integer compares --
// 1e4 x 1.2 M = 1.2e10 per level.  The feature maximum per voxel is torch_scatter.scatter_max (:742).
//
// Here the same sets come from hashing, in HBM-bound passes over the inputs:
//   k_ag_mark       N0*k offsets: the mask (:683-688) -> list of candidate slots (block-aggregated append) and their count
//   k_ag_box        candidates: positions (:698) and voxels (:709) -> their voxel bounding box           [host read: count, box]
//   k_ag_insert     candidates -> 64-bit voxel key packed to the box (x most significant: the key's order is torch.unique's row order),
//                   atomicCAS insert into an open-addressing set of >= 2 C slots; each candidate remembers its slot
//   k_ag_probe      the N existing anchors quantised the same way (:706) probe the set: a hit marks the voxel dead (:714-729)
//   k_ag_survivors  set slots -> the distinct voxels' count (:711) and the compacted live keys           [host read: V, U]
//   sort            the U live keys: LSD radix on the low and the high word (binning.hip's pair sort, value = position)
//   k_ag_emit       rank r -> new_anchor[r] = voxel * cur_size (:730); the voxel's set slot learns r
//   k_ag_features   C*F threads: atomicMax of an order-preserving integer image of anchor_feat[anchor of candidate][f] into new_feat[r][f]
//   k_ag_unmap      the image back to floats, in place
// Arithmetic that decides a voxel is written exactly as torch evaluates it: float32 product then sum for the position (no
// contraction: this file is built with -ffp-contract=off), the quotient either as x * float32(1 / cur_size), the reciprocal taken in double (torch's device kernel for
// tensor / python scalar, measured on this GPU: tools/div_convention.py, profiles/r06_div_convention.txt) or as an IEEE division (torch's CPU kernel; LIDARGS_AG_EXACT_DIVISION), round half to even, cast.
#include "lidargs_common.h"
#include "../../include/lidargs_rasterizer.h"
#include "../../include/lidargs_anchor_growing.h"
#include <limits.h>
#include <algorithm>

namespace lg {

#define AG_HDR 16          // u32 words: [0] candidates, [1..3] voxel min, [4..6] voxel max, [8] distinct voxels, [9] live voxels
#define AG_EMPTY 0xFFFFFFFFFFFFFFFFull

struct AgLevel {
    int N, N0, k, F;
    float thr, rthr, size, inv;
    int exact;
};
struct AgKey { int mn[3]; int mx[3]; int sy, sx; uint32_t mask; };   // key = dx << sx | dy << sy | dz; mask = table slots - 1

__device__ __forceinline__ int ag_voxel(float x, const AgLevel& p) {
    const float q = p.exact ? x / p.size : x * p.inv;
    return (int)__builtin_rintf(q);
}
__device__ __forceinline__ void ag_candidate_voxel(const AgLevel& p, uint32_t slot, const float* __restrict__ anchor, const float* __restrict__ offset,
                                                   const float* __restrict__ scaling, int g[3]) {
    const uint32_t a = slot / (uint32_t)p.k;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float t = offset[3 * (size_t)slot + c] * scaling[6 * (size_t)a + c];      // :698, product rounded, then the sum
        g[c] = ag_voxel(anchor[3 * (size_t)a + c] + t, p);
    }
}
__device__ __forceinline__ unsigned long long ag_pack(const AgKey& kd, const int g[3]) {
    return ((unsigned long long)((uint32_t)g[0] - (uint32_t)kd.mn[0]) << kd.sx) | ((unsigned long long)((uint32_t)g[1] - (uint32_t)kd.mn[1]) << kd.sy) |
           (unsigned long long)((uint32_t)g[2] - (uint32_t)kd.mn[2]);
}
__device__ __forceinline__ uint32_t ag_hash(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (uint32_t)k;
}
// Block-aggregated append (256 threads): thread t brings `n` items (its own count), gets the position of its first one.  ONE global
// atomic per block: with a wave-level append every wave of a 7.2 M-thread launch hit the same counter word (measured: 0.8 ms for a
// 65-MB pass that needs 15 us).
__device__ __forceinline__ uint32_t ag_block_append(uint32_t n, uint32_t* counter, uint32_t* s_wave /* [5] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t0 = s_wave[0], t1 = s_wave[1], t2 = s_wave[2], t3 = s_wave[3], tot = t0 + t1 + t2 + t3;
        const uint32_t base = tot ? atomicAdd(counter, tot) : 0u;
        s_wave[0] = base; s_wave[1] = base + t0; s_wave[2] = base + t0 + t1; s_wave[3] = base + t0 + t1 + t2;
    }
    __syncthreads();
    const uint32_t pos = s_wave[wave] + incl - n;
    __syncthreads();                                                     // (s_wave may be reused by the caller)
    return pos;
}

__global__ void k_ag_init(uint32_t* hdr) {
    const int t = threadIdx.x;
    if (t < AG_HDR) hdr[t] = (t >= 1 && t <= 3) ? (uint32_t)INT_MAX : (t >= 4 && t <= 6) ? (uint32_t)INT_MIN : 0u;
}

#define AG_MARK_ITEMS 16     // offsets per thread of the mask pass: 4096 per block, 1758 blocks at 7.2 M offsets
__global__ void __launch_bounds__(256) k_ag_mark(AgLevel p, const float* __restrict__ anchor, const float* __restrict__ offset, const float* __restrict__ scaling,
                                                 const float* __restrict__ grads, const uint8_t* __restrict__ omask, const float* __restrict__ rnd,
                                                 uint32_t* __restrict__ hdr, uint32_t* __restrict__ list, int vec) {
    __shared__ uint32_t s_wave[5];
    const size_t slots = (size_t)p.N0 * p.k;
    // thread t takes 16 CONSECUTIVE offsets: four 16-byte loads of the gradients, four of the random draws, one of the mask bytes
    // (48 scalar loads before: the pass ran at 1.2 TB/s); `vec` = the three arrays are 16-byte aligned (torch's allocations are)
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * AG_MARK_ITEMS;
    uint32_t bits = 0;
    if (vec && base + AG_MARK_ITEMS <= slots) {
        const float4* g4 = reinterpret_cast<const float4*>(grads + base);
        const uint4 m4 = *reinterpret_cast<const uint4*>(omask + base);
        float4 g[4], r[4];
#pragma unroll
        for (int q = 0; q < 4; q++) g[q] = g4[q];
        if (rnd) {
            const float4* r4 = reinterpret_cast<const float4*>(rnd + base);
#pragma unroll
            for (int q = 0; q < 4; q++) r[q] = r4[q];
        }
        const uint32_t mw[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float gv[4] = {g[q].x, g[q].y, g[q].z, g[q].w};
            const float rv[4] = {r[q].x, r[q].y, r[q].z, r[q].w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                bool cand = gv[e] >= p.thr && ((mw[q] >> (8 * e)) & 0xFFu) != 0;      // :683-684 (a NaN gradient is no candidate: the compare is false)
                if (rnd) cand = cand && rv[e] > p.rthr;                                 // :687-689
                bits |= (cand ? 1u : 0u) << (4 * q + e);
            }
        }
    } else {
#pragma unroll 4
        for (int it = 0; it < AG_MARK_ITEMS; it++) {
            const size_t s = base + (size_t)it;
            bool cand = false;
            if (s < slots) {
                cand = grads[s] >= p.thr && omask[s] != 0;
                if (cand && rnd) cand = rnd[s] > p.rthr;
            }
            bits |= (cand ? 1u : 0u) << it;
        }
    }
    uint32_t pos = ag_block_append((uint32_t)__builtin_popcount(bits), hdr, s_wave);
    for (uint32_t b = bits; b; b &= b - 1) list[pos++] = (uint32_t)(base + (size_t)__builtin_ctz(b));
}

// the candidates' voxel bounding box: one candidate per thread, every gather of the launch in flight at once (inside k_ag_mark the few
// lanes with candidates walked theirs one dependent round trip after the other: 52-74 us for a pass that streams in 15)
__global__ void __launch_bounds__(256) k_ag_box(AgLevel p, const float* __restrict__ anchor, const float* __restrict__ offset, const float* __restrict__ scaling,
                                                uint32_t* __restrict__ hdr, const uint32_t* __restrict__ list) {
    __shared__ int s_box[6];
    if (threadIdx.x < 6) s_box[threadIdx.x] = threadIdx.x < 3 ? INT_MAX : INT_MIN;
    __syncthreads();
    const uint32_t C = hdr[0];
    int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
    for (uint32_t c = blockIdx.x * 256 + threadIdx.x; c < C; c += gridDim.x * 256) {
        int g[3];
        ag_candidate_voxel(p, list[c], anchor, offset, scaling, g);
#pragma unroll
        for (int q = 0; q < 3; q++) { lo[q] = min(lo[q], g[q]); hi[q] = max(hi[q], g[q]); }
    }
#pragma unroll
    for (int q = 0; q < 3; q++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo[q] = min(lo[q], __shfl_xor(lo[q], o)); hi[q] = max(hi[q], __shfl_xor(hi[q], o)); }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 3; q++) { atomicMin(&s_box[q], lo[q]); atomicMax(&s_box[3 + q], hi[q]); }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int v = s_box[threadIdx.x];
        if (threadIdx.x < 3) { if (v != INT_MAX) atomicMin((int*)hdr + 1 + threadIdx.x, v); }
        else if (v != INT_MIN) atomicMax((int*)hdr + 1 + threadIdx.x, v);
    }
}

__global__ void __launch_bounds__(256) k_ag_insert(AgLevel p, AgKey kd, uint32_t C, const float* __restrict__ anchor, const float* __restrict__ offset,
                                                   const float* __restrict__ scaling, const uint32_t* __restrict__ list, unsigned long long* __restrict__ keys,
                                                   uint32_t* __restrict__ cand_slot) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    int g[3];
    ag_candidate_voxel(p, list[c], anchor, offset, scaling, g);
    const unsigned long long key = ag_pack(kd, g);
    uint32_t h = ag_hash(key) & kd.mask;
    for (;;) {
        const unsigned long long prev = atomicCAS(&keys[h], AG_EMPTY, key);
        if (prev == AG_EMPTY || prev == key) break;
        h = (h + 1) & kd.mask;
    }
    cand_slot[c] = h;
}

__global__ void __launch_bounds__(256) k_ag_probe(AgLevel p, AgKey kd, const float* __restrict__ anchor, const unsigned long long* __restrict__ keys,
                                                  int* __restrict__ vals) {
    const uint32_t a = blockIdx.x * 256 + threadIdx.x;
    if (a >= (uint32_t)p.N) return;
    int g[3];
    bool in = true;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        g[c] = ag_voxel(anchor[3 * (size_t)a + c], p);                   // :706
        in = in && g[c] >= kd.mn[c] && g[c] <= kd.mx[c];
    }
    if (!in) return;                                                     // outside the candidates' box: equal to none of them
    const unsigned long long key = ag_pack(kd, g);
    uint32_t h = ag_hash(key) & kd.mask;
    for (;;) {
        const unsigned long long cur = keys[h];
        if (cur == key) { vals[h] = -1; return; }                        // :714-729 remove_duplicates
        if (cur == AG_EMPTY) return;
        h = (h + 1) & kd.mask;
    }
}

#define AG_SURV_ITEMS 8
__global__ void __launch_bounds__(256) k_ag_survivors(uint32_t T, const unsigned long long* __restrict__ keys, const int* __restrict__ vals,
                                                      uint32_t* __restrict__ hdr, unsigned long long* __restrict__ surv) {
    __shared__ uint32_t s_wave[5];
    const uint32_t base = blockIdx.x * (256 * AG_SURV_ITEMS) + threadIdx.x;
    unsigned long long key[AG_SURV_ITEMS];
    uint32_t live = 0, used = 0;
#pragma unroll
    for (int it = 0; it < AG_SURV_ITEMS; it++) {
        const uint32_t h = base + it * 256;
        key[it] = h < T ? keys[h] : AG_EMPTY;
        if (key[it] != AG_EMPTY) { used++; if (vals[h] == 0) live |= 1u << it; }
    }
    (void)ag_block_append(used, hdr + 8, s_wave);                        // distinct voxels (:711)
    uint32_t pos = ag_block_append((uint32_t)__builtin_popcount(live), hdr + 9, s_wave);
#pragma unroll
    for (int it = 0; it < AG_SURV_ITEMS; it++) if (live >> it & 1u) surv[pos++] = key[it];
}

__global__ void __launch_bounds__(256) k_ag_split(uint32_t U, const unsigned long long* __restrict__ surv, const uint32_t* __restrict__ perm, int hi, uint32_t* __restrict__ out,
                                                  uint32_t* __restrict__ ident) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= U) return;
    const unsigned long long key = surv[perm ? perm[i] : i];
    out[i] = hi ? (uint32_t)(key >> 32) : (uint32_t)key;
    if (ident) ident[i] = i;                                             // a key of zero bits (one voxel) is not sorted at all: the permutation must exist anyway
}

__global__ void __launch_bounds__(256) k_ag_emit(AgLevel p, AgKey kd, uint32_t U, const unsigned long long* __restrict__ surv, const uint32_t* __restrict__ perm,
                                                 const unsigned long long* __restrict__ keys, int* __restrict__ vals, float* __restrict__ new_anchor) {
    const uint32_t r = blockIdx.x * 256 + threadIdx.x;
    if (r >= U) return;
    const unsigned long long key = surv[perm[r]];
    const uint32_t dz = (uint32_t)(key & ((1ull << kd.sy) - 1));
    const uint32_t dy = (uint32_t)((key >> kd.sy) & ((1ull << (kd.sx - kd.sy)) - 1));
    const uint32_t dx = (uint32_t)(key >> kd.sx);
    const int g[3] = {(int)((uint32_t)kd.mn[0] + dx), (int)((uint32_t)kd.mn[1] + dy), (int)((uint32_t)kd.mn[2] + dz)};
#pragma unroll
    for (int c = 0; c < 3; c++) new_anchor[3 * (size_t)r + c] = (float)g[c] * p.size;   // :730 int32 -> float32, times float32(cur_size)
    uint32_t h = ag_hash(key) & kd.mask;
    while (keys[h] != key) h = (h + 1) & kd.mask;                        // the key is in the set
    vals[h] = (int)r + 1;
}

// order-preserving image of a float in uint32 (x < y  <=>  image(x) < image(y); -0 below +0); 0 is below every image
__device__ __forceinline__ uint32_t ag_image(float x) { const uint32_t u = __float_as_uint(x); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ag_unimage(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

__global__ void __launch_bounds__(256) k_ag_features(int k, int F, uint32_t C, const uint32_t* __restrict__ list, const uint32_t* __restrict__ cand_slot,
                                                     const int* __restrict__ vals, const float* __restrict__ feat, uint32_t* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t c = (uint32_t)(t / (unsigned)F);
    if (c >= C) return;
    const int f = (int)(t % (unsigned)F);
    const int r = vals[cand_slot[c]];
    if (r <= 0) return;                                                  // the candidate's voxel already holds an anchor
    const uint32_t a = list[c] / (uint32_t)k;                            // :740 the candidate's own anchor's feature row
    atomicMax(out + (size_t)(r - 1) * F + f, ag_image(feat[(size_t)a * F + f]));    // :742 scatter_max
}

__global__ void __launch_bounds__(256) k_ag_unmap(size_t n, uint32_t* __restrict__ io) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n) io[t] = __float_as_uint(ag_unimage(io[t]));
}

}  // namespace lg

#define AG_HIP(call) do { hipError_t e_ = (hipError_t)(call); if (e_ != hipSuccess) return lg::api_fail(LIDARGS_ERR_HIP, hipGetErrorString(e_)); } while (0)

extern "C" {

size_t lidargs_ag_scratch_bytes(int N0, int n_offsets) {
    if (N0 < 0 || n_offsets < 1) return 0;
    return 256 + (size_t)AG_HDR * 4 + 128 + (size_t)N0 * n_offsets * sizeof(uint32_t);
}

int lidargs_anchor_growing_level(int N, int N0, int n_offsets, int feat_dim, const float* anchor, const float* offset, const float* scaling,
                                 const float* anchor_feat, const float* grads, const uint8_t* offset_mask, const float* rnd,
                                 float grad_threshold, float rand_threshold, double cur_size, int flags, char* scratch, size_t scratch_bytes,
                                 lidargs_alloc_fn alloc_work, void* work_user, lidargs_alloc_fn alloc_anchor, void* anchor_user,
                                 lidargs_alloc_fn alloc_feat, void* feat_user, int* counts_host, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (counts_host) counts_host[0] = counts_host[1] = counts_host[2] = 0;
    if (N < 0 || N0 < 0 || N0 > N || n_offsets < 1 || feat_dim < 1 || feat_dim > 256) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "anchor_growing: bad sizes");
    if ((size_t)N * n_offsets >= ((size_t)1 << 31)) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "anchor_growing: N * n_offsets must be below 2^31");
    if (!(cur_size > 0.0) || !(cur_size < (double)INFINITY) || !((float)cur_size > 0.0f)) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "anchor_growing: cur_size must be positive and finite");
    if (N0 == 0) return 0;
    if (!anchor || !offset || !scaling || !anchor_feat || !grads || !offset_mask || !scratch || !alloc_work || !alloc_anchor || !alloc_feat)
        return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "anchor_growing: NULL pointer");
    if (scratch_bytes < lidargs_ag_scratch_bytes(N0, n_offsets)) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "anchor_growing: scratch too small");
    lg::AgLevel p;
    p.N = N; p.N0 = N0; p.k = n_offsets; p.F = feat_dim;
    p.thr = grad_threshold; p.rthr = rand_threshold; p.size = (float)cur_size;
    p.inv = (float)(1.0 / cur_size);       // the reciprocal of the Python double, THEN rounded: what torch-ROCm's tensor / scalar kernel multiplies by (measured: tools/div_convention.py)
    p.exact = (flags & LIDARGS_AG_EXACT_DIVISION) ? 1 : 0;
    lg::Carver cv(scratch);
    uint32_t* hdr = cv.take<uint32_t>(AG_HDR);
    uint32_t* list = cv.take<uint32_t>((size_t)N0 * n_offsets);
    const size_t slots = (size_t)N0 * n_offsets;

    const int vec = (((uintptr_t)grads | (uintptr_t)offset_mask | (uintptr_t)(rnd ? rnd : grads)) & 15u) == 0 ? 1 : 0;
    hipLaunchKernelGGL(lg::k_ag_init, dim3(1), dim3(64), 0, stream, hdr);
    hipLaunchKernelGGL(lg::k_ag_mark, dim3((unsigned)((slots + 256 * AG_MARK_ITEMS - 1) / (256 * AG_MARK_ITEMS))), dim3(256), 0, stream, p, anchor, offset, scaling, grads, offset_mask, rnd, hdr, list, vec);
    hipLaunchKernelGGL(lg::k_ag_box, dim3((unsigned)std::min<size_t>(1024, (slots + 255) / 256)), dim3(256), 0, stream, p, anchor, offset, scaling, hdr, list);
    AG_HIP(hipGetLastError());
    uint32_t h[AG_HDR];
    AG_HIP(lg::api_read_words_zero_behind(hdr, 8, h, nullptr, 0, stream));
    const uint32_t C = h[0];
    if (counts_host) counts_host[0] = (int)C;
    if (C == 0) return 0;
    if (C > (1u << 30)) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "anchor_growing: more than 2^30 candidates (the voxel set's slot index is 32 bits)");

    lg::AgKey kd;
    int bits[3];
    for (int c = 0; c < 3; c++) {
        kd.mn[c] = (int)h[1 + c]; kd.mx[c] = (int)h[4 + c];
        const unsigned long long ext = (unsigned long long)((long long)kd.mx[c] - (long long)kd.mn[c]) + 1ull;    // <= 2^32
        int b = 0;
        while (b < 33 && (1ull << b) < ext) b++;
        bits[c] = b;
    }
    const int total_bits = bits[0] + bits[1] + bits[2];
    if (total_bits > 63) return lg::api_fail(LIDARGS_ERR_INVALID_ARGUMENT, "anchor_growing: the candidates' voxel box needs more than 63 key bits");
    kd.sy = bits[2]; kd.sx = bits[2] + bits[1];
    size_t T = 256;
    while (T < 2 * (size_t)C) T <<= 1;
    kd.mask = (uint32_t)(T - 1);

    const size_t sort_words = lg::sort_scratch_words(C);
    const size_t work_bytes = 128 * 10 + T * 12 + (size_t)C * (4 + 8 + 16) + sort_words * 4;
    char* work = alloc_work(work_user, work_bytes);
    if (!work) return lg::api_fail(LIDARGS_ERR_ALLOC, "anchor_growing: the work allocator returned NULL");
    lg::Carver wc(work);
    unsigned long long* keys = wc.take<unsigned long long>(T);
    int* vals = wc.take<int>(T);
    uint32_t* cand_slot = wc.take<uint32_t>(C);
    unsigned long long* surv = wc.take<unsigned long long>(C);
    uint32_t* ka = wc.take<uint32_t>(C); uint32_t* kb = wc.take<uint32_t>(C);
    uint32_t* va = wc.take<uint32_t>(C); uint32_t* vb = wc.take<uint32_t>(C);
    uint32_t* sort_scratch = wc.take<uint32_t>(sort_words);

    AG_HIP(hipMemsetAsync(keys, 0xFF, T * sizeof(unsigned long long), stream));
    AG_HIP(hipMemsetAsync(vals, 0, T * sizeof(int), stream));
    hipLaunchKernelGGL(lg::k_ag_insert, dim3((C + 255) / 256), dim3(256), 0, stream, p, kd, C, anchor, offset, scaling, list, keys, cand_slot);
    hipLaunchKernelGGL(lg::k_ag_probe, dim3(((unsigned)N + 255) / 256), dim3(256), 0, stream, p, kd, anchor, keys, vals);
    hipLaunchKernelGGL(lg::k_ag_survivors, dim3((unsigned)((T + 256 * AG_SURV_ITEMS - 1) / (256 * AG_SURV_ITEMS))), dim3(256), 0, stream, (uint32_t)T, keys, vals, hdr, surv);
    AG_HIP(hipGetLastError());
    AG_HIP(lg::api_read_words_zero_behind(hdr + 8, 2, h, nullptr, 0, stream));
    const uint32_t V = h[0], U = h[1];
    if (counts_host) { counts_host[1] = (int)V; counts_host[2] = (int)U; }
    if (U == 0) return 0;

    // the live keys in ascending order: LSD on the low word, then on the high word (the permutation is the value)
    const unsigned ub = (U + 255) / 256;
    hipLaunchKernelGGL(lg::k_ag_split, dim3(ub), dim3(256), 0, stream, U, surv, (const uint32_t*)nullptr, 0, ka, va);
    int side = lg::launch_radix_sort_pairs(ka, kb, va, vb, U, total_bits < 32 ? total_bits : 32, sort_scratch, stream, 0, nullptr, 0, true);
    uint32_t* perm = side ? vb : va;
    if (total_bits > 32) {
        uint32_t* k2a = side ? kb : ka; uint32_t* k2b = side ? ka : kb; uint32_t* v2b = side ? va : vb;
        hipLaunchKernelGGL(lg::k_ag_split, dim3(ub), dim3(256), 0, stream, U, surv, perm, 1, k2a, (uint32_t*)nullptr);
        const int side2 = lg::launch_radix_sort_pairs(k2a, k2b, perm, v2b, U, total_bits - 32, sort_scratch, stream);
        perm = side2 ? v2b : perm;
    }
    AG_HIP(hipGetLastError());

    float* new_anchor = (float*)alloc_anchor(anchor_user, (size_t)U * 3 * sizeof(float));
    uint32_t* new_feat = (uint32_t*)alloc_feat(feat_user, (size_t)U * feat_dim * sizeof(float));
    if (!new_anchor || !new_feat) return lg::api_fail(LIDARGS_ERR_ALLOC, "anchor_growing: an output allocator returned NULL");
    const size_t nf = (size_t)U * feat_dim;
    AG_HIP(hipMemsetAsync(new_feat, 0, nf * sizeof(float), stream));
    hipLaunchKernelGGL(lg::k_ag_emit, dim3(ub), dim3(256), 0, stream, p, kd, U, surv, perm, keys, vals, new_anchor);
    const size_t ft = (size_t)C * feat_dim;
    hipLaunchKernelGGL(lg::k_ag_features, dim3((unsigned)((ft + 255) / 256)), dim3(256), 0, stream, n_offsets, feat_dim, C, list, cand_slot, vals, anchor_feat, new_feat);
    hipLaunchKernelGGL(lg::k_ag_unmap, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, stream, nf, new_feat);
    AG_HIP(hipGetLastError());
    return (int)U;
}

}  // extern "C"
