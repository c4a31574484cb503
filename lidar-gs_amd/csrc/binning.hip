// binning.hip -- range sort + per-tile radix bin (gfx950), replacing the reference's
// scan -> duplicateWithKeys -> cub::DeviceRadixSort::SortPairs(64-bit) -> identifyTileRanges
// chain (R3/cr/rasterizer_impl.cu:288-332).
//
// The reference sorts R_ref = sum(16x1 tiles touched) 64-bit (tile|range) keys -- its largest
// pure-bandwidth stage.  Here the same ordering (tile, range bits, id)
// is produced in two much smaller steps:
//   1. stable LSD radix sort of the P Gaussians by their 32-bit range key (ties keep id order);
//   2. instances are EMITTED IN RANGE ORDER, one per (Gaussian, 16xTH tile), and a stable LSD
//      radix sort on the tile id only (ceil(log2(tiles)) bits, 1-2 passes) bins them.
// A stable bin of a range-ordered stream leaves every tile's list ordered by (range, id): the
// cub sort's tie-break (R3/cr/rasterizer_impl.cu:317-322, stable on identical keys).
//
// Ranks inside a wave come from ballot-matching the digit bits (wave-wide match-any), so every pass is
// stable by construction; blocks sort locally in LDS and write whole digit runs.
#include "lidargs_common.h"
#include <stdlib.h>

namespace lg {

// ------------------------------------------------------------------------------------------------
// Exclusive scan (u32), three small kernels: block reduce -> serial-ish scan of partials -> block scan.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t n = __shfl_up(v, o);
        if (lane >= o) v += n;
    }
    return v;
}

__global__ void __launch_bounds__(256) k_scan_reduce(const uint32_t* __restrict__ in, size_t n, uint32_t* __restrict__ partial) {
    __shared__ uint32_t ws[4];
    const size_t base = (size_t)blockIdx.x * SCAN_BLOCK;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const size_t i = base + (size_t)k * 256 + threadIdx.x;
        if (i < n) s += in[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// single block: exclusive scan of `nb` partials in place, grand total -> *total_out (may be null)
__global__ void __launch_bounds__(1024) k_scan_partials(uint32_t* __restrict__ partial, size_t nb, uint32_t* __restrict__ total_out) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (size_t base = 0; base < nb; base += 1024) {
        const size_t i = base + threadIdx.x;
        const uint32_t v = i < nb ? partial[i] : 0;
        const uint32_t inc = wave_incl_scan(v, lane);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        uint32_t off = carry_s;
        for (int k = 0; k < w; k++) off += wsum[k];
        if (i < nb) partial[i] = off + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = off + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry_s;
}

// `in` may alias `out` (in-place scan of the radix histograms): each thread reads its own 4 words first.
__global__ void __launch_bounds__(256) k_scan_apply(const uint32_t* in, uint32_t* out, size_t n,
                                                     const uint32_t* __restrict__ partial) {
    __shared__ uint32_t wsum[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // blocked arrangement: thread t owns 4 consecutive elements
    const size_t i0 = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * 4;
    uint32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = (i0 + k < n) ? in[i0 + k] : 0;
    const uint32_t tsum = v[0] + v[1] + v[2] + v[3];
    const uint32_t inc = wave_incl_scan(tsum, lane);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t off = partial[blockIdx.x] + inc - tsum;
    for (int k = 0; k < w; k++) off += wsum[k];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (i0 + k < n) out[i0 + k] = off;
        off += v[k];
    }
}

void launch_exclusive_scan(const uint32_t* in, uint32_t* out, size_t n, uint32_t* total_out, uint32_t* scratch, hipStream_t s) {
    if (n == 0) {
        if (total_out) hipMemsetAsync(total_out, 0, sizeof(uint32_t), s);
        return;
    }
    const size_t nb = scan_blocks(n);
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(256), 0, s, in, n, scratch);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, s, scratch, nb, total_out);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(256), 0, s, in, out, n, scratch);
}

// ------------------------------------------------------------------------------------------------
// LSD radix sort pass on (u32 key, u32 value) pairs.  A block = 4 waves = 4096 consecutive keys, wave w
// owning the w-th 1024-key slice so that the order (wave, round, lane) IS the key order (stability).
//   k_radix_hist          per-block digit histogram -> hist[digit][block]
//   k_radix_digit_prefix  one workgroup per digit: exclusive prefix over the blocks in place, digit totals -> tot[digit]
//   k_radix_scatter       ranks its keys (ballot match-any per round, per-wave LDS counters), sorts the block into
//                         LDS, then streams it out: consecutive threads write consecutive addresses inside each digit
//                         run, so the pass writes whole lines instead of 4-byte crumbs (measured 5x write
//                         amplification with direct per-key scatter).
// `n_dev` (nullable): the pair count lives on the device (enqueue-only forward: the host never learns it); the launch then covers the
// capacity `n` and the blocks behind *n_dev see no keys (their histograms are zero, their scatter retires).
// The value a pass takes its digit from: the key itself, or (range sort with a bias) key - kmin with the culled key 0xFFFFFFFF mapped
// just above the largest valid one.
// Third form (round 5, the MSD pass of the bucketed range sort below): the digit is the key's LINEAR range bucket,
//     digit = min(BINS - 2, (uint)((range - rmin) * (BINS - 1) / (rmax - rmin))),     culled key -> BINS - 1,
// with rmin / rmax folded by every block from the slots the preprocess left on the device (no host knowledge).  Monotone in the key
// (fp32 subtraction, multiplication and truncation are), so the buckets partition the frame's range order.
struct KeyMap {
    uint32_t kmin, cull; bool on;                                      // by value in the kernel arguments (host-side: KeyBias)
    const uint32_t* lin_span;                                          // non-NULL: linear buckets; [LG_INST_SLOTS][2] = (~smallest, largest) visible key
    __device__ __forceinline__ uint32_t operator()(uint32_t k) const { return on ? (k == 0xFFFFFFFFu ? cull : k - kmin) : k; }
};
static inline KeyMap key_map(const KeyBias* b) {
    KeyMap m; m.on = b != nullptr && b->lin_span == nullptr; m.kmin = b ? b->kmin : 0u; m.cull = b ? b->cull : 0xFFFFFFFFu; m.lin_span = b ? b->lin_span : nullptr;
    return m;
}
struct LinMap { float rmin, scale; uint32_t kmin, kmax; };
// every wave folds the 64 slots itself (two loads per lane, twelve shuffles): no launch, no LDS, no host
__device__ __forceinline__ LinMap lin_map_load(const uint32_t* __restrict__ span, int lane, int bins) {
    uint32_t kinv = span[2 * (lane & (LG_INST_SLOTS - 1))], kmx = span[2 * (lane & (LG_INST_SLOTS - 1)) + 1];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { kinv = max(kinv, (uint32_t)__shfl_xor((int)kinv, o)); kmx = max(kmx, (uint32_t)__shfl_xor((int)kmx, o)); }
    LinMap m;
    m.kmin = ~kinv; m.kmax = kmx;
    if (m.kmax < m.kmin) { m.kmin = 0u; m.kmax = 0u; }                 // no visible Gaussian: every key is the culled one
    m.rmin = __uint_as_float(m.kmin);
    const float w = __uint_as_float(m.kmax) - m.rmin;
    m.scale = w > 0.f ? (float)(bins - 1) / w : 0.f;
    return m;
}
template <int BINS>
__device__ __forceinline__ uint32_t lin_digit(const LinMap& m, uint32_t k) {
    if (k == 0xFFFFFFFFu) return (uint32_t)(BINS - 1);
    const float x = (__uint_as_float(k) - m.rmin) * m.scale;
    return min((uint32_t)(BINS - 2), (uint32_t)fmaxf(x, 0.f));
}
template <int BINS>
__device__ __forceinline__ uint32_t key_digit(const KeyMap& km, const LinMap& lin, uint32_t k, int shift) {
    return km.lin_span ? lin_digit<BINS>(lin, k) : ((km(k) >> shift) & (uint32_t)(BINS - 1));
}

template <int BITS, int ITEMS, typename KT = uint32_t>
__global__ void __launch_bounds__(256) k_radix_hist(const KT* __restrict__ keys, size_t n, const uint32_t* __restrict__ n_dev, int shift,
                                                    uint32_t* __restrict__ hist, unsigned nblocks, const KeyMap km) {
    constexpr int BINS = 1 << BITS;
    if (n_dev) n = min(n, (size_t)*n_dev);
    __shared__ uint32_t cnt[BINS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t base = (size_t)blockIdx.x * (256 * ITEMS) + (size_t)w * (64 * ITEMS);
    uint32_t k[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const size_t i = base + (size_t)r * 64 + lane;
        k[r] = i < n ? (uint32_t)keys[i] : 0u;
    }
    for (int d = tid; d < BINS; d += 256) cnt[d] = 0;
    LinMap lin = LinMap();
    if (km.lin_span) lin = lin_map_load(km.lin_span, lane, BINS);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const size_t i = base + (size_t)r * 64 + lane;
        if (i < n) atomicAdd(&cnt[key_digit<BINS>(km, lin, k[r], shift)], 1u);
    }
    __syncthreads();
    for (int d = tid; d < BINS; d += 256) hist[(size_t)d * nblocks + blockIdx.x] = cnt[d];
}

// One workgroup (4 waves) per (digit, chunk of `chunk` blocks): exclusive prefix of the block histograms inside the chunk, in place,
// and the chunk's sum -> part[digit][chunk].  With a single chunk (chunk >= nblocks) the sum is the digit total itself.
// Each wave takes a quarter of the chunk (WG groups of 64 blocks, WG = 2, 4 or 8 picked at launch): every load is issued before
// the first scan and the groups' scans are independent chains; only the carries are sequential, and the quarters meet through LDS.
// (One wave per row walked 22 groups one after the other at 1381 blocks: 12 us of a 50-us sort pass on 64 waves.)
template <int WG>
__global__ void __launch_bounds__(256) k_radix_digit_prefix(uint32_t* __restrict__ hist, unsigned nblocks, int bins, unsigned chunk,
                                                            unsigned chunks, uint32_t* __restrict__ part) {
    __shared__ uint32_t wtot[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned d = blockIdx.x / chunks, c = blockIdx.x - d * chunks;
    uint32_t* row = hist + (size_t)d * nblocks;
    const unsigned lo = c * chunk, hi = min(nblocks, lo + chunk);
    const unsigned wlo = lo + (unsigned)w * (WG * 64);                 // this wave's quarter
    uint32_t v[WG], inc[WG];
#pragma unroll
    for (int g = 0; g < WG; g++) { const unsigned b = wlo + (unsigned)g * 64 + lane; v[g] = b < hi ? row[b] : 0u; }
#pragma unroll
    for (int g = 0; g < WG; g++) inc[g] = wave_incl_scan(v[g], lane);
    uint32_t carry = 0, excl[WG];
#pragma unroll
    for (int g = 0; g < WG; g++) {
        excl[g] = carry + inc[g] - v[g];
        carry += __shfl(inc[g], 63);
    }
    if (lane == 0) wtot[w] = carry;
    __syncthreads();
    uint32_t off = 0;
    for (int q = 0; q < w; q++) off += wtot[q];
#pragma unroll
    for (int g = 0; g < WG; g++) {
        const unsigned b = wlo + (unsigned)g * 64 + lane;
        if (b < hi) row[b] = off + excl[g];
    }
    if (threadIdx.x == 0) part[(size_t)d * chunks + c] = wtot[0] + wtot[1] + wtot[2] + wtot[3];
}
// Second level (only when there is more than one chunk): exclusive prefix of the chunk sums per digit, in place; digit total.
// A single wave used to walk all the blocks of a digit: 33 k blocks at 138 M keys = 0.3 ms per pass on 128 waves.
__global__ void __launch_bounds__(256) k_radix_chunk_prefix(uint32_t* __restrict__ part, int bins, unsigned chunks, uint32_t* __restrict__ tot) {
    const int lane = threadIdx.x & 63;
    const int d = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (d >= bins) return;
    uint32_t* row = part + (size_t)d * chunks;
    uint32_t carry = 0;
    for (unsigned c0 = 0; c0 < chunks; c0 += 64) {
        const unsigned c = c0 + lane;
        const uint32_t v = c < chunks ? row[c] : 0u;
        const uint32_t inc = wave_incl_scan(v, lane);
        if (c < chunks) row[c] = carry + inc - v;
        carry += __shfl(inc, 63);
    }
    if (lane == 0) tot[d] = carry;
}

// Phase clocks of k_radix_scatter for tools/micro/scatter_phases.cpp (compiled in with -DLG_PHASE_CLOCKS only; the product build has none):
// wave 0 of the first 4096 workgroups stores the 100-MHz wall clock at each phase boundary.
#ifdef LG_PHASE_CLOCKS
__device__ unsigned long long lg_phase_clk[8 * 4096];
#define LG_PHASE_CLK(slot) do { if (threadIdx.x == 0 && blockIdx.x < 4096) lg_phase_clk[blockIdx.x * 8 + (slot)] = wall_clock64(); } while (0)
void phase_clocks_read(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(lg_phase_clk), sizeof(unsigned long long) * 8 * 4096); }
#else
#define LG_PHASE_CLK(slot) do { } while (0)
#endif
// AUX (round 6, the bucketed range sort's pass): a second 32-bit value per key travels with the pair -- aux_in[i] in input order,
// aux_out at the pair's destination (the Gaussians' span records: the bucket launch then finds them beside the ids instead of gathering
// 2 M random words of an 8-MB table, 128 MB of line traffic for 8 MB of data).
template <int BITS, int ITEMS, typename KT = uint32_t, bool AUX = false>
__global__ void __launch_bounds__(256) k_radix_scatter(const KT* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                       KT* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                       size_t n, const uint32_t* __restrict__ n_dev, int shift, const uint32_t* __restrict__ hist,
                                                       const uint32_t* __restrict__ tot, unsigned nblocks,
                                                       const uint32_t* __restrict__ part, unsigned chunk, unsigned chunks, const RadixTail tail,
                                                       const KeyMap km, const uint32_t* __restrict__ aux_in = nullptr, uint32_t* __restrict__ aux_out = nullptr) {
    constexpr int BINS = 1 << BITS;
    constexpr int CHUNK = 256 * ITEMS;                    // keys per block
    constexpr int PER = BINS > 256 ? BINS / 256 : 1;      // bins per thread in the block-wide scans (thread t owns bins [t*PER, t*PER+PER))
    if (n_dev) n = min(n, (size_t)*n_dev);
    if ((size_t)blockIdx.x * CHUNK >= n) return;     // wave-uniform, before any barrier
    LG_PHASE_CLK(0);
    __shared__ uint32_t run[SORT_WAVES][BINS];   // per-wave running digit counts, then wave bases
    __shared__ uint32_t dbase[BINS];             // block-local start of each digit run
    __shared__ uint32_t gbase[BINS];             // global start of this block's run of each digit
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t s_key[CHUNK];
    __shared__ uint32_t s_val[CHUNK];
    __shared__ uint32_t s_aux[AUX ? CHUNK : 1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t blk_base = (size_t)blockIdx.x * CHUNK;
    const size_t base = blk_base + (size_t)w * (64 * ITEMS);
    uint32_t k[ITEMS], v[ITEMS], pos[ITEMS], ax[AUX ? ITEMS : 1];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const size_t i = base + (size_t)r * 64 + lane;
        const bool valid = i < n;
        k[r] = valid ? (uint32_t)keys_in[i] : 0u;
        v[r] = valid ? (vals_in ? vals_in[i] : (uint32_t)i) : 0u;    // vals_in == nullptr: the values are the positions (first pass of an id sort)
        if constexpr (AUX) ax[r] = valid ? aux_in[i] : 0u;
    }
    // global digit bases: exclusive scan of the digit totals (every block repeats this tiny scan)
    {
        uint32_t my_tot[PER], my_hist[PER], tsum = 0;
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int d = tid * PER + q;
            my_tot[q] = 0; my_hist[q] = 0;
            if (d < BINS) {
                my_tot[q] = tot[d]; my_hist[q] = hist[(size_t)d * nblocks + blockIdx.x];
                if (part) my_hist[q] += part[(size_t)d * chunks + blockIdx.x / chunk];      // two-level cross-block prefix
            }
            tsum += my_tot[q];
        }
        const uint32_t inc = wave_incl_scan(tsum, lane);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        uint32_t off = inc - tsum;
        for (int q = 0; q < w; q++) off += wsum[q];
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int d = tid * PER + q;
            if (d < BINS) gbase[d] = off + my_hist[q];
            off += my_tot[q];
        }
    }
    for (int d = lane; d < BINS; d += 64) run[w][d] = 0;
    LinMap lin = LinMap();
    if (km.lin_span) lin = lin_map_load(km.lin_span, lane, BINS);
    __syncthreads();
    LG_PHASE_CLK(1);                                  // keys requested, digit bases scanned

    // A. wave-local stable ranks
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        const bool valid = base + (size_t)r * 64 + lane < n;
        const uint32_t d = key_digit<BINS>(km, lin, k[r], shift);
        // peers = the lanes holding the same digit: for every digit bit, keep the lanes whose bit equals mine.  With `mine` = the bit
        // spread over a word (0 / ~0), that is peers & ~(ballot ^ mine) -- ONE v_bitop3_b32 per 32 lanes and bit on gfx950
        // (truth table 0x90 = a & ~(b ^ c)) instead of a select, an xor and an and.
        const unsigned long long v0 = __ballot(valid);
        uint32_t plo = (uint32_t)v0, phi = (uint32_t)(v0 >> 32);
#pragma unroll
        for (int b = 0; b < BITS; b++) {
            const uint32_t mine = (uint32_t)(((int)(d << (31 - b))) >> 31);
            const unsigned long long m = __ballot(mine != 0u);
            plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t)m, mine, 0x90);
            phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t)(m >> 32), mine, 0x90);
        }
        const unsigned long long peers = ((unsigned long long)phi << 32) | plo;
        const uint32_t rank = (uint32_t)__popcll(peers & lt);
        // one wave owns run[w][]: its LDS operations execute in program order (read, then the leaders' update)
        pos[r] = run[w][d] + rank;
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) run[w][d] += (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    LG_PHASE_CLK(2);                                  // ranked

    // B. block-local digit starts: digit-major, then wave-major inside a digit
    {
        uint32_t c0[PER], c1[PER], c2[PER], c3[PER], tsum = 0;
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int d = tid * PER + q;
            c0[q] = c1[q] = c2[q] = c3[q] = 0;
            if (d < BINS) { c0[q] = run[0][d]; c1[q] = run[1][d]; c2[q] = run[2][d]; c3[q] = run[3][d]; }
            tsum += c0[q] + c1[q] + c2[q] + c3[q];
        }
        const uint32_t inc = wave_incl_scan(tsum, lane);
        __syncthreads();                       // wsum reuse; every read of run[][] above is done
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        uint32_t start = inc - tsum;
        for (int q = 0; q < w; q++) start += wsum[q];
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int d = tid * PER + q;
            if (d < BINS) {
                dbase[d] = start;
                run[0][d] = start; run[1][d] = start + c0[q]; run[2][d] = start + c0[q] + c1[q]; run[3][d] = start + c0[q] + c1[q] + c2[q];
            }
            start += c0[q] + c1[q] + c2[q] + c3[q];
        }
    }
    __syncthreads();
    LG_PHASE_CLK(3);                                  // block-local digit starts

    // C. park the block in LDS in sorted order
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        if (base + (size_t)r * 64 + lane < n) {
            const uint32_t d = key_digit<BINS>(km, lin, k[r], shift);
            const uint32_t p = run[w][d] + pos[r];
            s_key[p] = k[r]; s_val[p] = v[r];
            if constexpr (AUX) s_aux[p] = ax[r];
        }
    }
    __syncthreads();
    LG_PHASE_CLK(4);                                  // parked

    // D. stream out: thread i writes element i of the sorted block to its digit run
    const uint32_t count = (uint32_t)(n - blk_base < (size_t)CHUNK ? n - blk_base : (size_t)CHUNK);
    if (tail.mode == 0) {
        for (uint32_t i = tid; i < count; i += 256) {
            const uint32_t kk = s_key[i];
            const uint32_t d = key_digit<BINS>(km, lin, kk, shift);
            const size_t g = (size_t)gbase[d] + (i - dbase[d]);
            keys_out[g] = (KT)kk; vals_out[g] = s_val[i];
            if constexpr (AUX) aux_out[g] = s_aux[i];
        }
    } else {
        // last pass with a tail: the sorted keys are not written; the record of every value is gathered (all of a thread's gathers
        // in flight at once) and written at the value's final position
        uint32_t v[ITEMS]; size_t g[ITEMS];
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const uint32_t i = tid + 256u * (uint32_t)r;
            const uint32_t ii = i < count ? i : 0u;
            const uint32_t kk = s_key[ii];
            const uint32_t d = key_digit<BINS>(km, lin, kk, shift);
            g[r] = (size_t)gbase[d] + (ii - dbase[d]);
            v[r] = s_val[ii];
        }
        if (tail.mode == 1) {
            uint32_t sp[ITEMS];
#pragma unroll
            for (int r = 0; r < ITEMS; r++) sp[r] = (tid + 256u * (uint32_t)r < count) ? static_cast<const uint32_t*>(tail.src)[v[r]] : 0u;
#pragma unroll
            for (int r = 0; r < ITEMS; r++)
                if (tid + 256u * (uint32_t)r < count) { vals_out[g[r]] = v[r]; static_cast<uint32_t*>(tail.dst)[g[r]] = sp[r]; }
        } else {
            uint4 sp[ITEMS];
#pragma unroll
            for (int r = 0; r < ITEMS; r++) sp[r] = (tid + 256u * (uint32_t)r < count) ? static_cast<const uint4*>(tail.src)[v[r]] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int r = 0; r < ITEMS; r++)
                if (tid + 256u * (uint32_t)r < count) { vals_out[g[r]] = v[r]; static_cast<uint2*>(tail.dst)[g[r]] = make_uint2(sp[r].y, sp[r].x); }
        }
    }
#ifdef LG_PHASE_CLOCKS
    __builtin_amdgcn_s_waitcnt(0);                    // the stores' completion belongs to the last phase
    LG_PHASE_CLK(5);
#endif
}

// Keys per sort block: 4096 (16 per lane), or 2048 for inputs up to 4 M pairs when the scratch has room for twice the blocks -- at
// 2 M keys 489 blocks of 4 waves leave half of the 1024 SIMDs without a wave, and each block's load -> rank -> park -> stream-out
// chain is the launch's length.
template <int BITS, int ITEMS, typename KT>
static void radix_pass_items(const KT* kin, const uint32_t* vin, KT* kout, uint32_t* vout, size_t n, const uint32_t* n_dev, int shift,
                             uint32_t* scratch, hipStream_t s, int scratch_bits, const RadixTail& tail, const KeyBias* bias) {
    const size_t cap_bins = (size_t)1 << scratch_bits;           // the scratch layout of sort_scratch_words(n, scratch_bits)
    constexpr size_t CHUNK = 256 * ITEMS;
    const unsigned nb = (unsigned)((n + CHUNK - 1) / CHUNK);
    constexpr int BINS = 1 << BITS;
    uint32_t* hist = scratch;
    uint32_t* tot = scratch + cap_bins * sort_blocks(n);
    uint32_t* part = tot + cap_bins;
    const unsigned chunks_all = (nb + SORT_PREFIX_CHUNK - 1) / SORT_PREFIX_CHUNK;
    const bool two_level = chunks_all > 2;
    const unsigned chunk = two_level ? SORT_PREFIX_CHUNK : nb, chunks = two_level ? chunks_all : 1u;
    hipLaunchKernelGGL((k_radix_hist<BITS, ITEMS, KT>), dim3(nb), dim3(256), 0, s, kin, n, n_dev, shift, hist, nb, key_map(bias));
    // single level: the chunk sums ARE the digit totals, written straight to `tot`
    const dim3 pgrid(BINS * chunks);
    uint32_t* const pout = two_level ? part : tot;
    if (chunk <= 8 * 64) hipLaunchKernelGGL(k_radix_digit_prefix<2>, pgrid, dim3(256), 0, s, hist, nb, BINS, chunk, chunks, pout);
    else if (chunk <= 16 * 64) hipLaunchKernelGGL(k_radix_digit_prefix<4>, pgrid, dim3(256), 0, s, hist, nb, BINS, chunk, chunks, pout);
    else hipLaunchKernelGGL(k_radix_digit_prefix<8>, pgrid, dim3(256), 0, s, hist, nb, BINS, chunk, chunks, pout);
    if (two_level) hipLaunchKernelGGL(k_radix_chunk_prefix, dim3((BINS + 3) / 4), dim3(256), 0, s, part, BINS, chunks, tot);
    hipLaunchKernelGGL((k_radix_scatter<BITS, ITEMS, KT>), dim3(nb), dim3(256), 0, s, kin, vin, kout, vout, n, n_dev, shift, hist, tot, nb,
                       two_level ? part : (const uint32_t*)nullptr, chunk, chunks, tail, key_map(bias));
}
// the same pass with a second value per key (k_radix_scatter<..., AUX = true>); no tail
template <int BITS, int ITEMS>
static void radix_pass_items_aux(const uint32_t* kin, const uint32_t* vin, uint32_t* kout, uint32_t* vout, const uint32_t* aux_in, uint32_t* aux_out, size_t n, int shift,
                                 uint32_t* scratch, hipStream_t s, int scratch_bits, const KeyBias* bias) {
    const size_t cap_bins = (size_t)1 << scratch_bits;
    constexpr size_t CHUNK = 256 * ITEMS;
    const unsigned nb = (unsigned)((n + CHUNK - 1) / CHUNK);
    constexpr int BINS = 1 << BITS;
    uint32_t* hist = scratch;
    uint32_t* tot = scratch + cap_bins * sort_blocks(n);
    uint32_t* part = tot + cap_bins;
    const unsigned chunks_all = (nb + SORT_PREFIX_CHUNK - 1) / SORT_PREFIX_CHUNK;
    const bool two_level = chunks_all > 2;
    const unsigned chunk = two_level ? SORT_PREFIX_CHUNK : nb, chunks = two_level ? chunks_all : 1u;
    hipLaunchKernelGGL((k_radix_hist<BITS, ITEMS, uint32_t>), dim3(nb), dim3(256), 0, s, kin, n, (const uint32_t*)nullptr, shift, hist, nb, key_map(bias));
    const dim3 pgrid(BINS * chunks);
    uint32_t* const pout = two_level ? part : tot;
    if (chunk <= 8 * 64) hipLaunchKernelGGL(k_radix_digit_prefix<2>, pgrid, dim3(256), 0, s, hist, nb, BINS, chunk, chunks, pout);
    else if (chunk <= 16 * 64) hipLaunchKernelGGL(k_radix_digit_prefix<4>, pgrid, dim3(256), 0, s, hist, nb, BINS, chunk, chunks, pout);
    else hipLaunchKernelGGL(k_radix_digit_prefix<8>, pgrid, dim3(256), 0, s, hist, nb, BINS, chunk, chunks, pout);
    if (two_level) hipLaunchKernelGGL(k_radix_chunk_prefix, dim3((BINS + 3) / 4), dim3(256), 0, s, part, BINS, chunks, tot);
    hipLaunchKernelGGL((k_radix_scatter<BITS, ITEMS, uint32_t, true>), dim3(nb), dim3(256), 0, s, kin, vin, kout, vout, n, (const uint32_t*)nullptr, shift, hist, tot, nb,
                       two_level ? part : (const uint32_t*)nullptr, chunk, chunks, RadixTail(), key_map(bias), aux_in, aux_out);
}
// ------------------------------------------------------------------------------------------------
// Small inputs (<= SMALL_SORT_MAX pairs: small scenes, tests): the WHOLE multi-pass sort in ONE launch of one 1024-thread workgroup.
// The general form spends three launches per pass on what is a few microseconds of work each at this size (a 10 k-Gaussian frame:
// 12 + 6 launches, 80 us of a 0.2-ms frame).  Same algorithm, with the whole array as one block: wave w owns the w-th contiguous
// slice (so (wave, round, lane) is the key order: stable), ballot-match ranks per 64-key round, per-(digit, wave) counts and their
// digit-major scan in LDS, scatter through global memory (the ping-pong buffers are L2-resident); a workgroup barrier separates the
// passes (workgroup-scope release: the waves of one workgroup share their CU's L1, which is coherent for them -- an agent-scope fence
// by 1024 threads cost ~20 us per pass).
// One workgroup has nothing to hide a memory round trip (~1.2 us from L2 here) behind but its own loads: every dependent
// load-then-use step of a loop costs that much (a plain 14-iteration copy loop: 17 us).  So the rounds run in groups of four, the next
// group's keys (and values) in flight while this one is ranked, with plain unconditional loads the compiler can count (the first forms
// -- all rounds in registers behind workgroup-scope atomic loads, then one round ahead -- waited for every load: 30 us per pass).
constexpr int SMALL_SORT_WAVES = 16;
constexpr size_t SMALL_SORT_MAX = 16384;                               // pairs (a size choice: 16 rounds per wave)
// ... and up to where it is used (LIDARGS_SMALL_SORT_MAX): one CU ranks ~0.5 pairs per ns (16 waves x ~100 instructions per 64 pairs and
// phase on four SIMDs: issue-bound, tools/micro/small_sort_bench.cpp), so at 14 k pairs a pass takes 25-30 us against ~15 us for the
// three launches of the general form; below ~4 k pairs the single launch wins (4000 pairs: 12 us per pass)
constexpr size_t SMALL_SORT_DEFAULT = 4096;
constexpr int SMALL_SORT_GROUP = 4;                                    // rounds ranked per set of loads in flight
__device__ __forceinline__ uint32_t load_l2(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <typename KT>
__device__ __forceinline__ uint32_t load_key_l2(const KT* p) { return (uint32_t)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// the lanes of the wave that hold the digit d (valid lanes only): eight independent ballots, each folded with one v_bitop3_b32 per half
// (a & ~(b ^ c), as in k_radix_scatter); bits above the pass's width are 0 in every lane and change nothing
__device__ __forceinline__ unsigned long long small_sort_peers(uint32_t d, bool valid) {
    const unsigned long long v0 = __ballot(valid);
    uint32_t plo = (uint32_t)v0, phi = (uint32_t)(v0 >> 32);
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const uint32_t mine = (uint32_t)(((int)(d << (31 - q))) >> 31);
        const unsigned long long m = __ballot(mine != 0u);
        plo = __builtin_amdgcn_bitop3_b32(plo, (uint32_t)m, mine, 0x90);
        phi = __builtin_amdgcn_bitop3_b32(phi, (uint32_t)(m >> 32), mine, 0x90);
    }
    return ((unsigned long long)phi << 32) | plo;
}
// (the body as a device function: the bucketed range sort below runs it on the sub-range of a bucket too big for its LDS path)
template <typename KT = uint32_t, int W = SMALL_SORT_WAVES>
__device__ __forceinline__ void wg_radix_sort(KT* key_a, KT* key_b, uint32_t* val_a, uint32_t* val_b, uint32_t n, int begin_bit, int end_bit, int max_bits,
                                              int vals_are_positions, const RadixTail tail, const KeyMap km,
                                              uint32_t (*cnt)[W + 1], uint32_t* wsum) {
    constexpr int MAXB = 8, BINS = 1 << MAXB;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t per = ((n + W - 1) / W + 63u) & ~63u;               // keys per wave, whole rounds
    const uint32_t lo = min(n, (uint32_t)w * per), hi = min(n, lo + per);
    const uint32_t rounds = (hi - lo + 63u) >> 6;                      // wave-uniform
    const unsigned long long lt = (1ull << lane) - 1ull;
    int cur = 0, shift = begin_bit;
#pragma clang loop unroll(disable)
    while (shift < end_bit) {
        const int left = end_bit - shift;
        const int passes_left = (left + max_bits - 1) / max_bits;
        const int bits = (left + passes_left - 1) / passes_left;
        const uint32_t mask = (1u << bits) - 1u;
        const KT* kin = cur ? key_b : key_a; const uint32_t* vin = cur ? val_b : val_a;
        KT* kout = cur ? key_a : key_b; uint32_t* vout = cur ? val_a : val_b;
        const bool positions = shift == 0 && vals_are_positions;
        for (int i = tid; i < BINS * (W + 1); i += 64 * W) (&cnt[0][0])[i] = 0u;
        __syncthreads();
        // rounds in groups of SMALL_SORT_GROUP, the next group's keys (and values) in flight while this one is ranked; the loads are
        // unconditional (index clamped into the array) so that the compiler counts them and waits for exactly the older group
        constexpr int G = SMALL_SORT_GROUP;
        const uint32_t last_i = n - 1u;                                // n > 0 here
        auto load_keys = [&](uint32_t base, uint32_t (&k)[G]) {
#pragma unroll
            for (int q = 0; q < G; q++) k[q] = (uint32_t)kin[min(base + (uint32_t)q * 64u + lane, last_i)];
        };
        auto load_vals = [&](uint32_t base, uint32_t (&v)[G]) {
#pragma unroll
            for (int q = 0; q < G; q++) { const uint32_t i = min(base + (uint32_t)q * 64u + lane, last_i); v[q] = positions ? i : vin[i]; }
        };
        // 1. per-(digit, wave) counts (one wave owns its column: plain LDS read-modify-writes in program order, through the peers' leader)
        {
            uint32_t ka[G], kb[G];
            load_keys(lo, ka);
#pragma clang loop unroll(disable)
            for (uint32_t g = 0; g < rounds; g += G) {
                load_keys(lo + (g + G) * 64u, kb);
#pragma unroll
                for (int q = 0; q < G; q++) {
                    const uint32_t i = lo + (g + (uint32_t)q) * 64u + lane;
                    const bool valid = i < hi;
                    const uint32_t d = valid ? (km(ka[q]) >> shift) & mask : 0u;
                    const unsigned long long peers = small_sort_peers(d, valid);
                    if (valid && (peers & lt) == 0ull) cnt[d][w] += (uint32_t)__popcll(peers);
                    __builtin_amdgcn_wave_barrier();
                }
#pragma unroll
                for (int q = 0; q < G; q++) ka[q] = kb[q];
            }
        }
        __syncthreads();
        // 2. exclusive scan, digit-major then wave-major: thread t owns BINS * W / 1024 = 4 consecutive (digit, wave) cells
        {
            constexpr int PER = BINS * W / (64 * W);                   // 4
            uint32_t c[PER], tsum = 0;
#pragma unroll
            for (int q = 0; q < PER; q++) { const int cell = tid * PER + q; c[q] = cnt[cell / W][cell % W]; tsum += c[q]; }
            const uint32_t inc = wave_incl_scan(tsum, lane);
            if (lane == 63) wsum[w] = inc;
            __syncthreads();
            uint32_t off = inc - tsum;
            for (int q = 0; q < w; q++) off += wsum[q];
#pragma unroll
            for (int q = 0; q < PER; q++) { const int cell = tid * PER + q; cnt[cell / W][cell % W] = off; off += c[q]; }
        }
        __syncthreads();
        // 3. scatter in key order
        {
            uint32_t ka[G], kb[G], va[G], vb[G];
            load_keys(lo, ka); load_vals(lo, va);
#pragma clang loop unroll(disable)
            for (uint32_t g = 0; g < rounds; g += G) {
                load_keys(lo + (g + G) * 64u, kb); load_vals(lo + (g + G) * 64u, vb);
#pragma unroll
                for (int q = 0; q < G; q++) {
                    const uint32_t i = lo + (g + (uint32_t)q) * 64u + lane;
                    const bool valid = i < hi;
                    const uint32_t k = ka[q], v = va[q];
                    const uint32_t d = valid ? (km(k) >> shift) & mask : 0u;
                    const unsigned long long peers = small_sort_peers(d, valid);
                    const uint32_t rank = (uint32_t)__popcll(peers & lt);
                    const uint32_t pos = valid ? cnt[d][w] + rank : 0u;
                    __builtin_amdgcn_wave_barrier();
                    if (valid && rank == 0) cnt[d][w] += (uint32_t)__popcll(peers);
                    __builtin_amdgcn_wave_barrier();
                    if (valid) { kout[pos] = (KT)k; vout[pos] = v; }
                }
#pragma unroll
                for (int q = 0; q < G; q++) { ka[q] = kb[q]; va[q] = vb[q]; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");         // the pass's stores are visible to the workgroup's waves before the next pass reads them
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");         // (plain loads behind it: a workgroup's waves share their CU's L1)
        shift += bits;
        cur ^= 1;
    }
    // an odd number of passes left the pairs on the b side: bring them home here (the host then always finds them on the a side; two
    // device-to-device copy launches cost a 10 k-Gaussian frame 20 us).  Not with a tail: its last pass wrote to the tail's own arrays.
    if (n == 0) return;
    constexpr int CG = 4;                                              // loads in flight per thread in the two loops below
    const uint32_t last_n = n - 1u;
    if (cur && tail.mode == 0) {
#pragma clang loop unroll(disable)
        for (uint32_t i0 = tid; i0 < n; i0 += CG * 64 * W) {
            uint32_t k[CG], v[CG];
#pragma unroll
            for (int q = 0; q < CG; q++) { const uint32_t i = min(i0 + (uint32_t)q * 64u * W, last_n); k[q] = (uint32_t)key_b[i]; v[q] = val_b[i]; }
#pragma unroll
            for (int q = 0; q < CG; q++) { const uint32_t i = i0 + (uint32_t)q * 64u * W; if (i < n) { key_a[i] = (KT)k[q]; val_a[i] = v[q]; } }
        }
    }
    // the tail (a gather by the sorted values: each value's record to its final place) as its own loop behind the last pass
    if (tail.mode != 0) {
        const uint32_t* vfin = cur ? val_b : val_a;
#pragma clang loop unroll(disable)
        for (uint32_t i0 = tid; i0 < n; i0 += CG * 64 * W) {
            uint32_t v[CG];
#pragma unroll
            for (int q = 0; q < CG; q++) v[q] = vfin[min(i0 + (uint32_t)q * 64u * W, last_n)];
            if (tail.mode == 1) {
                uint32_t x[CG];
#pragma unroll
                for (int q = 0; q < CG; q++) x[q] = static_cast<const uint32_t*>(tail.src)[v[q]];
#pragma unroll
                for (int q = 0; q < CG; q++) { const uint32_t i = i0 + (uint32_t)q * 64u * W; if (i < n) static_cast<uint32_t*>(tail.dst)[i] = x[q]; }
            } else {
                uint4 x[CG];
#pragma unroll
                for (int q = 0; q < CG; q++) x[q] = static_cast<const uint4*>(tail.src)[v[q]];
#pragma unroll
                for (int q = 0; q < CG; q++) { const uint32_t i = i0 + (uint32_t)q * 64u * W; if (i < n) static_cast<uint2*>(tail.dst)[i] = make_uint2(x[q].y, x[q].x); }
            }
        }
    }
}

template <typename KT = uint32_t>
__global__ void __launch_bounds__(64 * SMALL_SORT_WAVES) k_radix_sort_small(KT* key_a, KT* key_b, uint32_t* val_a, uint32_t* val_b, uint32_t n,
                                                                            const uint32_t* __restrict__ n_dev, int begin_bit, int end_bit, int max_bits,
                                                                            int vals_are_positions, const RadixTail tail, const KeyMap km) {
    __shared__ uint32_t cnt[256][SMALL_SORT_WAVES + 1];                // [digit][wave] counts, then bases (+1: no bank conflicts down a column)
    __shared__ uint32_t wsum[SMALL_SORT_WAVES];
    if (n_dev) n = min(n, *n_dev);
    wg_radix_sort<KT>(key_a, key_b, val_a, val_b, n, begin_bit, end_bit, max_bits, vals_are_positions, tail, km, cnt, wsum);
}

template <int BITS, typename KT>
static void radix_pass(const KT* kin, const uint32_t* vin, KT* kout, uint32_t* vout, size_t n, const uint32_t* n_dev, int shift,
                       uint32_t* scratch, hipStream_t s, int scratch_bits, const RadixTail& tail, const KeyBias* bias) {
    static const int env = [] { const char* e = getenv("LIDARGS_SORT_ITEMS"); return e ? atoi(e) : 0; }();   // 8 / 16 forces the block size
    const bool room = BITS + 1 <= scratch_bits;                  // twice the blocks x BINS <= the histogram area (and the chunk sums likewise)
    const bool half = room && (env ? env == 8 : n <= ((size_t)4 << 20));
    if constexpr (BITS <= 10) {
        if (half) { radix_pass_items<BITS, SORT_ITEMS / 2, KT>(kin, vin, kout, vout, n, n_dev, shift, scratch, s, scratch_bits, tail, bias); return; }
    }
    radix_pass_items<BITS, SORT_ITEMS, KT>(kin, vin, kout, vout, n, n_dev, shift, scratch, s, scratch_bits, tail, bias);
}

template <typename KT>
static int radix_sort_pairs_t(KT* key_a, KT* key_b, uint32_t* val_a, uint32_t* val_b, size_t n, int end_bit,
                            uint32_t* scratch, hipStream_t s, int max_bits, const uint32_t* n_dev, int scratch_bits, bool vals_are_positions,
                            RadixTail tail, int begin_bit, const KeyBias* bias) {
    if (n == 0 || end_bit <= begin_bit) return 0;
    if (max_bits < 1 || max_bits > SORT_MAX_RADIX_BITS) max_bits = SORT_RADIX_BITS;
    static const int small_off = [] { const char* e = getenv("LIDARGS_NO_SMALL_SORT"); return e ? atoi(e) : 0; }();
    static const size_t small_max = [] { const char* e = getenv("LIDARGS_SMALL_SORT_MAX"); const long v = e ? atol(e) : (long)SMALL_SORT_DEFAULT;
                                         return (size_t)(v < 0 ? 0 : (v > (long)SMALL_SORT_MAX ? (long)SMALL_SORT_MAX : v)); }();
    if (n <= small_max && !small_off) {                                // the whole sort in one launch (k_radix_sort_small)
        const int mb = max_bits > 8 ? 8 : max_bits;
        int passes = 0;
        for (int sh = begin_bit; sh < end_bit;) { const int left = end_bit - sh, pl = (left + mb - 1) / mb; sh += (left + pl - 1) / pl; passes++; }
        hipLaunchKernelGGL(k_radix_sort_small<KT>, dim3(1), dim3(64 * SMALL_SORT_WAVES), 0, s, key_a, key_b, val_a, val_b, (uint32_t)n, n_dev, begin_bit, end_bit,
                           mb, vals_are_positions ? 1 : 0, tail, key_map(bias));
        return tail.mode == 0 ? 0 : (passes & 1);                       // without a tail the kernel brings the pairs home itself
    }
    if (scratch_bits < max_bits) scratch_bits = max_bits;
    int cur = 0;
    int shift = begin_bit;
    while (shift < end_bit) {
        const int left = end_bit - shift;
        KT* kin = cur ? key_b : key_a; uint32_t* vin = cur ? val_b : val_a;
        if (shift == 0 && vals_are_positions) vin = nullptr;              // val_a is not read (and need not have been written)
        const RadixTail none;
        KT* kout = cur ? key_a : key_b; uint32_t* vout = cur ? val_a : val_b;
        // split the remaining bits evenly over the remaining passes (e.g. 12 bits -> 6+6, not 8+4; 31 bits at 11 -> 11+10+10)
        const int passes_left = (left + max_bits - 1) / max_bits;
        const int bits = (left + passes_left - 1) / passes_left;
        const bool last = shift + bits >= end_bit;                       // the tail (a gather by the sorted values) rides on the last pass
        switch (bits) {
            case 1: radix_pass<1, KT>(kin, vin, kout, vout, n, n_dev, shift, scratch, s, scratch_bits, last ? tail : none, bias); break;
            case 2: radix_pass<2, KT>(kin, vin, kout, vout, n, n_dev, shift, scratch, s, scratch_bits, last ? tail : none, bias); break;
            case 3: radix_pass<3, KT>(kin, vin, kout, vout, n, n_dev, shift, scratch, s, scratch_bits, last ? tail : none, bias); break;
            case 4: radix_pass<4, KT>(kin, vin, kout, vout, n, n_dev, shift, scratch, s, scratch_bits, last ? tail : none, bias); break;
            case 5: radix_pass<5, KT>(kin, vin, kout, vout, n, n_dev, shift, scratch, s, scratch_bits, last ? tail : none, bias); break;
            case 6: radix_pass<6, KT>(kin, vin, kout, vout, n, n_dev, shift, scratch, s, scratch_bits, last ? tail : none, bias); break;
            case 7: radix_pass<7, KT>(kin, vin, kout, vout, n, n_dev, shift, scratch, s, scratch_bits, last ? tail : none, bias); break;
            case 8: radix_pass<8, KT>(kin, vin, kout, vout, n, n_dev, shift, scratch, s, scratch_bits, last ? tail : none, bias); break;
            case 9: if constexpr (sizeof(KT) == 2) { break; } else radix_pass<9, KT>(kin, vin, kout, vout, n, n_dev, shift, scratch, s, scratch_bits, last ? tail : none, bias); break;
            case 10: if constexpr (sizeof(KT) == 2) { break; } else radix_pass<10, KT>(kin, vin, kout, vout, n, n_dev, shift, scratch, s, scratch_bits, last ? tail : none, bias); break;
            default: if constexpr (sizeof(KT) == 2) { break; } else radix_pass<11, KT>(kin, vin, kout, vout, n, n_dev, shift, scratch, s, scratch_bits, last ? tail : none, bias); break;
        }
        shift += bits;
        cur ^= 1;
    }
    return cur;
}

// Where launch_radix_sort_pairs[16](..., end_bit, max_bits = default) without a tail will leave the sorted pairs: 0 = the a side, 1 = the
// b side (an odd number of passes of the general form; the single-launch form brings them home itself).  A caller that wants them
// on a given side produces the input on the other one when this says 1 (api.hip: the emit of a one-pass tile sort writes to the b side).
int radix_sort_result_side(size_t n, int end_bit) {
    if (n == 0 || end_bit <= 0) return 0;
    static const int small_off = [] { const char* e = getenv("LIDARGS_NO_SMALL_SORT"); return e ? atoi(e) : 0; }();
    static const size_t small_max = [] { const char* e = getenv("LIDARGS_SMALL_SORT_MAX"); const long v = e ? atol(e) : (long)SMALL_SORT_DEFAULT;
                                         return (size_t)(v < 0 ? 0 : (v > (long)SMALL_SORT_MAX ? (long)SMALL_SORT_MAX : v)); }();
    if (n <= small_max && !small_off) return 0;
    return ((end_bit + SORT_RADIX_BITS - 1) / SORT_RADIX_BITS) & 1;
}
int launch_radix_sort_pairs(uint32_t* key_a, uint32_t* key_b, uint32_t* val_a, uint32_t* val_b, size_t n, int end_bit,
                            uint32_t* scratch, hipStream_t s, int max_bits, const uint32_t* n_dev, int scratch_bits, bool vals_are_positions,
                            RadixTail tail, int begin_bit, const KeyBias* bias) {
    return radix_sort_pairs_t<uint32_t>(key_a, key_b, val_a, val_b, n, end_bit, scratch, s, max_bits, n_dev, scratch_bits, vals_are_positions, tail, begin_bit, bias);
}
// 16-bit keys (the tile sort whenever the image has at most 65536 list tiles): 6 instead of 8 bytes per pair moved by a pass, 2 instead
// of 4 read by its histogram.  Digits of at most 8 bits.
int launch_radix_sort_pairs16(uint16_t* key_a, uint16_t* key_b, uint32_t* val_a, uint32_t* val_b, size_t n, int end_bit, uint32_t* scratch, hipStream_t s,
                              const uint32_t* n_dev) {
    return radix_sort_pairs_t<uint16_t>(key_a, key_b, val_a, val_b, n, end_bit, scratch, s, SORT_RADIX_BITS, n_dev, 0, false, RadixTail(), 0, nullptr);
}

// ------------------------------------------------------------------------------------------------
// Bucketed range sort of the P Gaussians (round 5).  The LSD sort above moves all P pairs through HBM once per 8-9 key bits: three
// passes, nine launches, 107 us of the 2 M-Gaussian frame for 48 MB of necessary traffic.  Here:
//   1. ONE stable scatter pass (the LSD pass's three launches, with the linear range bucket as the digit: KeyMap::lin_span) cuts the frame
//      into BUCKET_BINS consecutive range intervals of equal WIDTH -- for surfaces seen from the sensor the count per interval varies
//      by a small factor over the frame's span, where equal intervals of the key's BITS (an MSD digit) would vary a thousandfold
//      between 2 m and 80 m;
//   2. one launch sorts every bucket COMPLETELY, a 512-thread workgroup per bucket: the pairs in registers, stable 8-bit LSD passes
//      through LDS on key - (the bucket's smallest key) -- 2 passes for a bucket's 13-16 significant bits at 20-80 m, 3 at 2 m --, then
//      the ids (and the span records they select: the RadixTail) written at their final places.
// No host knowledge is needed (the frame's range span is folded on the device), so the whole sort is queued behind the preprocess.
// A bucket with more than BSORT_CAP pairs (a frame whose Gaussians crowd into a thousandth of its range span) is sorted by the same
// workgroup through global memory (wg_radix_sort on the bucket's sub-range): correct, slower.  The culled Gaussians (key 0xFFFFFFFF) are
// the last bucket: all keys equal, nothing to sort.
constexpr int BUCKET_BITS = 10, BUCKET_BITS_BIG = 11;     // 1024 intervals up to 4 M Gaussians, 2048 above (round 6): the LDS path holds 7168 pairs per bucket
constexpr int BSORT_WAVES = 8, BSORT_ITEMS = 14, BSORT_CAP = 64 * BSORT_WAVES * BSORT_ITEMS;   // 7168 pairs per bucket on the LDS path (66 KB of LDS: two workgroups per CU)
constexpr uint32_t BSORT_CULL_SLICE = 4096;                              // culled Gaussians written per workgroup of the launch's tail
constexpr size_t BUCKET_SORT_MAX = (size_t)4 << 20;                    // beyond: the plain LSD passes.  Round 6 measured the 2048-interval form (BUCKET_BITS_BIG, kept behind
                                                                       // LIDARGS_RANGE_SORT_BUCKET_BITS=11 and tested) at 8 M Gaussians: 409 us against the LSD passes' 336 -- with 2048 digits a 4096-key
                                                                       // block's digit runs are 2 keys long (scatter 185 us: every pair a 4-byte write of its own), the bucket launch 166 us

// (its own function, not inlined: the rare path's registers must not count against the LDS path's occupancy)
__device__ __attribute__((noinline)) void bucket_sort_slow(uint32_t* ka, uint32_t* kb, uint32_t* va, uint32_t* vb, uint32_t n, int bits, uint32_t kmn,
                                                           uint32_t (*cnt)[BSORT_WAVES + 1], uint32_t* wsum) {
    KeyMap km; km.kmin = kmn; km.cull = 0u; km.on = true; km.lin_span = nullptr;
    wg_radix_sort<uint32_t, BSORT_WAVES>(ka, kb, va, vb, n, 0, bits, 8, 0, RadixTail(), km, cnt, wsum);
}
template <int BUCKET_BINS>
__global__ void __launch_bounds__(64 * BSORT_WAVES) __attribute__((amdgpu_waves_per_eu(4, 4))) k_bucket_sort(uint32_t* keys, uint32_t* ids, uint32_t* key_tmp, uint32_t* id_tmp,
                                                                       const uint32_t* __restrict__ tot, uint32_t* ids_out, const RadixTail tail, const uint32_t P) {
    constexpr int W = BSORT_WAVES, ITEMS = BSORT_ITEMS, BINS = 256;
    if (blockIdx.x >= (unsigned)(BUCKET_BINS - 1)) {
        // the culled Gaussians (the last bucket, the frame's last positions): nothing to sort and no record to gather -- "no instances" at
        // every position, a slice per workgroup (they can be most of a frame: one workgroup walking them all would be the launch's length)
        const uint32_t nc = tot[BUCKET_BINS - 1], first = P - nc;
        const uint32_t lo = (blockIdx.x - (unsigned)(BUCKET_BINS - 1)) * BSORT_CULL_SLICE, hi = min(nc, lo + BSORT_CULL_SLICE);
        for (uint32_t i = lo + threadIdx.x; i < hi; i += 64 * W) {
            ids_out[first + i] = ids[first + i];
            if (tail.mode == 1) static_cast<uint32_t*>(tail.dst)[first + i] = 0xFFFFFFFFu;          // span_pack's "no instances"
            else if (tail.mode == 2) static_cast<uint2*>(tail.dst)[first + i] = make_uint2(0u, 0u);   // an empty column span
        }
        return;
    }
    __shared__ uint32_t s_key[BSORT_CAP];
    __shared__ uint32_t s_val[BSORT_CAP];
    __shared__ uint32_t cnt[BINS][W + 1];
    __shared__ uint32_t wsum[W];
    __shared__ uint32_t s_red[3][W];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned d = blockIdx.x;
    const uint32_t n = tot[d];
    if (n == 0) return;                                                // (uniform over the workgroup)
    // the bucket's first position: the sum of the totals in front of it (at most one per thread)
    uint32_t part = 0u;
    for (unsigned j = (unsigned)tid; j < d; j += 64 * W) part += tot[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    const bool fits = n <= (uint32_t)BSORT_CAP;
    // the pairs, in (wave, round, lane) = key order, and the bucket's smallest / largest key
    const uint32_t per = ((n + W - 1) / W + 63u) & ~63u;               // pairs per wave, whole rounds (LDS path)
    const uint32_t lo = min(n, (uint32_t)w * per), hi = min(n, lo + per);
    uint32_t kmn = 0xFFFFFFFFu, kmx = 0u;
    if (lane == 0) s_red[0][w] = part;
    __syncthreads();
    uint32_t start = 0;
#pragma unroll
    for (int q = 0; q < W; q++) start += s_red[0][q];
    uint32_t k[ITEMS], v[ITEMS];
    if (fits) {
        // every pair is requested before the first is looked at, from an index any lane may read (its own, or the bucket's last): with the
        // smallest / largest key folded inside the loading loop each round waited for its key before the next was asked for -- fourteen
        // round trips one behind the other (ISA of the first form: `G | G w1` x 14)
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const uint32_t i = lo + (uint32_t)r * 64u + lane;
            const uint32_t ic = i < n ? i : n - 1u;
            k[r] = keys[start + ic]; v[r] = ids[start + ic];
        }
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const bool valid = lo + (uint32_t)r * 64u + lane < hi;
            if (valid) { kmn = min(kmn, k[r]); kmx = max(kmx, k[r]); }
            else { k[r] = 0u; v[r] = 0u; }
        }
    } else {
        for (uint32_t i = tid; i < n; i += 64 * W) { const uint32_t kk = keys[start + i]; kmn = min(kmn, kk); kmx = max(kmx, kk); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { kmn = min(kmn, (uint32_t)__shfl_xor((int)kmn, o)); kmx = max(kmx, (uint32_t)__shfl_xor((int)kmx, o)); }
    if (lane == 0) { s_red[1][w] = kmn; s_red[2][w] = kmx; }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < W; q++) { kmn = min(kmn, s_red[1][q]); kmx = max(kmx, s_red[2][q]); }
    const uint32_t span = kmx - kmn;
    const int bits = span ? 32 - __builtin_clz(span) : 0;              // significant bits of key - kmn (0: all keys equal -- the culled bucket)
    const uint32_t* sorted_ids = ids + start;                          // where the bucket's ids stand in final order (global memory), unless `in_lds`
    bool in_lds = false;
    if (bits > 0 && !fits) {
        bucket_sort_slow(keys + start, key_tmp + start, ids + start, id_tmp + start, n, bits, kmn, cnt, wsum);   // brings the pairs home
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __syncthreads(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else if (bits > 0) {
        const unsigned long long lt = (1ull << lane) - 1ull;
        const int passes = (bits + 7) / 8;
        int shift = 0;
        for (int pass = 0; pass < passes; pass++) {
            const int left = bits - shift, pl = passes - pass;
            const int pb = (left + pl - 1) / pl;                       // the remaining bits split evenly over the remaining passes
            const uint32_t mask = (1u << pb) - 1u;
            for (int i = tid; i < BINS * (W + 1); i += 64 * W) (&cnt[0][0])[i] = 0u;
            __syncthreads();
            uint32_t pos[ITEMS];
#pragma unroll
            for (int r = 0; r < ITEMS; r++) {
                if ((uint32_t)r * 64u >= per) break;                   // (wave-uniform)
                const bool valid = lo + (uint32_t)r * 64u + lane < hi;
                const uint32_t dg = valid ? ((k[r] - kmn) >> shift) & mask : 0u;
                const unsigned long long peers = small_sort_peers(dg, valid);
                const uint32_t rank = (uint32_t)__popcll(peers & lt);
                pos[r] = valid ? cnt[dg][w] + rank : 0u;               // one wave owns its column: LDS operations in program order
                __builtin_amdgcn_wave_barrier();
                if (valid && rank == 0) cnt[dg][w] += (uint32_t)__popcll(peers);
                __builtin_amdgcn_wave_barrier();
            }
            __syncthreads();
            {   // exclusive scan, digit-major then wave-major: thread t owns 4 consecutive (digit, wave) cells
                constexpr int PER = BINS * W / (64 * W);
                uint32_t c[PER], tsum = 0;
#pragma unroll
                for (int q = 0; q < PER; q++) { const int cell = tid * PER + q; c[q] = cnt[cell / W][cell % W]; tsum += c[q]; }
                const uint32_t inc = wave_incl_scan(tsum, lane);
                if (lane == 63) wsum[w] = inc;
                __syncthreads();
                uint32_t off = inc - tsum;
                for (int q = 0; q < w; q++) off += wsum[q];
#pragma unroll
                for (int q = 0; q < PER; q++) { const int cell = tid * PER + q; cnt[cell / W][cell % W] = off; off += c[q]; }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ITEMS; r++) {
                if ((uint32_t)r * 64u >= per) break;
                if (lo + (uint32_t)r * 64u + lane < hi) {
                    const uint32_t dg = ((k[r] - kmn) >> shift) & mask;
                    const uint32_t p = cnt[dg][w] + pos[r];
                    s_key[p] = k[r]; s_val[p] = v[r];
                }
            }
            __syncthreads();
            shift += pb;
            if (pass + 1 < passes) {
#pragma unroll
                for (int r = 0; r < ITEMS; r++) {
                    if ((uint32_t)r * 64u >= per) break;
                    const uint32_t i = lo + (uint32_t)r * 64u + lane;
                    if (i < hi) { k[r] = s_key[i]; v[r] = s_val[i]; }
                }
                __syncthreads();                                       // (cnt and the LDS arrays are rewritten by the next pass)
            }
        }
        in_lds = true;
    }
    // the ids at their final places, and the record each selects (the RadixTail: the spans in range order)
    for (uint32_t i0 = tid; i0 < n; i0 += 4 * 64 * W) {
        uint32_t g[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { const uint32_t i = min(i0 + (uint32_t)q * 64u * W, n - 1u); g[q] = in_lds ? s_val[i] : sorted_ids[i]; }
        if (tail.mode == 1) {
            uint32_t x[4];
#pragma unroll
            for (int q = 0; q < 4; q++) x[q] = static_cast<const uint32_t*>(tail.src)[g[q]];
#pragma unroll
            for (int q = 0; q < 4; q++) { const uint32_t i = i0 + (uint32_t)q * 64u * W; if (i < n) { ids_out[start + i] = g[q]; static_cast<uint32_t*>(tail.dst)[start + i] = x[q]; } }
        } else if (tail.mode == 2) {
            uint4 x[4];
#pragma unroll
            for (int q = 0; q < 4; q++) x[q] = static_cast<const uint4*>(tail.src)[g[q]];
#pragma unroll
            for (int q = 0; q < 4; q++) { const uint32_t i = i0 + (uint32_t)q * 64u * W; if (i < n) { ids_out[start + i] = g[q]; static_cast<uint2*>(tail.dst)[start + i] = make_uint2(x[q].y, x[q].x); } }
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) { const uint32_t i = i0 + (uint32_t)q * 64u * W; if (i < n) ids_out[start + i] = g[q]; }
        }
    }
}

// Round 6: the same launch when the bucket pass carried the span records along (k_radix_scatter<AUX>): `aux` holds them beside the ids.
// A bucket's pairs are sorted as ONE 32-bit word each, (key - bucket minimum) << 13 | position inside the bucket: half the LDS traffic and
// registers of the (key, id) pairs, 37 KB of LDS instead of 66 (four workgroups per CU), stable by construction; the ids and span records
// are then read through the sorted positions from the bucket's own 2 x 28-KB window (lines that are fetched once and used whole)
// instead of gathering 2 M random words of the 8-MB span table by id (128 MB of line traffic, the old launch's length).
// Buckets whose keys span more than 19 bits, or that do not fit (n > 7168), take the old path: sorted through global memory, spans by id.
constexpr int BPK_IDX_BITS = 13;
static_assert((1 << BPK_IDX_BITS) >= BSORT_CAP, "positions inside a bucket must fit the word's low bits");
template <int BUCKET_BINS>
__global__ void __launch_bounds__(64 * BSORT_WAVES) __attribute__((amdgpu_waves_per_eu(4, 4))) k_bucket_sort_packed(uint32_t* keys, uint32_t* ids, const uint32_t* __restrict__ aux, uint32_t* key_tmp, uint32_t* id_tmp,
                                                                         const uint32_t* __restrict__ tot, uint32_t* ids_out, uint32_t* span_out, const uint32_t* __restrict__ span_table,
                                                                         const uint32_t P) {
    constexpr int W = BSORT_WAVES, ITEMS = BSORT_ITEMS, BINS = 256;
    if (blockIdx.x >= (unsigned)(BUCKET_BINS - 1)) {                   // the culled Gaussians: as k_bucket_sort
        const uint32_t nc = tot[BUCKET_BINS - 1], first = P - nc;
        const uint32_t lo = (blockIdx.x - (unsigned)(BUCKET_BINS - 1)) * BSORT_CULL_SLICE, hi = min(nc, lo + BSORT_CULL_SLICE);
        for (uint32_t i = lo + threadIdx.x; i < hi; i += 64 * W) { ids_out[first + i] = ids[first + i]; span_out[first + i] = 0xFFFFFFFFu; }
        return;
    }
    __shared__ uint32_t s_w[BSORT_CAP];
    __shared__ uint32_t cnt[BINS][W + 1];
    __shared__ uint32_t wsum[W];
    __shared__ uint32_t s_red[3][W];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned d = blockIdx.x;
    const uint32_t n = tot[d];
    if (n == 0) return;
    uint32_t part = 0u;
    for (unsigned j = (unsigned)tid; j < d; j += 64 * W) part += tot[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    const bool fits = n <= (uint32_t)BSORT_CAP;
    const uint32_t per = ((n + W - 1) / W + 63u) & ~63u;
    const uint32_t lo = min(n, (uint32_t)w * per), hi = min(n, lo + per);
    uint32_t kmn = 0xFFFFFFFFu, kmx = 0u;
    if (lane == 0) s_red[0][w] = part;
    __syncthreads();
    uint32_t start = 0;
#pragma unroll
    for (int q = 0; q < W; q++) start += s_red[0][q];
    uint32_t k[ITEMS];
    if (fits) {
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const uint32_t i = lo + (uint32_t)r * 64u + lane;
            k[r] = keys[start + (i < n ? i : n - 1u)];
        }
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            if (lo + (uint32_t)r * 64u + lane < hi) { kmn = min(kmn, k[r]); kmx = max(kmx, k[r]); }
            else k[r] = 0u;
        }
    } else {
        for (uint32_t i = tid; i < n; i += 64 * W) { const uint32_t kk = keys[start + i]; kmn = min(kmn, kk); kmx = max(kmx, kk); }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { kmn = min(kmn, (uint32_t)__shfl_xor((int)kmn, o)); kmx = max(kmx, (uint32_t)__shfl_xor((int)kmx, o)); }
    if (lane == 0) { s_red[1][w] = kmn; s_red[2][w] = kmx; }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < W; q++) { kmn = min(kmn, s_red[1][q]); kmx = max(kmx, s_red[2][q]); }
    const uint32_t span = kmx - kmn;
    const int bits = span ? 32 - __builtin_clz(span) : 0;
    if (bits > 0 && (!fits || bits + BPK_IDX_BITS > 32)) {
        // the old path: the pairs sorted through global memory by this workgroup, the span records gathered by id
        bucket_sort_slow(keys + start, key_tmp + start, ids + start, id_tmp + start, n, bits, kmn, cnt, wsum);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __syncthreads(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (uint32_t i = tid; i < n; i += 64 * W) { const uint32_t g = ids[start + i]; ids_out[start + i] = g; span_out[start + i] = span_table[g]; }
        return;
    }
    if (bits > 0) {
        const unsigned long long lt = (1ull << lane) - 1ull;
        // the words: (key - kmn) << 13 | position; every pass below is a stable counting pass on 8 (or fewer) bits of the key part
#pragma unroll
        for (int r = 0; r < ITEMS; r++) {
            const uint32_t i = lo + (uint32_t)r * 64u + lane;
            k[r] = i < hi ? (((k[r] - kmn) << BPK_IDX_BITS) | i) : 0u;
        }
        const int passes = (bits + 7) / 8;
        int shift = BPK_IDX_BITS;
        for (int pass = 0; pass < passes; pass++) {
            const int left = bits + BPK_IDX_BITS - shift, pl = passes - pass;
            const int pb = (left + pl - 1) / pl;
            const uint32_t mask = (1u << pb) - 1u;
            for (int i = tid; i < BINS * (W + 1); i += 64 * W) (&cnt[0][0])[i] = 0u;
            __syncthreads();
            uint32_t pos[ITEMS];
#pragma unroll
            for (int r = 0; r < ITEMS; r++) {
                if ((uint32_t)r * 64u >= per) break;
                const bool valid = lo + (uint32_t)r * 64u + lane < hi;
                const uint32_t dg = valid ? (k[r] >> shift) & mask : 0u;
                const unsigned long long peers = small_sort_peers(dg, valid);
                const uint32_t rank = (uint32_t)__popcll(peers & lt);
                pos[r] = valid ? cnt[dg][w] + rank : 0u;
                __builtin_amdgcn_wave_barrier();
                if (valid && rank == 0) cnt[dg][w] += (uint32_t)__popcll(peers);
                __builtin_amdgcn_wave_barrier();
            }
            __syncthreads();
            {
                constexpr int PER = BINS * W / (64 * W);
                uint32_t c[PER], tsum = 0;
#pragma unroll
                for (int q = 0; q < PER; q++) { const int cell = tid * PER + q; c[q] = cnt[cell / W][cell % W]; tsum += c[q]; }
                const uint32_t inc = wave_incl_scan(tsum, lane);
                if (lane == 63) wsum[w] = inc;
                __syncthreads();
                uint32_t off = inc - tsum;
                for (int q = 0; q < w; q++) off += wsum[q];
#pragma unroll
                for (int q = 0; q < PER; q++) { const int cell = tid * PER + q; cnt[cell / W][cell % W] = off; off += c[q]; }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ITEMS; r++) {
                if ((uint32_t)r * 64u >= per) break;
                if (lo + (uint32_t)r * 64u + lane < hi) s_w[cnt[(k[r] >> shift) & mask][w] + pos[r]] = k[r];
            }
            __syncthreads();
            shift += pb;
            if (pass + 1 < passes) {
#pragma unroll
                for (int r = 0; r < ITEMS; r++) {
                    if ((uint32_t)r * 64u >= per) break;
                    const uint32_t i = lo + (uint32_t)r * 64u + lane;
                    if (i < hi) k[r] = s_w[i];
                }
                __syncthreads();
            }
        }
    }
    // the ids and span records through the sorted positions (bits == 0: every key equal, the input order stands)
    for (uint32_t i0 = tid; i0 < n; i0 += 4 * 64 * W) {
        uint32_t li[4], g[4], x[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { const uint32_t i = min(i0 + (uint32_t)q * 64u * W, n - 1u); li[q] = bits > 0 ? (s_w[i] & ((1u << BPK_IDX_BITS) - 1u)) : i; }
#pragma unroll
        for (int q = 0; q < 4; q++) { g[q] = ids[start + li[q]]; x[q] = aux[start + li[q]]; }
#pragma unroll
        for (int q = 0; q < 4; q++) { const uint32_t i = i0 + (uint32_t)q * 64u * W; if (i < n) { ids_out[start + i] = g[q]; span_out[start + i] = x[q]; } }
    }
}

// LIDARGS_RANGE_SORT_BUCKETS=0: the LSD passes again (A/B, tests).  LIDARGS_RANGE_SORT_BUCKET_BITS=11: the 2048-interval form at any size
// (tests: the CPU checker cannot run frames above 4 M Gaussians; read per call, the suite switches it inside one process).
static int forced_bucket_bits() { const char* e = getenv("LIDARGS_RANGE_SORT_BUCKET_BITS"); const int v = e ? atoi(e) : 0; return (v == BUCKET_BITS || v == BUCKET_BITS_BIG) ? v : 0; }
bool range_sort_buckets_ok(size_t P) {
    static const bool on = [] { const char* e = getenv("LIDARGS_RANGE_SORT_BUCKETS"); return !e || atoi(e) != 0; }();
    static const size_t small_max = [] { const char* e = getenv("LIDARGS_SMALL_SORT_MAX"); const long v = e ? atol(e) : (long)SMALL_SORT_DEFAULT;
                                         return (size_t)(v < 0 ? 0 : (v > (long)SMALL_SORT_MAX ? (long)SMALL_SORT_MAX : v)); }();
    static const int small_off = [] { const char* e = getenv("LIDARGS_NO_SMALL_SORT"); return e ? atoi(e) : 0; }();
    return on && (P <= BUCKET_SORT_MAX || forced_bucket_bits() == BUCKET_BITS_BIG) && (small_off || P > small_max);
}
// (key_a, positions) -> ids in range order in id_a, tail.dst = the records of tail.src in range order.  key_b / id_b hold the bucketed
// pairs; `scratch` as for launch_radix_sort_pairs with scratch_bits = SORT_MAX_RADIX_BITS.  key_span: GeomView totals + LG_TOTALS_KEYSPAN_WORD.
template <int BB>
static void range_sort_buckets_t(uint32_t* key_a, uint32_t* key_b, uint32_t* id_a, uint32_t* id_b, size_t P, uint32_t* scratch, const KeyBias* kb, RadixTail tail, hipStream_t s) {
    const uint32_t* tot = scratch + ((size_t)1 << SORT_MAX_RADIX_BITS) * sort_blocks(P);       // where the pass leaves the digit totals (radix_pass_items)
    const unsigned cull_blocks = (unsigned)((P + BSORT_CULL_SLICE - 1) / BSORT_CULL_SLICE);     // (as many as a frame of culled Gaussians only would need: the others leave at once)
    static const bool packed_on = [] { const char* e = getenv("LIDARGS_RANGE_SORT_PACKED"); return !e || atoi(e) != 0; }();   // 0: round 5's form (A/B)
    if (tail.mode == 1 && packed_on) {
        // 4-byte span records: they ride along with the pairs (the second half of the span_sorted allocation -- u32x2[P], of which the
        // compact form uses the first P words -- holds them between the two launches)
        uint32_t* aux_b = static_cast<uint32_t*>(tail.dst) + P;
        radix_pass_items_aux<BB, SORT_ITEMS>(key_a, nullptr, key_b, id_b, static_cast<const uint32_t*>(tail.src), aux_b, P, 0, scratch, s, SORT_MAX_RADIX_BITS, kb);
        hipLaunchKernelGGL(k_bucket_sort_packed<(1 << BB)>, dim3((1 << BB) - 1 + cull_blocks), dim3(64 * BSORT_WAVES), 0, s, key_b, id_b, aux_b, key_a, id_a, tot, id_a,
                           static_cast<uint32_t*>(tail.dst), static_cast<const uint32_t*>(tail.src), (uint32_t)P);
        return;
    }
    // (4096-key blocks: with 1024 digits a block's digit runs are 4 keys long, 2 in a half-size block -- every pair a write of its own:
    //  scatter 36.3 -> 28.7 us, histogram 12.7 -> 10.2 us at 2 M keys)
    radix_pass_items<BB, SORT_ITEMS, uint32_t>(key_a, nullptr, key_b, id_b, P, nullptr, 0, scratch, s, SORT_MAX_RADIX_BITS, RadixTail(), kb);
    hipLaunchKernelGGL(k_bucket_sort<(1 << BB)>, dim3((1 << BB) - 1 + cull_blocks), dim3(64 * BSORT_WAVES), 0, s, key_b, id_b, key_a, id_a, tot, id_a, tail, (uint32_t)P);
}
void launch_range_sort_buckets(uint32_t* key_a, uint32_t* key_b, uint32_t* id_a, uint32_t* id_b, size_t P, uint32_t* scratch, const uint32_t* key_span,
                               RadixTail tail, hipStream_t s) {
    if (P == 0) return;
    KeyBias kb; kb.kmin = 0u; kb.cull = 0xFFFFFFFFu; kb.lin_span = key_span;
    const int forced = forced_bucket_bits();
    if (forced ? forced == BUCKET_BITS_BIG : P > BUCKET_SORT_MAX) range_sort_buckets_t<BUCKET_BITS_BIG>(key_a, key_b, id_a, id_b, P, scratch, &kb, tail, s);
    else range_sort_buckets_t<BUCKET_BITS>(key_a, key_b, id_a, id_b, P, scratch, &kb, tail, s);
}

// ------------------------------------------------------------------------------------------------
// tiles of a pruned rect for tile height TH (a power of two): (column span) x (rows of tiles the row span touches); an empty
// column span = 0
__device__ __forceinline__ uint32_t span_tiles(uint32_t xs, uint32_t rs, int th_shift) {
    const uint32_t cols = (xs >> 16) - (xs & 0xFFFFu);
    return cols ? cols * ((((rs >> 16) - 1u) >> th_shift) - ((rs & 0xFFFFu) >> th_shift) + 1u) : 0u;
}

// Instance offsets without a per-Gaussian scan array.  A block of SCAN_BLOCK range-consecutive Gaussians:
//   (range sort)     its last pass leaves every Gaussian's span record in range order (RadixTail)
//   k_span_block_sums  the block's instance count -> block_sum[block]
//   k_scan_partials  exclusive prefix of the block sums, grand total (= R) -> what the host reads
//   k_emit_instances re-derives the counts from the range-ordered spans and scans them inside the block
// Block sums of the instance counts from the spans in range order (the last pass of the range sort gathered them there, RadixTail;
// COMPACT: 4-byte span records, lidargs_common.h span_pack).
template <bool COMPACT>
__global__ void __launch_bounds__(256) k_span_block_sums(const void* __restrict__ span_sorted_, int th_shift, uint32_t* __restrict__ block_sum, size_t P) {
    __shared__ uint32_t ws[4];
    const size_t base = (size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    uint32_t sum = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const size_t i = base + (size_t)r * 256;
        if (i < P) {
            const uint2 xr = COMPACT ? span_unpack(static_cast<const uint32_t*>(span_sorted_)[i]) : static_cast<const uint2*>(span_sorted_)[i];
            sum += span_tiles(xr.x, xr.y, th_shift);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) block_sum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
// spans in range order + exclusive block offsets in `block_off` (scan_blocks(P) words) + the instance total in *total_out
void launch_instance_offsets(const void* span_sorted, bool compact, int TH, uint32_t* block_off, uint32_t* total_out, size_t P, hipStream_t s, bool scan) {
    int sh = 0;
    while ((1 << sh) < TH) sh++;
    const size_t nb = scan_blocks(P);
    if (compact) hipLaunchKernelGGL(k_span_block_sums<true>, dim3((unsigned)nb), dim3(256), 0, s, span_sorted, sh, block_off, P);
    else hipLaunchKernelGGL(k_span_block_sums<false>, dim3((unsigned)nb), dim3(256), 0, s, span_sorted, sh, block_off, P);
    if (scan) hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, s, block_off, nb, total_out);   // (otherwise the emit adds the sums up itself)
}

// Load-balanced expansion: each wave owns 64 range-consecutive Gaussians at a time and writes their
// instances cooperatively, 64 consecutive output slots per step (coalesced 256-B stores),
// instead of one thread looping over its own rect (the reference's duplicateWithKeys,
// R3/cr/rasterizer_impl.cu:70-112, whose per-thread trip count varies 1..100s).
template <bool COMPACT, typename KT = uint32_t>
__global__ void __launch_bounds__(SCAN_BLOCK) k_emit_instances(const uint32_t* __restrict__ ids_sorted, const uint32_t* __restrict__ block_off,
                                                               const void* __restrict__ span_sorted_, size_t P, int th_shift, int tiles_x,
                                                               KT* __restrict__ inst_tile, uint32_t* __restrict__ inst_val, uint32_t cap,
                                                               uint2* __restrict__ ranges, uint32_t tiles, const int sums_unscanned) {
    // every tile's list range starts out empty, as the reference pre-zeroes `ranges` (R3/cr/rasterizer_impl.cu:324): k_tile_ranges, two
    // sorts later, writes the non-empty ones.  (It used to zero the tiles a step of the sorted keys skips, one thread per step: a far range
    // shell's frame, whose lists leave a thousand tiles empty in a row, spent 42 us there.)
    for (size_t t = (size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x; t < tiles; t += (size_t)gridDim.x * SCAN_BLOCK) ranges[t] = make_uint2(0u, 0u);
    __shared__ uint32_t s_tot[SCAN_BLOCK / 64];                        // instance count of each wave's 64 Gaussians
    __shared__ uint32_t s_own[SCAN_BLOCK / 64][64];                    // per wave: the lane whose instances start at each slot of the window
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    const bool valid = i < P;
    uint2 sp = make_uint2(0u, 0u);                                     // (xspan, rowspan); an empty column span = no instances
    if (valid) sp = COMPACT ? span_unpack(static_cast<const uint32_t*>(span_sorted_)[i]) : static_cast<const uint2*>(span_sorted_)[i];
    const uint32_t cnt = span_tiles(sp.x, sp.y, th_shift);
    const uint32_t inc = wave_incl_scan(cnt, lane);
    if (lane == 63) s_tot[w] = inc;
    uint32_t g = 0, nx = 1, x0 = 0, ty0 = 0;
    if (cnt) {
        g = ids_sorted[i];
        x0 = sp.x & 0xFFFFu; nx = (sp.x >> 16) - x0;
        ty0 = (sp.y & 0xFFFFu) >> th_shift;
    }
    // sums_unscanned: block_off holds the blocks' instance COUNTS and this block adds up the ones in front of it itself (<= 2 loads per
    // thread at 2 M Gaussians) -- the single-workgroup scan launch between the block sums and this one (k_scan_partials) is gone
    __shared__ uint32_t s_pre[SCAN_BLOCK / 64];
    uint32_t pre = 0;
    if (sums_unscanned) {
        for (uint32_t j = threadIdx.x; j < blockIdx.x; j += SCAN_BLOCK) pre += block_off[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pre += __shfl_xor(pre, o);
        if (lane == 0) s_pre[w] = pre;
    }
    __syncthreads();
    uint32_t wave_base = sums_unscanned ? 0u : block_off[blockIdx.x];
    if (sums_unscanned) for (int q = 0; q < SCAN_BLOCK / 64; q++) wave_base += s_pre[q];
    for (int q = 0; q < w; q++) wave_base += s_tot[q];
    const uint32_t wave_total = __shfl(inc, 63);
    const uint32_t lo = valid ? inc - cnt : 0xFFFFFFFFu;               // local exclusive prefix (+inf past the end)
    const uint32_t t_end = (wave_total + 63u) & ~63u;                  // every lane takes part in the shuffles
    uint32_t carry = 0;                                                // owner of the slot in front of the window
    for (uint32_t t = lane; t < t_end; t += 64) {
        // owner = largest lane L with cnt_L > 0 and lo_L <= t.  Every such lane marks the slot its instances start at (inside this
        // window of 64 slots) with its index; an inclusive max-scan over the window, seeded with the owner in front of it, is the
        // owner of every slot -- 6 row-shift / broadcast steps instead of a 6-step binary search through 6 lane permutes.
        // (One wave owns s_own[w][]: its LDS operations execute in program order.)
        const uint32_t t0 = t - (uint32_t)lane;
        s_own[w][lane] = 0u;
        __builtin_amdgcn_wave_barrier();
        if (cnt && lo >= t0 && lo < t0 + 64u) s_own[w][lo - t0] = (uint32_t)lane;
        __builtin_amdgcn_wave_barrier();
        uint32_t m = s_own[w][lane];
        m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x111, 0xF, 0xF, false));   // row_shr:1
        m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x112, 0xF, 0xF, false));   // row_shr:2
        m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x114, 0xF, 0xF, false));   // row_shr:4
        m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x118, 0xF, 0xF, false));   // row_shr:8
        m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x142, 0xA, 0xF, false));   // row_bcast:15 -> rows 1, 3
        m = max(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x143, 0xC, 0xF, false));   // row_bcast:31 -> rows 2, 3
        m = max(m, carry);
        carry = __shfl(m, 63);
        const int a = (int)m;
        const uint32_t o_lo = __shfl(lo, a), o_g = __shfl(g, a), o_nx = __shfl(nx, a), o_x0 = __shfl(x0, a), o_ty0 = __shfl(ty0, a);
        if (t < wave_total && wave_base + t < cap) {                  // cap: the binning buffer's capacity (enqueue-only forward); UINT_MAX otherwise
            const uint32_t jj = t - o_lo;
            // jj / o_nx without the ~25-instruction integer division: float estimate (exact operands below 2^22, beyond which the
            // integer division runs), one correction step either way makes it exact
            uint32_t ry;
            if (jj < (1u << 22)) {
                ry = (uint32_t)(((float)jj + 0.5f) * __builtin_amdgcn_rcpf((float)o_nx));
                ry -= (ry * o_nx > jj) ? 1u : 0u;
                ry += ((ry + 1u) * o_nx <= jj) ? 1u : 0u;
            } else {
                ry = jj / o_nx;
            }
            const uint32_t rx = jj - ry * o_nx;
            uint32_t tx = o_x0 + rx;                                   // a span may run across the seam of the panorama (surfel.hip sf_prune): x1 > tiles_x
            tx -= tx >= (uint32_t)tiles_x ? (uint32_t)tiles_x : 0u;
            inst_tile[(size_t)wave_base + t] = (KT)((o_ty0 + ry) * (uint32_t)tiles_x + tx);
            inst_val[(size_t)wave_base + t] = o_g;
        }
    }
}

void launch_emit_instances(const uint32_t* ids_sorted, const uint32_t* block_off, const void* span_sorted, bool compact, size_t P, TileGrid grid,
                           uint32_t* inst_tile, uint32_t* inst_val, hipStream_t s, uint32_t cap, bool key16, uint2* ranges, bool sums_unscanned) {
    const int su = sums_unscanned ? 1 : 0;
    const uint32_t tiles = ranges ? (uint32_t)grid.num_tiles() : 0u;
    int sh = 0;
    while ((1 << sh) < grid.TH) sh++;
    if (key16) {
        uint16_t* t16 = reinterpret_cast<uint16_t*>(inst_tile);
        if (compact) hipLaunchKernelGGL((k_emit_instances<true, uint16_t>), dim3((unsigned)scan_blocks(P)), dim3(SCAN_BLOCK), 0, s, ids_sorted, block_off, span_sorted,
                                        P, sh, grid.tiles_x, t16, inst_val, cap, ranges, tiles, su);
        else hipLaunchKernelGGL((k_emit_instances<false, uint16_t>), dim3((unsigned)scan_blocks(P)), dim3(SCAN_BLOCK), 0, s, ids_sorted, block_off, span_sorted,
                                P, sh, grid.tiles_x, t16, inst_val, cap, ranges, tiles, su);
        return;
    }
    if (compact) hipLaunchKernelGGL(k_emit_instances<true>, dim3((unsigned)scan_blocks(P)), dim3(SCAN_BLOCK), 0, s, ids_sorted, block_off, span_sorted,
                                    P, sh, grid.tiles_x, inst_tile, inst_val, cap, ranges, tiles, su);
    else hipLaunchKernelGGL(k_emit_instances<false>, dim3((unsigned)scan_blocks(P)), dim3(SCAN_BLOCK), 0, s, ids_sorted, block_off, span_sorted,
                            P, sh, grid.tiles_x, inst_tile, inst_val, cap, ranges, tiles, su);
}

// R3/cr/rasterizer_impl.cu:117-139 identifyTileRanges on 32-bit tile keys.  The reference pre-zeroes `ranges` (:324) so that tiles
// without instances read (0, 0); here the thread at a boundary writes the empty ranges of the tiles it skips over (and the
// first / last thread those before the first / behind the last key): every entry is written, no separate fill launch.
// Eight consecutive keys per thread (one or two 16-byte loads; one key per thread made cfg4's 35 M instances a 43-us launch).
constexpr int RANGES_ITEMS = 8;
template <typename KT = uint32_t>
__global__ void __launch_bounds__(256) k_tile_ranges(const KT* __restrict__ tile_sorted, size_t R, const uint32_t* __restrict__ R_dev,
                                                     uint2* __restrict__ ranges, uint32_t tiles, uint32_t* __restrict__ zero, int n_zero, bool prezeroed) {
    const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * RANGES_ITEMS;
    if (blockIdx.x == 0) for (int q = threadIdx.x; q < n_zero; q += 256) zero[q] = 0u;    // the work list's counters (lidargs_common.h WorkList)
    if (R_dev) R = min(R, (size_t)*R_dev);                             // enqueue-only forward: the count lives on the device
    if (R == 0) {                                                      // nothing binned: every tile is empty
        if (blockIdx.x == 0) for (uint32_t t = threadIdx.x; t < tiles; t += 256) ranges[t] = make_uint2(0u, 0u);
        return;
    }
    if (i0 >= R) return;
    // the arrays are carved with room behind their last element (the next array of the binning buffer at worst): the vector loads may
    // read past R, what they bring is not looked at
    uint32_t key[RANGES_ITEMS];
    if constexpr (sizeof(KT) == 2) {
        const uint4 v = *reinterpret_cast<const uint4*>(tile_sorted + i0);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; q++) { key[2 * q] = w[q] & 0xFFFFu; key[2 * q + 1] = w[q] >> 16; }
    } else {
        const uint4 v0 = *reinterpret_cast<const uint4*>(tile_sorted + i0), v1 = *reinterpret_cast<const uint4*>(tile_sorted + i0 + 4);
        key[0] = v0.x; key[1] = v0.y; key[2] = v0.z; key[3] = v0.w; key[4] = v1.x; key[5] = v1.y; key[6] = v1.z; key[7] = v1.w;
    }
    uint32_t prev = i0 ? (uint32_t)tile_sorted[i0 - 1] : 0u;
#pragma unroll
    for (int q = 0; q < RANGES_ITEMS; q++) {
        const size_t i = i0 + (size_t)q;
        if (i >= R) break;
        const uint32_t cur = key[q];
        // prezeroed: the emit launch cleared every range; otherwise the thread at a step clears the tiles it skips
        if (i == 0) {
            if (!prezeroed) for (uint32_t t = 0; t < cur && t < tiles; t++) ranges[t] = make_uint2(0u, 0u);
            ranges[cur].x = 0;
        } else if (cur != prev) {
            ranges[prev].y = (uint32_t)i; ranges[cur].x = (uint32_t)i;
            if (!prezeroed) for (uint32_t t = prev + 1; t < cur; t++) ranges[t] = make_uint2(0u, 0u);
        }
        if (i == R - 1) {
            ranges[cur].y = (uint32_t)R;
            if (!prezeroed) for (uint32_t t = cur + 1; t < tiles; t++) ranges[t] = make_uint2(0u, 0u);
        }
        prev = cur;
    }
}

void launch_tile_ranges(const uint32_t* tile_sorted, size_t R, uint2* ranges, int tiles, hipStream_t s, const uint32_t* R_dev, bool key16,
                        uint32_t* zero, int n_zero, bool prezeroed) {
    if (!zero) n_zero = 0;
    if (R && key16) hipLaunchKernelGGL(k_tile_ranges<uint16_t>, dim3((unsigned)((R + 256 * RANGES_ITEMS - 1) / (256 * RANGES_ITEMS))), dim3(256), 0, s,
                                       reinterpret_cast<const uint16_t*>(tile_sorted), R, R_dev, ranges, (uint32_t)tiles, zero, n_zero, prezeroed);
    else if (R) hipLaunchKernelGGL(k_tile_ranges<uint32_t>, dim3((unsigned)((R + 256 * RANGES_ITEMS - 1) / (256 * RANGES_ITEMS))), dim3(256), 0, s, tile_sorted, R, R_dev,
                                   ranges, (uint32_t)tiles, zero, n_zero, prezeroed);
    else {
        hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)tiles, s);
        if (n_zero) hipMemsetAsync(zero, 0, sizeof(uint32_t) * (size_t)n_zero, s);
    }
}

// Enqueue-only forward: what the host would have read, folded on the device.  status[0] = instances the frame needs at the chosen
// tile height (the scan total), [1] = instances binned = min(needed, capacity), [2..7] = 64-bit instance totals for tile heights
// 4 / 8 / 16, [8] = 1 if the capacity was too small (instances were dropped: the frame is wrong and must be redone), [9] = capacity.
__global__ void __launch_bounds__(64) k_finish_totals(const uint32_t* __restrict__ totals, const unsigned long long* __restrict__ slots, uint32_t cap,
                                                      uint32_t* __restrict__ status) {
    const int lane = threadIdx.x;
    unsigned long long v[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        v[q] = lane < LG_INST_SLOTS ? slots[4 * lane + q] : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[q] += __shfl_xor(v[q], o);
    }
    if (lane == 0) {
        const uint32_t need = totals[0];
        status[0] = need; status[1] = need < cap ? need : cap;
        for (int q = 0; q < 3; q++) { status[2 + 2 * q] = (uint32_t)v[q]; status[3 + 2 * q] = (uint32_t)(v[q] >> 32); }
        status[8] = need > cap ? 1u : 0u; status[9] = cap;
        status[10] = (uint32_t)v[3]; status[11] = (uint32_t)(v[3] >> 32);       // ... and for tile height 32
    }
}
void launch_finish_totals(const uint32_t* totals, const unsigned long long* slots, uint32_t cap, uint32_t* status, hipStream_t s) {
    hipLaunchKernelGGL(k_finish_totals, dim3(1), dim3(64), 0, s, totals, slots, cap, status);
}

}  // namespace lg
