// render.hip -- forward and backward alpha-compositing of the range image (gfx950).
//
// Replaces renderCUDA fwd (R3/cr/forward.cu:502-641) and bwd (R3/cr/backward.cu:535-791).
//
// Pixel mapping.  The reference runs 16-thread blocks (a quarter of a wave64) on 16x1 tiles and
// re-evaluates cos/sin of the pixel ray for every (pixel, Gaussian) pair.  Here one wave64 owns a
// 16-column x 4-row pixel patch; a list tile is 16 columns x TH rows (TH/4 "sub" patches).  The
// 16-pixel tile WIDTH is kept equal to the reference's BLOCK_X so that "is this pixel inside the
// Gaussian's rect" -- which is observable, because the rect truncates the footprint at ~3 sigma
// where alpha can still exceed 1/255 -- reduces to a per-lane test on the pixel ROW only.
//
// List segments.  Per-pixel compositing is a serial chain through T, and a 64 x 2650 image has only
// 2656 patches for 1024 SIMDs, so a single walk per patch is bound by the LONGEST list times the
// per-entry latency (measured: 16k entries x ~180 cycles).  Every tile list is therefore cut into S
// equal segments and each (patch, segment) is its own 64-thread workgroup:
//   pass 1  (T only)  every segment walks its entries from T = 1            -> Tpass[patch][seg][lane]
//   pass 2  (full)    segment k starts from T_in = prod_{j<k} Tpass[j]; the reference's T < 1e-4 stop is
//                     applied to this GLOBAL transmittance, so a pixel's walk is exactly the serial one;
//                     segments behind the stop see T_in < 1e-4 and retire immediately
//   combine           per patch: sum the segments' partial sums, pick T_final, write the image
//   backward          one workgroup per segment again; its "colour behind" recurrences are seeded with the
//                     partial sums of the segments (and, multi-GPU, range shells) behind it.
// If a segment's walk trips T < 1e-4, Tpass holds the value that tripped it (< 1e-4), so every later
// segment is shut off; otherwise Tpass is the exact product of its (1 - alpha).  This is the same
// two-phase scheme lidargs_dist.py uses across GPUs, applied across the CUs of one GPU.
//
// Per chunk of 64 list entries: lane l gathers entry l's 64-byte splat record (one aligned segment)
// + row span into registers while the previous chunk is being composited, then parks it in LDS
// component-major; the inner loop reads entry j with four broadcast ds_read_b128.
//
// Backward sums.  The reference issues 20 global float atomics per contributing (pixel, Gaussian)
// pair.  Here the 64 pixels of a wave handle the SAME Gaussian in the same step, so the 16 sums that
// are needed (dL/dsphere follows by linearity from dL/dmean2D, see preprocess.hip) go through a
// wave-level butterfly REDUCE-SCATTER (8+4+2+1+1+1 exchanges instead of 16x6), after which 16 lanes
// each hold one finished sum and issue ONE atomic instruction into a packed 64-byte accumulator line
// of that Gaussian.  Entries no lane contributes to are skipped wholesale.
#include "lidargs_common.h"
#ifdef LG_PRECISE_EXP       /* experiment (tools/residue_ab.sh): the library's expf instead of the hardware exponential */
#define LG_EXPF(x) expf(x)
#else
#define LG_EXPF(x) __expf(x)
#endif
#include <stdlib.h>

namespace lg {

#define LG_CHUNK 64

// Lane statistics of the walks (tools/lane_stats.py; compiled in with -DLG_LANE_STATS only, the product build has none): per walked
// entry, how many of the wave's 64 pixels take it, and how many of its four 16-pixel rows hold one that does.
#ifdef LG_LANE_STATS
__device__ unsigned long long lg_lane_stats[16];
__device__ __forceinline__ void lane_stat(int base, unsigned long long m) {
    if (threadIdx.x % 64 == 0) {
        const int rows = ((m & 0xFFFFull) != 0) + ((m & 0xFFFF0000ull) != 0) + ((m & 0xFFFF00000000ull) != 0) + ((m >> 48) != 0);
        atomicAdd(&lg_lane_stats[base], 1ull); atomicAdd(&lg_lane_stats[base + 1], (unsigned long long)__popcll(m));
        atomicAdd(&lg_lane_stats[base + 2], (unsigned long long)rows); atomicAdd(&lg_lane_stats[base + 3], m ? 1ull : 0ull);
    }
}
void lane_stats_read(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(lg_lane_stats), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(lg_lane_stats), z, sizeof z); }
}
#define LG_LANE_STAT(base, m) lane_stat(base, m)
#else
#define LG_LANE_STAT(base, m) do { } while (0)
#endif

// (u1-part, u2-part) pairs: the splat record interleaves the two tangent directions so that everything the two projections
// share is one packed-fp32 operation (v_pk_fma_f32 / v_pk_mul_f32) on adjacent registers, without moves to pair them up.
typedef float v2f __attribute__((ext_vector_type(2)));

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

// The list entry (a Gaussian id) is fetched for every entry of the chunk, in parallel with its contribution flag, and only the
// 64-byte record waits for both: one round trip less on the critical path of every chunk.
__device__ __forceinline__ Staged gather_record(const float4* __restrict__ rec, const uint32_t* __restrict__ rowspan, uint32_t g, bool valid) {
    Staged s;
    s.a0 = s.a1 = s.a2 = s.a3 = make_float4(0.f, 0.f, 0.f, 0.f);
    s.span = 0;                                                        // empty row span: no pixel matches
    if (valid) {
        const float4* r = rec + 4 * (size_t)g;
        s.a0 = r[0]; s.a1 = r[1]; s.a2 = r[2]; s.a3 = r[3];
        s.span = rowspan[g];
    }
    return s;
}

// A chunk's records parked in LDS, component-major, for the walks' broadcast reads.  The conic's B moves next to the direction
// (slot 0 = s.xyz | B) and the range next to the colours (slot 3 = range, -, colour0, colour1): the T-only walk then reads three
// full 16-byte slots and nothing it does not use -- on gfx950 a ds_read_b96 costs 8 LDS cycles per wave, a ds_read_b128 4
// (MI355X_MICROARCH.md, LDS table), and four SIMDs share the pipe: 14 instead of 20 LDS cycles per wave-entry beside 22 VALU
// instructions.
__device__ __forceinline__ void park_record(float4* s_rec, int lane, const Staged& st) {
    s_rec[lane] = make_float4(st.a0.x, st.a0.y, st.a0.z, st.a3.x);
    s_rec[LG_CHUNK + lane] = st.a1;
    s_rec[2 * LG_CHUNK + lane] = st.a2;
    s_rec[3 * LG_CHUNK + lane] = make_float4(st.a0.w, 0.f, st.a3.z, st.a3.w);
}

// ------------------------------------------------------------------------------------------------
// The walk over the flagged entries of one chunk parked in LDS, shared by the forward kernels.
struct WalkState {
    float T, T_break;                  // transmittance after the last blend | after the last hit (the hand-over value, R3/cr/forward.cu:613-618)
    v2f C01; float D;                  // colour and range sums
    uint32_t last;                     // 1-based segment position of the last entry blended
    bool done;
    unsigned long long took;           // T-only walk: entries of the chunk some pixel took
};

// Software-pipelined and branch-free: every per-pixel decision is a select (no exec-mask branching), and two register sets (a, b)
// are used in turn -- while one entry is evaluated, the record of the next flagged entry is already on its way from LDS into the
// other set.  The reads are volatile, which keeps each where it is written: a single set handed over at the loop's end costs 13
// register moves per entry, and plain loads get folded into the top of the next iteration, where every entry waits out the LDS
// latency.  Only what the pass uses is read (the T-only walk needs neither depth nor colours, nobody the record's opacity): a register
// that is loaded but never used gets reused as a temporary, and the walk would wait for the load to land first.
// `stop` is the transmittance below which a hit ends the walk (R3/cr/forward.cu:608-613): 1e-4 for a walk that carries the true T;
// the fused kernel's T-only walks, which start every segment from 1, pass 1e-4 / (T at the round's start) per lane.
template <bool T_ONLY, bool TRACK = T_ONLY>
__device__ __forceinline__ void walk_flagged(unsigned long long todo, const float4* s_rec, const float* oprow, const v2f qxy, const float qz,
                                             const uint32_t chunk_base, WalkState& w, const float stop = 0.0001f) {
    struct Rec { float4 r0, r1, r2, r3; float op; };
    auto read = [&](int jj) {
        Rec r;
        const float* f3 = reinterpret_cast<const float*>(&s_rec[3 * LG_CHUNK + jj]);
        const float4 s0 = lds_ahead(&s_rec[jj]);                       // s.xyz | B  (park_record)
        if (T_ONLY) {
            r.r0 = make_float4(s0.x, s0.y, s0.z, 0.f);
            r.r3 = make_float4(s0.w, 0.f, 0.f, 0.f);
        } else {
            const v2f col = *(LG_LDS_VOLATILE(v2f))(f3 + 2);
            r.r0 = make_float4(s0.x, s0.y, s0.z, lds_ahead(f3));       // range
            r.r3 = make_float4(s0.w, 0.f, col.x, col.y);
        }
        r.r1 = lds_ahead(&s_rec[LG_CHUNK + jj]); r.r2 = lds_ahead(&s_rec[2 * LG_CHUNK + jj]);
        r.op = lds_ahead(&oprow[4 * jj]);
        return r;
    };
    auto evaluate = [&](const Rec& r, int jj) {
        const v2f exy = v2f{r.r0.x, r.r0.y} - qxy;
        const float ez = r.r0.z - qz;
        const v2f d = exy.x * v2f{r.r1.x, r.r1.y} + exy.y * v2f{r.r1.z, r.r1.w} + ez * v2f{r.r2.x, r.r2.y};   // (d.x, d.y) = delta . (u1', u2')
        const v2f qd = v2f{r.r2.z, r.r2.w} * d * d;                                          // (A dx^2, C dy^2)
        const float power = -0.5f * (qd.x + qd.y) - r.r3.x * d.x * d.y;                       // :601
        // `op` is the entry's opacity on this pixel's row, 0 outside its row span: alpha = 0 there fails the 1/255 test, which is the
        // row test of the rect (R3/cr/forward.cu:580-583 via the tile lists) without two compares per pair.  A lane that is done,
        // or has power > 0 (:602), gets exp(-inf) = 0 the same way, so that `hit` is ONE compare whose lane mask is the ballot itself.
        const float pw = (!w.done && power <= 0.0f) ? power : -INFINITY;
        const float alpha = fminf(0.99f, r.op * LG_EXPF(pw));
        const bool hit = alpha >= 1.0f / 255.0f;
        const float test_T = w.T * (1.f - alpha);
        const bool trip = hit && (test_T < stop);
        if (T_ONLY) {
            // only the hand-over value is kept: T takes the tripping value too, and the lane is done from there on
            w.T = hit ? test_T : w.T;
        } else {
            const bool blend = hit != trip;
            const float wt = blend ? alpha * w.T : 0.f;
            w.C01 += v2f{r.r3.z, r.r3.w} * wt; w.D += r.r0.w * wt;
            w.T = blend ? test_T : w.T;
            w.T_break = hit ? test_T : w.T_break;
            w.last = blend ? (chunk_base + (uint32_t)jj + 1u) : w.last;
        }
        if (TRACK) w.took |= (__ballot(hit) != 0ull) ? (1ull << jj) : 0ull;
        w.done = w.done || trip;
    };
    if (T_ONLY && (todo & (todo + 1ull)) == 0ull) {
        // The T-only walks visit EVERY entry of the chunk: `todo` is a run of ones from bit 0, and the loop is a counted one -- no
        // find-first-set / clear-lowest / compare on a 64-bit scalar mask per entry (a third of the walk's ~22 scalar instructions).
        const int cnt = __builtin_popcountll(todo);
        Rec ra = read(0), rb;
        for (int j = 0; j < cnt; j += 2) {
            const int jb = min(j + 1, cnt - 1);
            rb = read(jb);
            evaluate(ra, j);
            if (j + 1 >= cnt) break;
            ra = read(min(j + 2, cnt - 1));
            evaluate(rb, jb);
        }
        return;
    }
    int ja = __builtin_ctzll(todo);
    todo &= todo - 1ull;
    Rec ra = read(ja), rb;
    while (true) {
        // (the read is not conditional on there being a next entry -- the last one is simply read again: behind a branch, the
        //  compiler's wait counts have to cover the path without new reads in flight, and the walk waits for every read at once)
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

// The T-only walk over entries [0, cnt) of a chunk, second form (round 4).  What the first form (walk_flagged<true>) pays per entry
// beside its ~21 vector instructions is ~14 scalar ones and 3-4 waits -- loop control, the clamped look-ahead index and its address,
// the 64-bit took mask, and the `done` mask, which sits ON the loop-carried path: compare -> s_and / s_or -> the next entry's select
// in front of its exp (a VALU -> SGPR -> SALU -> VALU round trip per entry).  A wave that is alone on its SIMD issues one instruction
// per ~4.5 clocks whatever its kind, so those count like the arithmetic.  Here:
//   * alpha of an entry does not depend on the entries before it; only T does, through one multiply per entry (a lane whose walk
//     has tripped keeps multiplying: any value below `stop` is an equally good hand-over, every consumer only compares it with 1e-4
//     or multiplies on);
//   * entries come in groups of four with constant LDS offsets off one advancing address; the lanes behind the chunk's count parked
//     zero records (alpha = 0), so a group needs no tail handling;
//   * `dead` (the lanes that are out: outside the image, or T below `stop`) is refreshed once per group and only gates the
//     contribution flags (a superset by at most three entries per trip: pass 2 applies the exact rule to the flagged entries);
//   * the took bits are shifted into a scalar accumulator (s_andn2 sets SCC, s_addc shifts it in), newest entry at bit 0.
// Same products in the same order as the first form while nothing trips: the Tpass planes are bit-identical.
template <bool TRACK, int G = 4>
__device__ __forceinline__ void walk_T_only_v2(const int cnt, const float4* s_rec, const float* oprow, const v2f qxy, const float qz,
                                               WalkState& w, const unsigned long long dead0, const float stop = 0.0001f) {
    struct Rec { float4 r0, r1, r2; float op; };
    auto read = [&](int jj) {
        Rec r;
        r.r0 = lds_ahead(&s_rec[jj]);                                  // s.xyz | B  (park_record)
        r.r1 = lds_ahead(&s_rec[LG_CHUNK + jj]); r.r2 = lds_ahead(&s_rec[2 * LG_CHUNK + jj]);
        r.op = lds_ahead(&oprow[4 * jj]);
        return r;
    };
    // -> the factor (1 - alpha) of this pixel for the entry (1 when it does not take it) and the lanes that take it
    auto factor = [&](const Rec& r, unsigned long long& hitmask) {
        const v2f exy = v2f{r.r0.x, r.r0.y} - qxy;
        const float ez = r.r0.z - qz;
        const v2f d = exy.x * v2f{r.r1.x, r.r1.y} + exy.y * v2f{r.r1.z, r.r1.w} + ez * v2f{r.r2.x, r.r2.y};
        const v2f qd = v2f{r.r2.z, r.r2.w} * d * d;
        const float power = -0.5f * (qd.x + qd.y) - r.r0.w * d.x * d.y;   // :601
        const float pw = (power <= 0.0f) ? power : -INFINITY;             // :602 (exp(-inf) = 0: fails the 1/255 test)
        const float alpha = fminf(0.99f, r.op * LG_EXPF(pw));
        const bool hit = alpha >= 1.0f / 255.0f;
        hitmask = __ballot(hit);
        LG_LANE_STAT(0, hitmask);
        return hit ? 1.f - alpha : 1.f;
    };
    float T = w.T;
    unsigned long long dead = dead0 | __ballot(T < stop);
    uint32_t acc_lo = 0u, acc_hi = 0u;                                 // took bits, newest entry at bit 0
    // acc = (acc << 1) | ((hitmask & ~dead) != 0) in three scalar instructions: s_andn2 leaves "result != 0" in SCC, the two s_addc shift
    // it in (x + x + carry) and carry bit 31 of the low word into the high one.  (Written out: the compiler's form of the same goes
    // through a v_cndmask and a v_readfirstlane per entry.)
    auto note = [&](unsigned long long hitmask) {
        if (TRACK) {
            unsigned long long tmp;
            asm volatile("s_andn2_b64 %2, %3, %4\n\ts_addc_u32 %0, %0, %0\n\ts_addc_u32 %1, %1, %1"
                         : "+s"(acc_lo), "+s"(acc_hi), "=&s"(tmp) : "s"(hitmask), "s"(dead) : "scc");
        }
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
        dead = dead0 | __ballot(T < stop);
    }
    w.T = T;
    w.done = w.done || (T < stop);
    if (TRACK) {
        // entry e of the chunk sits at bit G ng - 1 - e of the accumulator
        const unsigned long long acc = ((unsigned long long)acc_hi << 32) | acc_lo;
        const unsigned long long rev = __brevll(acc) >> (64 - G * ng);
        w.took = rev;
    }
}

// The full walk, second form: over entries [0, cnt) of a chunk whose visited records were parked COMPACTED (park_compact below: the
// flagged entries first, in list order, zero records behind them), in groups of four like walk_T_only_v2.  No mask scan per entry
// (find-first-set, clear, compare, the look-ahead's clamp and address: ~10 scalar instructions), and no `done` mask on the
// loop-carried path.  The per-pixel state is three things:
//   Tw   the working transmittance, NEGATED once the pixel is out (outside the image, started below 1e-4, or its walk tripped): then
//        test = Tw (1 - alpha) <= 0 fails `test >= stop` by itself, nothing is blended any more, and |Tw| stays the transmittance
//        after the last blended entry (R3/cr/forward.cu:608-618);
//   C01, D   the sums;   last   the 1-based position of the last blended entry (its position rides in the parked record).
// An entry a live pixel does not take has alpha_eff = 0: test = Tw exactly, blended with weight 0.  The same products and sums in
// the same order as the first form.  T_break (only ever compared with 1e-4 by its consumers -- the combine's stop test, the shells'
// compose) is T while the pixel is live and 0 once its walk has tripped here.
template <bool TRACK>
__device__ __forceinline__ void walk_full_v2(const int cnt, const float4* s_rec, const float* oprow, const v2f qxy, const float qz, WalkState& w) {
    struct Rec { float4 r0, r1, r2, r3; float op; };
    auto read = [&](int jj) {
        Rec r;
        r.r0 = lds_ahead(&s_rec[jj]);                                  // s.xyz | B
        r.r1 = lds_ahead(&s_rec[LG_CHUNK + jj]); r.r2 = lds_ahead(&s_rec[2 * LG_CHUNK + jj]);
        r.r3 = lds_ahead(&s_rec[3 * LG_CHUNK + jj]);                   // range, position, colour0, colour1
        r.op = lds_ahead(&oprow[4 * jj]);
        return r;
    };
    auto alpha_eff = [&](const Rec& r) {
        const v2f exy = v2f{r.r0.x, r.r0.y} - qxy;
        const float ez = r.r0.z - qz;
        const v2f d = exy.x * v2f{r.r1.x, r.r1.y} + exy.y * v2f{r.r1.z, r.r1.w} + ez * v2f{r.r2.x, r.r2.y};
        const v2f qd = v2f{r.r2.z, r.r2.w} * d * d;
        const float power = -0.5f * (qd.x + qd.y) - r.r0.w * d.x * d.y;   // :601
        const float pw = (power <= 0.0f) ? power : -INFINITY;             // :602
        const float alpha = fminf(0.99f, r.op * LG_EXPF(pw));
        return (alpha >= 1.0f / 255.0f) ? alpha : 0.f;                    // :605
    };
    const bool was_done = w.done;
    float Tw = was_done ? -w.T : w.T;
    v2f C01 = w.C01; float D = w.D; uint32_t last = w.last;
    uint32_t acc_lo = 0u, acc_hi = 0u;
    auto blend = [&](const Rec& r, const float a) {
        const float test = Tw * (1.f - a);
        const bool ok = test >= 0.0001f;                               // live, and this entry does not trip it (:608-613)
        const float wt = ok ? a * Tw : 0.f;
        C01 += v2f{r.r3.z, r.r3.w} * wt; D += r.r3.x * wt;             // :615-617
        const bool blended = wt > 0.f;
        LG_LANE_STAT(4, __ballot(blended));
        last = blended ? __float_as_uint(r.r3.y) : last;
        Tw = ok ? test : -fabsf(Tw);
        if (TRACK) {
            const unsigned long long m = __ballot(blended);
            asm volatile("s_cmp_lg_u64 %2, 0\n\ts_addc_u32 %0, %0, %0\n\ts_addc_u32 %1, %1, %1" : "+s"(acc_lo), "+s"(acc_hi) : "s"(m) : "scc");
        }
    };
    const int ng = (cnt + 3) >> 2;
    Rec ra = read(0), rb = read(1);
    for (int g = 0; g < ng; g++) {
        const int j = 4 * g;
        const float a0 = alpha_eff(ra); const Rec r0 = ra; ra = read(j + 2);
        const float a1 = alpha_eff(rb); const Rec r1 = rb; rb = read(j + 3);
        blend(r0, a0); blend(r1, a1);
        const float a2 = alpha_eff(ra); const Rec r2 = ra; ra = read(j + 4);
        const float a3 = alpha_eff(rb); const Rec r3 = rb; rb = read(j + 5);
        blend(r2, a2); blend(r3, a3);
    }
    const bool now_done = (__float_as_uint(Tw) >> 31) != 0u;
    w.T = fabsf(Tw);
    w.T_break = now_done ? (was_done ? w.T_break : 0.f) : Tw;
    w.done = now_done;
    w.C01 = C01; w.D = D; w.last = last;
    if (TRACK) {
        const unsigned long long acc = ((unsigned long long)acc_hi << 32) | acc_lo;
        w.took = __brevll(acc) >> (64 - 4 * ng);
    }
}

// Parks a chunk's records for the second form of the walks: lane `lane` holds entry `lane` of the chunk (zeros if it is not to be
// visited); the visited ones go to slots 0 .. cnt-1 in list order, the others behind them, so that every slot is written and the
// slots behind cnt hold zero records (alpha = 0).  `pos`: the entry's 1-based position in its segment.  Returns cnt.
__device__ __forceinline__ int park_compact(float4* s_rec, float4* s_oprow, const int lane, const bool have, const Staged& st, const uint32_t pos, const int y0) {
    const unsigned long long todo = __ballot(have);
    const int cnt = __builtin_popcountll(todo);
    const uint32_t lo = (uint32_t)todo, hi = (uint32_t)(todo >> 32);
    const int before = (int)__builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));   // visited entries in front of this lane
    const int slot = have ? before : cnt + (lane - before);
    s_rec[slot] = make_float4(st.a0.x, st.a0.y, st.a0.z, st.a3.x);
    s_rec[LG_CHUNK + slot] = st.a1;
    s_rec[2 * LG_CHUNK + slot] = st.a2;
    s_rec[3 * LG_CHUNK + slot] = make_float4(st.a0.w, __uint_as_float(pos), st.a3.z, st.a3.w);
    s_oprow[slot] = rows_opacity(st.span, st.a3.y, y0);
    return cnt;
}

// ------------------------------------------------------------------------------------------------
// One workgroup = (patch, segment).  T_ONLY: pass 1.  Otherwise pass 2.
// Register budget of six waves per SIMD (80 / 67 VGPRs, no spills; unconstrained the full walk takes 108, the T-only one 92): measured -5 us / -12 us on pass 2 of the
// cfg3 frame at opacity scale 1 / 0.1, no spills (EXPERIMENTS.md "occupancy of the blend kernels"; 8 waves and any cap on the backward are slower).
#ifndef LG_FWD_WAVES
#define LG_FWD_WAVES 6
#endif
template <bool T_ONLY, bool V2 = false>
__attribute__((amdgpu_waves_per_eu(LG_FWD_WAVES, LG_FWD_WAVES)))
__global__ void __launch_bounds__(64) k_render_forward(const RenderFwdArgs a) {
    __shared__ float4 s_rec[4 * LG_CHUNK + 2];                         // (+2: the second forms' look-ahead reads run two slots past the last component)
    __shared__ float4 s_oprow[LG_CHUNK + 2];                           // the entry's opacity per pixel row of this patch, 0 outside its row span (+2: as above)
    const int lane = threadIdx.x;
    const int S = a.S;
    const int wpt = a.grid.waves_per_tile;
    int patch, seg;                                                    // patch = tile * waves_per_tile + sub
    if (!block_patch_segment(blockIdx.x, a.grid.window_patches(), a.seg_hi - a.seg_lo, patch, seg)) return;
    patch = a.grid.global_patch(patch);
    seg += a.seg_lo;
    const int tile = patch / wpt, sub = patch - tile * wpt;
    const uint2 tr = a.ranges[tile];
    const int St = segment_count(tr, S, a.seg_len);
    if (seg >= St) return;
    if (a.alive && seg >= (int)a.alive[patch]) return;                  // every pixel saturated before this segment (never walked)
    const PixelSetup px = pixel_setup(a.grid, a.coltab, a.rowtab, tile, sub, lane);
    const uint2 sr = segment_range(tr, St, seg);
    const uint32_t n = sr.y - sr.x;
    float* segbase = a.seg + ((size_t)patch * S + seg) * (LG_SEG_PLANES * 64);

    const int y0 = (tile / a.grid.tiles_x) * a.grid.TH + sub * LG_WAVE_ROWS;       // first pixel row of the patch
    const float* oprow = reinterpret_cast<const float*>(s_oprow) + (lane >> 4);
    const v2f qxy = v2f{px.q.x, px.q.y};

    uint8_t* fl = a.flags ? a.flags + (size_t)sub * a.R + sr.x : nullptr;
    // pass 2, second form: the first chunk's list entries and flags are requested in front of the transmittance planes (a loop of
    // loads and waits), so that the two round trips -- ids, then records -- overlap it instead of following it
    uint32_t g_first = 0u; bool have_first = false;
    if (!T_ONLY && V2 && n > 0) {
        g_first = (uint32_t)lane < n ? a.point_list[sr.x + lane] : 0u;
        have_first = (uint32_t)lane < n && (!fl || fl[lane] != 0);
    }
    float T = 1.0f;
    if (!T_ONLY) {
        if (a.T_in && px.inside) T = a.T_in[px.pix];
        const float* tp = a.seg + (size_t)patch * S * (LG_SEG_PLANES * 64) + LG_SEG_TPASS * 64 + lane;
        T = plane_product(T, tp, LG_SEG_PLANES * 64, seg);
    }
    // a lane that starts below the threshold can never blend again: every contributing entry trips T < 1e-4
    WalkState w{T, T, v2f{0.f, 0.f}, 0.f, 0u, !px.inside || (!T_ONLY && T < 0.0001f), 0ull};

    const uint32_t nchunks = (n + LG_CHUNK - 1) / LG_CHUNK;
    // Contribution flags: pass 1 records, per (sub-patch, entry), whether ANY pixel took the entry; pass 2 (and the
    // backward) then touch only flagged entries -- no record gather, no LDS traffic, no evaluation for the rest.
    // Pass 1 starts from T = 1 (>= the true transmittance), so every pixel is active at least as long as in
    // pass 2: the flagged set is a superset of what pass 2 / backward can ever blend.
    uint32_t c_done = 0;                                               // chunks whose flags pass 1 has written
    if (__ballot(!w.done) != 0ull && n > 0) {
        auto entry_valid = [&](uint32_t k) { return k < n && (T_ONLY || !fl || fl[k] != 0); };
        uint32_t g_fetched = 0u;                                       // the Gaussian of the entry this lane fetched last (pass 1 marks it as touched with its flag)
        auto fetch = [&](uint32_t k, bool& have) {
            const uint32_t g = k < n ? a.point_list[sr.x + k] : 0u;    // not waiting for the flag
            have = entry_valid(k);
            g_fetched = g;
            return gather_record(a.rec, a.rowspan, g, have);
        };
        bool have;
        Staged st;
        if (!T_ONLY && V2) { have = have_first; st = gather_record(a.rec, a.rowspan, g_first, have); }
        else st = fetch((uint32_t)lane, have);
        for (uint32_t c = 0; c < nchunks; c++) {
            const uint32_t g_mine = g_fetched;                         // entry c * 64 + lane's
            __syncthreads();
            unsigned long long todo = __ballot(have);                  // entries of this chunk worth visiting
            int cnt = 0;
            if (!T_ONLY && V2) cnt = park_compact(s_rec, s_oprow, lane, have, st, c * LG_CHUNK + (uint32_t)lane + 1u, y0);
            else {
                park_record(s_rec, lane, st);
                s_oprow[lane] = rows_opacity(st.span, st.a3.y, y0);
            }
            __syncthreads();
            if (c + 1 < nchunks) {
                st = fetch((c + 1) * LG_CHUNK + lane, have);
            }
            if (__ballot(!w.done) == 0ull) break;                       // R3/cr/forward.cu:559-561 early-out
            w.took = 0ull;
            if (todo) {
                if (T_ONLY && V2) walk_T_only_v2<true>(__builtin_popcountll(todo), s_rec, oprow, qxy, px.q.z, w, __ballot(!px.inside));
                else if (V2) walk_full_v2<false>(cnt, s_rec, oprow, qxy, px.q.z, w);
                else walk_flagged<T_ONLY>(todo, s_rec, oprow, qxy, px.q.z, c * LG_CHUNK, w);
            }
            if (T_ONLY && fl) {
                const uint32_t k = c * LG_CHUNK + lane;
                const bool took = (w.took >> lane) & 1ull;
                if (k < n) fl[k] = (uint8_t)took;
                if (a.touched && took && k < n) a.touched[g_mine] = 1;  // (same value from every patch that takes it: a plain byte store)
                c_done = c + 1;
            }
        }
    }
    if (T_ONLY && fl) {                                                // entries pass 1 never reached: nobody can take them
        for (uint32_t c = c_done; c < nchunks; c++) {
            const uint32_t k = c * LG_CHUNK + lane;
            if (k < n) fl[k] = 0;
        }
    }

    if (T_ONLY) {
        segbase[LG_SEG_TPASS * 64 + lane] = w.T;
    } else {
        segbase[LG_SEG_C0 * 64 + lane] = w.C01.x;
        segbase[LG_SEG_C1 * 64 + lane] = w.C01.y;
        segbase[LG_SEG_D * 64 + lane] = w.D;
        segbase[LG_SEG_TEND * 64 + lane] = w.T;
        segbase[LG_SEG_TBREAK * 64 + lane] = w.T_break;
        reinterpret_cast<uint32_t*>(segbase)[LG_SEG_LAST * 64 + lane] = w.last;
    }
}

// ------------------------------------------------------------------------------------------------
// Pass 2 over GROUPS of consecutive segments.  A (patch, segment) workgroup of pass 2 walks only the flagged entries of ~63 list
// positions -- a few microseconds of work behind three dependent memory round trips (tile range -> ids + flags -> records).  Here a
// workgroup takes `G` consecutive segments of its patch and walks them one after the other, exactly like the serial reference
// walk: T is carried across the segment boundaries (only the group's first segment starts from the pass-1 products), the records
// of the next segment's first chunk are gathered while the current segment's last chunk is composited, and every segment still
// gets its own planes (partial sums, T_end, T_break, last), which is all the combine and the backward look at.
//
// FIRST: the head of every list -- its first G segments -- walked ONCE, instead of a T-only walk per segment followed by pass 2 over
// what was flagged.  The head starts from the true transmittance (1, or T_in), so this walk IS the serial reference walk: it visits
// every entry, records the contribution flags itself (exact ones: a subset of what the T-only walks, which restart from T = 1 in
// every segment, would flag), and leaves in the head's T_pass planes the hand-over value (in segment 0's; 1 in the others), so that
// the product pass 2 and k_render_alive form over the segments in front of the tail is the transmittance the tail starts from.
template <bool FIRST, bool V2 = false>
__global__ void __launch_bounds__(64) k_render_pass2_grouped(const RenderFwdArgs a, const int G) {
    __shared__ float4 s_rec[4 * LG_CHUNK + 2];                         // (+2: walk_full_v2's look-ahead reads)
    __shared__ float4 s_oprow[LG_CHUNK + 2];
    const int lane = threadIdx.x;
    const int S = a.S;
    const int wpt = a.grid.waves_per_tile;
    const int groups = FIRST ? 1 : (((S - a.seg_lo + G - 1) / G) | 1); // odd, like S: keeps the group index decorrelated from the XCD (b % 8)
    int patch, grp;
    if (!block_patch_segment(blockIdx.x, a.grid.window_patches(), groups, patch, grp)) return;
    patch = a.grid.global_patch(patch);
    const int tile = patch / wpt, sub = patch - tile * wpt;
    const uint2 tr = a.ranges[tile];
    const int St = segment_count(tr, S, a.seg_len);
    const int limit = (!FIRST && a.alive) ? min(St, (int)a.alive[patch]) : St;   // segments behind the limit were never walked by pass 1
    const int s0 = (FIRST ? 0 : a.seg_lo) + grp * G;                   // (pass 2 may start behind a head that was walked once)
    if (s0 >= limit) return;
    const int s1 = min(limit, s0 + G);
    const PixelSetup px = pixel_setup(a.grid, a.coltab, a.rowtab, tile, sub, lane);
    const size_t pstride = LG_SEG_PLANES * 64;
    float* pbase = a.seg + (size_t)patch * S * pstride;

    float T = 1.0f;
    if (a.T_in && px.inside) T = a.T_in[px.pix];
    const float T_start = T;
    T = plane_product(T, pbase + LG_SEG_TPASS * 64 + lane, pstride, s0);
    WalkState w{T, T, v2f{0.f, 0.f}, 0.f, 0u, !px.inside || T < 0.0001f, 0ull};
    const int y0 = (tile / a.grid.tiles_x) * a.grid.TH + sub * LG_WAVE_ROWS;       // first pixel row of the patch
    const float* oprow = reinterpret_cast<const float*>(s_oprow) + (lane >> 4);
    const v2f qxy = v2f{px.q.x, px.q.y};
    uint8_t* flp = a.flags ? a.flags + (size_t)sub * a.R : nullptr;

    uint32_t g_fetched = 0u;                                           // FIRST: the Gaussian of the entry this lane fetched last (marked as touched with its flag)
    auto fetch = [&](uint2 sr, uint32_t n, uint32_t c, bool& have) {
        const uint32_t k = c * LG_CHUNK + (uint32_t)lane;
        const uint32_t g = k < n ? a.point_list[sr.x + k] : 0u;        // not waiting for the flag
        have = k < n && (FIRST || !flp || flp[sr.x + k] != 0);
        g_fetched = g;
        return gather_record(a.rec, a.rowspan, g, have);
    };
    uint2 sr = segment_range(tr, St, s0);
    bool have;
    Staged st = fetch(sr, sr.y - sr.x, 0u, have);
    bool all_done = false;
    float hand = T;                                                    // FIRST: what a walk behind this group starts from (< 1e-4: stopped)
    for (int sg = s0; sg < s1; sg++) {
        sr = segment_range(tr, St, sg);
        const uint32_t n = sr.y - sr.x;
        const uint32_t nchunks = (n + LG_CHUNK - 1) / LG_CHUNK;
        w.T_break = w.T; w.C01 = v2f{0.f, 0.f}; w.D = 0.f; w.last = 0u;
        uint32_t c_done = 0;                                           // FIRST: chunks of this segment whose flags are written
        if (!all_done) {
            for (uint32_t c = 0; c < nchunks; c++) {
                const uint32_t g_mine = g_fetched;                     // entry c * 64 + lane's
                __syncthreads();
                const unsigned long long todo = __ballot(have);
                int cnt = 0;
                if (V2) cnt = park_compact(s_rec, s_oprow, lane, have, st, c * LG_CHUNK + (uint32_t)lane + 1u, y0);   // (FIRST: every entry is visited, slot = lane)
                else {
                    park_record(s_rec, lane, st);
                    s_oprow[lane] = rows_opacity(st.span, st.a3.y, y0);
                }
                __syncthreads();
                if (c + 1 < nchunks) st = fetch(sr, n, c + 1, have);
                else if (sg + 1 < s1) { const uint2 nsr = segment_range(tr, St, sg + 1); st = fetch(nsr, nsr.y - nsr.x, 0u, have); }
                if (__ballot(!w.done) == 0ull) { all_done = true; break; }   // R3/cr/forward.cu:559-561 early-out
                w.took = 0ull;
                if (todo) {
                    if (V2) walk_full_v2<FIRST>(cnt, s_rec, oprow, qxy, px.q.z, w);
                    else walk_flagged<false, FIRST>(todo, s_rec, oprow, qxy, px.q.z, c * LG_CHUNK, w);
                }
                if (FIRST && flp) {
                    const uint32_t k = c * LG_CHUNK + lane;
                    const bool took = (w.took >> lane) & 1ull;
                    if (k < n) flp[sr.x + k] = (uint8_t)took;
                    if (a.touched && took && k < n) a.touched[g_mine] = 1;
                    c_done = c + 1;
                }
            }
        }
        if (FIRST && flp) {                                            // entries the walk never reached: nobody takes them
            for (uint32_t c = c_done; c < nchunks; c++) {
                const uint32_t k = c * LG_CHUNK + lane;
                if (k < n) flp[sr.x + k] = 0;
            }
        }
        float* segbase = pbase + (size_t)sg * pstride;
        segbase[LG_SEG_C0 * 64 + lane] = w.C01.x;
        segbase[LG_SEG_C1 * 64 + lane] = w.C01.y;
        segbase[LG_SEG_D * 64 + lane] = w.D;
        segbase[LG_SEG_TEND * 64 + lane] = w.T;
        segbase[LG_SEG_TBREAK * 64 + lane] = w.T_break;
        reinterpret_cast<uint32_t*>(segbase)[LG_SEG_LAST * 64 + lane] = w.last;
        if (FIRST) {
            hand = hand < 0.0001f ? hand : w.T_break;                  // T after the segment's last hit: the tripping value if it stopped here
            segbase[LG_SEG_TPASS * 64 + lane] = 1.0f;
        }
    }
    if (FIRST) {
        // the consumers multiply T_in by the product of the planes in front of them: this plane carries the head's factor
        pbase[LG_SEG_TPASS * 64 + lane] = (T_start > 0.f) ? hand / T_start : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// The whole forward blend of a patch in ONE workgroup of NW waves (the 64 x 2650 frames' plan: 64-entry segments).
//
// The multi-launch form above -- T-only walks of the first segments, k_render_alive, T-only walks of the rest, pass 2, combine --
// gathers every walked record two or three times (measured: 420 MB of fabric traffic for 84 MB of needed records), launches S
// workgroups per patch of which most retire at once, and can only notice that a patch has saturated at a launch boundary.  Here the
// patch's segments are walked in ROUNDS of NW consecutive segments, one per wave:
//   1. wave w walks segment k = r NW + w transmittance-only from T = 1 (the product is what the segments behind it need) and records
//      the contribution flags; its stop test is against 1e-4 / carry, carry = the true transmittance at the round's start -- the
//      true T of any later point is <= carry x (local product), so a lane is done here no later than in the serial walk;
//   2. the waves exchange their products through LDS: T_in(k) = carry x prod_{v < w} Tpass(r NW + v), multiplied in list order, i.e.
//      the very sequence of products plane_product() forms in the multi-launch form -- the images of the two forms are bit-identical;
//   3. wave w walks the FLAGGED entries of its segment again from T_in(k) with the reference's stop rule, accumulating colour and
//      range; a 64-entry segment is still parked in the wave's LDS slot from step 1 (no second gather), longer ones are re-gathered
//      chunk by chunk (they were fetched microseconds ago);
//   4. every wave folds the round's NW partial results in list order (the combine's logic), the carry advances, and the patch
//      leaves as soon as every pixel has stopped -- at most one round behind the point where the serial walk ends.
// The per-segment planes, the flags and the patch's limit are written exactly as the backward expects them (k_render_backward).
template <int NW>
__global__ void __launch_bounds__(64 * NW) k_render_fused(const RenderFwdArgs a) {
    __shared__ float4 s_rec_all[NW][4 * LG_CHUNK + 2];                 // (+2: the second forms' look-ahead reads stay inside the wave's own slot)
    __shared__ float4 s_oprow_all[NW][LG_CHUNK + 2];                 // (+2: walk_T_only_v2's look-ahead reads)
    __shared__ float s_x[NW][6][64];                                   // per wave: Tpass | C0, C1, D, T_end, T_break of its segment
    __shared__ unsigned long long s_took[NW][8];                       // contribution masks of the chunks of the wave's segment (<= 8 kept)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float4* s_rec = s_rec_all[w];
    float4* s_oprow = s_oprow_all[w];
    const int S = a.S;
    const int wpt = a.grid.waves_per_tile;
    const int patch = a.grid.global_patch(blockIdx.x);
    const int tile = patch / wpt, sub = patch - tile * wpt;
    const uint2 tr = a.ranges[tile];
    const int St = segment_count(tr, S, a.seg_len);
    const PixelSetup px = pixel_setup(a.grid, a.coltab, a.rowtab, tile, sub, lane);
    const int y0 = (tile / a.grid.tiles_x) * a.grid.TH + sub * LG_WAVE_ROWS;
    const float* oprow = reinterpret_cast<const float*>(s_oprow) + (lane >> 4);
    const v2f qxy = v2f{px.q.x, px.q.y};
    const size_t pstride = LG_SEG_PLANES * 64;
    float* pbase = a.seg + (size_t)patch * S * pstride;
    uint8_t* flp = a.flags + (size_t)sub * a.R;

    float carry = (a.T_in && px.inside) ? a.T_in[px.pix] : 1.0f;       // true transmittance in front of the current round
    // what the combine folds, kept by every wave (identical values): running sums, T after the last blended entry, hand-over value
    float C0 = 0.f, C1 = 0.f, D = 0.f, T_final = carry, T_hand = carry;
    bool stopped = !px.inside;
    int walked = 0;                                                    // segments whose planes are written

    for (int k0 = 0; k0 < St; k0 += NW) {                              // (an empty list has one empty segment: its planes are written too)
        const int k = k0 + w;
        const bool mine = k < St;
        uint2 sr = make_uint2(0u, 0u);
        uint32_t n = 0, nchunks = 0;
        if (mine) { sr = segment_range(tr, St, k); n = sr.y - sr.x; nchunks = (n + LG_CHUNK - 1) / LG_CHUNK; }
        // ---- 1. transmittance-only walk from 1, flags -------------------------------------------------------------------------
        float Tp = 1.0f;
        if (mine) {
            // (a hair low: a smaller threshold only walks further, never stops a lane the serial walk would still blend)
            const float stop = carry > 0.0001f ? (0.0001f / carry) * 0.999f : 2.0f;   // carry already below: every hit ends the walk
            WalkState ws{1.0f, 1.0f, v2f{0.f, 0.f}, 0.f, 0u, !px.inside || carry < 0.0001f, 0ull};
            const unsigned long long dead0 = __ballot(ws.done);
            uint32_t c_done = 0;
            if (__ballot(!ws.done) != 0ull && n > 0) {
                bool have;
                uint32_t g_fetched = 0u;
                auto fetch = [&](uint32_t kk, bool& hv) {
                    const uint32_t g = kk < n ? a.point_list[sr.x + kk] : 0u;
                    hv = kk < n;
                    g_fetched = g;
                    return gather_record(a.rec, a.rowspan, g, hv);
                };
                Staged st = fetch((uint32_t)lane, have);
                for (uint32_t c = 0; c < nchunks; c++) {
                    const uint32_t g_mine = g_fetched;                 // entry c * 64 + lane's
                    __builtin_amdgcn_wave_barrier();
                    park_record(s_rec, lane, st);
                    s_oprow[lane] = rows_opacity(st.span, st.a3.y, y0);
                    const unsigned long long todo = __ballot(have);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (c + 1 < nchunks) st = fetch((c + 1) * LG_CHUNK + lane, have);
                    if (__ballot(!ws.done) == 0ull) break;
                    ws.took = 0ull;
                    if (todo) {
                        if (a.walk2) walk_T_only_v2<true>(__builtin_popcountll(todo), s_rec, oprow, qxy, px.q.z, ws, dead0, stop);
                        else walk_flagged<true>(todo, s_rec, oprow, qxy, px.q.z, c * LG_CHUNK, ws, stop);
                    }
                    const uint32_t kk = c * LG_CHUNK + lane;
                    const bool took = (ws.took >> lane) & 1ull;
                    if (kk < n) flp[sr.x + kk] = (uint8_t)took;
                    if (a.touched && took && kk < n) a.touched[g_mine] = 1;
                    if (lane == 0 && c < 8) s_took[w][c] = ws.took;
                    c_done = c + 1;
                }
            }
            for (uint32_t c = c_done; c < nchunks; c++) {              // entries the walk never reached: nobody can take them
                const uint32_t kk = c * LG_CHUNK + lane;
                if (kk < n) flp[sr.x + kk] = 0;
                if (lane == 0 && c < 8) s_took[w][c] = 0ull;
            }
            Tp = ws.T;
        }
        s_x[w][0][lane] = Tp;
        __syncthreads();
        // ---- 2. the transmittance this wave's segment starts from ---------------------------------------------------------------
        float T = carry;
        for (int v = 0; v < w; v++) T *= s_x[v][0][lane];
        // ---- 3. full walk of the flagged entries from the true T -------------------------------------------------------------
        WalkState wf{T, T, v2f{0.f, 0.f}, 0.f, 0u, !px.inside || T < 0.0001f, 0ull};
        if (mine && __ballot(!wf.done) != 0ull && n > 0) {
            if (nchunks == 1) {
                const unsigned long long todo = s_took[w][0];         // (own wave's write, in program order)
                if (todo) walk_flagged<false>(todo, s_rec, oprow, qxy, px.q.z, 0u, wf);
            } else {
                for (uint32_t c = 0; c < nchunks; c++) {
                    unsigned long long todo;
                    bool have;
                    const uint32_t kk = c * LG_CHUNK + lane;
                    if (c < 8) { todo = s_took[w][c]; have = (todo >> lane) & 1ull; }
                    else { have = kk < n && flp[sr.x + kk] != 0; todo = __ballot(have); }
                    if (todo == 0ull) continue;
                    const uint32_t g = have ? a.point_list[sr.x + kk] : 0u;
                    const Staged st = gather_record(a.rec, a.rowspan, g, have);
                    __builtin_amdgcn_wave_barrier();
                    park_record(s_rec, lane, st);
                    s_oprow[lane] = rows_opacity(st.span, st.a3.y, y0);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (__ballot(!wf.done) == 0ull) break;
                    walk_flagged<false>(todo, s_rec, oprow, qxy, px.q.z, c * LG_CHUNK, wf);
                }
            }
        }
        if (mine) {
            float* segbase = pbase + (size_t)k * pstride;
            segbase[LG_SEG_TPASS * 64 + lane] = Tp;
            segbase[LG_SEG_C0 * 64 + lane] = wf.C01.x;
            segbase[LG_SEG_C1 * 64 + lane] = wf.C01.y;
            segbase[LG_SEG_D * 64 + lane] = wf.D;
            segbase[LG_SEG_TEND * 64 + lane] = wf.T;
            segbase[LG_SEG_TBREAK * 64 + lane] = wf.T_break;
            reinterpret_cast<uint32_t*>(segbase)[LG_SEG_LAST * 64 + lane] = wf.last;
        }
        s_x[w][1][lane] = wf.C01.x; s_x[w][2][lane] = wf.C01.y; s_x[w][3][lane] = wf.D; s_x[w][4][lane] = wf.T; s_x[w][5][lane] = wf.T_break;
        __syncthreads();
        // ---- 4. fold the round in list order (k_render_combine's logic), advance the carry ------------------------------------
        const int nseg = min(NW, St - k0);
        for (int v = 0; v < nseg; v++) {
            const bool use = !stopped;
            C0 += use ? s_x[v][1][lane] : 0.f; C1 += use ? s_x[v][2][lane] : 0.f; D += use ? s_x[v][3][lane] : 0.f;
            T_final = use ? s_x[v][4][lane] : T_final;
            const float tb = s_x[v][5][lane];
            T_hand = use ? tb : T_hand;
            stopped = stopped || (use && tb < 0.0001f);
            carry *= s_x[v][0][lane];
        }
        walked = k0 + nseg;
        const bool all_stopped = __ballot(!stopped) == 0ull;
        __syncthreads();                                               // s_x is rewritten by the next round
        if (all_stopped) break;
    }
    if (w == 0) {
        if (lane == 0) a.alive[patch] = (walked >= St) ? 255 : (uint8_t)min(walked, 254);
        if (px.inside) {
            const size_t N = (size_t)a.grid.W * a.grid.H;
            a.final_T[px.pix] = T_final;
            if (a.T_pass) a.T_pass[px.pix] = T_hand;
            const float b0 = a.bg ? a.bg[0] : 0.f, b1 = a.bg ? a.bg[1] : 0.f;
            a.out_color[px.pix] = C0 + T_final * b0;
            a.out_color[N + px.pix] = C1 + T_final * b1;
            a.out_depth[px.pix] = D;
            a.out_occ[px.pix] = 1.f - T_final;
        }
    }
}

// Pass 1 runs in rounds of growing depth; after the round that ends at segment `front`, a patch stays open (limit 255) if
// some pixel's transmittance through those segments (product of the T-only walks, i.e. the hand-over value of a walk from
// T = 1) is still >= 1e-4 and its list goes on; otherwise its limit becomes `front` and nothing behind is ever walked.
// Pass 1 starts from T = 1 >= the true transmittance, so what is closed here is closed for pass 2 and the backward too.
__global__ void __launch_bounds__(64) k_render_alive(const RenderFwdArgs a) {
    const int lane = threadIdx.x;
    const int S = a.S;
    const int patch = a.grid.global_patch(blockIdx.x);
    if (a.seg_lo > 0 && a.alive[patch] != 255) return;                 // closed by an earlier round
    const int wpt = a.grid.waves_per_tile;
    const int tile = patch / wpt, sub = patch - tile * wpt;
    const TileGrid& g = a.grid;
    const int x = (tile % g.tiles_x) * LG_TILE_W + (lane & 15);
    const int y = (tile / g.tiles_x) * g.TH + sub * LG_WAVE_ROWS + (lane >> 4);
    const bool inside = x < g.W && y < g.H;
    const int St = segment_count(a.ranges[tile], S, a.seg_len);
    const float* sb = a.seg + (size_t)patch * S * (LG_SEG_PLANES * 64) + lane;
    const size_t stride = LG_SEG_PLANES * 64;
    float T = 1.f;
    const int kf = min(a.front, St);
    for (int k = 0; k < kf && T >= 0.0001f; k++) T *= sb[k * stride + LG_SEG_TPASS * 64];
    const unsigned long long open = __ballot(inside && T >= 0.0001f);
    if (lane == 0) a.alive[patch] = (St > a.front && open != 0ull) ? 255 : (uint8_t)min(a.front, 254);
}

// Per patch: fold the segments (pass 2 results, or pass 1 products when t_only) into the image planes.
__global__ void __launch_bounds__(64) k_render_combine(const RenderFwdArgs a) {
    const int lane = threadIdx.x;
    const int S = a.S;
    const int patch = a.grid.global_patch(blockIdx.x);
    const int wpt = a.grid.waves_per_tile;
    const int tile = patch / wpt, sub = patch - tile * wpt;
    const TileGrid& g = a.grid;
    const int x = (tile % g.tiles_x) * LG_TILE_W + (lane & 15);
    const int y = (tile / g.tiles_x) * g.TH + sub * LG_WAVE_ROWS + (lane >> 4);
    if (x >= g.W || y >= g.H) return;
    const int pix = y * g.W + x;
    const float* sb = a.seg + (size_t)patch * S * (LG_SEG_PLANES * 64) + lane;
    const size_t stride = LG_SEG_PLANES * 64;
    int St = segment_count(a.ranges[tile], S, a.seg_len);
    if (a.alive) St = min(St, (int)a.alive[patch]);                    // the planes behind a patch's limit were never written

    if (a.transmittance_only) {
        // hand-over value of the whole list: product of the segments' (a tripped segment makes it < 1e-4)
        float T = 1.f;
        // stop at the first product below the threshold: the planes behind it may never have been written (dead patch),
        // and every consumer of T_pass only asks whether it is < 1e-4
        for (int k = 0; k < St && T >= 0.0001f; k++) T *= sb[k * stride + LG_SEG_TPASS * 64];
        if (a.T_pass) a.T_pass[pix] = T;
        return;
    }
    float C0 = 0.f, C1 = 0.f, D = 0.f;
    float T_final = a.T_in ? a.T_in[pix] : 1.f, T_hand = T_final;
    bool stopped = false;
    int taken = 0, kend = 0;                                           // segments this pixel's walk went through | segments the patch's loop loaded
    // four segments per step: their twenty loads are issued together (every plane below St was written by pass 2), and a pixel
    // whose walk has ended simply stops taking them; the patch leaves once all of its pixels have
    for (int k0 = 0; k0 < St; k0 += 4) {
        float c0[4], c1[4], dd[4], te[4], tb[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int k = min(k0 + i, St - 1);
            c0[i] = sb[k * stride + LG_SEG_C0 * 64]; c1[i] = sb[k * stride + LG_SEG_C1 * 64]; dd[i] = sb[k * stride + LG_SEG_D * 64];
            te[i] = sb[k * stride + LG_SEG_TEND * 64]; tb[i] = sb[k * stride + LG_SEG_TBREAK * 64];
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool use = !stopped && (k0 + i < St);
            C0 += use ? c0[i] : 0.f; C1 += use ? c1[i] : 0.f; D += use ? dd[i] : 0.f;
            T_final = use ? te[i] : T_final; T_hand = use ? tb[i] : T_hand;
            taken += use ? 1 : 0;
            stopped = stopped || (use && tb[i] < 0.0001f);             // the walk ended inside segment k
        }
        kend = min(St, k0 + 4);
        if (__ballot(!stopped) == 0ull) break;
    }
    if (a.fill.cnt) {
        // the backward's work list: the segments some pixel of the patch walked through (nothing was blended behind them).  Some pixel
        // was still walking when the last step began, so the count lies in that step; one atomic per patch, by the first lane that is
        // inside the image (the others left above)
        int n = 0;
        for (int k = kend; k > kend - 4 && k > 0 && n == 0; k--) if (__ballot(taken >= k) != 0ull) n = k;
        if (n > 0) {
            const unsigned long long act = __ballot(true);
            const int leader = __ffsll((long long)act) - 1;
            const uint32_t r = (uint32_t)patch % LG_WORK_REGIONS;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(a.fill.cnt + r * LG_WORK_CNT_STRIDE, (uint32_t)n);
            base = (uint32_t)__shfl((int)base, leader);
            const int rank = __popcll(act & ((1ull << lane) - 1ull)), nact = __popcll(act);
            for (int k = rank; k < n; k += nact) a.fill.items[(size_t)r * a.fill.cap + base + (uint32_t)k] = ((uint32_t)patch << 8) | (uint32_t)k;
        }
    }
    const size_t N = (size_t)g.W * g.H;
    a.final_T[pix] = T_final;
    if (a.T_end_out) a.T_end_out[pix] = T_final;
    if (a.T_pass) a.T_pass[pix] = T_hand;
    const float b0 = a.bg ? a.bg[0] : 0.f, b1 = a.bg ? a.bg[1] : 0.f;
    a.out_color[pix] = C0 + T_final * b0;                              // :637
    a.out_color[N + pix] = C1 + T_final * b1;
    a.out_depth[pix] = D;
    a.out_occ[pix] = 1.f - T_final;
}

// LIDARGS_WALK2 (bits; default all): 1 = the second form of the T-only walk, 2 = of the full walk, 4 = of the backward walk; 0 = the first forms (A/B)
static int walk2() {
    static const int m = [] { const char* e = getenv("LIDARGS_WALK2"); return e ? atoi(e) : 7; }();
    return m;
}
// waves per workgroup of the fused forward blend: LIDARGS_FUSED_WAVES = 4, 8 (default) or 16
void launch_render_fused(const RenderFwdArgs& a_, hipStream_t s) {
    RenderFwdArgs a = a_; a.walk2 = walk2() & 1;
    static const int env = [] { const char* e = getenv("LIDARGS_FUSED_WAVES"); return e ? atoi(e) : 0; }();
    const unsigned patches = (unsigned)a.grid.window_patches();
    if (env == 4) hipLaunchKernelGGL(k_render_fused<4>, dim3(patches), dim3(256), 0, s, a);
    else if (env == 16) hipLaunchKernelGGL(k_render_fused<16>, dim3(patches), dim3(1024), 0, s, a);
    else hipLaunchKernelGGL(k_render_fused<8>, dim3(patches), dim3(512), 0, s, a);
}
void launch_render_alive(const RenderFwdArgs& a, hipStream_t s) {
    const unsigned patches = (unsigned)a.grid.window_patches();
    hipLaunchKernelGGL(k_render_alive, dim3(patches), dim3(64), 0, s, a);
}
void launch_render_pass1(const RenderFwdArgs& a, hipStream_t s) {
    const unsigned blocks = segment_grid(a.grid.window_patches(), a.seg_hi - a.seg_lo);
    if (walk2() & 1) hipLaunchKernelGGL((k_render_forward<true, true>), dim3(blocks), dim3(64), 0, s, a);
    else hipLaunchKernelGGL((k_render_forward<true, false>), dim3(blocks), dim3(64), 0, s, a);
}
// Segments a pass-2 workgroup walks in a row.  Measured (r02): on the 64-entry plan of the 64x2650 frames grouping LOSES (cfg3 pass 2
// 0.069 ms at 1, 0.090 at 2, 0.146 at 4: the patches that never saturate set the launch's length, and their walk becomes G times as
// long), so it stays at 1 = one workgroup per (patch, segment); on the 128-entry plan of the big frames 8 wins (cfg4 0.104 ->
// 0.077 ms).  LIDARGS_P2_GROUP overrides.
static int pass2_group(int seg_len) {
    static const int env = [] { const char* e = getenv("LIDARGS_P2_GROUP"); const int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v > 64 ? 64 : v); }();
    return env ? env : (seg_len >= LG_SEG_LEN_DEFAULT ? 8 : 1);
}
void launch_render_pass2(const RenderFwdArgs& a, hipStream_t s) {
    const int G = pass2_group(a.seg_len);
    if (G <= 1 || a.seg_hi != a.S) {
        const unsigned blocks = segment_grid(a.grid.window_patches(), a.seg_hi - a.seg_lo);
        if (walk2() & 2) hipLaunchKernelGGL((k_render_forward<false, true>), dim3(blocks), dim3(64), 0, s, a);
        else hipLaunchKernelGGL((k_render_forward<false, false>), dim3(blocks), dim3(64), 0, s, a);
        return;
    }
    const unsigned blocks = segment_grid(a.grid.window_patches(), ((a.S - a.seg_lo + G - 1) / G) | 1);
    if (walk2() & 2) hipLaunchKernelGGL((k_render_pass2_grouped<false, true>), dim3(blocks), dim3(64), 0, s, a, G);
    else hipLaunchKernelGGL((k_render_pass2_grouped<false, false>), dim3(blocks), dim3(64), 0, s, a, G);
}
// the first `head` segments of every list, walked once from the true transmittance (see k_render_pass2_grouped<true>)
void launch_render_head(const RenderFwdArgs& a, int head, hipStream_t s) {
    const unsigned blocks = segment_grid(a.grid.window_patches(), 1);
    if (walk2() & 2) hipLaunchKernelGGL((k_render_pass2_grouped<true, true>), dim3(blocks), dim3(64), 0, s, a, head);
    else hipLaunchKernelGGL((k_render_pass2_grouped<true, false>), dim3(blocks), dim3(64), 0, s, a, head);
}
void launch_render_combine(const RenderFwdArgs& a, hipStream_t s) {
    const unsigned patches = (unsigned)a.grid.window_patches();
    hipLaunchKernelGGL(k_render_combine, dim3(patches), dim3(64), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// Cross-lane exchanges for the reduce-scatter below.
__device__ __forceinline__ float xchg_xor1(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float xchg_xor2(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
}
template <int XORMASK>
__device__ __forceinline__ float xchg_swz(float x) {
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), (XORMASK << 10) | 0x1F));
}

// Reduce-scatter of 16 per-lane values over the 64 lanes of a wave.  On return lane L owns, in v[0], the wave-wide sum of value slot
//   id(L) = 8*bit5(L) + 4*bit4(L) + 2*bit0(L) + bit1(L)          (lanes that differ only in bits 2, 3 hold the same)
// The two wide steps come first, when there are most values to fold: a lane swap + an add each, instead of two selects + a
// cross-lane add.  lane ^ 1 and lane ^ 2 partners then come from DPP quad permutes, the sums over the row's quads from DPP row rotations.
__device__ __forceinline__ float reduce_scatter16(float (&v)[16], int lane) {
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = fold_halves32(v[k], v[k + 8]);
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = fold_halves16(v[k], v[k + 4]);
    {
        const bool hi = lane & 1;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const float keep = hi ? v[k + 2] : v[k], send = hi ? v[k] : v[k + 2];
            v[k] = keep + xchg_xor1(send);
        }
    }
    {
        const bool hi = lane & 2;
        const float keep = hi ? v[1] : v[0], send = hi ? v[0] : v[1];
        v[0] = keep + xchg_xor2(send);
    }
    // the four quads of a 16-lane row hold partial sums of the same slots: two row rotations (DPP, no LDS crossbar round trip on the
    // entry's dependency chain) leave every lane with the row total
    v[0] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[0]), 0x124, 0xF, 0xF, true));   // row_ror:4
    v[0] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[0]), 0x128, 0xF, 0xF, true));   // row_ror:8
    return v[0];
}

// V2 (round 4): the visited records of a chunk are parked compacted (the flagged entries in slots 0 .. cnt-1, in list order, zero
// records behind them; every array two slots further up, so that the look-ahead reads below slot 0 stay inside it) and walked from
// slot cnt-1 (or cnt, a zero record, when cnt is odd) down in pairs: a counted loop, the entry's position and Gaussian id riding in LDS
// -- no find-last-set / clear / compare / clamp on a 64-bit scalar mask per entry (~10 scalar instructions of the ~45 a visited entry
// that contributes nothing costs).  The arithmetic per entry is the first form's, in the same order.
template <bool V2>
#ifdef LG_BWD_WAVES      /* experiment (tools/waves_ab.sh): force the register budget of N waves per SIMD */
__attribute__((amdgpu_waves_per_eu(LG_BWD_WAVES, LG_BWD_WAVES)))
#endif
__global__ void __launch_bounds__(64) k_render_backward(const RenderBwdArgs a) {
    constexpr int OFS = V2 ? 2 : 0;                                    // slack slots below slot 0 (V2's look-ahead reads)
    __shared__ float4 s_rec_[4 * (LG_CHUNK + OFS)];
    __shared__ float4 s_oprow_[LG_CHUNK + OFS];                        // opacity per pixel row of the patch, 0 outside the entry's row span
    __shared__ uint32_t s_gid_[LG_CHUNK + OFS];
    constexpr int CS = LG_CHUNK + OFS;                                 // component stride of s_rec
    float4* const s_rec = s_rec_ + OFS;
    float4* const s_oprow = s_oprow_ + OFS;
    uint32_t* const s_gid = s_gid_ + OFS;
    const int lane = threadIdx.x;
    const int S = a.S;
    const int wpt = a.grid.waves_per_tile;
    int patch, seg;
    if (a.walk.cnt) {
        // item blockIdx / R of region blockIdx % R of the list k_render_combine filled (lidargs_common.h WorkList); the workgroups behind a
        // region's count -- the tail of the grid -- leave on one scalar load
        const unsigned r = blockIdx.x % LG_WORK_REGIONS, i = blockIdx.x / LG_WORK_REGIONS;
        if (i >= a.walk.cnt[r * LG_WORK_CNT_STRIDE]) return;
        const uint32_t it = a.walk.items[(size_t)r * a.walk.cap + i];
        patch = (int)(it >> 8); seg = (int)(it & 255u);
    } else {
        if (!block_patch_segment(blockIdx.x, a.grid.window_patches(), S, patch, seg)) return;
        patch = a.grid.global_patch(patch);
    }
    const int tile = patch / wpt, sub = patch - tile * wpt;
    const size_t stride = LG_SEG_PLANES * 64;
    // the three things every workgroup decides on are fetched together (the plane address is valid for any slot; what an unwalked
    // slot holds is never looked at): one round trip before the decision instead of three
    const float* sb = a.seg + (size_t)patch * S * stride + lane;      // this patch's segment planes, this lane
    const uint2 tr = a.ranges[tile];
    const int limit = a.alive ? (int)a.alive[patch] : 255;
    const uint32_t n_lane = reinterpret_cast<const uint32_t*>(sb)[(size_t)seg * stride + LG_SEG_LAST * 64];
    const int St = segment_count(tr, S, a.seg_len);
    if (seg >= St) return;
    if (seg >= limit) return;                                          // never walked by the forward: nothing blended there
    uint32_t n_max = n_lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_max = max(n_max, (uint32_t)__shfl_xor((int)n_max, o));
    if (n_max == 0) return;                                            // nothing blended in this segment

    const uint2 sr = segment_range(tr, St, seg);
    // The first chunk's list entries and flags are requested FIRST: its records hang on them (a second round trip), and everything
    // between here and the walk -- the pixel's rays, upstream gradients, T planes, the sums over the segments behind -- waits for
    // loads of its own that can be in flight at the same time.  (In program order the gather used to come after the plane sums'
    // loop: one or more round trips later.)
    const int c_last = (int)((n_max - 1) / LG_CHUNK);
    const uint8_t* fl = a.flags ? a.flags + (size_t)sub * a.R + sr.x : nullptr;
    uint32_t g_first = 0u; bool have_first = false;
    if (V2) {
        const uint32_t k = (uint32_t)c_last * LG_CHUNK + lane;
        const bool in = k < n_max;
        g_first = in ? a.point_list[sr.x + k] : 0u;
        have_first = in && (!fl || fl[k] != 0);
    }
    const PixelSetup px = pixel_setup(a.grid, a.coltab, a.rowtab, tile, sub, lane);
    const size_t N = (size_t)a.grid.W * a.grid.H;

    // per-pixel state of the back-to-front walk (R3/cr/backward.cu:590-615), restricted to this segment:
    // T starts at the segment's own end value; the "colour behind" recurrences are seeded with everything
    // composited behind the segment (later segments of this list + farther range shells), as seen from here.
    float T = sb[(size_t)seg * stride + LG_SEG_TEND * 64];
    const float T_final = px.inside ? (a.T_final_global ? a.T_final_global[px.pix] : a.final_T[px.pix]) : 0.f;
    float g0 = 0.f, g1 = 0.f, gd = 0.f, go = 0.f;
    if (px.inside) { g0 = a.dL_dpix[px.pix]; g1 = a.dL_dpix[N + px.pix]; gd = a.dL_ddepth[px.pix]; go = a.dL_docc[px.pix]; }
    const float bgdot = a.bg ? (a.bg[0] * g0 + a.bg[1] * g1) : 0.f;
    Staged st; uint32_t gid = 0u; bool have = false;
    if (V2) { have = have_first; gid = have ? g_first : 0u; st = gather_record(a.rec, a.rowspan, g_first, have); }   // behind the ids, in front of the plane sums
    // accum_rec[2] | accum_red, accum_reo: the four "colour behind" recurrences run as two packed pairs
    v2f acc01 = v2f{0.f, 0.f}, accdo = v2f{0.f, 0.f};
    const v2f g01 = v2f{g0, g1}, gdo = v2f{gd, go};
    {
        float b0 = 0.f, b1 = 0.f, bd = 0.f;
        const int Send = a.alive ? min(St, (int)a.alive[patch]) : St;  // nothing was walked behind the patch's limit
        {
            float acc[3] = {0.f, 0.f, 0.f};
            const int planes[3] = {LG_SEG_C0, LG_SEG_C1, LG_SEG_D};
            plane_sums<3>(acc, sb + (size_t)(seg + 1) * stride, stride, planes, Send - (seg + 1));
            b0 = acc[0]; b1 = acc[1]; bd = acc[2];
        }
        if (a.behind && px.inside) { b0 += a.behind[px.pix]; b1 += a.behind[N + px.pix]; bd += a.behind[2 * N + px.pix]; }
        if (T > 0.f) {
            const float inv = 1.f / T;
            acc01 = v2f{b0 * inv, b1 * inv};
            accdo = v2f{bd * inv, 1.f - T_final * inv};
        }
    }
    float last_alpha = 0.f;
    v2f lc01 = v2f{0.f, 0.f}, ldo = v2f{0.f, 1.f};                    // last entry's (colour0, colour1) | (range, 1)
    float* const gacc_slot = a.gacc + (8 * ((lane >> 5) & 1) + 4 * ((lane >> 4) & 1) + 2 * (lane & 1) + ((lane >> 1) & 1));   // this lane's slot of a Gaussian's packed line (reduce_scatter16)

    const int y0 = (tile / a.grid.tiles_x) * a.grid.TH + sub * LG_WAVE_ROWS;       // first pixel row of the patch
    const float* oprow = reinterpret_cast<const float*>(s_oprow) + (lane >> 4);
    const v2f qxy = v2f{px.q.x, px.q.y};
    auto gather = [&](int c, Staged& st, uint32_t& gid, bool& have) {
        const uint32_t k = (uint32_t)c * LG_CHUNK + lane;
        const bool in = k < n_max;
        const uint32_t g = in ? a.point_list[sr.x + k] : 0u;           // not waiting for the flag
        have = in && (!fl || fl[k] != 0);
        gid = have ? g : 0u;
        st = gather_record(a.rec, a.rowspan, g, have);
    };
    if (!V2) gather(c_last, st, gid, have);
    for (int c = c_last; c >= 0; c--) {
        __syncthreads();
        unsigned long long todo = __ballot(have);
        int cnt = 0;
        if (V2) {
            cnt = __builtin_popcountll(todo);
            const int before = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(todo >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)todo, 0u));
            const int slot = have ? before : cnt + (lane - before);
            s_rec[slot] = make_float4(st.a0.x, st.a0.y, st.a0.z, st.a3.x);
            s_rec[CS + slot] = st.a1;
            s_rec[2 * CS + slot] = st.a2;
            s_rec[3 * CS + slot] = make_float4(st.a0.w, __uint_as_float((uint32_t)c * LG_CHUNK + (uint32_t)lane), st.a3.z, st.a3.w);   // range, 0-based position, colours
            s_oprow[slot] = rows_opacity(st.span, st.a3.y, y0); s_gid[slot] = gid;
        } else {
            park_record(s_rec, lane, st);
            s_oprow[lane] = rows_opacity(st.span, st.a3.y, y0); s_gid[lane] = gid;
        }
        __syncthreads();
        if (c > 0) gather(c - 1, st, gid, have);
        if (todo == 0ull) continue;
        // back to front, software-pipelined like the forward walk (walk_flagged): two register sets, unconditional look-ahead reads
        struct Rec { float4 r0, r1, r2, r3; float op; uint32_t gid; };
        auto read = [&](int jj) {
            Rec r;
            r.gid = lds_ahead(&s_gid[jj]);                             // with the look-ahead reads: fetched where the atomic needs it, the owners waited out an LDS round trip per entry
            const float* f3 = reinterpret_cast<const float*>(&s_rec[3 * CS + jj]);
            const float4 s0 = lds_ahead(&s_rec[jj]);                   // s.xyz | B  (park_record)
            r.r1 = lds_ahead(&s_rec[CS + jj]); r.r2 = lds_ahead(&s_rec[2 * CS + jj]);
            const v2f col = *(LG_LDS_VOLATILE(v2f))(f3 + 2);
            r.r0 = make_float4(s0.x, s0.y, s0.z, lds_ahead(f3));       // range
            r.r3 = make_float4(s0.w, V2 ? lds_ahead(f3 + 1) : 0.f, col.x, col.y);   // B, (V2: position), colours
            r.op = lds_ahead(&oprow[4 * jj]);
            return r;
        };
        auto evaluate = [&](const Rec& r, int j) {
            const uint32_t e = V2 ? __float_as_uint(r.r3.y) : (uint32_t)c * LG_CHUNK + j;   // 0-based position inside the segment
            const v2f exy = v2f{r.r0.x, r.r0.y} - qxy;
            const float ex = exy.x, ey = exy.y, ez = r.r0.z - px.q.z;
            const v2f ux = v2f{r.r1.x, r.r1.y}, uy = v2f{r.r1.z, r.r1.w}, uz = v2f{r.r2.x, r.r2.y};        // (u1', u2') by component
            const v2f d = ex * ux + ey * uy + ez * uz;
            const float dx = d.x, dy = d.y;
            const float A = r.r2.z, B = r.r3.x, Cc = r.r2.w;
            const float op = r.op;                                    // the entry's opacity on this pixel's row, 0 outside its row span
            const float power = -0.5f * (A * dx * dx + Cc * dy * dy) - B * dx * dy;
            // :650 skip entries behind the last contributor; :673-679 the forward's skips.  As in the forward walk the tests are folded
            // into the exponent (exp(-inf) = 0 -> alpha 0) and the row test into `op`, so that `contrib` is one compare = the ballot.
            const float pw = ((e < n_lane) && (power <= 0.0f)) ? power : -INFINITY;
            const float G = LG_EXPF(pw);
            const float alpha_raw = fminf(0.99f, op * G);
            const bool contrib = alpha_raw >= 1.0f / 255.0f;
            LG_LANE_STAT(8, __ballot(contrib));
            if (__ballot(contrib) != 0ull) {                           // wave-uniform
                // A pixel that does not blend this entry treats it as an alpha = 0 entry: T / (1 - 0) = T, the "colour
                // behind" recurrences commit the previous entry (the same operation, just earlier) and then carry
                // (alpha 0, this colour), which the next step folds away exactly (0 * c + 1 * acc) -- bit-identical to
                // skipping it, and no per-pixel select on any of the nine state registers or the sixteen sums.
                const float alpha = contrib ? alpha_raw : 0.f;
                const float inv = __builtin_amdgcn_rcpf(1.f - alpha);  // 1 ulp; (1 - alpha) >= 0.01
                const float Tn = T * inv;                              // :681
                const float w = alpha * Tn;
                const float keep = 1.f - last_alpha;
                const v2f a01 = last_alpha * lc01 + keep * acc01;      // :694-714
                const v2f ado = last_alpha * ldo + keep * accdo;
                const v2f c01 = v2f{r.r3.z, r.r3.w}, cdo = v2f{r.r0.w, 1.f};
                const v2f t4 = (c01 - a01) * g01 + (cdo - ado) * gdo;
                float dL_dalpha = (t4.x + t4.y) * Tn;
                dL_dalpha -= T_final * inv * bgdot;                    // :727
                dL_dalpha = contrib ? dL_dalpha : 0.f;                 // every sum below carries dL_dalpha or w as a factor
                const float dL_dG = op * dL_dalpha;
                const v2f gd2 = G * d;                                 // (G dx, G dy)
                const float gdx = gd2.x, gdy = gd2.y;
                const v2f gxy = -dL_dG * (gd2 * v2f{A, Cc} + v2f{gdy, gdx} * B);   // dL/dmean2D  (:734,:753)
                const float gx = gxy.x, gy = gxy.y;
                // per-pixel sphere-gradient norm statistic (:759-779): |gx u1' + gy u2'|.  u1, u2 are orthonormal (u1 normalised, u2 = dir x u1;
                // u_i' = u_i / (u_i.u_i)), so the norm is sqrt(gx^2 + gy^2) up to their rounding (1e-7): three instructions instead of ten.
                // At a pole (u1 = 0, the only degenerate case) d = 0 and with it gx = gy = 0 in both forms.
                // dL/du1 = gx (delta/uu1 - 2 dx u1') per pixel (:738-750), with dx = delta . u1'.  Its sum over the pixels is
                //   |u1'|^2 G1 - 2 u1' (u1' . G1),   G1 = sum gx delta   (and the same with gy, u2' for dL/du2),
                // and u1', u2' are per-Gaussian: the pixels only accumulate the two moment vectors G1, G2 (three packed
                // multiplies instead of ten packed operations); k_gaussian_backward finishes the expression once per Gaussian

                float v[16];
                v[0] = gx;
                v[1] = gy;
                // hardware square root (1 ulp, one instruction; sqrtf's correctly rounded expansion is 17): this slot is the sum of
                // per-pixel norms that feeds the densification statistic, not a gradient, and a last-bit error per term is 1e-7 of it
                v[2] = __builtin_amdgcn_sqrtf(gx * gx + gy * gy);
                const float mh = -0.5f * dL_dG;
                const v2f cac = mh * gd2 * d;                          // conic A, C (:783)
                const v2f wc = w * g01;                                // colours (:702)
                v[3] = cac.x;
                v[4] = mh * gdx * dy;                                  // conic B
                v[5] = cac.y;
                v[6] = G * dL_dalpha;                                  // opacity (:788)
                v[7] = wc.x;
                v[8] = wc.y;
                v[9] = w * gd;                                         // range (:711)
                v[10] = gx * ex; v[11] = gx * ey; v[12] = gx * ez;     // G1 = sum gx delta  (-> dL/du1)   (six plain multiplies: the packed
                v[13] = gy * ex; v[14] = gy * ey; v[15] = gy * ez;     // G2 = sum gy delta  (-> dL/du2)    form cost five moves to pair delta up)
                T = Tn;
                acc01 = a01; accdo = ado;
                lc01 = c01; ldo = cdo;
                last_alpha = alpha;
                // (the slot's address before the reduction, in every lane: independent work for the wait states between its dependent DPP steps)
                float* const dst = gacc_slot + 16 * (size_t)r.gid;
                const float mine = reduce_scatter16(v, lane);
                if ((lane & 12) == 0) atomicAdd(dst, mine);             // one owner per slot
            }
        };
        if (V2) {
            // pairs from the top slot down; an odd count starts one slot higher, on a zero record (alpha = 0: nothing to do)
            int j = ((cnt + 1) & ~1) - 1;
            Rec ra = read(j), rb;
            for (; j > 0; j -= 2) {
                rb = read(j - 1);
                evaluate(ra, j);
                ra = read(j - 2);                                      // (below slot 0 on the last trip: the slack slots)
                evaluate(rb, j - 1);
            }
            continue;
        }
        auto top = [&](unsigned long long& m) { const int jj = 63 - __builtin_clzll(m); m &= ~(1ull << jj); return jj; };
        int ja = top(todo);
        Rec ra = read(ja), rb;
        while (true) {
            const bool more_b = todo != 0ull;
            int jb = ja;
            if (more_b) jb = top(todo);
            rb = read(jb);
            evaluate(ra, ja);
            if (!more_b) break;
            const bool more_a = todo != 0ull;
            ja = jb;
            if (more_a) ja = top(todo);
            ra = read(ja);
            evaluate(rb, jb);
            if (!more_a) break;
        }
    }
}

// Diagnostics (lidargs_counters_enable + lidargs_last_counters, outside any timed region): the list entries k_render_backward gathers for a frame -- per (patch,
// segment) the flagged entries in front of the segment's last blended one, exactly its own selection (above: n_max, the flags, the
// patch's limit).  What bench.py prices the launch's algorithmic bytes on.
__global__ void __launch_bounds__(64) k_count_backward_entries(const RenderBwdArgs a, unsigned long long* __restrict__ out) {
    const int lane = threadIdx.x;
    const int S = a.S, wpt = a.grid.waves_per_tile;
    int patch, seg;
    if (!block_patch_segment(blockIdx.x, a.grid.window_patches(), S, patch, seg)) return;
    patch = a.grid.global_patch(patch);
    const int tile = patch / wpt, sub = patch - tile * wpt;
    const size_t stride = LG_SEG_PLANES * 64;
    const uint2 tr = a.ranges[tile];
    const int St = segment_count(tr, S, a.seg_len);
    const int limit = a.alive ? (int)a.alive[patch] : 255;
    if (seg >= St || seg >= limit) return;
    uint32_t n_max = reinterpret_cast<const uint32_t*>(a.seg + (size_t)patch * S * stride + lane)[(size_t)seg * stride + LG_SEG_LAST * 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_max = max(n_max, (uint32_t)__shfl_xor((int)n_max, o));
    if (n_max == 0) return;
    const uint2 sr = segment_range(tr, St, seg);
    const uint8_t* fl = a.flags ? a.flags + (size_t)sub * a.R + sr.x : nullptr;
    uint32_t c = 0;
    for (uint32_t k = lane; k < n_max; k += 64) c += (!fl || fl[k] != 0) ? 1u : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (lane == 0 && c) atomicAdd(out, (unsigned long long)c);
}
void launch_count_backward_entries(const RenderBwdArgs& a, unsigned long long* out, hipStream_t s) {
    hipLaunchKernelGGL(k_count_backward_entries, dim3(segment_grid(a.grid.window_patches(), a.S)), dim3(64), 0, s, a, out);
}

void launch_render_backward(const RenderBwdArgs& a, hipStream_t s) {
    const unsigned blocks = a.walk.cnt ? (unsigned)LG_WORK_REGIONS * a.walk.cap : segment_grid(a.grid.window_patches(), a.S);
    if (walk2() & 4) hipLaunchKernelGGL(k_render_backward<true>, dim3(blocks), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(k_render_backward<false>, dim3(blocks), dim3(64), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// Multi-GPU glue (lidargs_dist): per-pixel folds over the G range shells, one launch each instead of a dozen
// elementwise framework ops on a 0.2 ms critical path.
__global__ void __launch_bounds__(256) k_shell_transmittance(int G, int rank, int N, size_t row_stride, const float* __restrict__ all_T, float* __restrict__ T_in) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float T = 1.f;
    for (int g = 0; g < rank && g < G; g++) T *= all_T[(size_t)g * row_stride + i];
    T_in[i] = T;
}

// planes[g] = (C0, C1, D, T_end, T_hand) of shell g.  The walk stopped in the first shell whose hand-over value fell
// below the reference's 1e-4 threshold; T_final is that shell's T_end (the last shell's if none stopped).
__global__ void __launch_bounds__(256) k_shell_compose(int G, int rank, int N, const float* __restrict__ planes, const float* __restrict__ bg,
                                                       float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ out_occ,
                                                       float* __restrict__ T_final, float* __restrict__ behind) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float c0 = 0.f, c1 = 0.f, d = 0.f, b0 = 0.f, b1 = 0.f, bd = 0.f, Tf = 1.f;
    bool stopped = false;
    for (int g = 0; g < G; g++) {
        const float* p = planes + (size_t)g * 5 * N + i;
        const float pc0 = p[0], pc1 = p[(size_t)N], pd = p[2 * (size_t)N];
        c0 += pc0; c1 += pc1; d += pd;
        if (g > rank) { b0 += pc0; b1 += pc1; bd += pd; }
        if (!stopped) { Tf = p[3 * (size_t)N]; stopped = p[4 * (size_t)N] < 0.0001f; }
    }
    const float g0 = bg ? bg[0] : 0.f, g1 = bg ? bg[1] : 0.f;
    out_color[i] = c0 + Tf * g0; out_color[(size_t)N + i] = c1 + Tf * g1;
    out_depth[i] = d; out_occ[i] = 1.f - Tf; T_final[i] = Tf;
    behind[i] = b0; behind[(size_t)N + i] = b1; behind[2 * (size_t)N + i] = bd;
}

// Column wedges: a rank's own pixel columns [c0, c1) of the four image planes (colour 0/1, depth, occupancy) as one dense
// [4][H][wmax] block (what the image all-gather ships; columns >= c1 - c0 are padding), and back: G such blocks -> full planes.
__global__ void __launch_bounds__(256) k_wedge_pack_columns(int H, int W, int c0, int c1, int wmax, const float* __restrict__ color,
                                                            const float* __restrict__ depth, const float* __restrict__ occ, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = 4 * H * wmax;
    if (i >= n) return;
    const int x = i % wmax, y = (i / wmax) % H, pl = i / (wmax * H);
    const int col = c0 + x;
    float v = 0.f;
    if (col < c1) {
        const size_t pix = (size_t)y * W + col;
        v = pl < 2 ? color[(size_t)pl * H * W + pix] : (pl == 2 ? depth[pix] : occ[pix]);
    }
    out[i] = v;
}
struct WedgeEdges { int e[65]; };
__global__ void __launch_bounds__(256) k_wedge_unpack_columns(int G, int H, int W, int wmax, size_t stride, WedgeEdges ed, const float* __restrict__ blocks,
                                                              float* __restrict__ color, float* __restrict__ depth, float* __restrict__ occ) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)4 * H * W;
    if (i >= n) return;
    const int col = (int)(i % W), y = (int)((i / W) % H), pl = (int)(i / ((size_t)W * H));
    int g = 0;
    while (g + 1 < G && col >= ed.e[g + 1]) g++;
    const float v = blocks[(size_t)g * stride + ((size_t)pl * H + y) * wmax + (col - ed.e[g])];
    const size_t pix = (size_t)y * W + col;
    if (pl < 2) color[(size_t)pl * H * W + pix] = v;
    else if (pl == 2) depth[pix] = v;
    else occ[pix] = v;
}
void launch_wedge_pack_columns(int H, int W, int c0, int c1, int wmax, const float* color, const float* depth, const float* occ, float* out, hipStream_t s) {
    const int n = 4 * H * wmax;
    hipLaunchKernelGGL(k_wedge_pack_columns, dim3((n + 255) / 256), dim3(256), 0, s, H, W, c0, c1, wmax, color, depth, occ, out);
}
void launch_wedge_unpack_columns(int G, int H, int W, int wmax, size_t stride, const int* edges, const float* blocks, float* color, float* depth,
                                 float* occ, hipStream_t s) {
    WedgeEdges ed;
    for (int g = 0; g <= G && g < 65; g++) ed.e[g] = edges[g];
    const size_t n = (size_t)4 * H * W;
    hipLaunchKernelGGL(k_wedge_unpack_columns, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, G, H, W, wmax, stride, ed, blocks, color, depth, occ);
}

void launch_shell_transmittance(int G, int rank, int N, size_t row_stride, const float* all_T, float* T_in, hipStream_t s) {
    hipLaunchKernelGGL(k_shell_transmittance, dim3((N + 255) / 256), dim3(256), 0, s, G, rank, N, row_stride, all_T, T_in);
}
void launch_shell_compose(int G, int rank, int N, const float* planes, const float* bg, float* out_color, float* out_depth, float* out_occ,
                          float* T_final, float* behind, hipStream_t s) {
    hipLaunchKernelGGL(k_shell_compose, dim3((N + 255) / 256), dim3(256), 0, s, G, rank, N, planes, bg, out_color, out_depth, out_occ, T_final, behind);
}

}  // namespace lg
