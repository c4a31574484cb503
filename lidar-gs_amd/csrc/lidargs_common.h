// lidargs_common.h -- shared declarations of the gfx950 LiDAR Gaussian rasterizer.
//
// Data layout in HBM (all private to this library; the reference's counterpart is the
// GeometryState / BinningState / ImageState carving of R3/cr/rasterizer_impl.h:21-76):
//
//   geometry buffer  (sized by P)
//     rec      float4[4P]  one 64-byte "splat record" per Gaussian, written once by preprocess and
//                          gathered as one aligned 64-B segment per (tile, Gaussian) instance:
//                            rec[0] = (s.x, s.y, s.z, range)       s = p_view/|p_view|
//                            rec[1] = (u1'.x, u2'.x, u1'.y, u2'.y)     u_i' = u_i/(u_i.u_i); the two tangent directions are
//                            rec[2] = (u1'.z, u2'.z, A, C)             interleaved by component and the conic's diagonal is a
//                            rec[3] = (B, opacity, colour0, colour1)   pair: both projections run as packed-fp32 operations
//     rowspan  u32[P]      ymin | ymax<<16 of the (pruned) pixel-row rect (R3/cr/auxiliary.h:80-92): read per list entry by the blend
//     spans    u32[P] (in a u32x4[P] area)   the pruned rect as ONE word, x0 | (nx - 1) << 8 | first row << 16 | last row << 24
//                           (0xFFFFFFFF = no instances), while the image has <= 256 tile columns and <= 256 rows; beyond, u32x4
//                           (rowspan, xspan = xmin | xmax<<16 in 16-pixel tile columns (0 = no instances), -, -): ONE gather per
//                           Gaussian when the lists are built (span_pack / compact_spans below)
//     span_sorted u32[P] | u32x2[P]  the same in range order (compact | (xspan, rowspan)): the instance emit reads no per-Gaussian
//                           array at random
//     key_a    u32[P]      float bits of the range (sort key), 0xFFFFFFFF when culled
//     sort ping/pong, sorted ids, per-sorted-Gaussian instance offsets, scan/sort scratch
//   binning buffer   (sized by R = #instances)
//     tile keys ping/pong u32[R], Gaussian ids ping/pong u32[R], sort scratch
//     -> point_list u32[R]: Gaussian ids sorted by (tile, range, id)
//   image buffer     (sized by W*H)
//     final_T f32[N], T_pass f32[N], ranges uint2[tiles], per-row / per-column ray tables
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define LG_TILE_W 16          // = the reference's BLOCK_X (R3/cr/config.h:16); rect x-units are shared
#define LG_WAVE_ROWS 4        // one wave64 = 16 columns x 4 rows of pixels
#define LG_CHANNELS 2

// per-(patch, segment) planes of 64 floats written by the forward blend (render.hip)
#define LG_SEG_TPASS 0    // pass 1: transmittance handed to the next segment (< 1e-4 if the walk tripped)
#define LG_SEG_C0 1       // pass 2: partial sums, already weighted by the global transmittance
#define LG_SEG_C1 2
#define LG_SEG_D 3
#define LG_SEG_TEND 4     // T after the segment's last blended entry
#define LG_SEG_TBREAK 5   // = TEND, or the value that tripped T < 1e-4 inside the segment
#define LG_SEG_LAST 6     // u32: entries of the segment consumed up to the last blended one
#define LG_SEG_PLANES 7

#define LG_REGION 256     // Gaussians per region of the backward's touched lists (k_zero_touched, k_gaussian_backward)

namespace lg {

struct Carver {
    char* p;
    explicit Carver(char* base) : p(base) {}
    template <typename T> T* take(size_t n) {
        uintptr_t a = (reinterpret_cast<uintptr_t>(p) + 127) & ~uintptr_t(127);
        T* r = reinterpret_cast<T*>(a);
        p = reinterpret_cast<char*>(a) + n * sizeof(T);
        return r;
    }
};

// ---- radix sort / scan scratch sizing (binning.hip) --------------------------------------------
constexpr int SORT_ITEMS = 16;                  // rounds of 64 keys per wave
constexpr int SORT_WAVES = 4;                   // waves per sort block
constexpr int SORT_CHUNK = 64 * SORT_ITEMS * SORT_WAVES;   // keys per block (4096)
constexpr int SORT_RADIX_BITS = 8;               // default digit width (tile binning: short digit runs must stay whole lines)
constexpr int SORT_MAX_RADIX_BITS = 11;          // the range sort of the P Gaussians: 31 key bits in 3 passes (11 + 10 + 10)
constexpr int SORT_MAX_BINS = 1 << SORT_MAX_RADIX_BITS;
constexpr int SCAN_BLOCK = 1024;                // elements per scan block (256 threads x 4)

inline size_t sort_blocks(size_t n) { return (n + SORT_CHUNK - 1) / SORT_CHUNK; }
inline size_t scan_blocks(size_t n) { return (n + SCAN_BLOCK - 1) / SCAN_BLOCK; }
// u32 words of scratch needed to sort n pairs: digit histogram [BINS x blocks] + per-digit totals
constexpr unsigned SORT_PREFIX_CHUNK = 1024;     // blocks per wave in the two-level cross-block prefix (used above 2 chunks)
inline size_t sort_scratch_words(size_t n, int max_bits = SORT_RADIX_BITS) {
    const size_t bins = (size_t)1 << max_bits;
    size_t h = bins * sort_blocks(n);
    size_t chunks = (sort_blocks(n) + SORT_PREFIX_CHUNK - 1) / SORT_PREFIX_CHUNK;
    return h + bins + bins * chunks + 64;
}
inline size_t scan_scratch_words(size_t n) { return scan_blocks(n) + 64; }

// Per-Gaussian span record the instance offsets and the emit are built from.  Full form: u32x4 (rowspan = lo | hi << 16 in pixel
// rows, xspan = x0 | x1 << 16 in 16-pixel tile columns, 0 = no instances, -, -).  COMPACT form, whenever the image has at most 256
// tile columns and 256 pixel rows (64x2650 and 128x4096 do): ONE u32 = x0 | (nx - 1) << 8 | lo << 16 | (hi - 1) << 24, 0xFFFFFFFF =
// no instances.  The spans are gathered at random in range order (by the range sort's last pass): with 16-byte records every 128-byte line of a
// 32-MB table is fetched for one record (measured 243 MB of fabric traffic for 2 M Gaussians); the 8-MB table of compact records
// stays in the L2s / MALL.
inline bool compact_spans(int tiles_x, int H) { return tiles_x <= 256 && H <= 256; }
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t span_pack(uint32_t rs, uint32_t xs) {
    if (!xs) return 0xFFFFFFFFu;
    const uint32_t x0 = xs & 0xFFFFu, x1 = xs >> 16, lo = rs & 0xFFFFu, hi = rs >> 16;
    return x0 | ((x1 - x0 - 1u) << 8) | (lo << 16) | ((hi - 1u) << 24);
}
__device__ __forceinline__ uint2 span_unpack(uint32_t w) {            // -> (xspan, rowspan)
    if (w == 0xFFFFFFFFu) return make_uint2(0u, 0u);
    const uint32_t x0 = w & 255u, nx = ((w >> 8) & 255u) + 1u, lo = (w >> 16) & 255u, hi = (w >> 24) + 1u;
    return make_uint2(x0 | ((x0 + nx) << 16), lo | (hi << 16));
}
#endif

// totals: word 0 = instance total of the scan; from word LG_TOTALS_SLOT_WORD on, LG_INST_SLOTS slots of four 64-bit sums (the
// instance counts for tile heights 4 / 8 / 16 / 32), one 32-byte slot per group of preprocess blocks
#define LG_INST_SLOTS 64
#define LG_TOTALS_SLOT_WORD 8
// (word 4 of the totals was rounds 2-4's "gradient lines dirty" mark: every backward now clears the touched Gaussians' lines itself)
// behind the status words: LG_INST_SLOTS slots of (~smallest, largest) range key of the frame's visible Gaussians (atomicMax of the
// preprocess blocks, spread over the slots like the instance totals; all start at 0, so an empty frame reads kmin = 0xFFFFFFFF,
// kmax = 0).  The host folds them after its one read: the range sort then works on key - kmin and needs only as many passes as the
// span has bits.
// behind the slots: the 16 status words of an enqueue-only forward (binning.hip k_finish_totals)
#define LG_TOTALS_STATUS_WORD (LG_TOTALS_SLOT_WORD + 8 * LG_INST_SLOTS)
#define LG_STATUS_WORDS 16
#define LG_TOTALS_KEYSPAN_WORD (LG_TOTALS_STATUS_WORD + LG_STATUS_WORDS)
#define LG_TOTALS_READ_WORDS (LG_TOTALS_KEYSPAN_WORD + 2 * LG_INST_SLOTS)   // what the forward's host read copies
// behind that: LG_INST_SLOTS diagnostic slots of two 64-bit sums (visible Gaussians, reference 16x1 tiles_touched): read only by
// lidargs_last_counters
#define LG_TOTALS_DIAG_WORD LG_TOTALS_READ_WORDS
#define LG_TOTALS_WORDS (LG_TOTALS_DIAG_WORD + 4 * LG_INST_SLOTS)
struct GeomView {
    float4* rec;
    uint32_t* rowspan;
    uint4* spans;                       // (rowspan, xspan, instances at 4-row tiles, reference tiles_touched)
    uint2* span_sorted;                 // (xspan, rowspan) of the i-th Gaussian in range order
    uint32_t* key_a; uint32_t* key_b;   // range keys ping/pong
    uint32_t* id_a; uint32_t* id_b;     // Gaussian ids ping/pong (id_sorted ends in id_a)
    uint32_t* block_off;                // [scan_blocks(P)] exclusive instance offset of each block of SCAN_BLOCK range-consecutive Gaussians
    uint32_t* totals;                   // [0]=#instances of the scan, [8..] instance-total slots
    float* gacc;                        // [16P] packed per-Gaussian gradient accumulators (backward)
    uint8_t* touched;                   // [P] 1 = some pixel's walk took the Gaussian (the forward's contribution flags, a superset of what the
                                        //     backward blends): only these have a gradient, and only their gacc lines are ever zeroed, added to or read
    uint8_t* tlist; uint16_t* tcount;   // the backward's lists of them: per region of LG_REGION consecutive Gaussians, the touched ones' offsets
                                        //     in the region (tlist[region * LG_REGION + j], j < tcount[region]); written by k_zero_touched
    uint32_t* scratch;                  // sort + scan scratch
    size_t scratch_words;
};

inline size_t geom_carve(char* base, size_t P, GeomView* v) {
    Carver c(base);
    GeomView g;
    g.rec = c.take<float4>(4 * P);
    g.rowspan = c.take<uint32_t>(P);
    g.spans = c.take<uint4>(P);
    g.span_sorted = c.take<uint2>(P);
    g.key_a = c.take<uint32_t>(P); g.key_b = c.take<uint32_t>(P);
    g.id_a = c.take<uint32_t>(P); g.id_b = c.take<uint32_t>(P);
    g.block_off = c.take<uint32_t>(scan_blocks(P) + 64);
    g.totals = c.take<uint32_t>(LG_TOTALS_WORDS);
    g.gacc = c.take<float>(16 * P);
    g.touched = c.take<uint8_t>(P + 64);
    g.tlist = c.take<uint8_t>(P + LG_REGION); g.tcount = c.take<uint16_t>(P / LG_REGION + 64);
    g.scratch_words = sort_scratch_words(P, SORT_MAX_RADIX_BITS) + scan_scratch_words(P);
    g.scratch = c.take<uint32_t>(g.scratch_words);
    if (v) *v = g;
    return (size_t)(c.p - base) + 128;
}

struct BinView {
    uint32_t* tile_a; uint32_t* tile_b;
    uint32_t* val_a; uint32_t* val_b;
    uint32_t* scratch;
    size_t scratch_words;
    float* seg;                          // [patches][S][LG_SEG_PLANES][64]
    uint8_t* flags;                      // [waves_per_tile][R]: pass 1 saw >= 1 pixel of the patch take this entry
    uint8_t* alive;                      // [patches]: number of list segments pass 1 walked (255 = all)
    uint32_t* work;                      // the backward's work list (see WorkList): counters, then items [LG_WORK_REGIONS][cap]
    uint32_t work_cap;                   // items per region
};

// ---- the reference's tile rect of one Gaussian, in 16x1 tiles ------------------------------------------------------------------
// R3/cr/auxiliary.h:80-92 getRect_lidar (3-D variant): x truncates, y rounds half away from zero.  Every operation in the order the
// expression there is written: `p.x + rx + BLOCK_X - 1` is ((p.x + rx) + 16) - 1 in fp32, TWO roundings -- for p.x + rx an ulp
// under 17, 49, 113, ... the first one ties up to the next integer (16.999998 + 16 -> 33) and the rect reaches one tile further than
// with + 15.f in one step (round 3).  tests/ hold it bit for bit on adversarial inputs through lidargs_debug_rects.
__device__ __forceinline__ void rect_lidar(float p_c, float p_r, int rx, int ry, int gx, int gy, int& xmin, int& ymin, int& xmax, int& ymax) {
    xmin = min(gx, max(0, (int)((p_c - (float)rx) / 16.f)));
    xmax = min(gx, max(0, (int)((((p_c + (float)rx) + 16.f) - 1.f) / 16.f)));
    ymin = min(gy, max(0, (int)roundf(p_r - (float)ry)));
    ymax = min(gy, max(0, (int)fmaxf(roundf(p_r + (float)ry), roundf(p_r) + 1.f)));
}
// R2/cr/auxiliary.h:99-112 (surfel variant): x and ymin truncate, ymax = round(p.y + ry)
__device__ __forceinline__ void rect_surfel(float p_c, float p_r, int rx, int ry, int gx, int gy, int& xmin, int& ymin, int& xmax, int& ymax) {
    xmin = min(gx, max(0, (int)((p_c - (float)rx) / 16.f)));
    xmax = min(gx, max(0, (int)((((p_c + (float)rx) + 16.f) - 1.f) / 16.f)));
    ymin = min(gy, max(0, (int)(p_r - (float)ry)));
    ymax = min(gy, max(0, (int)roundf(p_r + (float)ry)));
}

// Work list of the backward blend.  A blend launch over (patch, segment) slots carries patches x S single-wave workgroups of which,
// on the BASELINE frames, five in six (cfg4: 49 in 50) have nothing to walk (behind the list's end or the patch's saturation point): each
// costs a dispatch (~0.22 ns, tools/micro/empty_wg.hip) and a wave slot for the round trip of the three loads it decides on.  The
// backward knows its live slots beforehand -- k_render_combine sees, per patch, how many segments some pixel walked through -- so it walks
// a LIST of them: item b / R of region b % R, the workgroups behind a region's count leaving on one scalar load at the END of the grid,
// where their dispatch hides behind the live ones.  The combine appends a patch's slots with ONE atomic on the counter of region
// patch % LG_WORK_REGIONS; the counters sit 128 bytes apart (same-line atomics serialise at ~11 ns each: 64 counters in two lines cost
// cfg4's combine 22 us).  A region's items are in patch order up to the arrival order of the atomics: which workgroup walks which slot
// changes nothing in what is computed.  (Lists for the forward launches -- the tail of pass 1, pass 2 -- were built and measured too:
// no gain, those launches are as long as their longest segment walk)
#define LG_WORK_REGIONS 64
#define LG_WORK_CNT_STRIDE 32            // words between two regions' counters (one 128-byte line each)
struct WorkList {
    uint32_t* cnt;                       // [LG_WORK_REGIONS] at stride LG_WORK_CNT_STRIDE
    uint32_t* items;                     // [LG_WORK_REGIONS][cap]: (patch << 8) | segment
    uint32_t cap;
};
inline uint32_t work_cap(size_t patches, int S) { return (uint32_t)(((patches + LG_WORK_REGIONS - 1) / LG_WORK_REGIONS) * (size_t)S); }
inline size_t work_words(size_t patches, int S) { return (size_t)LG_WORK_REGIONS * (LG_WORK_CNT_STRIDE + (size_t)work_cap(patches, S)); }
inline bool work_lists_fit(size_t patches, int S) { return S <= 255 && patches < ((size_t)1 << 24); }

// List segments: the launch provides `max_segments` workgroups per patch; each tile uses ceil(len / seg_len) of them
// (render.hip segment_count), so long lists of a skewed frame (far range shells, street canyons) are split as finely as
// the short ones are left alone.  The count is a pure function of the tile's range: backward recomputes it.
#define LG_SEG_LEN_DEFAULT 128
inline int choose_segments(size_t R, int max_segments) {
    if (R == 0) return 1;
    return max_segments < 1 ? 1 : max_segments;
}

inline size_t bin_carve(char* base, size_t R, size_t patches, int waves_per_tile, int S, BinView* v) {
    Carver c(base);
    BinView b;
    size_t n = R ? R : 1;
    b.tile_a = c.take<uint32_t>(n); b.tile_b = c.take<uint32_t>(n);
    b.val_a = c.take<uint32_t>(n); b.val_b = c.take<uint32_t>(n);
    b.scratch_words = sort_scratch_words(n);
    b.scratch = c.take<uint32_t>(b.scratch_words);
    b.seg = c.take<float>(patches * (size_t)S * LG_SEG_PLANES * 64);
    b.flags = c.take<uint8_t>((size_t)waves_per_tile * n + 64);
    b.alive = c.take<uint8_t>(patches + 64);
    b.work_cap = work_cap(patches, S);
    b.work = c.take<uint32_t>(work_words(patches, S));
    if (v) *v = b;
    return (size_t)(c.p - base) + 128;
}

inline WorkList work_list(const BinView& b) {
    WorkList w;
    w.cnt = b.work;
    w.items = b.work + (size_t)LG_WORK_REGIONS * LG_WORK_CNT_STRIDE;
    w.cap = b.work_cap;
    return w;
}

struct ImgView {
    float* final_T;       // T at the end of this call's list (after early-out)
    float* T_pass;        // transmittance handed to the next range shell (multi-GPU only)
    uint2* ranges;        // [tiles]
    float2* coltab;       // [W]  (cos beta, sin beta)
    float2* rowtab;       // [H]  (cos alpha, sin alpha) of pixel row y
};

inline size_t img_carve(char* base, int W, int H, int tiles, ImgView* v) {
    Carver c(base);
    ImgView m;
    size_t N = (size_t)W * H;
    m.final_T = c.take<float>(N);
    m.T_pass = c.take<float>(N);
    m.ranges = c.take<uint2>(tiles);
    m.coltab = c.take<float2>(W);
    m.rowtab = c.take<float2>(H);
    if (v) *v = m;
    return (size_t)(c.p - base) + 128;
}

// Tile grid: 16 columns x TH rows (TH multiple of 4); each tile is rendered by TH/4 waves.
struct TileGrid {
    int W, H, TH;
    int tiles_x, tiles_y;   // list tiles
    int ref_tiles_x;        // == tiles_x (16-wide), reference grid.x
    int waves_per_tile;
    int x_lo, x_n;          // tile-column window the blend launches cover (a column wedge's own tiles); 0, tiles_x otherwise
    __host__ __device__ int num_tiles() const { return tiles_x * tiles_y; }
    // the blend kernels are launched over the window's patches only; this maps a launch-local patch index to the global one
    // (tile = ty * tiles_x + tx, patch = tile * waves_per_tile + sub), which is what ranges / alive / segment planes are indexed by
    __host__ __device__ int window_patches() const { return x_n * tiles_y * waves_per_tile; }
    __host__ __device__ int global_patch(int lp) const {
        if (x_n == tiles_x) return lp;                     // the whole image (wave-uniform): no index arithmetic on the common path
        const int lt = lp / waves_per_tile, sub = lp - lt * waves_per_tile;
        const int ty = lt / x_n, tx = x_lo + (lt - ty * x_n);
        return (ty * tiles_x + tx) * waves_per_tile + sub;
    }
};

inline TileGrid make_grid(int W, int H, int TH) {
    TileGrid g;
    g.W = W; g.H = H; g.TH = TH;
    g.tiles_x = (W + LG_TILE_W - 1) / LG_TILE_W;
    g.tiles_y = (H + TH - 1) / TH;
    g.ref_tiles_x = g.tiles_x;
    g.waves_per_tile = TH / LG_WAVE_ROWS;
    g.x_lo = 0; g.x_n = g.tiles_x;
    return g;
}

struct PreprocessParams {
    int P, W, H, TH, tiles_x, tiles_y;
    float scale_modifier;
    float near_f, far_f;        // reference int near/far converted to float (R3/cr/forward.cu:304)
    float shell_lo, shell_hi;   // extra float range shell: keep lo <= range < hi (multi-GPU); +-inf otherwise
    int tile_x_lo, tile_x_hi;   // tile-column window [lo, hi) this call bins and renders (multi-GPU column wedges); 0, tiles_x otherwise
    int compact;                // 4-byte span records (compact_spans(tiles_x, H))
    int prune;                  // conservative footprint pruning on (default); 0 = bin the whole reference rect (LIDARGS_NO_PRUNE=1: the
                                // results must not depend on it -- tests/test_beam_tables_gpu.py, tools/diag_prune.py)
    float col_step;             // 2*pi/W        (float, as the reference evaluates it)
    float inv_col_step;         // a bound from above on 1 / col_step (footprint pruning only)
    float tan_col_step;         // tanf(2*pi/W)  (host libm)
    const float* view;          // DEVICE pointer to the 16 floats (wave-uniform -> scalar loads)
    const uint32_t* n_valid = nullptr;   // DEVICE word, or NULL: rows [*n_valid, P) are padding of a capacity-sized selection (multi-GPU,
                                         // enqueue-only rank frames) and are culled before any of their attributes is read
};

#ifdef LG_LANE_STATS
void lane_stats_read(unsigned long long* out, int reset);              // render.hip (instrumented builds only: tools/lane_stats.py)
#endif
// helpers exported by api.hip for the other entry-point files
int api_fail(int code, const char* msg);
int api_check_launch(hipStream_t s, int debug, const char* what);
int api_tile_rows();
bool api_prune_footprints();
int api_ceil_log2(uint32_t n);
int api_range_sort_bits();
struct SegPlan { int seg_len, max_segments, n_rounds, rounds[8]; int head; int fused; };   // api.hip plan_segments (head: walk round 1
                                                                                          // completely; fused: one workgroup per patch does it all)
SegPlan api_plan_segments(size_t R, int waves_per_tile, int surfel);
// Device -> host read of `n` (<= 1024) words with `zero_bytes` at `zero` cleared BEHIND the copy on the same stream: the host waits
// for the copy only.  Returns a hipError_t.
int api_read_words_zero_behind(const uint32_t* dev, int n, uint32_t* out, void* zero, size_t zero_bytes, hipStream_t s);
int api_read_words_begin(const uint32_t* dev, int n, hipStream_t s);    // the same read in two halves: queue the copy ...
int api_read_words_end(int n, uint32_t* out);                           // ... and, with more work queued behind it, wait for it
// num_rendered = instance capacity (multiple of 4) | tile-height code: all a later call on the forward's buffers needs (api.hip)
// per-stage HIP-event timing (lidargs_profile_*): kind 0 = forward-like call, 1 = backward
void api_prof_begin(hipStream_t s, int kind);
void api_prof_mark(const char* name, hipStream_t s);
void api_note_forward(long long P, long long R, int TH, int tiles, int S, const void* spans, const uint8_t* flags, size_t flags_stride,
                      int flags_planes, const uint8_t* touched, hipStream_t s);   // host-side counters; with lidargs_counters_enable(1) also queues the counting launches
int api_encode_rendered(size_t R, int TH);
size_t api_rendered_capacity(int num_rendered);
int api_rendered_tile_rows(int num_rendered);
// First launch of a backward (preprocess.hip k_zero_touched): clears the packed gradient line (`line_f4` float4: 4 = 64 bytes, the 3-D
// variant; 8 = 128 bytes, the surfel variant) of every Gaussian the forward marked as touched -- nobody reads or adds to the others' --,
// lists the touched Gaussians per region (tlist / tcount of the geometry view) and zeroes every row of the caller's gradient arrays.
struct ZeroRows { static constexpr int MAX = 12; float* p[MAX]; int w[MAX]; int n = 0;   // arrays of P rows of w floats (w = 1, 2, 3, 4, 6 or 9)
                  void add(float* q, int width) { if (q && n < MAX) { p[n] = q; w[n] = width; n++; } } };
void launch_zero_touched(const uint8_t* touched, float4* acc, int line_f4, size_t P, uint8_t* tlist, uint16_t* tcount, const ZeroRows& zr, hipStream_t s);
// touched[i] = 1 for all P (frames whose forward writes no contribution flags: every listed entry is walked, so every visible Gaussian counts)
void launch_touch_all(uint8_t* touched, const int* radii, size_t P, hipStream_t s);   // marks the Gaussians with radii > 0

// kernels / launchers (defined in the .hip files)
void launch_setup_tables(const float* beams, int W, int H, ImgView img, hipStream_t s);
void launch_preprocess(const PreprocessParams& pp, const float* means3D, const float* scales, const float* rotations,
                       const float* opacities, const float* colors, const float* cov3D_precomp, const float* beams,
                       int* radii, int* radii_xy, GeomView g, const ImgView* tables, bool filter_only, hipStream_t s);
void launch_mark_visible(int P, const float* means3D, const float* view, unsigned char* present, hipStream_t s);
void launch_debug_rects(int n, int surfel, const float* p_cr, const int* r_xy, int gx, int gy, int* rects, hipStream_t s);   // test hook

void launch_shell_pack_rows(int M, const float* g_m3, const float* g_m2, const float* g_col, const float* g_op, const float* g_sc,
                            const float* g_rot, const int* idx, float* rows, hipStream_t s);
void launch_shell_unpack_rows(int n, const float* rows, int P, float* dense, int blocked, hipStream_t s, int base = 0);
void launch_shell_pack_rows_live(bool write, int M, const float* g_m3, const float* g_m2, const float* g_col, const float* g_op, const float* g_sc, const float* g_rot,
                                 const int* idx, int P, int chunk_rows, int world, uint32_t* counts, uint32_t* cursor, float* rows_out, hipStream_t s);
void launch_shell_chunk_counts(int M, const int* idx, int chunk, int world, float* counts, hipStream_t s);
void launch_shell_scatter_i32(int M, const int* idx, const int* src, int P, int* dst, hipStream_t s);
void launch_shell_transmittance(int G, int rank, int N, size_t row_stride, const float* all_T, float* T_in, hipStream_t s);
void launch_shell_compose(int G, int rank, int N, const float* planes, const float* bg, float* out_color, float* out_depth, float* out_occ,
                          float* T_final, float* behind, hipStream_t s);
void launch_wedge_pack_columns(int H, int W, int c0, int c1, int wmax, const float* color, const float* depth, const float* occ, float* out, hipStream_t s);
void launch_wedge_unpack_columns(int G, int H, int W, int wmax, size_t stride, const int* edges, const float* blocks, float* color, float* depth,
                                 float* occ, hipStream_t s);
// the one-launch selection of a rank's Gaussians (preprocess.hip k_select_fused, round 6)
#define SEL_ITEMS 4
#define SEL_BLOCK (256 * SEL_ITEMS)
struct SelArgs {
    int P; const float* means; const float* colors; const float* opac; const float* scales; const float* rot; const float* vm;
    float lo, hi;                                                      // shell: range in [lo, hi)
    float mod, inv_col_step, inv_tan_step, col_lo, col_hi;             // wedge: the reach bound of k_wedge_flags
    uint32_t cap; int* idx_out; float* o_means; float* o_colors; float* o_opac; float* o_scales; float* o_rot;
    uint32_t* n_valid_out; int chunk_rows, world; float* chunk_counts;
    unsigned long long* status; uint32_t* ticket;                      // [blocks] (flag << 32 | value), [2]: ticket, finished blocks -- zeroed by the caller
    unsigned blocks;
};
inline size_t select_fused_words(size_t P) { return 2 * ((P + SEL_BLOCK - 1) / SEL_BLOCK) + 8; }      // u32 words of zeroed scratch: status (u64 per block) + ticket + finished
void launch_select_fused(SelArgs a, bool wedge, hipStream_t s);
void launch_shell_flags(int P, const float* means3D, const float* view, float lo, float hi, uint32_t* flags, hipStream_t s);
void launch_wedge_flags(int P, const float* means3D, const float* scales, const float* rotations, float scale_modifier, const float* view,
                        int W, int col_lo, int col_hi, uint32_t* flags, hipStream_t s);
void launch_shell_unpack_rows_add(int n, const float* rows, int P, float* dense, hipStream_t s, int base = 0);
void launch_shell_gather(int P, const uint32_t* flags, const uint32_t* offs, const float* means3D, const float* colors, const float* opacities,
                         const float* scales, const float* rotations, int* idx_out, float* o_means, float* o_colors, float* o_opac,
                         float* o_scales, float* o_rot, hipStream_t s, uint32_t cap = 0xFFFFFFFFu, const uint32_t* total = nullptr,
                         uint32_t* n_valid_out = nullptr, int chunk_rows = 0, int world = 0, float* chunk_counts = nullptr);
void launch_exclusive_scan(const uint32_t* in, uint32_t* out, size_t n, uint32_t* total_out, uint32_t* scratch, hipStream_t s);
// What the LAST pass of a sort may do instead of writing the sorted keys (which the range sort's callers never read): gather a
// per-value record by the sorted value and write it at the value's final position -- mode 1: u32 src[val] -> u32 dst[pos] (compact
// span records), mode 2: (y, x) of u32x4 src[val] -> u32x2 dst[pos].  The random gather then rides on a launch that exists anyway.
struct RadixTail { const void* src = nullptr; void* dst = nullptr; int mode = 0; };
// sorts (key,val) pairs on key bits [0,end_bit) in digits of at most max_bits (<= SORT_MAX_RADIX_BITS; 0 = SORT_RADIX_BITS);
// result ends in (key_a,val_a) or (key_b,val_b): returns 0 for a, 1 for b
// n_dev (nullable): the pair count lives on the device and n is only the capacity the launches cover
// scratch_bits: the scratch holds sort_scratch_words(n, scratch_bits) words (0 = max_bits); room beyond the digit width lets small
// inputs be sorted in half-size blocks (binning.hip radix_pass)
// begin_bit: the sort runs on key bits [begin_bit, end_bit) (an LSD sort cut in two calls: the input of the second is the first's output)
// bias (nullable): the passes sort on key - bias->kmin, with 0xFFFFFFFF (a culled Gaussian) mapped to bias->cull, so that end_bit only
// has to cover the frame's key SPAN.  With kmin a multiple of 256 the low byte of key - kmin is the key's own low byte: a first pass
// over bits [0, 8) needs no bias and can be queued before the host knows the span.
struct KeyBias { uint32_t kmin, cull; const uint32_t* lin_span = nullptr; };   // lin_span (device, binning.hip KeyMap): linear range buckets instead
int launch_radix_sort_pairs(uint32_t* key_a, uint32_t* key_b, uint32_t* val_a, uint32_t* val_b, size_t n, int end_bit,
                            uint32_t* scratch, hipStream_t s, int max_bits = 0, const uint32_t* n_dev = nullptr, int scratch_bits = 0,
                            bool vals_are_positions = false,    // true: the values are 0..n-1 and val_a is never read
                            RadixTail tail = RadixTail(), int begin_bit = 0, const KeyBias* bias = nullptr);
void launch_finish_totals(const uint32_t* totals, const unsigned long long* slots, uint32_t cap, uint32_t* status, hipStream_t s);
// block instance offsets + the instance total from the spans in range order (filled by the range sort's last pass); compact: span_pack
// scan = false: block_off is left holding the blocks' instance COUNTS (launch_emit_instances(..., sums_unscanned = true) adds them up itself;
// *total_out is then not written)
void launch_instance_offsets(const void* span_sorted, bool compact, int TH, uint32_t* block_off, uint32_t* total_out, size_t P, hipStream_t s, bool scan = true);
// key16: the tile keys are 16-bit (the array is the same allocation, half used): every image with at most 65536 list tiles
void launch_emit_instances(const uint32_t* ids_sorted, const uint32_t* block_off, const void* span_sorted, bool compact, size_t P, TileGrid grid,
                           uint32_t* inst_tile, uint32_t* inst_val, hipStream_t s, uint32_t cap = 0xFFFFFFFFu, bool key16 = false,
                           uint2* ranges = nullptr,    // ranges: the launch also clears every tile's list range (then launch_tile_ranges(..., prezeroed = true))
                           bool sums_unscanned = false);
// zero / n_zero: words this launch also clears (the work lists' counters: nothing before the blends touches them)
void launch_tile_ranges(const uint32_t* tile_sorted, size_t R, uint2* ranges, int tiles, hipStream_t s, const uint32_t* R_dev = nullptr, bool key16 = false,
                        uint32_t* zero = nullptr, int n_zero = 0, bool prezeroed = false);
int launch_radix_sort_pairs16(uint16_t* key_a, uint16_t* key_b, uint32_t* val_a, uint32_t* val_b, size_t n, int end_bit, uint32_t* scratch, hipStream_t s,
                              const uint32_t* n_dev = nullptr);
int radix_sort_result_side(size_t n, int end_bit);
// the range sort of the P Gaussians as one bucket pass + one launch that sorts every bucket completely (binning.hip): ids in range order
// -> id_a, the tail's records in range order -> tail.dst; no host knowledge needed.  `ok`: P is in the range this form is used for.
bool range_sort_buckets_ok(size_t P);
void launch_range_sort_buckets(uint32_t* key_a, uint32_t* key_b, uint32_t* id_a, uint32_t* id_b, size_t P, uint32_t* scratch, const uint32_t* key_span,
                               RadixTail tail, hipStream_t s);   // side the two functions above (default digit width, no tail) leave the result on

#ifdef __HIPCC__
// blockIdx -> (patch, segment): segment-fastest, with S ODD.  Workgroups are dealt round-robin to the 8 XCDs (b % 8) and
// the work of a frame is front-loaded (pass 2 and the backward retire the segments behind the T < 1e-4 stop at once), so
// with S a multiple of 8 every first segment lands on the same XCD (measured: backward blend 0.30 -> 2.56 ms at S = 32).
// With S odd the segment index is decorrelated from b % 8.  Dealing spatial groups of patches to XCDs instead (for L2
// reuse between neighbouring tiles) measured slower on pass 1 (0.39 vs 0.36 ms), so the plain numbering stays.
// The launch's stride per patch is S | 1: workgroups go to the eight XCDs round robin (b % 8), and with an even stride a given XCD would
// always walk the same residue class of segment indices -- the early segments (which saturate and leave) on some XCDs, the late ones on
// others.  Found on the second pass-1 round of the semi-transparent cfg3 frame (45 - 5 = 40 segments: 0.221 ms, with 39 or 41: 0.208);
// the padding workgroup of an even S retires on its index.
#ifdef LG_XCD_CHUNK     /* experiment (tools/xcd_ab.sh): LG_XCD_CHUNK consecutive patches stay on one XCD (workgroup b runs on XCD b % 8) */
__device__ __forceinline__ bool block_patch_segment(unsigned b, int patches, int S, int& patch, int& seg) {
    const unsigned stride = (unsigned)S | 1u, per = (unsigned)LG_XCD_CHUNK * stride;
    const unsigned xcd = b & 7u, j = b >> 3, c = j / per, o = j - c * per;
    patch = (int)((c * 8u + xcd) * (unsigned)LG_XCD_CHUNK + o / stride);
    seg = (int)(o % stride);
    return patch < patches && seg < S;
}
inline unsigned segment_grid(int patches, int S) {
    const unsigned group = 8u * (unsigned)LG_XCD_CHUNK;
    return (((unsigned)patches + group - 1u) / group) * group * ((unsigned)S | 1u);
}
#else
__device__ __forceinline__ bool block_patch_segment(unsigned b, int patches, int S, int& patch, int& seg) {
    const unsigned stride = (unsigned)S | 1u;
    seg = (int)(b % stride);
    patch = (int)(b / stride);
#if defined(LG_PATCH_ORDER) && LG_PATCH_ORDER == 1     /* experiment (tools/xcd_ab.sh order): the image's last patches first */
    patch = patches - 1 - patch;
    return patch >= 0 && seg < S;
#elif defined(LG_PATCH_ORDER) && LG_PATCH_ORDER == 2   /* experiment: the segment index outermost (all patches' segment 0, then all segment 1s, ...) */
    seg = (int)(b / (unsigned)patches); patch = (int)(b % (unsigned)patches);
    return seg < S;
#endif
    return patch < patches && seg < S;
}
inline unsigned segment_grid(int patches, int S) { return (unsigned)patches * ((unsigned)S | 1u); }
#endif


// Segments of a tile list: ceil(L / seg_len) of them, at most S (the launch provides S workgroups per patch; the
// surplus ones retire at once and never touch the segment planes).  A pure function of the tile's range, so every
// kernel of a frame (and the backward) recomputes the same split.
__device__ __forceinline__ int segment_count(uint2 range, int S, int seg_len) {
    const uint32_t L = range.y - range.x;
    const uint32_t want = (L + (uint32_t)seg_len - 1u) / (uint32_t)seg_len;
    return (int)min((uint32_t)S, max(1u, want));
}
// [start, end) of segment `seg` (< segment_count) of a tile list
__device__ __forceinline__ uint2 segment_range(uint2 range, int St, int seg) {
    const uint32_t L = range.y - range.x;
    const uint32_t len = (L + (uint32_t)St - 1u) / (uint32_t)St;
    const uint32_t a = min(range.y, range.x + (uint32_t)seg * len);
    const uint32_t b = min(range.y, a + len);
    return make_uint2(a, b);
}

#endif  // __HIPCC__

struct RenderFwdArgs {
    TileGrid grid;
    const uint2* ranges; const uint32_t* point_list; const float4* rec; const uint32_t* rowspan;
    const float2* coltab; const float2* rowtab;
    const float* bg;          // device [2] or nullptr (= 0)
    const float* T_in;        // nullptr = 1
    float* final_T; float* T_pass;
    float* out_color; float* out_depth; float* out_occ;
    float* seg; int S;        // per-(patch, segment) planes, segment slots per list
    int seg_len;              // target entries per segment (a tile uses min(S, ceil(len / seg_len)) segments)
    uint8_t* flags; size_t R; // per-(sub, entry) contribution flags written by pass 1 (nullptr when pass 1 never runs)
    uint8_t* touched = nullptr;   // [P]: set to 1 for the Gaussian of every entry whose flag is set (GeomView::touched); written only where flags are
    int seg_lo, seg_hi;       // segment slots [seg_lo, seg_hi) this launch covers
    int front;                // launch_render_alive: the segment the finished round ends at
    uint8_t* alive;           // [patches] (nullptr = no gating): number of segments pass 1 walked (255 = all of them)
    int run_pass1;            // run the T-only pass (needed when S > 1 or for a shell's phase 1)
    int transmittance_only;   // phase 1 of the multi-GPU shell render: only T_pass is produced
    WorkList fill;            // k_render_combine: the backward's work list (cnt == nullptr: none)
    int walk2;                // k_render_fused: the second form of the T-only walk (set by its launcher)
    float* T_end_out = nullptr;   // k_render_combine: a second copy of final_T (lidargs_render_shell's T_end_out), or NULL
};
void launch_render_pass1(const RenderFwdArgs& a, hipStream_t s);     // T-only walk of every segment
void launch_render_pass2(const RenderFwdArgs& a, hipStream_t s);
void launch_render_head(const RenderFwdArgs& a, int head, hipStream_t s);   // the first `head` segments of every list, walked once     // full walk from the true T_in
void launch_render_alive(const RenderFwdArgs& a, hipStream_t s);     // which patches are still unsaturated behind the front segments
void launch_render_combine(const RenderFwdArgs& a, hipStream_t s);   // fold the segments into the image planes
void launch_render_fused(const RenderFwdArgs& a, hipStream_t s);     // all of the above for a patch in one workgroup (64-entry plan)

struct RenderBwdArgs {
    TileGrid grid;
    const uint2* ranges; const uint32_t* point_list; const float4* rec; const uint32_t* rowspan;
    const float2* coltab; const float2* rowtab;
    const float* bg;
    const float* final_T;
    const float* seg; int S; int seg_len;
    const uint8_t* flags; size_t R;
    const uint8_t* alive;              // as in RenderFwdArgs: segments >= alive[patch] were never walked
    const float* T_final_global;   // nullptr = final_T (single GPU)
    const float* behind;           // nullptr or f32[3*N]: colour0, colour1, depth sums of farther shells
    const float* dL_dpix; const float* dL_ddepth; const float* dL_docc;
    float* gacc;                   // [16P], zeroed: slots 0-2 mean2D.xyz, 3-5 conic A,B,C, 6 opacity, 7-8 colour,
                                   //               9 range, 10-12 G1 = sum gx delta, 13-15 G2 = sum gy delta (moments of dL/du1, dL/du2)
    WorkList walk;                 // the list k_render_combine filled; cnt == nullptr: the slot grid
};
void launch_render_backward(const RenderBwdArgs& a, hipStream_t s);
void launch_count_backward_entries(const RenderBwdArgs& a, unsigned long long* out, hipStream_t s);   // diagnostics: adds the list entries k_render_backward gathers to *out

struct GaussBwdArgs {
    int P; float scale_modifier; const float* view;   // device pointer
    const float* means3D; const float* scales; const float* rotations; const float* cov3D_precomp; const int* radii;
    const float* gacc;             // packed sums from the backward blend (lines of touched Gaussians only)
    const uint8_t* tlist; const uint16_t* tcount;   // GeomView::tlist / tcount: the Gaussians with a gradient; the others' rows stay zero
    float* dL_dmean2D; float* dL_dconic; float* dL_dopacity; float* dL_dcolor; float* dL_ddepths;
    float* dL_dbasis_u1; float* dL_dbasis_u2;
    float* dL_dsphere; float* dL_dmean3D; float* dL_dcov3D; float* dL_dscale; float* dL_drot;
};
void launch_gaussian_backward(const GaussBwdArgs& a, hipStream_t s);

// ---- device-side helpers shared by the blend kernels (render.hip, surfel.hip) ----------------------------------------
#ifdef __HIPCC__
// Whole-row stores and loads of the per-Gaussian arrays (rows of 2, 3, 4 floats at 4-byte alignment: the caller's tensors are carved
// out of one slab): one dwordx2/x3/x4 instruction per row instead of one dword per element -- a wave's dword store of a [P, 4] column
// touches the same sixteen 64-byte lines as the whole-row store does, for a quarter of the data.
typedef float row4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float row3 __attribute__((ext_vector_type(3), aligned(4)));
typedef float row2 __attribute__((ext_vector_type(2), aligned(4)));
#if defined(LG_GB_VARIANT) && LG_GB_VARIANT == 5    /* experiment: rows are only stored when a value is NaN (never) */
__device__ __forceinline__ void put4(float* p, size_t i, float x, float y, float z, float w) { if (x != x || w * 0.f != 0.f) *reinterpret_cast<row4*>(p + 4 * i) = row4{x, y, z, w}; }
__device__ __forceinline__ void put3(float* p, size_t i, float x, float y, float z) { if (x != x || z * 0.f != 0.f) *reinterpret_cast<row3*>(p + 3 * i) = row3{x, y, z}; }
__device__ __forceinline__ void put2(float* p, size_t i, float x, float y) { if (x != x || y * 0.f != 0.f) *reinterpret_cast<row2*>(p + 2 * i) = row2{x, y}; }
#else
__device__ __forceinline__ void put4(float* p, size_t i, float x, float y, float z, float w) { *reinterpret_cast<row4*>(p + 4 * i) = row4{x, y, z, w}; }
__device__ __forceinline__ void put3(float* p, size_t i, float x, float y, float z) { *reinterpret_cast<row3*>(p + 3 * i) = row3{x, y, z}; }
__device__ __forceinline__ void put2(float* p, size_t i, float x, float y) { *reinterpret_cast<row2*>(p + 2 * i) = row2{x, y}; }
#endif
#if defined(LG_GB_VARIANT) && LG_GB_VARIANT == 6    /* experiment: inputs made up from the index instead of loaded */
__device__ __forceinline__ float3 get3(const float* p, size_t i) { const float f = 1.f + (float)(i & 1023) * 1e-3f; return make_float3(f, 0.5f * f, 0.25f + f); }
__device__ __forceinline__ float4 get4(const float* p, size_t i) { const float f = 1.f + (float)(i & 1023) * 1e-3f; return make_float4(0.5f, 0.5f * f, 0.5f, 0.5f / f); }
#else
__device__ __forceinline__ float3 get3(const float* p, size_t i) { const row3 v = *reinterpret_cast<const row3*>(p + 3 * i); return make_float3(v.x, v.y, v.z); }
__device__ __forceinline__ float4 get4(const float* p, size_t i) { const row4 v = *reinterpret_cast<const row4*>(p + 4 * i); return make_float4(v.x, v.y, v.z, v.w); }
#endif

// LDS reads that stay where they are written (see walk_flagged in render.hip)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v3f __attribute__((ext_vector_type(3)));
#define LG_LDS_VOLATILE(T) const volatile __attribute__((address_space(3))) T*
__device__ __forceinline__ float4 lds_ahead(const float4* p) {
    const v4f v = *(LG_LDS_VOLATILE(v4f))p;
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float lds_ahead(const float* p) { return *(LG_LDS_VOLATILE(float))p; }
__device__ __forceinline__ uint32_t lds_ahead(const uint32_t* p) { return *(LG_LDS_VOLATILE(uint32_t))p; }

// gfx950 lane swaps: {a', b'} with a' = (a's lower half | b's lower half), b' = (a's upper half | b's upper half) for halves of
// 32 lanes (v_permlane32_swap) or, row pair by row pair, of 16 (v_permlane16_swap).  a' + b' is then one reduce-scatter step
// with no select: the lower half of the lanes owns the sum of a, the upper half the sum of b.
__device__ __forceinline__ float fold_halves32(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float fold_halves16(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Product / sum over a run of per-segment planes with EIGHT loads in flight: the plain loops (`for k: T *= plane[k]`) compile to a
// load, a wait and a multiply per iteration, i.e. one memory round trip per segment in front of (or behind) the workgroup, before
// its walk can start.  Same operations in the same order as the plain loop.
__device__ __forceinline__ float plane_product(float T, const float* first, size_t stride, int n) {
    int k = 0;
    for (; k + 8 <= n; k += 8) {
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = first[(size_t)(k + i) * stride];
#pragma unroll
        for (int i = 0; i < 8; i++) T *= x[i];
    }
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = (k + i < n) ? first[(size_t)(k + i) * stride] : 1.f;
#pragma unroll
    for (int i = 0; i < 8; i++) if (k + i < n) T *= x[i];
    return T;
}
template <int NP>
__device__ __forceinline__ void plane_sums(float (&acc)[NP], const float* first, size_t stride, const int (&plane)[NP], int n) {
    int k = 0;
    for (; k + 4 <= n; k += 4) {
        float x[4][NP];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int p = 0; p < NP; p++) x[i][p] = first[(size_t)(k + i) * stride + (size_t)plane[p] * 64];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int p = 0; p < NP; p++) acc[p] += x[i][p];
    }
    for (; k < n; k++)
#pragma unroll
        for (int p = 0; p < NP; p++) acc[p] += first[(size_t)k * stride + (size_t)plane[p] * 64];
}

// The entry's opacity for each of the four pixel rows [y0, y0 + 4) of a patch: 0 on the rows outside its row span [lo, hi).
__device__ __forceinline__ float4 rows_opacity(uint32_t span, float opacity, int y0) {
    const int lo = (int)(span & 0xFFFFu), hi = (int)(span >> 16);
    float4 o;
    o.x = (y0 >= lo && y0 < hi) ? opacity : 0.f;
    o.y = (y0 + 1 >= lo && y0 + 1 < hi) ? opacity : 0.f;
    o.z = (y0 + 2 >= lo && y0 + 2 < hi) ? opacity : 0.f;
    o.w = (y0 + 3 >= lo && y0 + 3 < hi) ? opacity : 0.f;
    return o;
}
#endif

}  // namespace lg
