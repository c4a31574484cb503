/*
 * lidargs_rasterizer.h -- C ABI of the MI355X-native LiDAR Gaussian rasterizer.
 *
 * This is the drop-in boundary for the reference's native rasterizer core
 *   CudaRasterizer::Rasterizer::{forward, backward, visible_filter, markVisible}
 *   (/root/reference/submodules/diff_lidargs_rasterization/cuda_rasterizer/rasterizer.h:18-124,
 *    "R3/cr/rasterizer.h" below).
 * Every entry point takes plain device pointers and sizes (no torch / C++ types), the HIP
 * stream to run on, and -- where the reference takes std::function<char*(size_t)> resizers
 * (R3/cr/rasterizer.h:32-34) -- a C callback + user pointer, so the caller (torch, or any
 * other allocator) owns the three opaque scratch buffers and keeps them alive until backward.
 *
 * Conventions shared by all entry points
 *   - all array arguments are DEVICE pointers (gfx950 HBM), float32 / int32, contiguous;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); kernels are only
 *     enqueued on it, the single host wait is the 2-KB read of the instance totals in forward (queued right behind
 *     the preprocess, waited for while the range sort runs);
 *   - return value: >= 0 on success, one of LIDARGS_ERR_* (< 0) on failure, with a message
 *     available from lidargs_last_error() (thread-local);
 *   - argument order and meaning mirror the reference signatures line by line; arguments the
 *     reference's LiDAR path accepts but never reads (D, M, shs, projmatrix, cam_pos,
 *     prefiltered, tan_fov*) are accepted and ignored here too.
 *
 * The three scratch buffers are opaque: their layout, the meaning of the int returned by
 * lidargs_forward ("num_rendered") and the per-pixel contributor counts are private to this
 * library and only consumed by lidargs_backward, exactly as geomBuffer/binningBuffer/imgBuffer
 * are private to the reference's backward (R3/cr/rasterizer_impl.cu:469-471).
 *   num_rendered = Rp | code: Rp = the number of (Gaussian, 16xTH-tile) instances this
 *   implementation binned (NOT the reference's 16x1 count), rounded up to a multiple of 4;
 *   code (low two bits) = log2(TH / 4), the tile height the forward chose for the frame.
 * Like the reference's backward (`fromChunk` on (P, R, W*H)), lidargs_backward rebuilds its whole
 * view of the buffers from (P, R, width, height) and the buffers' CONTENTS: nothing is keyed by a
 * buffer's address and the library keeps no per-forward host state, so the saved buffers may be
 * cloned, moved, offloaded and brought back before the backward runs, and a backward may run
 * more than once on them (the "gradient lines still zero" mark is a word inside the geometry buffer).
 */
#ifndef LIDARGS_RASTERIZER_H
#define LIDARGS_RASTERIZER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIDARGS_ABI_VERSION 2 /* 2 (round 5): round 4 changed lidargs_shell_transmittance (row_stride), lidargs_shell_select_gather (chunk_rows, world, chunk_counts) and the chamfer scratch layout without a bump; a caller built against version 1 must be rebuilt */
#define LIDARGS_NUM_CHANNELS 2 /* R3/cr/config.h:15 NUM_CHANNELS (intensity, ray-drop) */

enum {
    LIDARGS_OK = 0,
    LIDARGS_ERR_INVALID_ARGUMENT = -1, /* bad sizes / NULL required pointer            */
    LIDARGS_ERR_NO_COLORS = -2,        /* colors_precomp == NULL, R3/cr/rasterizer_impl.cu:249-252 */
    LIDARGS_ERR_ALLOC = -3,            /* an allocator callback returned NULL           */
    LIDARGS_ERR_HIP = -4,              /* a HIP runtime call or kernel launch failed    */
    LIDARGS_ERR_STATE = -5,            /* backward: buffers do not match (P, R, W*H)    */
    LIDARGS_ERR_NO_DEVICE = -6         /* no gfx950 device / HIP runtime unavailable    */
};

/* Replaces std::function<char*(size_t N)> (R3/cr/rasterizer.h:32-34; the torch-side lambda is
 * R3/rasterize_points.cu:27-33).  Must return a device pointer to at least `bytes` bytes,
 * 128-byte aligned, valid until the matching backward has run.  Called at most once per
 * buffer per lidargs_forward / lidargs_visible_filter call. */
typedef char* (*lidargs_alloc_fn)(void* user, size_t bytes);

int lidargs_abi_version(void);
const char* lidargs_last_error(void);

/* Rasterizer::forward -- R3/cr/rasterizer.h:31-58, R3/cr/rasterizer_impl.cu:202-359.
 * Outputs (caller-allocated, R3/rasterize_points.cu:71-75): out_color f32[2*H*W],
 * out_depth f32[H*W], out_occ f32[H*W], radii i32[P], radii_xy i32[2P] (may be NULL: the reference's binding never
 * returns it, R3/rasterize_points.cu:75).
 * Returns num_rendered (>= 0) to be passed back to lidargs_backward as R. */
int lidargs_forward(
    lidargs_alloc_fn geometry_alloc, void* geometry_user,
    lidargs_alloc_fn binning_alloc, void* binning_user,
    lidargs_alloc_fn image_alloc, void* image_user,
    int P, int D, int M,
    const float* background,
    int width, int height,
    const float* means3D,
    const float* shs,
    const float* colors_precomp,
    const float* opacities,
    const float* scales,
    float scale_modifier,
    const float* rotations,
    const float* cov3D_precomp,
    const float* viewmatrix,
    const float* projmatrix,
    const float* cam_pos,
    const float* beam_inclinations,
    int prefiltered,
    int lidar_far,
    int lidar_near,
    float* out_color,
    float* out_depth,
    float* out_occ,
    int* radii,
    int* radii_xy,
    int debug,
    void* stream);

/* lidargs_forward without its one host wait (no reference counterpart: the reference blocks on a cudaMemcpy of the instance
 * count, R3/cr/rasterizer_impl.cu:292, to size the binning buffer).  The caller states the capacity instead:
 *   instance_capacity  (Gaussian, 16 x tile_rows tile) instances the binning buffer is sized for (e.g. 1.25 x what the previous
 *                      frame needed -- the int lidargs_forward returned, low two bits cleared);
 *   tile_rows          list-tile height 4, 8, 16 or 32 (what a previous lidargs_forward chose: 4 << (num_rendered & 3));
 *   status_host        NULL, or 16 words of PINNED host memory the stream fills behind the launches: [0] instances the frame
 *                      needed, [1] instances binned = min(needed, capacity), [2..7] 64-bit instance totals for tile heights
 *                      4 / 8 / 16, [8] = 1 if the capacity was too small, [9] capacity, [10..11] the total for tile height 32.  Valid once the
 *                      stream has passed the copy.
 * Every count the later stages need stays on the device, so the call only enqueues work (it can be captured in a HIP graph, and
 * the host can run ahead).  If [8] comes back 1, instances were dropped: that frame's outputs are wrong and it must be redone
 * with a larger capacity (or with lidargs_forward).  The returned int and the three buffers go to lidargs_backward as usual. */
int lidargs_forward_enqueue(
    lidargs_alloc_fn geometry_alloc, void* geometry_user,
    lidargs_alloc_fn binning_alloc, void* binning_user,
    lidargs_alloc_fn image_alloc, void* image_user,
    int P, int D, int M,
    const float* background,
    int width, int height,
    const float* means3D,
    const float* shs,
    const float* colors_precomp,
    const float* opacities,
    const float* scales,
    float scale_modifier,
    const float* rotations,
    const float* cov3D_precomp,
    const float* viewmatrix,
    const float* projmatrix,
    const float* cam_pos,
    const float* beam_inclinations,
    int prefiltered,
    int lidar_far,
    int lidar_near,
    float* out_color,
    float* out_depth,
    float* out_occ,
    int* radii,
    int* radii_xy,
    int debug,
    int instance_capacity,
    int tile_rows,
    unsigned* status_host,
    void* stream);

/* Rasterizer::backward -- R3/cr/rasterizer.h:86-122, R3/cr/rasterizer_impl.cu:431-549.
 * All dL_d* outputs are caller-allocated.  The reference zero-initialises them (R3/rasterize_points.cu:163-175)
 * and accumulates with atomics; this library instead WRITES every one of the P rows of every output (zeros for
 * Gaussians with radii <= 0), so they may be passed uninitialised:
 * dL_dmean2D f32[4P], dL_dconic f32[4P], dL_dopacity f32[P], dL_dcolor f32[2P],
 * dL_ddepths f32[P], dL_dmean3D f32[3P], dL_dsphere_means3D f32[3P], dL_dbasis_u1 f32[3P],
 * dL_dbasis_u2 f32[3P], dL_dcov3D f32[6P], dL_dsh (untouched), dL_dscale f32[3P], dL_drot f32[4P].
 * dL_dconic, dL_ddepths, dL_dsphere_means3D, dL_dbasis_u1 and dL_dbasis_u2 are scratch of the reference's two-kernel
 * backward that its binding never returns (R3/rasterize_points.cu:219); each of them may be NULL, in which case it is
 * not materialised (56 of 148 bytes per Gaussian less to write).  dL_dcov3D may be NULL as well when the covariance comes
 * from scales + rotations (cov3D_precomp == NULL): it is then only an intermediate of the scale / rotation gradients. */
int lidargs_backward(
    int P, int D, int M, int R,
    const float* background,
    int width, int height,
    const float* means3D,
    const float* shs,
    const float* colors_precomp,
    const float* scales,
    float scale_modifier,
    const float* rotations,
    const float* cov3D_precomp,
    const float* viewmatrix,
    const float* projmatrix,
    const float* campos,
    const float* beam_inclinations,
    float tan_fovx, float tan_fovy,
    const int* radii,
    char* geom_buffer,
    char* binning_buffer,
    char* image_buffer,
    const float* dL_dpix,
    const float* dL_dout_depth,
    const float* dL_dout_occ,
    float* dL_dmean2D,
    float* dL_dconic,
    float* dL_dopacity,
    float* dL_dcolor,
    float* dL_ddepths,
    float* dL_dmean3D,
    float* dL_dsphere_means3D,
    float* dL_dbasis_u1,
    float* dL_dbasis_u2,
    float* dL_dcov3D,
    float* dL_dsh,
    float* dL_dscale,
    float* dL_drot,
    int debug,
    void* stream);

/* Rasterizer::visible_filter -- R3/cr/rasterizer.h:60-84, R3/cr/rasterizer_impl.cu:362-426.
 * Only radii / radii_xy are produced; the allocators may be NULL (no scratch is needed here,
 * the reference allocates and discards it). */
int lidargs_visible_filter(
    lidargs_alloc_fn geometry_alloc, void* geometry_user,
    lidargs_alloc_fn binning_alloc, void* binning_user,
    lidargs_alloc_fn image_alloc, void* image_user,
    int P, int M,
    int width, int height,
    const float* means3D,
    const float* scales,
    float scale_modifier,
    const float* rotations,
    const float* cov3D_precomp,
    const float* viewmatrix,
    const float* projmatrix,
    const float* cam_pos,
    const float* beam_inclinations,
    float tan_fovx, float tan_fovy,
    int prefiltered,
    int lidar_far,
    int lidar_near,
    int* radii,
    int* radii_xy,
    int debug,
    void* stream);

/* Rasterizer::markVisible -- R3/cr/rasterizer.h:21-29, R3/cr/rasterizer_impl.cu:142-154.
 * present: bool (1 byte) [P]. */
int lidargs_mark_visible(
    int P,
    const float* means3D,
    const float* viewmatrix,
    const float* projmatrix,
    unsigned char* present,
    void* stream);

/* ---- multi-GPU range-shell variants (no reference counterpart) -------
 * A rank owns the Gaussians whose range lies in [shell_lo, shell_hi) (on top of the reference's
 * integer near/far cull), so every pixel's depth-sorted list is the concatenation of the ranks'
 * lists and per-Gaussian work and gradients stay local to one rank.
 *
 *   lidargs_forward_shell   preprocess + bin this shell, then composite it starting from
 *                           T_in f32[H*W] (NULL = 1).  transmittance_pass != 0 is phase 1 of the
 *                           two-phase render: only T_out f32[H*W] is produced, = the transmittance
 *                           this shell hands to the shells behind it (if the reference's
 *                           T < 1e-4 early-out fires inside the shell, the value that tripped it).
 *   lidargs_render_shell    phase 2: composite the already-binned shell again, now from the true
 *                           T_in = product of the nearer shells' T_out.  Outputs are this shell's
 *                           own partial sums, already weighted by the global transmittance
 *                           (pass background = NULL and add T_final*bg after the reduction);
 *                           T_end_out f32[H*W] (optional) = transmittance after this shell.
 *   lidargs_backward_shell  like lidargs_backward, seeded with what lies BEHIND the shell:
 *                           behind f32[3*H*W] = (colour0, colour1, depth) partial sums of all
 *                           farther shells; T_final_global f32[H*W].
 */
int lidargs_forward_shell(
    lidargs_alloc_fn geometry_alloc, void* geometry_user,
    lidargs_alloc_fn binning_alloc, void* binning_user,
    lidargs_alloc_fn image_alloc, void* image_user,
    int P, const float* background, int width, int height,
    const float* means3D, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* beam_inclinations,
    int lidar_far, int lidar_near, float shell_lo, float shell_hi,
    const float* T_in, int transmittance_pass,
    float* out_color, float* out_depth, float* out_occ, float* T_out,
    int* radii, int* radii_xy, int debug, void* stream);

/* lidargs_forward_shell without its host wait, as lidargs_forward_enqueue is to lidargs_forward (instance_capacity, tile_rows,
 * status_host: see there), for rank frames the host only enqueues.  n_valid (DEVICE word, or NULL = all P): the P rows are a
 * capacity-sized selection (lidargs_shell_select_enqueue) of which only the first *n_valid exist; the rest are culled unread. */
int lidargs_forward_shell_enqueue(
    lidargs_alloc_fn geometry_alloc, void* geometry_user,
    lidargs_alloc_fn binning_alloc, void* binning_user,
    lidargs_alloc_fn image_alloc, void* image_user,
    int P, const float* background, int width, int height,
    const float* means3D, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* beam_inclinations,
    int lidar_far, int lidar_near, float shell_lo, float shell_hi,
    const float* T_in, int transmittance_pass,
    float* out_color, float* out_depth, float* out_occ, float* T_out,
    int* radii, int* radii_xy, int debug,
    const unsigned* n_valid, int instance_capacity, int tile_rows, unsigned* status_host, void* stream);

int lidargs_render_shell(
    int P, int R, const float* background, int width, int height,
    char* geom_buffer, char* binning_buffer, char* image_buffer,
    const float* T_in, int transmittance_pass,
    float* out_color, float* out_depth, float* out_occ, float* T_out, float* T_end_out,
    int debug, void* stream);

int lidargs_backward_shell(
    int P, int R, const float* background, int width, int height,
    const float* means3D, const float* colors_precomp,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* beam_inclinations,
    const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
    const float* behind, const float* T_final_global,
    const float* dL_dpix, const float* dL_dout_depth, const float* dL_dout_occ,
    float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_ddepths,
    float* dL_dmean3D, float* dL_dsphere_means3D, float* dL_dbasis_u1, float* dL_dbasis_u2,
    float* dL_dcov3D, float* dL_dscale, float* dL_drot, int debug, void* stream);

/* ---- range-shell helpers (multi-GPU, lidar-gs_amd/lidargs_dist.py; no reference counterpart) -------------------------
 * lidargs_shell_select compacts the Gaussians whose view-space range lies in [shell_lo, shell_hi) -- the same float test
 * lidargs_forward_shell applies -- into dense arrays, in ascending index order, so that a rank's frame runs on ~P/N rows.
 * idx_out i32[P] and the five out_* arrays have room for P rows; scratch holds lidargs_shell_select_scratch_bytes(P) bytes.
 * Returns the number of selected Gaussians (one host read) or a negative error code.
 * lidargs_shell_select_count / _gather are its two halves: count (flags + scan + the host read, result left in `scratch`) returns
 * M, after which the caller allocates exactly M rows PER FRAME and gather fills them -- a forward's selection is saved for its
 * backward and must not be overwritten by the next forward's (several views per step, gradient accumulation, an eval render).
 * The gathers take chunk_counts f32[world] (or NULL): the number of selected rows whose index lies in each index chunk of
 * chunk_rows -- the split sizes of the gradient all-to-all, read off the scan (world <= 256).
 * lidargs_shell_transmittance: T_in[i] = prod_{g<rank} all_T[g*row_stride+i] (row_stride >= N floats: the gathered rows may
 * carry a tail behind their N transmittances).
 * lidargs_shell_compose folds the gathered per-shell planes (planes f32[G*5*N], per shell C0, C1, D, T_end, T_hand):
 * image = sum of partials (+ T_final * background), T_final = T_end of the first shell whose T_hand < 1e-4 (the last
 * shell's otherwise), behind f32[3N] = sum of the partials of the shells behind `rank`. */
size_t lidargs_shell_select_scratch_bytes(int P);
int lidargs_shell_select(int P, const float* means3D, const float* colors, const float* opacities, const float* scales,
                         const float* rotations, const float* viewmatrix, float shell_lo, float shell_hi,
                         int* idx_out, float* out_means3D, float* out_colors, float* out_opacities, float* out_scales,
                         float* out_rotations, char* scratch, size_t scratch_bytes, void* stream);
int lidargs_shell_select_count(int P, const float* means3D, const float* viewmatrix, float shell_lo, float shell_hi,
                               char* scratch, size_t scratch_bytes, void* stream);
int lidargs_shell_select_gather(int P, const float* means3D, const float* colors, const float* opacities, const float* scales,
                                const float* rotations, int* idx_out, float* out_means3D, float* out_colors,
                                float* out_opacities, float* out_scales, float* out_rotations, char* scratch,
                                size_t scratch_bytes, int chunk_rows, int world, float* chunk_counts, void* stream);
/* The selections without their host read (enqueue-only rank frames): the caller states a row CAPACITY (e.g. 1.25 x what the
 * previous frame selected) and gets capacity-row arrays of which the first min(selected, capacity) are filled.  idx_out's tail
 * holds 0x7F7F7F7F (ascending order kept; every consumer skips indices >= P).  n_valid_dev u32[2] (device): [0] rows gathered --
 * hand it to lidargs_forward_shell_enqueue / _wedge_enqueue as n_valid --, [1] rows selected; status_host (PINNED u32[2], optional)
 * receives both behind the launches: [1] > capacity means rows were dropped and the frame must be redone with more. */
int lidargs_shell_select_enqueue(int P, const float* means3D, const float* colors, const float* opacities, const float* scales,
                                 const float* rotations, const float* viewmatrix, float shell_lo, float shell_hi, int capacity,
                                 int* idx_out, float* out_means3D, float* out_colors, float* out_opacities, float* out_scales,
                                 float* out_rotations, unsigned* n_valid_dev, unsigned* status_host, char* scratch,
                                 size_t scratch_bytes, int chunk_rows, int world, float* chunk_counts, void* stream);
int lidargs_wedge_select_enqueue(int P, const float* means3D, const float* colors, const float* opacities, const float* scales,
                                 const float* rotations, float scale_modifier, const float* viewmatrix, int width, int col_lo,
                                 int col_hi, int capacity, int* idx_out, float* out_means3D, float* out_colors,
                                 float* out_opacities, float* out_scales, float* out_rotations, unsigned* n_valid_dev,
                                 unsigned* status_host, char* scratch, size_t scratch_bytes, int chunk_rows, int world,
                                 float* chunk_counts, void* stream);
/* Round 6 (experiment, not what lidargs_dist uses by default: measured no faster): the selection of an ORDINARY frame in ONE launch (test, scan
 * in index order by decoupled look-back over the blocks, gather -- k_select_fused) instead of flags + a three-launch scan + gather: the caller passes `capacity`-row arrays (capacity = P is always enough),
 * the call makes the one host read the two-step form makes as well and returns the rows gathered M; the first M rows are the selection,
 * ascending by index, bit-identical to the two-step form's. */
int lidargs_shell_select_sync(int P, const float* means3D, const float* colors, const float* opacities, const float* scales,
                              const float* rotations, const float* viewmatrix, float shell_lo, float shell_hi, int capacity, int* idx_out,
                              float* out_means3D, float* out_colors, float* out_opacities, float* out_scales, float* out_rotations,
                              unsigned* n_valid_dev, char* scratch, size_t scratch_bytes, int chunk_rows, int world, float* chunk_counts,
                              void* stream);
int lidargs_wedge_select_sync(int P, const float* means3D, const float* colors, const float* opacities, const float* scales,
                              const float* rotations, float scale_modifier, const float* viewmatrix, int width, int col_lo, int col_hi,
                              int capacity, int* idx_out, float* out_means3D, float* out_colors, float* out_opacities, float* out_scales,
                              float* out_rotations, unsigned* n_valid_dev, char* scratch, size_t scratch_bytes, int chunk_rows, int world,
                              float* chunk_counts, void* stream);
/* Round 6: the gradient exchange of the sharded frame ships only the rows that carry a gradient (a frame blends a fraction of the Gaussians a
 * rank preprocesses: every other row is 72 bytes of zeros).  _live_count: live rows per destination chunk (index / chunk_rows) into
 * counts_dev u32[world] and, when counts_host (HOST u32[world]) is given, to the host (the call then waits for the copy and returns their sum;
 * with NULL it returns 0 at once: a caller that gathers every rank's counts reads them all in one go).  _live: writes exactly that many [18]-float rows, grouped by destination in ascending chunk order (any order inside a
 * group); cursor_dev u32[world] is scratch.  idx: the rows' global indices (ascending); rows whose index is outside [0, P) never travel. */
int lidargs_shell_pack_grad_rows_live_count(int M, const float* dL_dmeans3D, const float* dL_dmeans2D, const float* dL_dcolors,
                                            const float* dL_dopacity, const float* dL_dscales, const float* dL_drotations, const int* idx, int P,
                                            int chunk_rows, int world, unsigned* counts_dev, unsigned* counts_host, void* stream);
int lidargs_shell_pack_grad_rows_live(int M, const float* dL_dmeans3D, const float* dL_dmeans2D, const float* dL_dcolors, const float* dL_dopacity,
                                      const float* dL_dscales, const float* dL_drotations, const int* idx, int P, int chunk_rows, int world,
                                      unsigned* counts_dev, unsigned* cursor_dev, float* rows, void* stream);
/* Round 6, gradient mode "shard" of lidargs_dist: the [n, 18] gradient rows a rank received for its own index chunk
 * [base, base + chunk_rows) unpacked into dense f32[17][chunk_rows] (six contiguous blocks: means3D 3, means2D 4, colours 2, opacity 1, scales 3,
 * rotations 4 columns of chunk_rows rows each; zero-filled here); add != 0 adds rows of equal index (column wedges). */
int lidargs_shell_unpack_grad_rows_chunk(int n, const float* rows, int base, int chunk_rows, float* dense, int add, void* stream);
int lidargs_shell_transmittance(int G, int rank, int N, size_t row_stride, const float* all_T, float* T_in, void* stream);
int lidargs_shell_compose(int G, int rank, int N, const float* planes, const float* background, float* out_color,
                          float* out_depth, float* out_occ, float* T_final, float* behind, void* stream);

/* ---- column-wedge helpers (multi-GPU, lidar-gs_amd/lidargs_dist.py WedgeRasterizer; no reference counterpart) --------------------
 * Rank g owns the pixel columns [col_lo, col_hi) (whole 16-pixel tile columns; col_hi may be the image width).  Pixels are
 * independent, so a rank that bins every Gaussian that can reach its columns renders them exactly as a single GPU would -- same
 * lists, same order, no transmittance exchange.
 * lidargs_wedge_select_count flags the Gaussians whose reference rect can reach the wedge (a bound from above on the rect's
 * half-width from the largest scale; scales / rotations may be NULL = point-like / unit quaternions), leaves flags + offsets in
 * `scratch` (lidargs_shell_select_scratch_bytes(P) bytes) and returns their number (one host read); lidargs_shell_select_gather
 * then fills the caller's M-row arrays.
 * lidargs_forward_wedge = lidargs_forward restricted to the wedge's tile columns (binning AND blend launches cover them only):
 * out_* hold the wedge's pixel columns, the other columns are NOT written; radii / radii_xy are the Gaussians' full
 * (unrestricted) values.  Its buffers go to lidargs_backward_wedge with the same columns (lidargs_backward would walk patches
 * the forward never rendered); upstream gradients are full-size planes of which only the wedge's columns are read.
 * lidargs_wedge_unpack_grad_rows_add: as lidargs_shell_unpack_grad_rows (blocked layout), but rows carrying the same index are
 * ADDED: a Gaussian straddling a wedge boundary has partial gradient rows on both sides. */
int lidargs_wedge_select_count(int P, const float* means3D, const float* scales, const float* rotations, float scale_modifier,
                               const float* viewmatrix, int width, int col_lo, int col_hi, char* scratch, size_t scratch_bytes,
                               void* stream);
int lidargs_forward_wedge(
    lidargs_alloc_fn geometry_alloc, void* geometry_user,
    lidargs_alloc_fn binning_alloc, void* binning_user,
    lidargs_alloc_fn image_alloc, void* image_user,
    int P, const float* background, int width, int height,
    const float* means3D, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* beam_inclinations, int lidar_far, int lidar_near,
    int col_lo, int col_hi,
    float* out_color, float* out_depth, float* out_occ, int* radii, int* radii_xy, int debug, void* stream);
int lidargs_forward_wedge_enqueue(               /* lidargs_forward_wedge without its host wait: see lidargs_forward_shell_enqueue */
    lidargs_alloc_fn geometry_alloc, void* geometry_user,
    lidargs_alloc_fn binning_alloc, void* binning_user,
    lidargs_alloc_fn image_alloc, void* image_user,
    int P, const float* background, int width, int height,
    const float* means3D, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
    const float* viewmatrix, const float* beam_inclinations, int lidar_far, int lidar_near,
    int col_lo, int col_hi,
    float* out_color, float* out_depth, float* out_occ, int* radii, int* radii_xy, int debug,
    const unsigned* n_valid, int instance_capacity, int tile_rows, unsigned* status_host, void* stream);
int lidargs_backward_wedge(
    int P, int R, const float* background, int width, int height,
    const float* means3D, const float* colors_precomp, const float* scales, float scale_modifier,
    const float* rotations, const float* cov3D_precomp, const float* viewmatrix, const float* beam_inclinations,
    const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer, int col_lo, int col_hi,
    const float* dL_dpix, const float* dL_dout_depth, const float* dL_dout_occ,
    float* dL_dmean2D, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dscale,
    float* dL_drot, int debug, void* stream);
int lidargs_wedge_unpack_grad_rows_add(int n, const float* rows, int P, float* dense, void* stream);
/* The image exchange of the column wedges, one launch each way: pack = the rank's columns [col_lo, col_hi) of colour[2,H,W],
 * depth[H,W], occ[H,W] as a dense f32[4][H][wmax] block (zero padding behind col_hi - col_lo); unpack = G gathered blocks
 * (block g at blocks + g * block_stride floats) -> the full planes, edges_host = G + 1 HOST ints 0 = e_0 < ... < e_G = width. */
int lidargs_wedge_pack_columns(int height, int width, int col_lo, int col_hi, int wmax, const float* color, const float* depth,
                               const float* occ, float* out, void* stream);
int lidargs_wedge_unpack_columns(int G, int height, int width, int wmax, size_t block_stride, const int* edges_host,
                                 const float* blocks, float* color, float* depth, float* occ, void* stream);

/* Gradient exchange of the range shells (lidargs_dist step 6), one launch each:
 * lidargs_shell_pack_grad_rows    rows f32[M*18] = per selected Gaussian (dL_dmeans3D 3, dL_dmeans2D 4, dL_dcolors 2, dL_dopacity 1,
 *                                 dL_dscales 3, dL_drotations 4, bit pattern of its global index idx[i]): what the all-to-all ships.
 * lidargs_shell_unpack_grad_rows  dense f32[P*17] = 0, then row i of rows f32[n*18] written at the index in its column 17: as row
 *                                 [17] of dense (blocked = 0) or into six contiguous blocks [P,3][P,4][P,2][P,1][P,3][P,4] (blocked = 1).
 * lidargs_shell_chunk_counts      counts f32[world]: how many of the ascending idx i32[M] fall into each index chunk of chunk_rows
 *                                 rows (the split sizes of the all-to-all; exact as floats below 2^24).
 * lidargs_shell_scatter_radii     radii i32[P] = 0, then radii[idx[i]] = radii_shell[i]. */
int lidargs_shell_pack_grad_rows(int M, const float* dL_dmeans3D, const float* dL_dmeans2D, const float* dL_dcolors,
                                 const float* dL_dopacity, const float* dL_dscales, const float* dL_drotations, const int* idx,
                                 float* rows, void* stream);
int lidargs_shell_unpack_grad_rows(int n, const float* rows, int P, float* dense, int blocked, void* stream);
int lidargs_shell_chunk_counts(int M, const int* idx, int chunk_rows, int world, float* counts, void* stream);
int lidargs_shell_scatter_radii(int M, const int* idx, const int* radii_shell, int P, int* radii, void* stream);

/* ---- 2DGS "laser-surfel" variant (BASELINE config 5) ---------------------------------------------
 * Drop-in for CudaRasterizer::Rasterizer of /root/reference/submodules/diff_lidargs_surfel_rasterization
 * ("R2/"): forward R2/cuda_rasterizer/rasterizer.h:31-58, backward :60-94, visible_filter :96-114
 * (markVisible is identical to the 3-D variant's: use lidargs_mark_visible).  Same conventions as above.
 * scales is f32[2P] (vec2 per surfel, the reference reads the tensor with a row stride of 2 floats,
 * R2/cr/rasterizer_impl.cu:256).  out_others f32[7*H*W] = depth, alpha, normal xyz, median depth, distortion
 * (R2/cr/auxiliary.h:23-27).  `pixels` f32[P] is accepted and never written (neither does the reference:
 * the atomicAdd is commented out, R2/cr/forward.cu:522).  transMat_precomp f32[9P] (rows Tu, Tv, Tw per surfel) may be given NEXT TO
 * scales and rotations, as in the reference: the preprocess builds its own rows from scales / rotations whatever is passed (rect,
 * normal, sort depth, pixel centre: R2/cr/forward.cu:271-325) and the blends -- forward and backward -- read the precomputed rows
 * (R2/cr/rasterizer_impl.cu:332, :408); pass the same pointer to the backward.  Without scales / rotations it is refused
 * (LIDARGS_ERR_INVALID_ARGUMENT; the reference reads an empty tensor there and its backward prints "Ts_precomp error",
 * R2/cr/backward.cu:661-664).  Backward outputs (written for all P rows): dL_dmean2D f32[4P],
 * dL_dnormal f32[3P], dL_dopacity f32[P], dL_dcolor f32[2P], dL_dmean3D f32[3P], dL_dtransMat f32[9P],
 * dL_dtransMat_2dtemp f32[3P], dL_dscale f32[2P], dL_drot f32[4P], depth f32[P]; dL_depths is the gradient of
 * all 7 planes of out_others.  dL_dnormal, dL_dtransMat and dL_dtransMat_2dtemp are intermediates of the reference's
 * two-kernel backward (dL_dtransMat is also the gradient of transMat_precomp when that was given): each may be NULL, in which case
 * it is not materialised (60 bytes per surfel less to write). */
int lidargs_surfel_forward(
    lidargs_alloc_fn geometry_alloc, void* geometry_user,
    lidargs_alloc_fn binning_alloc, void* binning_user,
    lidargs_alloc_fn image_alloc, void* image_user,
    int P, int D, int M, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
    const float* scales, float scale_modifier, const float* rotations, const float* transMat_precomp,
    const float* viewmatrix, const float* projmatrix, const float* cam_pos, const float* beam_inclinations,
    int prefiltered, int lidar_far, int lidar_near,
    float* out_color, float* out_others, float* pixels, int* radii, int* radii_xy, int debug, void* stream);

int lidargs_surfel_backward(
    int P, int D, int M, int R, const float* background, int width, int height,
    const float* means3D, const float* shs, const float* colors_precomp,
    const float* scales, float scale_modifier, const float* rotations, const float* transMat_precomp,
    const float* viewmatrix, const float* projmatrix, const float* campos, const float* beam_inclinations,
    const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
    const float* dL_dpix, const float* dL_depths,
    float* dL_dmean2D, float* dL_dnormal, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D,
    float* dL_dtransMat, float* dL_dtransMat_2dtemp, float* dL_dsh, float* dL_dscale, float* dL_drot, float* depth,
    int debug, void* stream);

int lidargs_surfel_visible_filter(
    lidargs_alloc_fn geometry_alloc, void* geometry_user,
    lidargs_alloc_fn binning_alloc, void* binning_user,
    lidargs_alloc_fn image_alloc, void* image_user,
    int P, int M, int width, int height,
    const float* means3D, const float* scales, float scale_modifier, const float* rotations, const float* transMat_precomp,
    const float* viewmatrix, const float* projmatrix, const float* beam_inclinations,
    int prefiltered, int lidar_far, int lidar_near, int* radii, int* radii_xy, int debug, void* stream);

/* ---- introspection used by bench.py / tests (no reference counterpart) ---------------------
 * Per-stage HIP-event timing on the op's own stream.  While enabled, every forward/backward call
 * records one event per stage boundary (no host wait); lidargs_profile_read() synchronises the
 * LAST call's events and returns elapsed milliseconds per stage. */
#define LIDARGS_MAX_STAGES 24
void lidargs_profile_enable(int on);                         /* 0 off, 1 every call, N > 1 every N-th forward and every N-th backward */
int lidargs_profile_read(float* ms_out, int max_stages);      /* returns #stages written     */
const char* lidargs_profile_stage_name(int stage);            /* NULL past the last stage    */
/* Aggregate of all calls recorded since lidargs_profile_enable(1) (up to 512): per distinct stage
 * name the summed milliseconds and sample count; returns the number of names written.
 * (at most 256 calls are kept; lidargs_profile_enable(1) pre-creates the events and resets.) */
int lidargs_profile_summary(const char** names_out, float* total_ms_out, int* count_out, int max_stages);

/* Counters of the last forward ON THE CALLING THREAD (thread-local diagnostics: a call from another thread -- autograd's
 * backward thread, a monitoring thread -- sees that thread's own last forward, or zeros): [0]=P, [1]=visible Gaussians V,
 * [2]=instances binned by this library (num_rendered), [3]=R_ref = sum of the reference's
 * 16x1 tiles_touched (what SURVEY.md 8d's byte formula is written in), [4]=tile rows TH,
 * [5]=number of tiles, [6]=instances that at least one pixel of their patch took in pass 1 (-1 if the
 * T-only pass did not run), [7]=segments per list, [8]=Gaussians some pixel's walk took (the only ones with a gradient: the backward
 * clears, adds to and reads the packed sums of these alone; -1 if unknown), [9]=list entries the backward blend gathers (per patch and
 * segment the flagged entries in front of its last blended one; -1 if unknown: surfel variant, no instances).  Values [1], [3], [6], [8],
 * [9] live on the device: they are counted by small launches queued at the end of the forward itself, into a page the library owns, ONLY
 * while lidargs_counters_enable(1) is in force on the calling thread (otherwise they read -1); lidargs_last_counters then waits for
 * those launches.  Nothing about the caller's buffers is remembered between calls. */
void lidargs_counters_enable(int on);
int lidargs_last_counters(long long* out, int n);

/* Test hook (no reference counterpart as an entry point): the tile rect the two preprocess kernels give a Gaussian -- getRect_lidar,
 * R3/cr/auxiliary.h:80-92 (surfel = 0) and R2/cr/auxiliary.h:99-112 (surfel = 1) -- evaluated by the SAME device function on n
 * caller-supplied inputs: p_cr float[n][2] = (p.x, p.y), r_xy int[n][2] = (rx, ry), a tiles_x x tiles_y grid of 16x1 tiles;
 * rects int[n][4] = (xmin, ymin, xmax, ymax).  All device pointers.  tests/ feed it inputs within a few ulps of every truncation and
 * rounding boundary (which one random Gaussian in 1e8 hits) and require bit equality with the oracle's getRect. */
int lidargs_debug_rects(int n, int surfel, const float* p_cr, const int* r_xy, int tiles_x, int tiles_y, int* rects, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LIDARGS_RASTERIZER_H */
