// Internal declarations shared by the kernels and the host orchestration of libgsr_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/gsr.h"
#include "gsr_constants.h"

// ---------------------------------------------------------------- error plumbing
void gsr_set_error(const char* fmt, ...);
#define GSR_HIP_CHECK(expr)                                                         \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            gsr_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),    \
                          __FILE__, __LINE__);                                      \
            return GSR_E_HIP;                                                       \
        }                                                                           \
    } while (0)
#define GSR_LAUNCH_CHECK() GSR_HIP_CHECK(hipGetLastError())

// ---------------------------------------------------------------- profiler hooks
enum GsrKernelId {
    GSR_K_PREPROCESS_FWD = 0, GSR_K_SORT_HIST, GSR_K_SORT_SCATTER, GSR_K_SCAN, GSR_K_EMIT,
    GSR_K_FINALIZE, GSR_K_RENDER_FWD, GSR_K_RENDER_BWD, GSR_K_PREPROCESS_BWD, GSR_K_KNN,
    GSR_K_LOSS_FWD, GSR_K_LOSS_BWD, GSR_K_REG_FWD, GSR_K_REG_BWD, GSR_K_ADAM, GSR_K_COUNT
};
bool gsr_profile_on();
void gsr_profile_begin(int kernel, hipStream_t s);
void gsr_profile_end(int kernel, hipStream_t s);
struct GsrProfileScope {
    int k; hipStream_t s; bool on;
    GsrProfileScope(int k_, hipStream_t s_) : k(k_), s(s_), on(gsr_profile_on()) { if (on) gsr_profile_begin(k, s); }
    ~GsrProfileScope() { if (on) gsr_profile_end(k, s); }
};

// ---------------------------------------------------------------- buffer layouts
// One "splat record" per Gaussian, written by preprocess_fwd and gathered by the render kernels
// in 16-byte pieces: [Tu.xyz Tv.xyz Tw.xyz | xy | n.xyz opa | rgb | cull rect (4 x int16)]
// The cull rect is a conservative pixel bounding box of {alpha >= 1/255}: pairs outside it are
// skipped by the per-pixel alpha test anyway, so whole waves can skip the splat without evaluating it.
#define GSR_SPLAT_FLOATS 20
#define GSR_SP_TU 0
#define GSR_SP_TV 3
#define GSR_SP_TW 6
#define GSR_SP_XY 9
#define GSR_SP_NRM 11
#define GSR_SP_OPA 14
#define GSR_SP_RGB 15
#define GSR_SP_RECT 18   // two 32-bit words: (x0 | x1 << 16), (y0 | y1 << 16), signed 16-bit each

// One gradient row per (instance, 4x4 pixel block the forward blended it into), written by render_bwd (the 16 lanes of a
// DPP row own their block's row: no cross-wave combine, no atomics) and summed per Gaussian by reduce_rows.  A row is
// 72 bytes in two arrays: 16 floats (ONE aligned 64-byte store of the 16 lanes) [dTu.xyz dTv.xyz dTw.xyz | dn.xyz | dopa |
// drgb] and, in a second array, the two floats of the low-pass branch's centre gradient dxy.  The per-Gaussian sums
// are rows of GSR_GROW_FLOATS floats: the 16 columns, dxy, two unused.
#define GSR_GROW_FLOATS 20
#define GSR_GROW_MAIN 16      // floats of a row in the main array
#define GSR_GROW_XY 2         // floats of a row in the xy array
#define GSR_SUBROWS 16        // gradient sub-rows per instance: one per 4x4 pixel block of the 16x16 tile
#define GSR_GR_T 0
#define GSR_GR_NRM 9
#define GSR_GR_OPA 12
#define GSR_GR_RGB 13
#define GSR_GR_XY 16

static inline size_t gsr_align(size_t x) { return (x + 255) & ~size_t(255); }

struct GsrGeomLayout {
    size_t splat, clamped, tiles_touched, tile_rect, depth_key, order, offs, color_jac, total;
    // with_jac = false (forward-only calls, and calls whose colour comes from a cache that carries the Jacobian itself):
    // the last region is not reserved -- 36 B per Gaussian nobody would read or write.  The other offsets do not move.
    explicit GsrGeomLayout(int64_t N, bool with_jac = true) {
        size_t o = 0;
        splat = o;         o += gsr_align(size_t(N) * GSR_SPLAT_FLOATS * 4);
        clamped = o;       o += gsr_align(size_t(N) * 4);
        tiles_touched = o; o += gsr_align(size_t(N) * 4);
        tile_rect = o;     o += gsr_align(size_t(N) * 8);        // (x0 | y0 << 16, w | h << 16) on the tile grid
        depth_key = o;     o += gsr_align(size_t(N) * 4);
        order = o;         o += gsr_align(size_t(N) * 4);        // depth rank -> Gaussian id
        offs = o;          o += gsr_align(size_t(N + 1) * 4);    // depth rank -> first instance (emission order)
        // d(rgb)/d(view direction), 3x3 per Gaussian, left by the SH colour pass so that preprocess_bwd does not have to
        // read the 192 bytes of SH coefficients again for the view-direction term of dL/dmean
        color_jac = o;     o += with_jac ? gsr_align(size_t(N) * 9 * 4) : 0;
        total = o > 0 ? o : 256;
    }
};
size_t gsr_scan_workspace_bytes(int64_t n);
struct GsrBinLayout {
    size_t point_list, inst_row, ranges, covered, touch, slot_cnt, slot_off, row_scan_ws, total;
    // keep_for_backward = false (GSR_FLAG_FORWARD_ONLY): the touch words and the row-count bytes -- the two last regions,
    // 5 B per instance -- are not reserved (the forward-only kernels write neither)
    GsrBinLayout(int64_t D, int64_t tiles, bool keep_for_backward = true) {
        size_t o = 0;
        point_list = o; o += gsr_align(size_t(D) * 4);
        inst_row = o;   o += gsr_align(size_t(D) * 4);
        ranges = o;     o += gsr_align(size_t(tiles) * 8);
        // per (tile, quad): how many entries of the tile's list the forward staged for that quad before all its pixels
        // saturated -- the touch byte of that quad is DEFINED for exactly those entries (nothing behind them was blended,
        // nothing is written there): the backward's bookkeeping visits the walked prefix of a list, not all of it
        covered = o;    o += gsr_align(size_t(tiles) * 16);
        // one byte per (sorted instance, quad): did the forward blend it into >= 1 pixel of that quad?
        // The backward evaluates exactly those pairs (everything else has zero gradient).
        touch = o;      o += keep_for_backward ? gsr_align(size_t(D) * 4) : 0;
        // gradient rows per instance (emission order), one byte each: cleared by the forward's last binning kernel,
        // filled in by render_fwd (last wave of every tile) for the instances somebody walked
        slot_cnt = o;   o += keep_for_backward ? gsr_align(size_t(D)) : 0;
        // first gradient row of every instance (exclusive scan of slot_cnt, D + 1 entries) and the scan's workspace: here
        // rather than in the backward's scratch, so that kernels BETWEEN the forward and the backward can carry the scan
        // as a side job (GsrRowScanJob)
        slot_off = o;   o += keep_for_backward ? gsr_align((size_t(D) + 1) * 4) : 0;
        row_scan_ws = o; o += keep_for_backward ? gsr_scan_workspace_bytes(D > 0 ? D : 1) : 0;
        total = o > 0 ? o : 256;
    }
};
struct GsrImageLayout {
    size_t final_T, n_contrib, total;
    explicit GsrImageLayout(int64_t P) {
        size_t o = 0;
        final_T = o;   o += gsr_align(size_t(P) * 3 * 4);
        n_contrib = o; o += gsr_align(size_t(P) * 2 * 4);
        total = o > 0 ? o : 256;
    }
};

// ---------------------------------------------------------------- device primitives (binning.hip)
// exclusive scan of n u32 values; out[n] receives the total (out has n+1 entries).
// `gather` (may be NULL): in[i] is read as in[gather[i]].
int gsr_exclusive_scan_u32(const uint32_t* in, const uint32_t* gather, uint32_t* out, int64_t n,
                           void* ws, hipStream_t s);
int gsr_exclusive_scan_u8(const uint8_t* in, uint32_t* out, int64_t n, void* ws, hipStream_t s);
size_t gsr_sort_ws_bytes(int64_t n);
// vals2_*: optional second value array carried with the pairs (NULL: none)
int gsr_radix_sort_pairs(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out,
                         uint32_t* vals_out, uint32_t* keys_tmp, uint32_t* vals_tmp, int64_t n,
                         int begin_bit, int end_bit, void* ws, hipStream_t s,
                         const uint32_t* vals2_in = nullptr, uint32_t* vals2_out = nullptr, uint32_t* vals2_tmp = nullptr,
                         bool table_zeroed = false);
// the words of a sort workspace that must be zero when the sort starts (table_zeroed = true: the caller cleared them)
void gsr_sort_zero_region(void* ws, int64_t n, uint32_t** ptr, size_t* words);

// ---------------------------------------------------------------- kernel launchers
// `count_partial` (may be NULL; may be device-mapped host memory): one 64-bit sum of tiles_touched per workgroup,
// gsr_preprocess_fwd_blocks(N) of them; `zero`: words the launch also clears (e.g. gsr_sort_zero_region)
int gsr_preprocess_fwd_blocks(int N);
int gsr_launch_preprocess_fwd(const GsrView& v, const GsrGaussians& g, float* splat,
                              uint32_t* clamped, uint32_t* tiles_touched, uint2* tile_rect, uint32_t* depth_key,
                              int32_t* radii, unsigned long long* count_partial, uint32_t* zero, size_t zero_words,
                              hipStream_t s);
int gsr_launch_preprocess_color(const GsrView& v, const GsrGaussians& g, float* splat, uint32_t* clamped,
                                int32_t* radii, float* color_jac /* [N,9] or NULL */, hipStream_t s);
// does the colour pass of this call leave d(rgb)/d(dir) (forward and backward must agree)?
bool gsr_color_jac_available(const GsrView& v, const GsrGaussians& g);
// 64-bit partial sums of counts[0..N) (<= GSR_COUNT_PARTIALS of them, *n_partial says how many): the host adds them up
#define GSR_COUNT_PARTIALS 256
// (`partial` may be device-mapped host memory; `zero`: words this kernel also clears, e.g. gsr_sort_zero_region)
int gsr_launch_count_partials(const uint32_t* counts, int N, unsigned long long* partial, int* n_partial,
                              uint32_t* zero, size_t zero_words, hipStream_t s);
int gsr_launch_rank_gather_scan(int N, const uint32_t* order, const uint2* tile_rect, uint2* rank_rect, void* scan_ws,
                                hipStream_t s);
int gsr_launch_emit(int N, int grid_x, int grid_y, const uint32_t* order, const void* scan_ws, uint32_t* offs,
                    const uint2* rank_rect, uint32_t* tile_keys, uint32_t* emit_gid,
                    uint32_t* zero_a, size_t words_a, uint32_t* zero_b, size_t words_b, hipStream_t s);
int gsr_launch_finalize_bins(int D, int n_tiles, const uint32_t* tile_keys_sorted, uint32_t* ranges, uint8_t* slot_cnt,
                             hipStream_t s);
int gsr_launch_render_fwd(const GsrView& v, const uint32_t* ranges, const float* splat,
                          float* final_T, uint32_t* n_contrib, float* out_color,
                          float* out_allmap, uint8_t* touch, uint32_t* covered, const uint32_t* inst_row, uint8_t* slot_cnt,
                          const float* feat, const uint32_t* point_list, hipStream_t s);
int gsr_launch_render_bwd(const GsrView& v, const uint32_t* ranges, const uint32_t* covered, const uint32_t* inst_row,
                          const float* splat, const uint32_t* touch, const uint32_t* slot_off, const float* final_T,
                          const uint32_t* n_contrib, const float* dL_dcolor, const float* dL_dallmap, float* grad_rows,
                          float* grad_xy, int N, const uint32_t* offs, uint32_t* row_begin, const float* feat,
                          const uint32_t* point_list, float* feat_rows, size_t max_rows, hipStream_t s);
int gsr_launch_reduce_feat_rows(int N, int C, const uint32_t* order, const uint32_t* row_begin,
                                const float* feat_rows, float* dL_dcolors, hipStream_t s);
int gsr_launch_reduce_rows(int N, const uint32_t* order, const uint32_t* row_begin,
                           const float* grad_rows, const float* grad_xy, float* row_sums, hipStream_t s);
int gsr_launch_preprocess_bwd(const GsrView& v, const GsrGaussians& g, const int32_t* radii,
                              const float* splat, const uint32_t* clamped, const float* row_sums,
                              const float* color_jac /* [N,9] or NULL */, const GsrGrads& out, hipStream_t s);

// XCD-aware tile order for the image-space stencil kernels (loss, regularizer), as in render_fwd / render_bwd: the
// workgroups of a 1-D grid are dealt round-robin to the 8 XCDs, each with its own L2, so workgroup b takes tile
// (b % 8) * per_xcd + b / 8 of the row-major tile list -- every XCD owns one band of tile rows, and the halo lines two
// neighbouring tiles both read are fetched into ONE L2 (in launch order neighbours sat on different XCDs and every halo
// line crossed the fabric twice).  Launch gsr_xcd_tile_grid(gx * gy) workgroups; false: no tile for this workgroup.
static inline unsigned gsr_xcd_tile_grid(int n_tiles) { return 8u * (unsigned)((n_tiles + 7) / 8); }
#ifdef __HIPCC__
__device__ __forceinline__ bool gsr_xcd_tile(int gx, int gy, int& tile_x, int& tile_y) {
    const int n = gx * gy, per_xcd = (n + 7) / 8;
    const int lin = (int)(blockIdx.x & 7u) * per_xcd + (int)(blockIdx.x >> 3);
    if (lin >= n) return false;
    tile_y = lin / gx;
    tile_x = lin - tile_y * gx;
    return true;
}
#endif

// ---------------------------------------------------------------- small device helpers
#ifdef __HIPCC__
// v_rcp_f32 (1 ulp); the parity budget is 1e-4 relative
__device__ __forceinline__ float gsr_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// does the splat's cull rect (packed words wx, wy) overlap the pixel box [bx0, bx0+7] x [by0, by0+7]?
__device__ __forceinline__ bool gsr_rect_overlaps_quad(uint32_t wx, uint32_t wy, int bx0, int by0) {
    const int x0 = (int)(short)(wx & 0xFFFFu), x1 = (int)(short)(wx >> 16);
    const int y0 = (int)(short)(wy & 0xFFFFu), y1 = (int)(short)(wy >> 16);
    return x0 <= bx0 + 7 && x1 >= bx0 && y0 <= by0 + 7 && y1 >= by0;
}
// Second, tighter stage of the quad cull (render_fwd): can the splat reach alpha >= 1/255 anywhere in the pixel box
// [bx0, bx0+7] x [by0, by0+7]?  {alpha >= 1/255} = {rho3d <= c2} u {rho2d <= c2}, c2 = 2 ln(255 opa) (inflated like the
// cull rect of preprocess_fwd):
//   * {rho3d <= c2} is the screen-space image of the splat-space disc of radius sqrt(c2): an ELLIPSE with centre e and
//     shape matrix S (the same conic that gives the rect its extents sqrt(Sxx), sqrt(Syy); the rect ignores Sxy, so
//     for a rotated elongated splat it is mostly empty corners).  Exact box test: min over the box of
//     (x-e)^T adj(S) (x-e) <= det S; the minimiser is the centre (inside), or lies on the box edge x = clamp(e.x) or
//     y = clamp(e.y), where the 1-D minimum is closed form;
//   * {rho2d <= c2} is the low-pass disc of radius sqrt(c2 / 2) around the record's centre.
// Conservative by construction (box grown by 0.5 px + 1 % of the extents; anything degenerate or non-finite answers
// "overlaps"), so culling stays exact: tests/test_gpu_rasterizer.py::test_wave_culling_is_exact.
__device__ __forceinline__ bool gsr_tight_overlaps_quad(const float4 a0, const float4 a1, const float4 a2, float opa,
                                                        int bx0, int by0) {
    const float Tw0 = a1.z, Tw1 = a1.w, Tw2 = a2.x, cx = a2.y, cy = a2.z;
    // pixel frame translated to the record's centre: x - cx = ((Tu - cx Tw) . h) / (Tw . h).  The ellipse centre e is
    // then a few pixels at most and S = e e^T - M loses nothing to cancellation (in the image frame e^2 ~ 1e6 against
    // S ~ 25 costs three digits, enough to turn the thin axis of a needle-like ellipse into noise)
    const float Tu0 = fmaf(-cx, Tw0, a0.x), Tu1 = fmaf(-cx, Tw1, a0.y), Tu2 = fmaf(-cx, Tw2, a0.z);
    const float Tv0 = fmaf(-cy, Tw0, a0.w), Tv1 = fmaf(-cy, Tw1, a1.x), Tv2 = fmaf(-cy, Tw2, a1.y);
    const float c2 = (fmaxf(2.0f * __logf(255.0f * opa), 0.0f) + 0.05f) * 1.02f;
    const float dd = c2 * (Tw0 * Tw0 + Tw1 * Tw1) - Tw2 * Tw2;
    if (!(dd < 0.0f)) return true;
    const float idd = gsr_rcp(dd);
    const float g0 = c2 * idd, g2 = -idd;
    const float ex = g0 * (Tu0 * Tw0 + Tu1 * Tw1) + g2 * Tu2 * Tw2;
    const float ey = g0 * (Tv0 * Tw0 + Tv1 * Tw1) + g2 * Tv2 * Tw2;
    const float Sxx = ex * ex - (g0 * (Tu0 * Tu0 + Tu1 * Tu1) + g2 * Tu2 * Tu2);
    const float Syy = ey * ey - (g0 * (Tv0 * Tv0 + Tv1 * Tv1) + g2 * Tv2 * Tv2);
    const float Sxy = ex * ey - (g0 * (Tu0 * Tv0 + Tu1 * Tv1) + g2 * Tu2 * Tv2);
    const float pxy = Sxx * Syy, pdd = Sxy * Sxy;
    const float det = pxy - pdd;
    if (!(Sxx > 0.0f && Syy > 0.0f && det > 0.0f && pxy < 3.0e37f)) return true;
    const float m = 0.5f + 0.01f * (sqrtf(Sxx) + sqrtf(Syy));
    const float x0 = (float)bx0 - cx - m, x1 = (float)(bx0 + 7) - cx + m;
    const float y0 = (float)by0 - cy - m, y1 = (float)(by0 + 7) - cy + m;
    // ellipse vs box.  det and q are differences of products that nearly cancel for thin ellipses: a pair is culled only
    // when q exceeds det by more than what rounding (a few 1e-7 of the products; 2e-5 is charged) could account for
    const float lim = fmaf(2.0e-5f, pxy + pdd, det * 1.001f);
    const float dx1 = fminf(fmaxf(ex, x0), x1) - ex;
    const float dy1 = fminf(fmaxf(ey + Sxy * gsr_rcp(Sxx) * dx1, y0), y1) - ey;
    const float a1_ = Syy * dx1 * dx1, b1_ = 2.0f * Sxy * dx1 * dy1, c1_ = Sxx * dy1 * dy1;
    const bool out1 = (a1_ - b1_ + c1_) - 2.0e-5f * (a1_ + fabsf(b1_) + c1_) > lim;
    const float dy2 = fminf(fmaxf(ey, y0), y1) - ey;
    const float dx2 = fminf(fmaxf(ex + Sxy * gsr_rcp(Syy) * dy2, x0), x1) - ex;
    const float a2_ = Syy * dx2 * dx2, b2_ = 2.0f * Sxy * dx2 * dy2, c2_ = Sxx * dy2 * dy2;
    const bool out2 = (a2_ - b2_ + c2_) - 2.0e-5f * (a2_ + fabsf(b2_) + c2_) > lim;
    // low-pass disc (centred on the record's centre = the origin of this frame) vs box
    const float ddx = fmaxf(fmaxf(x0, -x1), 0.0f), ddy = fmaxf(fmaxf(y0, -y1), 0.0f);
    const bool disc_out = ddx * ddx + ddy * ddy > 0.5f * c2;
    return !(out1 && out2 && disc_out);
}
__device__ __forceinline__ void gsr_tile_rect(float cx, float cy, int radius, int gx, int gy,
                                              int& x0, int& y0, int& x1, int& y1) {
    // (int) casts truncate toward zero exactly like the recalled getRect
    x0 = min(gx, max(0, (int)((cx - radius) / GSR_TILE)));
    y0 = min(gy, max(0, (int)((cy - radius) / GSR_TILE)));
    x1 = min(gx, max(0, (int)((cx + radius + GSR_TILE - 1) / GSR_TILE)));
    y1 = min(gy, max(0, (int)((cy + radius + GSR_TILE - 1) / GSR_TILE)));
}
#endif
