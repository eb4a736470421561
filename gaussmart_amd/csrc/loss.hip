// Fused photometric loss of the training step:
//     loss = (1 - lambda) * mean|x - y| + lambda * (1 - mean SSIM(x, y))
// with the reference's SSIM (11x11 Gaussian window, sigma 1.5, zero padding, C1 = 0.01^2,
// C2 = 0.03^2; utils/loss_utils.py:16-57, train.py:113-114).  The torch formulation runs five
// depthwise 11x11 convolutions forward and their transposes backward through MIOpen (10 ms per
// step at 1080p, 56 % of the GPU time of the first correct path); here one kernel per direction
// does the separable filtering in LDS.
//
// forward : per 16x16 tile and channel, stage the 26x26 neighbourhood of x and y in LDS, filter
//           (x, y, x^2 + y^2, xy) horizontally then vertically, evaluate SSIM and its partial
//           derivatives w.r.t. (mu1, E[x^2], E[xy]) and write those three maps; per-block sums of
//           SSIM and |x-y| go to a partials array (summed deterministically by the caller).
// backward: dL/dx = -lambda/(CHW) * [ conv(dS/dmu1) + 2x conv(dS/dE[x^2]) + y conv(dS/dE[xy]) ]
//                   + (1-lambda)/(CHW) * sign(x - y), scaled by the incoming scalar gradient.
// HBM-bound streaming: forward reads 8 B and writes 12 B per element, backward reads 20 B and
// writes 4 B.
#include "gsr_common.h"
#include "scan_bodies.h"
#include <cmath>

#define LS_TILE 16                       // granularity of the partials array (shared with regularizer.hip)
#define LS_HALO 5
#define LT 32                            // pixels per side of a workgroup's tile
#define LR (LT + 2 * LS_HALO)            // 42: tile + halo
#define LSTR 44                          // LDS row stride of the staged planes (16-byte aligned rows)
#define LS_C1 0.0001f
#define LS_C2 0.0009f

struct LossWindow { float w[11]; };   // passed by value: lives in the kernarg segment (scalar loads)

// Row-scan side job (include/gsr.h: GsrRowScanJob): workgroups beyond the tiles' carry one half of the exclusive scan of
// the backward's per-instance row counts -- work no kernel between the forward and the backward depends on.
struct RowScanArgs { const uint8_t* counts; uint32_t* partial; uint32_t* slot_off; long long n; int blocks; };
static RowScanArgs make_row_scan(const GsrRowScanJob* job, int want_stage) {
    RowScanArgs a{nullptr, nullptr, nullptr, 0, 0};
    if (job && job->stage == want_stage && job->n > 0 && job->counts && job->slot_off && job->workspace) {
        a.counts = static_cast<const uint8_t*>(job->counts); a.partial = static_cast<uint32_t*>(job->workspace);
        a.slot_off = static_cast<uint32_t*>(job->slot_off); a.n = job->n;
        a.blocks = (int)((job->n + SCAN_TILE - 1) / SCAN_TILE);
    }
    return a;
}

static LossWindow make_window() {
    // the reference builds the 1-D window in fp32 (torch.Tensor of python floats, divided by its sum)
    LossWindow win;
    float gf[11], sf = 0.f;
    for (int i = 0; i < 11; ++i) gf[i] = (float)exp(-double((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5));
    for (int i = 0; i < 11; ++i) sf += gf[i];
    for (int i = 0; i < 11; ++i) win.w[i] = gf[i] / sf;
    return win;
}

// Both kernels: one workgroup (256 threads) per 32x32 tile, channels in a loop (the next channel's tile is fetched into
// registers while the current one is filtered).  Register-blocked separable filter:
//   horizontal: a task = 4 consecutive outputs of one staged row: 16 inputs per plane come in with four ds_read_b128
//               and serve 4 x 11 taps (2.75 LDS reads per output and plane instead of 11), results leave as one
//               ds_write_b128 per filtered quantity (lane t writes floats 4t..4t+3: conflict free);
//   vertical  : a thread = 4 consecutive rows of one column: 14 inputs serve 4 x 11 taps (3.5 reads per output), lanes
//               of a wave read consecutive columns.
// 16x16 tiles with one output per thread and tap-by-tap LDS reads took 84 + 74 us per step at 1080p; this: see
// profiles/.  Partials keep their 16x16 granularity (the regularizer shares the array size): a tile adds its sums
// to the slot of its top-left 16x16 cell and zeros the other three.

// stage a 42x42 window of one plane into registers (zero padding outside the image = the reference's conv2d padding)
#define LS_PER_THREAD ((LR * LR + 255) / 256)

__device__ __forceinline__ void loss_hpass_row4(const float* __restrict__ row, const LossWindow& win, float (&in)[16]) {
    const float4* r4 = reinterpret_cast<const float4*>(row);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float4 v = r4[i]; in[4 * i] = v.x; in[4 * i + 1] = v.y; in[4 * i + 2] = v.z; in[4 * i + 3] = v.w; }
}

__global__ void __launch_bounds__(256, 3) loss_fwd_kernel(const float* __restrict__ img, const float* __restrict__ gt,
                                                       int C, int H, int W, float* __restrict__ maps,
                                                       float* __restrict__ partials, LossWindow win, int tile_wgs,
                                                       RowScanArgs scan, float* __restrict__ invalidate5) {
    __shared__ __attribute__((aligned(16))) float sx[LR][LSTR];
    __shared__ __attribute__((aligned(16))) float sy[LR][LSTR];
    __shared__ __attribute__((aligned(16))) float sh[4][LR][LT];
    __shared__ float red[2][4];
    const int t = threadIdx.x;
    // deferred objective value (gsr_loss_backward_finish writes the five scalars): until then they read NaN, so that a value
    // taken before -- or without -- the backward is visibly invalid (a torch.full for this was a 6 us launch per step)
    if (invalidate5 && blockIdx.x == 0 && t < 5) invalidate5[t] = __builtin_nanf("");
    if ((int)blockIdx.x >= tile_wgs) {          // side job: first half of the row scan, one scan tile per workgroup
        __shared__ uint32_t wt[SCAN_BLOCK / 64];
        const int sb = (int)blockIdx.x - tile_wgs;
        if (sb < scan.blocks) scan_reduce_body<uint8_t>(scan.counts, nullptr, scan.partial, scan.n, sb, wt);
        return;
    }
    int tile_x, tile_y;
    if (!gsr_xcd_tile((W + LT - 1) / LT, (H + LT - 1) / LT, tile_x, tile_y)) return;
    const int x0 = tile_x * LT, y0 = tile_y * LT;
    const size_t HW = (size_t)H * W;
    float ssim_acc = 0.f, l1_acc = 0.f;

    float rx[LS_PER_THREAD], ry[LS_PER_THREAD];
    auto fetch = [&](int c) {
        const float* xi = img + c * HW;
        const float* yi = gt + c * HW;
#pragma unroll
        for (int k = 0; k < LS_PER_THREAD; ++k) {
            const int i = t + k * 256;
            const int r = i / LR, q = i - r * LR;
            const int gy = y0 + r - LS_HALO, gx = x0 + q - LS_HALO;
            const bool ok = i < LR * LR && gy >= 0 && gy < H && gx >= 0 && gx < W;
            rx[k] = 0.f; ry[k] = 0.f;
            if (ok) { rx[k] = xi[(size_t)gy * W + gx]; ry[k] = yi[(size_t)gy * W + gx]; }
        }
    };
    fetch(0);
    const int vc = t & (LT - 1), vr = (t >> 5) * 4;       // vertical pass: column, first of 4 rows
    for (int c = 0; c < C; ++c) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < LS_PER_THREAD; ++k) {
            const int i = t + k * 256;
            if (i < LR * LR) { const int r = i / LR, q = i - r * LR; sx[r][q] = rx[k]; sy[r][q] = ry[k]; }
        }
        __syncthreads();
        if (c + 1 < C) fetch(c + 1);
        // horizontal pass: 42 rows x 8 groups of 4 outputs
        for (int task = t; task < LR * (LT / 4); task += 256) {
            const int r = task >> 3, s4 = (task & 7) * 4;
            float xv[16], yv[16];
            loss_hpass_row4(&sx[r][s4], win, xv);
            loss_hpass_row4(&sy[r][s4], win, yv);
            // sigma_x^2 and sigma_y^2 enter SSIM only through their SUM (B2), and the filter is linear: ONE filtered
            // plane x^2 + y^2 replaces E[x^2] and E[y^2] -- four planes instead of five
            float ss[14], xy[14];
#pragma unroll
            for (int i = 0; i < 14; ++i) { ss[i] = fmaf(xv[i], xv[i], yv[i] * yv[i]); xy[i] = xv[i] * yv[i]; }
            float a[4][4];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) {
                    const float w = win.w[k];
                    a0 += w * xv[o + k]; a1 += w * yv[o + k]; a2 += w * ss[o + k]; a3 += w * xy[o + k];
                }
                a[0][o] = a0; a[1][o] = a1; a[2][o] = a2; a[3][o] = a3;
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
                *reinterpret_cast<float4*>(&sh[m][r][s4]) = make_float4(a[m][0], a[m][1], a[m][2], a[m][3]);
        }
        __syncthreads();
        // vertical pass: 4 rows of one column per thread
        float f[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float v[14];
#pragma unroll
            for (int k = 0; k < 14; ++k) v[k] = sh[m][vr + k][vc];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) acc += win.w[k] * v[o + k];
                f[m][o] = acc;
            }
        }
        const int px = x0 + vc;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int py = y0 + vr + o;
            if (px < W && py < H) {
                const float mu1 = f[0][o], mu2 = f[1][o], ess = f[2][o], e12 = f[3][o];
                const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
                const float s12 = e12 - mu12;
                const float A1 = 2.f * mu12 + LS_C1, A2 = 2.f * s12 + LS_C2;
                const float B1 = mu1_sq + mu2_sq + LS_C1, B2 = (ess - mu1_sq - mu2_sq) + LS_C2;
                // ONE reciprocal (v_rcp_f32 + a Newton step: <= 1 ulp) serves 1/(B1 B2), 1/B1 = B2 inv and 1/B2 = B1 inv;
                // four IEEE divisions here cost ~50 VALU instructions per pixel and channel
                const float den = B1 * B2;
                float inv = gsr_rcp(den);
                inv = fmaf(fmaf(-den, inv, 1.0f), inv, inv);
                const float S = A1 * A2 * inv;
                const size_t o_ = (size_t)py * W + px;
                // partial derivatives with (mu1, E[x^2], E[xy]) as the independent variables
                maps[(size_t)(0 * C + c) * HW + o_] = 2.f * mu2 * (A2 - A1) * inv - 2.f * mu1 * S * ((B2 - B1) * inv);
                maps[(size_t)(1 * C + c) * HW + o_] = -S * (B1 * inv);
                maps[(size_t)(2 * C + c) * HW + o_] = 2.f * A1 * inv;
                ssim_acc += S;
                l1_acc += fabsf(sx[vr + o + LS_HALO][vc + LS_HALO] - sy[vr + o + LS_HALO][vc + LS_HALO]);
            }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { ssim_acc += __shfl_down(ssim_acc, d, 64); l1_acc += __shfl_down(l1_acc, d, 64); }
    if ((t & 63) == 0) { red[0][t >> 6] = ssim_acc; red[1][t >> 6] = l1_acc; }
    __syncthreads();
    if (t < 4) {
        // the tile's sums go to the slot of its top-left 16x16 cell; the other cells it covers get zeros
        const int g16x = (W + LS_TILE - 1) / LS_TILE, g16y = (H + LS_TILE - 1) / LS_TILE;
        const int cx = 2 * tile_x + (t & 1), cy = 2 * tile_y + (t >> 1);
        if (cx < g16x && cy < g16y) {
            const int b = cy * g16x + cx;
            partials[2 * b + 0] = t == 0 ? (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]) : 0.f;
            partials[2 * b + 1] = t == 0 ? (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]) : 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Objective finish: turns the per-block partials of loss_fwd (and optionally reg_fwd) into the
// five scalars of the training objective in ONE tiny launch, instead of ~20 zero-dim torch kernels:
//   out[0] = total = (1-l) * l1 + l * (1 - ssim) + ln * normal + ld * dist
//   out[1] = l1, out[2] = ssim, out[3] = mean normal error, out[4] = mean distortion
// Fixed summation order (single workgroup, strided chunks, tree) => reproducible loss value.
struct FinishArgs {
    const float* pl; int n_pl; float inv_n_img;       // loss partials (pairs), their count, 1 / (C H W)
    const float* pr; int n_pr; float inv_n_pix;       // regularizer partials (pairs) or NULL / 0, 1 / (H W)
    float lambda_dssim, lambda_normal, lambda_dist;
    float* out;                                       // f32[5]; NULL: nothing to finish
};

// one 256-thread workgroup
__device__ __forceinline__ void objective_finish_body(const FinishArgs& f) {
    const float* __restrict__ pl = f.pl; const float* __restrict__ pr = f.pr;
    const int n_pl = f.n_pl, n_pr = f.n_pr;
    const float inv_n_img = f.inv_n_img, inv_n_pix = f.inv_n_pix;
    const float lambda_dssim = f.lambda_dssim, lambda_normal = f.lambda_normal, lambda_dist = f.lambda_dist;
    float* __restrict__ out = f.out;
    __shared__ float red[4][256];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    // eight independent 8-byte loads in flight per thread and round (one at a time made this 1-block kernel wait
    // ~30 memory latencies in a row); the per-thread summation order is unchanged
    const float2* pl2 = reinterpret_cast<const float2*>(pl);
    const float2* pr2 = reinterpret_cast<const float2*>(pr);
    for (int base = threadIdx.x; base < n_pl; base += 256 * 8) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = base + 256 * u; v[u] = i < n_pl ? pl2[i] : make_float2(0.f, 0.f); }
#pragma unroll
        for (int u = 0; u < 8; ++u) { s0 += v[u].x; s1 += v[u].y; }
    }
    for (int base = threadIdx.x; base < n_pr; base += 256 * 8) {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = base + 256 * u; v[u] = i < n_pr ? pr2[i] : make_float2(0.f, 0.f); }
#pragma unroll
        for (int u = 0; u < 8; ++u) { s2 += v[u].x; s3 += v[u].y; }
    }
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2; red[3][threadIdx.x] = s3;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d) {
#pragma unroll
            for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + d];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float ssim = red[0][0] * inv_n_img, l1 = red[1][0] * inv_n_img;
        const float nrm = red[2][0] * inv_n_pix, dst = red[3][0] * inv_n_pix;
        out[0] = (1.0f - lambda_dssim) * l1 + lambda_dssim * (1.0f - ssim) + lambda_normal * nrm + lambda_dist * dst;
        out[1] = l1; out[2] = ssim; out[3] = nrm; out[4] = dst;
    }
}

__global__ void __launch_bounds__(256) objective_finish_kernel(FinishArgs f) { objective_finish_body(f); }

__global__ void __launch_bounds__(256, 3) loss_bwd_kernel(const float* __restrict__ img, const float* __restrict__ gt,
                                                       const float* __restrict__ maps, int C, int H, int W,
                                                       float lambda, const float* __restrict__ grad_scale,
                                                       float* __restrict__ dimg, LossWindow win, FinishArgs fin, int tile_wgs,
                                                       RowScanArgs scan) {
    __shared__ __attribute__((aligned(16))) float sm[3][LR][LSTR];
    __shared__ __attribute__((aligned(16))) float sh[3][LR][LT];
    const int t = threadIdx.x;
    if ((int)blockIdx.x >= tile_wgs) {
        // gsr_loss_backward_finish: workgroups beyond the tiles' carry work this kernel does not depend on instead of it
        // being launches of its own -- the first turns the forward's partials into the five scalars of the objective,
        // the others run the second half of the row scan of the rasterizer's backward
        const int sb = (int)blockIdx.x - tile_wgs;
        if (sb == 0) {
            if (fin.out != nullptr) objective_finish_body(fin);
        } else if (sb - 1 < scan.blocks) {
            __shared__ uint32_t wt[SCAN_BLOCK / 64];
            scan_apply_body<uint8_t>(scan.counts, nullptr, scan.partial, scan.slot_off, scan.n, sb - 1, wt);
        }
        return;
    }
    int tile_x, tile_y;
    if (!gsr_xcd_tile((W + LT - 1) / LT, (H + LT - 1) / LT, tile_x, tile_y)) return;
    const int x0 = tile_x * LT, y0 = tile_y * LT;
    const size_t HW = (size_t)H * W;
    const float gs = grad_scale[0];
    const float inv_n = 1.0f / ((float)C * (float)H * (float)W);
    const float k_ssim = -lambda * inv_n * gs, k_l1 = (1.0f - lambda) * inv_n * gs;

    float rm[3][LS_PER_THREAD];
    auto fetch = [&](int c) {
#pragma unroll
        for (int k = 0; k < LS_PER_THREAD; ++k) {
            const int i = t + k * 256;
            const int r = i / LR, q = i - r * LR;
            const int gy = y0 + r - LS_HALO, gx = x0 + q - LS_HALO;
            const bool ok = i < LR * LR && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const size_t o = (size_t)gy * W + gx;
#pragma unroll
            for (int m = 0; m < 3; ++m) { rm[m][k] = 0.f; if (ok) rm[m][k] = maps[(size_t)(m * C + c) * HW + o]; }
        }
    };
    fetch(0);
    const int vc = t & (LT - 1), vr = (t >> 5) * 4;
    const int px = x0 + vc;
    for (int c = 0; c < C; ++c) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < LS_PER_THREAD; ++k) {
            const int i = t + k * 256;
            if (i < LR * LR) {
                const int r = i / LR, q = i - r * LR;
                sm[0][r][q] = rm[0][k]; sm[1][r][q] = rm[1][k]; sm[2][r][q] = rm[2][k];
            }
        }
        __syncthreads();
        if (c + 1 < C) fetch(c + 1);
        // this thread's image / target values are only needed after both filter passes: request them now
        float xv[4], yv[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int py = y0 + vr + o;
            xv[o] = 0.f; yv[o] = 0.f;
            if (px < W && py < H) { const size_t o_ = (size_t)py * W + px; xv[o] = img[c * HW + o_]; yv[o] = gt[c * HW + o_]; }
        }
        for (int task = t; task < LR * (LT / 4); task += 256) {
            const int r = task >> 3, s4 = (task & 7) * 4;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                float in[16];
                loss_hpass_row4(&sm[m][r][s4], win, in);
                float a[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < 11; ++k) acc += win.w[k] * in[o + k];
                    a[o] = acc;
                }
                *reinterpret_cast<float4*>(&sh[m][r][s4]) = make_float4(a[0], a[1], a[2], a[3]);
            }
        }
        __syncthreads();
        float g[3][4];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            float v[14];
#pragma unroll
            for (int k = 0; k < 14; ++k) v[k] = sh[m][vr + k][vc];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) acc += win.w[k] * v[o + k];
                g[m][o] = acc;
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int py = y0 + vr + o;
            if (px < W && py < H) {
                const float d = xv[o] - yv[o];
                const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                dimg[c * HW + (size_t)py * W + px] = k_ssim * (g[0][o] + 2.f * xv[o] * g[1][o] + yv[o] * g[2][o]) + k_l1 * sgn;
            }
        }
    }
}

extern "C" int32_t gsr_loss_num_partials(int32_t H, int32_t W) {
    return 2 * ((W + LS_TILE - 1) / LS_TILE) * ((H + LS_TILE - 1) / LS_TILE);
}

extern "C" int32_t gsr_loss_forward(const float* img, const float* gt, int32_t C, int32_t H, int32_t W,
                                    float* maps, float* partials, gsr_stream_t stream_) {
    return gsr_loss_forward_job(img, gt, C, H, W, maps, partials, nullptr, nullptr, stream_);
}

extern "C" int32_t gsr_loss_forward_job(const float* img, const float* gt, int32_t C, int32_t H, int32_t W,
                                        float* maps, float* partials, GsrRowScanJob* job, float* out5_invalidate,
                                        gsr_stream_t stream_) {
    if (!img || !gt || !maps || !partials || C <= 0 || H <= 0 || W <= 0) { gsr_set_error("bad loss_forward arguments"); return GSR_E_INVALID; }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    GsrProfileScope prof(GSR_K_LOSS_FWD, s);
    const unsigned tile_wgs = gsr_xcd_tile_grid(((W + LT - 1) / LT) * ((H + LT - 1) / LT));
    const RowScanArgs scan = make_row_scan(job, 0);
    hipLaunchKernelGGL(loss_fwd_kernel, dim3(tile_wgs + (unsigned)scan.blocks), dim3(256), 0, s, img, gt, C, H, W, maps, partials,
                       make_window(), (int)tile_wgs, scan, out5_invalidate);
    GSR_LAUNCH_CHECK();
    if (scan.blocks > 0) job->stage = 1;
    return GSR_OK;
}

extern "C" int32_t gsr_loss_backward(const float* img, const float* gt, const float* maps, int32_t C, int32_t H,
                                     int32_t W, float lambda_dssim, const float* grad_scale, float* dimg,
                                     gsr_stream_t stream_) {
    if (!img || !gt || !maps || !grad_scale || !dimg || C <= 0 || H <= 0 || W <= 0) { gsr_set_error("bad loss_backward arguments"); return GSR_E_INVALID; }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    GsrProfileScope prof(GSR_K_LOSS_BWD, s);
    const unsigned tile_wgs = gsr_xcd_tile_grid(((W + LT - 1) / LT) * ((H + LT - 1) / LT));
    FinishArgs none{};
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(tile_wgs), dim3(256), 0, s, img, gt, maps, C, H, W, lambda_dssim, grad_scale, dimg,
                       make_window(), none, (int)tile_wgs, make_row_scan(nullptr, 0));
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}

static FinishArgs make_finish(const float* loss_partials, int C, int H, int W, const float* reg_partials, float lambda_dssim,
                              float lambda_normal, float lambda_dist, float* out5) {
    FinishArgs f;
    const int n = gsr_loss_num_partials(H, W) / 2;
    f.pl = loss_partials; f.n_pl = n; f.inv_n_img = 1.0f / ((float)C * (float)H * (float)W);
    f.pr = reg_partials; f.n_pr = reg_partials ? n : 0; f.inv_n_pix = 1.0f / ((float)H * (float)W);
    f.lambda_dssim = lambda_dssim; f.lambda_normal = reg_partials ? lambda_normal : 0.f;
    f.lambda_dist = reg_partials ? lambda_dist : 0.f; f.out = out5;
    return f;
}

extern "C" int32_t gsr_loss_backward_finish(const float* img, const float* gt, const float* maps, int32_t C, int32_t H,
                                            int32_t W, float lambda_dssim, const float* grad_scale, float* dimg,
                                            const float* loss_partials, const float* reg_partials, float lambda_normal,
                                            float lambda_dist, float* out5, GsrRowScanJob* job, gsr_stream_t stream_) {
    if (!img || !gt || !maps || !grad_scale || !dimg || (out5 && !loss_partials) || C <= 0 || H <= 0 || W <= 0) {
        gsr_set_error("bad loss_backward_finish arguments");
        return GSR_E_INVALID;
    }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    GsrProfileScope prof(GSR_K_LOSS_BWD, s);
    const unsigned tile_wgs = gsr_xcd_tile_grid(((W + LT - 1) / LT) * ((H + LT - 1) / LT));
    const RowScanArgs scan = make_row_scan(job, 1);
    FinishArgs fin{};
    if (out5) fin = make_finish(loss_partials, C, H, W, reg_partials, lambda_dssim, lambda_normal, lambda_dist, out5);
    const unsigned extra = (out5 || scan.blocks > 0) ? 1u + (unsigned)scan.blocks : 0u;   // [finish][scan tiles...]
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(tile_wgs + extra), dim3(256), 0, s, img, gt, maps, C, H, W, lambda_dssim, grad_scale,
                       dimg, make_window(), fin, (int)tile_wgs, scan);
    GSR_LAUNCH_CHECK();
    if (scan.blocks > 0) job->stage = 2;
    return GSR_OK;
}

extern "C" int32_t gsr_objective_finish(const float* loss_partials, int32_t C, int32_t H, int32_t W,
                                        const float* reg_partials, float lambda_dssim, float lambda_normal,
                                        float lambda_dist, float* out5, gsr_stream_t stream_) {
    if (!loss_partials || !out5 || C <= 0 || H <= 0 || W <= 0) { gsr_set_error("bad objective_finish arguments"); return GSR_E_INVALID; }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(objective_finish_kernel, dim3(1), dim3(256), 0, s,
                       make_finish(loss_partials, C, H, W, reg_partials, lambda_dssim, lambda_normal, lambda_dist, out5));
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}
