// Fused photometric loss of the training step:
//     loss = (1 - lambda) * mean|x - y| + lambda * (1 - mean SSIM(x, y))
// with the reference's SSIM (11x11 Gaussian window, sigma 1.5, zero padding, C1 = 0.01^2,
// C2 = 0.03^2; utils/loss_utils.py:16-57, train.py:113-114).  The torch formulation runs five
// depthwise 11x11 convolutions forward and their transposes backward through MIOpen (10 ms per
// step at 1080p, 56 % of the GPU time of the first correct path); here one kernel per direction
// does the separable filtering in LDS.
//
// forward : per 16x16 tile and channel, stage the 26x26 neighbourhood of x and y in LDS, filter
//           (x, y, x^2, y^2, xy) horizontally then vertically, evaluate SSIM and its partial
//           derivatives w.r.t. (mu1, E[x^2], E[xy]) and write those three maps; per-block sums of
//           SSIM and |x-y| go to a partials array (summed deterministically by the caller).
// backward: dL/dx = -lambda/(CHW) * [ conv(dS/dmu1) + 2x conv(dS/dE[x^2]) + y conv(dS/dE[xy]) ]
//                   + (1-lambda)/(CHW) * sign(x - y), scaled by the incoming scalar gradient.
// HBM-bound streaming: forward reads 8 B and writes 12 B per element, backward reads 20 B and
// writes 4 B.
#include "gsr_common.h"
#include <cmath>

#define LS_TILE 16
#define LS_HALO 5
#define LS_REG (LS_TILE + 2 * LS_HALO)   // 26
#define LS_C1 0.0001f
#define LS_C2 0.0009f

struct LossWindow { float w[11]; };   // passed by value: lives in the kernarg segment (scalar loads)

static LossWindow make_window() {
    // the reference builds the 1-D window in fp32 (torch.Tensor of python floats, divided by its sum)
    LossWindow win;
    float gf[11], sf = 0.f;
    for (int i = 0; i < 11; ++i) gf[i] = (float)exp(-double((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5));
    for (int i = 0; i < 11; ++i) sf += gf[i];
    for (int i = 0; i < 11; ++i) win.w[i] = gf[i] / sf;
    return win;
}

__global__ void __launch_bounds__(256) loss_fwd_kernel(const float* __restrict__ img, const float* __restrict__ gt,
                                                       int C, int H, int W, float* __restrict__ maps,
                                                       float* __restrict__ partials, LossWindow win) {
    __shared__ float sx[LS_REG][LS_REG + 1], sy[LS_REG][LS_REG + 1];
    __shared__ float sh[5][LS_REG][LS_TILE + 1];
    __shared__ float red[2][4];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x0 = blockIdx.x * LS_TILE, y0 = blockIdx.y * LS_TILE;
    const int px = x0 + tx, py = y0 + ty;
    const bool inside = px < W && py < H;
    const size_t HW = (size_t)H * W;
    float ssim_acc = 0.f, l1_acc = 0.f;

    // tile + halo of one channel = 676 values per image: 3 per thread (the last pass partly idle);
    // the NEXT channel's values are fetched into registers while the current one is filtered
    constexpr int LS_PER_THREAD = (LS_REG * LS_REG + 255) / 256;
    float rx[LS_PER_THREAD], ry[LS_PER_THREAD];
    auto fetch = [&](int c) {
        const float* xi = img + c * HW;
        const float* yi = gt + c * HW;
#pragma unroll
        for (int k = 0; k < LS_PER_THREAD; ++k) {
            const int i = threadIdx.x + k * 256;
            const int r = i / LS_REG, q = i - r * LS_REG;
            const int gy = y0 + r - LS_HALO, gx = x0 + q - LS_HALO;
            const bool ok = i < LS_REG * LS_REG && gy >= 0 && gy < H && gx >= 0 && gx < W;
            rx[k] = 0.f; ry[k] = 0.f;
            if (ok) { rx[k] = xi[(size_t)gy * W + gx]; ry[k] = yi[(size_t)gy * W + gx]; }
        }
    };
    fetch(0);
    for (int c = 0; c < C; ++c) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < LS_PER_THREAD; ++k) {
            const int i = threadIdx.x + k * 256;
            if (i < LS_REG * LS_REG) { const int r = i / LS_REG, q = i - r * LS_REG; sx[r][q] = rx[k]; sy[r][q] = ry[k]; }
        }
        __syncthreads();
        if (c + 1 < C) fetch(c + 1);
        // horizontal pass: 26 rows x 16 columns
        for (int i = threadIdx.x; i < LS_REG * LS_TILE; i += 256) {
            const int r = i / LS_TILE, q = i - r * LS_TILE;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float w = win.w[k], xv = sx[r][q + k], yv = sy[r][q + k];
                a0 += w * xv; a1 += w * yv; a2 += w * xv * xv; a3 += w * yv * yv; a4 += w * xv * yv;
            }
            sh[0][r][q] = a0; sh[1][r][q] = a1; sh[2][r][q] = a2; sh[3][r][q] = a3; sh[4][r][q] = a4;
        }
        __syncthreads();
        float mu1 = 0.f, mu2 = 0.f, e1 = 0.f, e2 = 0.f, e12 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float w = win.w[k];
            mu1 += w * sh[0][ty + k][tx]; mu2 += w * sh[1][ty + k][tx];
            e1 += w * sh[2][ty + k][tx]; e2 += w * sh[3][ty + k][tx]; e12 += w * sh[4][ty + k][tx];
        }
        if (inside) {
            const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
            const float s1 = e1 - mu1_sq, s2 = e2 - mu2_sq, s12 = e12 - mu12;
            const float A1 = 2.f * mu12 + LS_C1, A2 = 2.f * s12 + LS_C2;
            const float B1 = mu1_sq + mu2_sq + LS_C1, B2 = s1 + s2 + LS_C2;
            const float inv = 1.0f / (B1 * B2);
            const float S = A1 * A2 * inv;
            const size_t o = (size_t)py * W + px;
            // partial derivatives with (mu1, E[x^2], E[xy]) as the independent variables
            maps[(size_t)(0 * C + c) * HW + o] = 2.f * mu2 * (A2 - A1) * inv - 2.f * mu1 * S * (1.f / B1 - 1.f / B2);
            maps[(size_t)(1 * C + c) * HW + o] = -S / B2;
            maps[(size_t)(2 * C + c) * HW + o] = 2.f * A1 * inv;
            ssim_acc += S;
            l1_acc += fabsf(sx[ty + LS_HALO][tx + LS_HALO] - sy[ty + LS_HALO][tx + LS_HALO]);
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { ssim_acc += __shfl_down(ssim_acc, d, 64); l1_acc += __shfl_down(l1_acc, d, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ssim_acc; red[1][threadIdx.x >> 6] = l1_acc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int b = blockIdx.y * gridDim.x + blockIdx.x;
        partials[2 * b + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partials[2 * b + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ void __launch_bounds__(256) loss_bwd_kernel(const float* __restrict__ img, const float* __restrict__ gt,
                                                       const float* __restrict__ maps, int C, int H, int W,
                                                       float lambda, const float* __restrict__ grad_scale,
                                                       float* __restrict__ dimg, LossWindow win) {
    __shared__ float sm[3][LS_REG][LS_REG + 1];
    __shared__ float sh[3][LS_REG][LS_TILE + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x0 = blockIdx.x * LS_TILE, y0 = blockIdx.y * LS_TILE;
    const int px = x0 + tx, py = y0 + ty;
    const bool inside = px < W && py < H;
    const size_t HW = (size_t)H * W;
    const float gs = grad_scale[0];
    const float inv_n = 1.0f / ((float)C * (float)H * (float)W);
    const float k_ssim = -lambda * inv_n * gs, k_l1 = (1.0f - lambda) * inv_n * gs;

    constexpr int LS_PER_THREAD = (LS_REG * LS_REG + 255) / 256;
    float rm[3][LS_PER_THREAD];
    auto fetch = [&](int c) {
#pragma unroll
        for (int k = 0; k < LS_PER_THREAD; ++k) {
            const int i = threadIdx.x + k * 256;
            const int r = i / LS_REG, q = i - r * LS_REG;
            const int gy = y0 + r - LS_HALO, gx = x0 + q - LS_HALO;
            const bool ok = i < LS_REG * LS_REG && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const size_t o = (size_t)gy * W + gx;
#pragma unroll
            for (int m = 0; m < 3; ++m) { rm[m][k] = 0.f; if (ok) rm[m][k] = maps[(size_t)(m * C + c) * HW + o]; }
        }
    };
    fetch(0);
    for (int c = 0; c < C; ++c) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < LS_PER_THREAD; ++k) {
            const int i = threadIdx.x + k * 256;
            if (i < LS_REG * LS_REG) {
                const int r = i / LS_REG, q = i - r * LS_REG;
                sm[0][r][q] = rm[0][k]; sm[1][r][q] = rm[1][k]; sm[2][r][q] = rm[2][k];
            }
        }
        __syncthreads();
        if (c + 1 < C) fetch(c + 1);
        // this pixel's image / target values are only needed after both filter passes: request them now
        float xv = 0.f, yv = 0.f;
        if (inside) { const size_t o = (size_t)py * W + px; xv = img[c * HW + o]; yv = gt[c * HW + o]; }
        for (int i = threadIdx.x; i < LS_REG * LS_TILE; i += 256) {
            const int r = i / LS_TILE, q = i - r * LS_TILE;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float w = win.w[k];
                a0 += w * sm[0][r][q + k]; a1 += w * sm[1][r][q + k]; a2 += w * sm[2][r][q + k];
            }
            sh[0][r][q] = a0; sh[1][r][q] = a1; sh[2][r][q] = a2;
        }
        __syncthreads();
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float w = win.w[k];
            g0 += w * sh[0][ty + k][tx]; g1 += w * sh[1][ty + k][tx]; g2 += w * sh[2][ty + k][tx];
        }
        if (inside) {
            const size_t o = (size_t)py * W + px;
            const float d = xv - yv;
            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            dimg[c * HW + o] = k_ssim * (g0 + 2.f * xv * g1 + yv * g2) + k_l1 * sgn;
        }
    }
}

extern "C" int32_t gsr_loss_num_partials(int32_t H, int32_t W) {
    return 2 * ((W + LS_TILE - 1) / LS_TILE) * ((H + LS_TILE - 1) / LS_TILE);
}

extern "C" int32_t gsr_loss_forward(const float* img, const float* gt, int32_t C, int32_t H, int32_t W,
                                    float* maps, float* partials, gsr_stream_t stream_) {
    if (!img || !gt || !maps || !partials || C <= 0 || H <= 0 || W <= 0) { gsr_set_error("bad loss_forward arguments"); return GSR_E_INVALID; }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    GsrProfileScope prof(GSR_K_LOSS_FWD, s);
    dim3 grid((W + LS_TILE - 1) / LS_TILE, (H + LS_TILE - 1) / LS_TILE);
    hipLaunchKernelGGL(loss_fwd_kernel, grid, dim3(256), 0, s, img, gt, C, H, W, maps, partials, make_window());
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}

extern "C" int32_t gsr_loss_backward(const float* img, const float* gt, const float* maps, int32_t C, int32_t H,
                                     int32_t W, float lambda_dssim, const float* grad_scale, float* dimg,
                                     gsr_stream_t stream_) {
    if (!img || !gt || !maps || !grad_scale || !dimg || C <= 0 || H <= 0 || W <= 0) { gsr_set_error("bad loss_backward arguments"); return GSR_E_INVALID; }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    GsrProfileScope prof(GSR_K_LOSS_BWD, s);
    dim3 grid((W + LS_TILE - 1) / LS_TILE, (H + LS_TILE - 1) / LS_TILE);
    hipLaunchKernelGGL(loss_bwd_kernel, grid, dim3(256), 0, s, img, gt, maps, C, H, W, lambda_dssim, grad_scale, dimg, make_window());
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}

// ------------------------------------------------------------------------------------------------
// Objective finish: turns the per-block partials of loss_fwd (and optionally reg_fwd) into the
// five scalars of the training objective in ONE tiny launch, instead of ~20 zero-dim torch kernels:
//   out[0] = total = (1-l) * l1 + l * (1 - ssim) + ln * normal + ld * dist
//   out[1] = l1, out[2] = ssim, out[3] = mean normal error, out[4] = mean distortion
// Fixed summation order (single workgroup, strided chunks, tree) => reproducible loss value.
__global__ void __launch_bounds__(256) objective_finish_kernel(const float* __restrict__ pl, int n_pl, float inv_n_img,
                                                               const float* __restrict__ pr, int n_pr, float inv_n_pix,
                                                               float lambda_dssim, float lambda_normal, float lambda_dist,
                                                               float* __restrict__ out) {
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

extern "C" int32_t gsr_objective_finish(const float* loss_partials, int32_t C, int32_t H, int32_t W,
                                        const float* reg_partials, float lambda_dssim, float lambda_normal,
                                        float lambda_dist, float* out5, gsr_stream_t stream_) {
    if (!loss_partials || !out5 || C <= 0 || H <= 0 || W <= 0) { gsr_set_error("bad objective_finish arguments"); return GSR_E_INVALID; }
    const int n = gsr_loss_num_partials(H, W) / 2;
    hipStream_t s = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(objective_finish_kernel, dim3(1), dim3(256), 0, s, loss_partials, n,
                       1.0f / ((float)C * (float)H * (float)W), reg_partials, reg_partials ? n : 0,
                       1.0f / ((float)H * (float)W), lambda_dssim, reg_partials ? lambda_normal : 0.f,
                       reg_partials ? lambda_dist : 0.f, out5);
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}
