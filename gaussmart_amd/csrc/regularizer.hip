// Fused surface regularizers of the training step: everything the
// reference does between the rasterizer's `allmap` and the two scalar regularization losses,
//     normal_loss = lambda_normal * mean(1 - sum_k rend_normal_k * surf_normal_k)
//     dist_loss   = lambda_dist   * mean(allmap[6])
// (gaussian_renderer/__init__.py:117-156, utils/point_utils.py:9-37, train.py:132-140), in one
// kernel per direction instead of ~70 elementwise / GEMM / reduction launches over 2 M pixels.
//
//   surf_depth  = (1-r) * nan_to_num(allmap[0] / allmap[1]) + r * nan_to_num(allmap[5])
//   P(x, y)     = surf_depth * K^-1 [x, y, 1]          (camera space; the dot product with the
//                                                        rendered normal is rotation invariant, so
//                                                        the reference's world-space detour drops out)
//   surf_normal = normalize((P[y+1,x]-P[y-1,x]) x (P[y,x+1]-P[y,x-1])) * alpha.detach(), 0 on the border
//   error       = 1 - allmap[2:5] . surf_normal
// Backward gathers: d surf_depth(p) collects from the 4 neighbours whose finite difference touches
// p; each neighbour's contribution needs ITS neighbours' points, i.e. a radius-2 stencil staged in
// LDS.  HBM-bound streaming (28 B read + 8 B of partials forward; 28 B read + 28 B written backward).
#include "gsr_common.h"

#define RG_T 16

struct RegParams {
    int W, H;
    float depth_ratio;
    float m[9];            // K^-1 (row-major): ray(x, y) = m * [x, y, 1]
};

__device__ __forceinline__ float rg_nan_to_num(float v) {   // torch.nan_to_num(v, 0, 0)
    return (isnan(v) || (isinf(v) && v > 0.f)) ? 0.f : v;
}
__device__ __forceinline__ float rg_surf_depth(const float* __restrict__ am, size_t HW, size_t o, float ratio) {
    const float D = am[o], A = am[HW + o], med = am[5 * HW + o];
    return (1.f - ratio) * rg_nan_to_num(D / A) + ratio * rg_nan_to_num(med);
}
__device__ __forceinline__ float3 rg_ray(const RegParams& p, int x, int y) {
    const float fx = (float)x, fy = (float)y;
    return make_float3(p.m[0] * fx + p.m[1] * fy + p.m[2], p.m[3] * fx + p.m[4] * fy + p.m[5],
                       p.m[6] * fx + p.m[7] * fy + p.m[8]);
}
__device__ __forceinline__ float3 rg_cross(float3 a, float3 b) {
    return make_float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// stage surf_depth of the tile plus `halo` pixels into LDS (0 outside the image).  (16 + 2 HALO)^2 <= 512 elements:
// every thread issues the loads of BOTH its elements before the first LDS write (they are independent; one element at
// a time left the block waiting two memory latencies in a row).
template <int HALO>
__device__ __forceinline__ void rg_stage_depth(const RegParams& p, const float* __restrict__ am, float (*sd)[RG_T + 2 * HALO + 1],
                                               int x0, int y0) {
    constexpr int R = RG_T + 2 * HALO;
    static_assert(R * R <= 512, "two elements per thread");
    const size_t HW = (size_t)p.W * p.H;
    float D[2], A[2], M[2];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = threadIdx.x + 256 * u;
        const int r = i / R, q = i - r * R;
        const int gy = y0 + r - HALO, gx = x0 + q - HALO;
        ok[u] = i < R * R && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        D[u] = 0.f; A[u] = 1.f; M[u] = 0.f;
        if (ok[u]) {
            const size_t o = (size_t)gy * p.W + gx;
            D[u] = am[o]; A[u] = am[HW + o]; M[u] = am[5 * HW + o];
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = threadIdx.x + 256 * u;
        if (i < R * R) {
            const int r = i / R, q = i - r * R;
            sd[r][q] = ok[u] ? (1.f - p.depth_ratio) * rg_nan_to_num(D[u] / A[u]) + p.depth_ratio * rg_nan_to_num(M[u]) : 0.f;
        }
    }
}

__global__ void __launch_bounds__(256) reg_fwd_kernel(RegParams p, const float* __restrict__ am,
                                                      float* __restrict__ partials) {
    __shared__ float sd[RG_T + 2][RG_T + 3];
    __shared__ float red[2][4];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int tiles_x = (p.W + RG_T - 1) / RG_T;
    int tile_x, tile_y;
    if (!gsr_xcd_tile(tiles_x, (p.H + RG_T - 1) / RG_T, tile_x, tile_y)) return;
    const int x0 = tile_x * RG_T, y0 = tile_y * RG_T;
    const int x = x0 + tx, y = y0 + ty;
    const size_t HW = (size_t)p.W * p.H;
    // this pixel's own planes are requested before the staging loads are waited for
    const bool in_img = x < p.W && y < p.H;
    const bool interior = in_img && x >= 1 && x <= p.W - 2 && y >= 1 && y <= p.H - 2;
    const size_t o = in_img ? (size_t)y * p.W + x : 0;
    float alpha = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f, dist = 0.f;
    if (interior) { alpha = am[HW + o]; n0 = am[2 * HW + o]; n1 = am[3 * HW + o]; n2 = am[4 * HW + o]; }
    if (in_img) dist = am[6 * HW + o];
    rg_stage_depth<1>(p, am, sd, x0, y0);
    __syncthreads();
    float err = 0.f;
    if (in_img) {
        float dotp = 0.f;
        if (interior) {
            const float3 ru = rg_ray(p, x, y + 1), rd = rg_ray(p, x, y - 1), rr = rg_ray(p, x + 1, y), rl = rg_ray(p, x - 1, y);
            const float du = sd[ty + 2][tx + 1], dd = sd[ty][tx + 1], dr = sd[ty + 1][tx + 2], dl = sd[ty + 1][tx];
            const float3 dx = make_float3(du * ru.x - dd * rd.x, du * ru.y - dd * rd.y, du * ru.z - dd * rd.z);
            const float3 dy = make_float3(dr * rr.x - dl * rl.x, dr * rr.y - dl * rl.y, dr * rr.z - dl * rl.z);
            const float3 c = rg_cross(dx, dy);
            const float inv = 1.0f / fmaxf(sqrtf(c.x * c.x + c.y * c.y + c.z * c.z), 1e-12f);
            dotp = (n0 * c.x + n1 * c.y + n2 * c.z) * inv * alpha;
        }
        err = 1.0f - dotp;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { err += __shfl_down(err, d, 64); dist += __shfl_down(dist, d, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = err; red[1][threadIdx.x >> 6] = dist; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int b = tile_y * tiles_x + tile_x;
        partials[2 * b + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partials[2 * b + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

__global__ void __launch_bounds__(256) reg_bwd_kernel(RegParams p, const float* __restrict__ am,
                                                      float lambda_normal, float lambda_dist,
                                                      const float* __restrict__ grad_scale,
                                                      float* __restrict__ dam, float* __restrict__ partials) {
    __shared__ float sd[RG_T + 4][RG_T + 5];          // surf_depth, halo 2
    __shared__ float sg[6][RG_T + 2][RG_T + 3];       // dL/ddx, dL/ddy of every pixel, halo 1
    __shared__ float sn_in[3][RG_T + 2][RG_T + 3];    // rendered normal of every pixel, halo 1 (only read with `partials`)
    __shared__ float red[2][4];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int tiles_x = (p.W + RG_T - 1) / RG_T;
    int tile_x, tile_y;
    if (!gsr_xcd_tile(tiles_x, (p.H + RG_T - 1) / RG_T, tile_x, tile_y)) return;
    const int x0 = tile_x * RG_T, y0 = tile_y * RG_T;
    const size_t HW = (size_t)p.W * p.H;
    const float inv_n = 1.0f / ((float)p.W * (float)p.H);
    const float gs = grad_scale[0];
    const float kn = lambda_normal * inv_n * gs, kd = lambda_dist * inv_n * gs;

    // phase-1 items of this thread (tile + halo 1 = 324 pixels: at most two per thread) and the final pixel: their
    // planes are requested before the staging loads are waited for
    constexpr int R1 = RG_T + 2;
    float p1_alpha[2], p1_n0[2], p1_n1[2], p1_n2[2];
    bool p1_in[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = threadIdx.x + 256 * u;
        const int r = i / R1, q = i - r * R1;
        const int y = y0 + r - 1, x = x0 + q - 1;
        p1_in[u] = i < R1 * R1 && x >= 1 && x <= p.W - 2 && y >= 1 && y <= p.H - 2;
        p1_alpha[u] = 0.f; p1_n0[u] = 0.f; p1_n1[u] = 0.f; p1_n2[u] = 0.f;
        if (p1_in[u]) {
            const size_t o = (size_t)y * p.W + x;
            p1_alpha[u] = am[HW + o]; p1_n0[u] = am[2 * HW + o]; p1_n1[u] = am[3 * HW + o]; p1_n2[u] = am[4 * HW + o];
        }
    }
    const int fx = x0 + tx, fy = y0 + ty;
    const bool f_in = fx < p.W && fy < p.H;
    const size_t fo = f_in ? (size_t)fy * p.W + fx : 0;
    float fD = 0.f, fA = 1.f, fmed = 0.f;
    if (f_in) { fD = am[fo]; fA = am[HW + fo]; fmed = am[5 * HW + fo]; }

    rg_stage_depth<2>(p, am, sd, x0, y0);
    __syncthreads();
    // phase 1: for every pixel q of the tile + halo 1, the gradient w.r.t. its two finite differences
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = threadIdx.x + 256 * u;
        if (i >= R1 * R1) continue;
        const int r = i / R1, q = i - r * R1;
        const int y = y0 + r - 1, x = x0 + q - 1;
        float3 gdx = make_float3(0.f, 0.f, 0.f), gdy = gdx;
        if (p1_in[u]) {
            const float3 ru = rg_ray(p, x, y + 1), rd = rg_ray(p, x, y - 1), rr = rg_ray(p, x + 1, y), rl = rg_ray(p, x - 1, y);
            const float du = sd[r + 2][q + 1], dd = sd[r][q + 1], dr = sd[r + 1][q + 2], dl = sd[r + 1][q];   // sd has halo 2
            const float3 dx = make_float3(du * ru.x - dd * rd.x, du * ru.y - dd * rd.y, du * ru.z - dd * rd.z);
            const float3 dy = make_float3(dr * rr.x - dl * rl.x, dr * rr.y - dl * rl.y, dr * rr.z - dl * rl.z);
            const float3 c = rg_cross(dx, dy);
            const float len = sqrtf(c.x * c.x + c.y * c.y + c.z * c.z);
            if (len > 1e-12f) {
                const float il = 1.0f / len;
                const float3 s = make_float3(c.x * il, c.y * il, c.z * il);
                const float alpha = p1_alpha[u];
                // L_q = -kn * alpha * (N . s)  (alpha detached)
                const float3 g = make_float3(-kn * alpha * p1_n0[u], -kn * alpha * p1_n1[u], -kn * alpha * p1_n2[u]);
                const float sg_ = s.x * g.x + s.y * g.y + s.z * g.z;
                const float3 gc = make_float3((g.x - s.x * sg_) * il, (g.y - s.y * sg_) * il, (g.z - s.z * sg_) * il);
                gdx = rg_cross(dy, gc);      // dL/ddx = dy x dL/dc
                gdy = rg_cross(gc, dx);      // dL/ddy = dL/dc x dx
            }
        }
        sg[0][r][q] = gdx.x; sg[1][r][q] = gdx.y; sg[2][r][q] = gdx.z;
        sg[3][r][q] = gdy.x; sg[4][r][q] = gdy.y; sg[5][r][q] = gdy.z;
        if (partials) { sn_in[0][r][q] = p1_n0[u]; sn_in[1][r][q] = p1_n1[u]; sn_in[2][r][q] = p1_n2[u]; }
    }
    __syncthreads();
    const int x = fx, y = fy;
    // `partials` (gsr_regularizer_backward_partials): this launch also leaves what reg_fwd would have -- the tile's sums of
    // the normal error and of the distortion -- so a training step that runs the backward anyway needs no forward launch
    // of the regularizer.  Same per-pixel expression, same reduction tree, same slot as reg_fwd_kernel.
    // the pixel's own finite differences: needed below for d/d(rendered normal) and, with `partials`, for the forward value
    const bool interior = f_in && x >= 1 && x <= p.W - 2 && y >= 1 && y <= p.H - 2;
    float3 c_px = make_float3(0.f, 0.f, 0.f);
    float inv_px = 0.f;
    if (interior) {
        const float3 ru = rg_ray(p, x, y + 1), rd = rg_ray(p, x, y - 1), rr = rg_ray(p, x + 1, y), rl = rg_ray(p, x - 1, y);
        const float du = sd[ty + 3][tx + 2], dd = sd[ty + 1][tx + 2], dr = sd[ty + 2][tx + 3], dl = sd[ty + 2][tx + 1];
        const float3 dx = make_float3(du * ru.x - dd * rd.x, du * ru.y - dd * rd.y, du * ru.z - dd * rd.z);
        const float3 dy = make_float3(dr * rr.x - dl * rl.x, dr * rr.y - dl * rl.y, dr * rr.z - dl * rl.z);
        c_px = rg_cross(dx, dy);
        inv_px = 1.0f / fmaxf(sqrtf(c_px.x * c_px.x + c_px.y * c_px.y + c_px.z * c_px.z), 1e-12f);
    }
    float err = 0.f, dist = 0.f;
    if (partials && f_in) {
        float dotp = 0.f;
        if (interior)
            dotp = (sn_in[0][ty + 1][tx + 1] * c_px.x + sn_in[1][ty + 1][tx + 1] * c_px.y + sn_in[2][ty + 1][tx + 1] * c_px.z) *
                   inv_px * fA;
        err = 1.0f - dotp;
        dist = am[6 * HW + fo];
    }
    if (partials) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { err += __shfl_down(err, d, 64); dist += __shfl_down(dist, d, 64); }
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = err; red[1][threadIdx.x >> 6] = dist; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int b = tile_y * tiles_x + tile_x;
            partials[2 * b + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
            partials[2 * b + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        }
    }
    if (!f_in) return;
    const size_t o = fo;
    const int r = ty + 1, q = tx + 1;     // position inside the halo-1 arrays
    // P(p) is the "upper" point of the pixel above... : dx(q) = P[y+1] - P[y-1]
    //   pixel (y-1) uses P(p) with +, pixel (y+1) with -, same for x through dy
    float3 gP;
    gP.x = sg[0][r - 1][q] - sg[0][r + 1][q] + sg[3][r][q - 1] - sg[3][r][q + 1];
    gP.y = sg[1][r - 1][q] - sg[1][r + 1][q] + sg[4][r][q - 1] - sg[4][r][q + 1];
    gP.z = sg[2][r - 1][q] - sg[2][r + 1][q] + sg[5][r][q - 1] - sg[5][r][q + 1];
    const float3 ray = rg_ray(p, x, y);
    const float g_sd = gP.x * ray.x + gP.y * ray.y + gP.z * ray.z;

    const float D = fD, A = fA, med = fmed;
    const float e = D / A;
    const bool e_ok = !(isnan(e) || isinf(e));
    const float g_e = (1.f - p.depth_ratio) * g_sd;
    dam[o] = e_ok ? g_e / A : 0.f;
    dam[HW + o] = e_ok ? -g_e * D / (A * A) : 0.f;
    dam[5 * HW + o] = (isnan(med) || isinf(med)) ? 0.f : p.depth_ratio * g_sd;
    dam[6 * HW + o] = kd;
    // rendered normal: error = 1 - N . (s * alpha)
    float3 sn = make_float3(0.f, 0.f, 0.f);
    if (interior) sn = make_float3(c_px.x * inv_px * A, c_px.y * inv_px * A, c_px.z * inv_px * A);
    dam[2 * HW + o] = -kn * sn.x;
    dam[3 * HW + o] = -kn * sn.y;
    dam[4 * HW + o] = -kn * sn.z;
}

// ---- the maps themselves (render()'s post-processing, gaussian_renderer/__init__.py:117-156) as one kernel each way ----------
// For callers that want the reference's dictionary -- rend_normal, surf_depth, surf_normal as tensors -- instead of the fused
// objective: out[0:3] = allmap[2:5] rotated to world space (n @ world_view[:3,:3].T), out[3] = surf_depth, out[4:7] =
// depth_to_normal(surf_depth) * alpha.detach() in WORLD space (p.m = c2w[:3,:3] K^-1: the rays the reference builds,
// utils/point_utils.py:9-24; the ray origin drops out of the differences).  The backward takes whatever gradients arrive for the
// seven planes.  F.normalize's eps regime (|cross| <= 1e-12: a plain scaling by 1e12) is followed on both sides.
struct MapRot { float a[9]; };        // world_view_transform[:3,:3], row-major: out_j = sum_i a[3 j + i] n_i

__global__ void __launch_bounds__(256) maps_fwd_kernel(RegParams p, MapRot rot, const float* __restrict__ am,
                                                       float* __restrict__ out) {
    __shared__ float sd[RG_T + 2][RG_T + 3];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int tiles_x = (p.W + RG_T - 1) / RG_T;
    int tile_x, tile_y;
    if (!gsr_xcd_tile(tiles_x, (p.H + RG_T - 1) / RG_T, tile_x, tile_y)) return;
    const int x0 = tile_x * RG_T, y0 = tile_y * RG_T;
    const int x = x0 + tx, y = y0 + ty;
    const size_t HW = (size_t)p.W * p.H;
    const bool in_img = x < p.W && y < p.H;
    const bool interior = in_img && x >= 1 && x <= p.W - 2 && y >= 1 && y <= p.H - 2;
    const size_t o = in_img ? (size_t)y * p.W + x : 0;
    float alpha = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f;
    if (in_img) { alpha = am[HW + o]; n0 = am[2 * HW + o]; n1 = am[3 * HW + o]; n2 = am[4 * HW + o]; }
    rg_stage_depth<1>(p, am, sd, x0, y0);
    __syncthreads();
    if (!in_img) return;
    float3 sn = make_float3(0.f, 0.f, 0.f);
    if (interior) {
        const float3 ru = rg_ray(p, x, y + 1), rd = rg_ray(p, x, y - 1), rr = rg_ray(p, x + 1, y), rl = rg_ray(p, x - 1, y);
        const float du = sd[ty + 2][tx + 1], dd = sd[ty][tx + 1], dr = sd[ty + 1][tx + 2], dl = sd[ty + 1][tx];
        const float3 dx = make_float3(du * ru.x - dd * rd.x, du * ru.y - dd * rd.y, du * ru.z - dd * rd.z);
        const float3 dy = make_float3(dr * rr.x - dl * rl.x, dr * rr.y - dl * rl.y, dr * rr.z - dl * rl.z);
        const float3 c = rg_cross(dx, dy);
        const float inv = 1.0f / fmaxf(sqrtf(c.x * c.x + c.y * c.y + c.z * c.z), 1e-12f);
        sn = make_float3(c.x * inv * alpha, c.y * inv * alpha, c.z * inv * alpha);
    }
    out[o] = rot.a[0] * n0 + rot.a[1] * n1 + rot.a[2] * n2;
    out[HW + o] = rot.a[3] * n0 + rot.a[4] * n1 + rot.a[5] * n2;
    out[2 * HW + o] = rot.a[6] * n0 + rot.a[7] * n1 + rot.a[8] * n2;
    out[3 * HW + o] = sd[ty + 1][tx + 1];
    out[4 * HW + o] = sn.x; out[5 * HW + o] = sn.y; out[6 * HW + o] = sn.z;
}

__global__ void __launch_bounds__(256) maps_bwd_kernel(RegParams p, MapRot rot, const float* __restrict__ am,
                                                       const float* __restrict__ gout, float* __restrict__ dam) {
    __shared__ float sd[RG_T + 4][RG_T + 5];          // surf_depth, halo 2
    __shared__ float sg[6][RG_T + 2][RG_T + 3];       // dL/ddx, dL/ddy of every pixel, halo 1
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int tiles_x = (p.W + RG_T - 1) / RG_T;
    int tile_x, tile_y;
    if (!gsr_xcd_tile(tiles_x, (p.H + RG_T - 1) / RG_T, tile_x, tile_y)) return;
    const int x0 = tile_x * RG_T, y0 = tile_y * RG_T;
    const size_t HW = (size_t)p.W * p.H;
    constexpr int R1 = RG_T + 2;
    float p1_alpha[2], p1_g0[2], p1_g1[2], p1_g2[2];
    bool p1_in[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = threadIdx.x + 256 * u;
        const int r = i / R1, q = i - r * R1;
        const int y = y0 + r - 1, x = x0 + q - 1;
        p1_in[u] = i < R1 * R1 && x >= 1 && x <= p.W - 2 && y >= 1 && y <= p.H - 2;
        p1_alpha[u] = 0.f; p1_g0[u] = 0.f; p1_g1[u] = 0.f; p1_g2[u] = 0.f;
        if (p1_in[u]) {
            const size_t o = (size_t)y * p.W + x;
            p1_alpha[u] = am[HW + o]; p1_g0[u] = gout[4 * HW + o]; p1_g1[u] = gout[5 * HW + o]; p1_g2[u] = gout[6 * HW + o];
        }
    }
    const int fx = x0 + tx, fy = y0 + ty;
    const bool f_in = fx < p.W && fy < p.H;
    const size_t fo = f_in ? (size_t)fy * p.W + fx : 0;
    float fD = 0.f, fA = 1.f, fmed = 0.f, g_rn0 = 0.f, g_rn1 = 0.f, g_rn2 = 0.f, g_sd_direct = 0.f;
    if (f_in) {
        fD = am[fo]; fA = am[HW + fo]; fmed = am[5 * HW + fo];
        g_rn0 = gout[fo]; g_rn1 = gout[HW + fo]; g_rn2 = gout[2 * HW + fo]; g_sd_direct = gout[3 * HW + fo];
    }
    rg_stage_depth<2>(p, am, sd, x0, y0);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = threadIdx.x + 256 * u;
        if (i >= R1 * R1) continue;
        const int r = i / R1, q = i - r * R1;
        const int y = y0 + r - 1, x = x0 + q - 1;
        float3 gdx = make_float3(0.f, 0.f, 0.f), gdy = gdx;
        if (p1_in[u]) {
            const float3 ru = rg_ray(p, x, y + 1), rd = rg_ray(p, x, y - 1), rr = rg_ray(p, x + 1, y), rl = rg_ray(p, x - 1, y);
            const float du = sd[r + 2][q + 1], dd = sd[r][q + 1], dr = sd[r + 1][q + 2], dl = sd[r + 1][q];
            const float3 dx = make_float3(du * ru.x - dd * rd.x, du * ru.y - dd * rd.y, du * ru.z - dd * rd.z);
            const float3 dy = make_float3(dr * rr.x - dl * rl.x, dr * rr.y - dl * rl.y, dr * rr.z - dl * rl.z);
            const float3 c = rg_cross(dx, dy);
            const float len = sqrtf(c.x * c.x + c.y * c.y + c.z * c.z);
            const float a = p1_alpha[u];
            const float3 g = make_float3(p1_g0[u] * a, p1_g1[u] * a, p1_g2[u] * a);      // (alpha detached)
            float3 gc;
            if (len > 1e-12f) {
                const float il = 1.0f / len;
                const float3 s = make_float3(c.x * il, c.y * il, c.z * il);
                const float sg_ = s.x * g.x + s.y * g.y + s.z * g.z;
                gc = make_float3((g.x - s.x * sg_) * il, (g.y - s.y * sg_) * il, (g.z - s.z * sg_) * il);
            } else {
                gc = make_float3(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f);                   // c / max(|c|, eps): the eps regime
            }
            gdx = rg_cross(dy, gc);
            gdy = rg_cross(gc, dx);
        }
        sg[0][r][q] = gdx.x; sg[1][r][q] = gdx.y; sg[2][r][q] = gdx.z;
        sg[3][r][q] = gdy.x; sg[4][r][q] = gdy.y; sg[5][r][q] = gdy.z;
    }
    __syncthreads();
    if (!f_in) return;
    const int x = fx, y = fy;
    const size_t o = fo;
    const int r = ty + 1, q = tx + 1;
    float3 gP;
    gP.x = sg[0][r - 1][q] - sg[0][r + 1][q] + sg[3][r][q - 1] - sg[3][r][q + 1];
    gP.y = sg[1][r - 1][q] - sg[1][r + 1][q] + sg[4][r][q - 1] - sg[4][r][q + 1];
    gP.z = sg[2][r - 1][q] - sg[2][r + 1][q] + sg[5][r][q - 1] - sg[5][r][q + 1];
    const float3 ray = rg_ray(p, x, y);
    const float g_sd = gP.x * ray.x + gP.y * ray.y + gP.z * ray.z + g_sd_direct;
    const float D = fD, A = fA, med = fmed;
    const float e = D / A;
    const bool e_ok = !(isnan(e) || isinf(e));
    const float g_e = (1.f - p.depth_ratio) * g_sd;
    // (where D / A is not finite -- alpha = 0 -- torch leaves 0 / 0 here; those pixels hold no splat and nobody reads them)
    dam[o] = e_ok ? g_e / A : 0.f;
    dam[HW + o] = e_ok ? -g_e * D / (A * A) : 0.f;
    dam[5 * HW + o] = (isnan(med) || isinf(med)) ? 0.f : p.depth_ratio * g_sd;
    dam[6 * HW + o] = 0.f;
    dam[2 * HW + o] = rot.a[0] * g_rn0 + rot.a[3] * g_rn1 + rot.a[6] * g_rn2;
    dam[3 * HW + o] = rot.a[1] * g_rn0 + rot.a[4] * g_rn1 + rot.a[7] * g_rn2;
    dam[4 * HW + o] = rot.a[2] * g_rn0 + rot.a[5] * g_rn1 + rot.a[8] * g_rn2;
}

static int fill_params(RegParams& p, int H, int W, float depth_ratio, const float* kinv) {
    if (H <= 0 || W <= 0 || !kinv) { gsr_set_error("bad regularizer arguments"); return GSR_E_INVALID; }
    p.W = W; p.H = H; p.depth_ratio = depth_ratio;
    for (int i = 0; i < 9; ++i) p.m[i] = kinv[i];
    return GSR_OK;
}

extern "C" int32_t gsr_regularizer_forward(const float* allmap, int32_t H, int32_t W, const float* kinv_host,
                                           float depth_ratio, float* partials, gsr_stream_t stream_) {
    RegParams p;
    int rc = fill_params(p, H, W, depth_ratio, kinv_host);
    if (rc != GSR_OK) return rc;
    if (!allmap || !partials) { gsr_set_error("bad regularizer arguments"); return GSR_E_INVALID; }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    GsrProfileScope prof(GSR_K_REG_FWD, s);
    dim3 grid(gsr_xcd_tile_grid(((W + RG_T - 1) / RG_T) * ((H + RG_T - 1) / RG_T)));
    hipLaunchKernelGGL(reg_fwd_kernel, grid, dim3(256), 0, s, p, allmap, partials);
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}

extern "C" int32_t gsr_regularizer_backward(const float* allmap, int32_t H, int32_t W, const float* kinv_host,
                                            float depth_ratio, float lambda_normal, float lambda_dist,
                                            const float* grad_scale, float* d_allmap, gsr_stream_t stream_) {
    return gsr_regularizer_backward_partials(allmap, H, W, kinv_host, depth_ratio, lambda_normal, lambda_dist, grad_scale,
                                             d_allmap, nullptr, stream_);
}

extern "C" int32_t gsr_regularizer_backward_partials(const float* allmap, int32_t H, int32_t W, const float* kinv_host,
                                                     float depth_ratio, float lambda_normal, float lambda_dist,
                                                     const float* grad_scale, float* d_allmap, float* partials,
                                                     gsr_stream_t stream_) {
    RegParams p;
    int rc = fill_params(p, H, W, depth_ratio, kinv_host);
    if (rc != GSR_OK) return rc;
    if (!allmap || !grad_scale || !d_allmap) { gsr_set_error("bad regularizer arguments"); return GSR_E_INVALID; }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    GsrProfileScope prof(GSR_K_REG_BWD, s);
    dim3 grid(gsr_xcd_tile_grid(((W + RG_T - 1) / RG_T) * ((H + RG_T - 1) / RG_T)));
    hipLaunchKernelGGL(reg_bwd_kernel, grid, dim3(256), 0, s, p, allmap, lambda_normal, lambda_dist, grad_scale, d_allmap,
                       partials);
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}

extern "C" int32_t gsr_surface_maps_forward(const float* allmap, int32_t H, int32_t W, const float* rays_world_host,
                                            const float* rot_host, float depth_ratio, float* out7, gsr_stream_t stream_) {
    RegParams p;
    int rc = fill_params(p, H, W, depth_ratio, rays_world_host);
    if (rc != GSR_OK) return rc;
    if (!allmap || !rot_host || !out7) { gsr_set_error("bad surface-maps arguments"); return GSR_E_INVALID; }
    MapRot rot;
    for (int i = 0; i < 9; ++i) rot.a[i] = rot_host[i];
    hipStream_t s = static_cast<hipStream_t>(stream_);
    GsrProfileScope prof(GSR_K_REG_FWD, s);
    dim3 grid(gsr_xcd_tile_grid(((W + RG_T - 1) / RG_T) * ((H + RG_T - 1) / RG_T)));
    hipLaunchKernelGGL(maps_fwd_kernel, grid, dim3(256), 0, s, p, rot, allmap, out7);
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}

extern "C" int32_t gsr_surface_maps_backward(const float* allmap, int32_t H, int32_t W, const float* rays_world_host,
                                             const float* rot_host, float depth_ratio, const float* d_out7, float* d_allmap,
                                             gsr_stream_t stream_) {
    RegParams p;
    int rc = fill_params(p, H, W, depth_ratio, rays_world_host);
    if (rc != GSR_OK) return rc;
    if (!allmap || !rot_host || !d_out7 || !d_allmap) { gsr_set_error("bad surface-maps arguments"); return GSR_E_INVALID; }
    MapRot rot;
    for (int i = 0; i < 9; ++i) rot.a[i] = rot_host[i];
    hipStream_t s = static_cast<hipStream_t>(stream_);
    GsrProfileScope prof(GSR_K_REG_BWD, s);
    dim3 grid(gsr_xcd_tile_grid(((W + RG_T - 1) / RG_T) * ((H + RG_T - 1) / RG_T)));
    hipLaunchKernelGGL(maps_bwd_kernel, grid, dim3(256), 0, s, p, rot, allmap, d_out7, d_allmap);
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}
