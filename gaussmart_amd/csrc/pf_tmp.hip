// K1 preprocess_fwd: one thread per Gaussian.
//   view cull -> quat -> T (splat->pixel homography, rows Tu,Tv,Tw) -> AABB radius + tile rect
//   -> view-space normal -> SH colour (+clamp mask) -> alpha-support cull rect -> 80-byte splat record.
// Restates the [U] preprocess of the un-vendored rasterizer; in-tree anchors:
//   T matrix / (W-1)/2 convention   gaussian_renderer/__init__.py:64-75
//   quaternion -> R                 utils/general_utils.py:78-99
//   SH basis, +0.5, clamp           utils/sh_utils.py:57-112
// HBM-bound streaming kernel: 232 B read (192 of them SH) + 96 B written per Gaussian.
// The [N,16,3] SH block of a wave (64 x 192 B = 12 KiB contiguous) is fetched with coalesced
// 16-byte loads into a wave-private LDS tile with 208-byte rows (conflict-free ds_read_b128 by
// row) instead of 48 strided dword loads per lane.
#include "gsr_common.h"
#include <cstdlib>
#include "sh_stage.h"
#include "wave_reduce.h"

#define PRE_BLOCK 256

struct PreParams {
    int N, W, H, gx, gy;
    int deg, M;
    float mod;
    const float* view; const float* proj; const float* campos;
    bool raw;
    const float* means; const float* shs; const float* shs_rest; const float* colors; const float* opac;
    const float* scales; const float* rots; const float* tprecomp;
    float* splat; uint32_t* clamped; uint32_t* tiles; uint2* rect; uint32_t* dkey; int32_t* radii;
    float* jac;      // colour pass: d(rgb)/d(dir) [N,9] (row c = channel, column a = x, y, z), or NULL
    // geometry pass: per-workgroup sums of tiles_touched (64-bit; may be device-mapped host memory) or NULL, and a few
    // words this launch clears for a later kernel (the depth sort's supergroup table)
    unsigned long long* count_partial; uint32_t* zero; int zero_words;
};

__device__ __forceinline__ float3 sh_to_rgb(int deg, const float* sh /*[M][3] in LDS or global*/,
                                            int stride3, float3 dir, uint32_t& clamp_bits) {
    // sh[k*stride3 + c]
    float x = dir.x, y = dir.y, z = dir.z;
    float r[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = GSR_SH_C0 * sh[0 * stride3 + c];
        if (deg > 0) {
            v = v - GSR_SH_C1 * y * sh[1 * stride3 + c] + GSR_SH_C1 * z * sh[2 * stride3 + c]
                  - GSR_SH_C1 * x * sh[3 * stride3 + c];
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                v = v + GSR_SH_C2_0 * xy * sh[4 * stride3 + c] + GSR_SH_C2_1 * yz * sh[5 * stride3 + c]
                      + GSR_SH_C2_2 * (2.0f * zz - xx - yy) * sh[6 * stride3 + c]
                      + GSR_SH_C2_3 * xz * sh[7 * stride3 + c]
                      + GSR_SH_C2_4 * (xx - yy) * sh[8 * stride3 + c];
                if (deg > 2) {
                    v = v + GSR_SH_C3_0 * y * (3.0f * xx - yy) * sh[9 * stride3 + c]
                          + GSR_SH_C3_1 * xy * z * sh[10 * stride3 + c]
                          + GSR_SH_C3_2 * y * (4.0f * zz - xx - yy) * sh[11 * stride3 + c]
                          + GSR_SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12 * stride3 + c]
                          + GSR_SH_C3_4 * x * (4.0f * zz - xx - yy) * sh[13 * stride3 + c]
                          + GSR_SH_C3_5 * z * (xx - yy) * sh[14 * stride3 + c]
                          + GSR_SH_C3_6 * x * (xx - 3.0f * yy) * sh[15 * stride3 + c];
                }
            }
        }
        v += 0.5f;
        if (v < 0.0f) clamp_bits |= (1u << c);
        r[c] = fmaxf(v, 0.0f);
    }
    return make_float3(r[0], r[1], r[2]);
}

// WITH_SH: evaluate the SH colour in this kernel (single-launch form).  The library normally runs
// the geometry part first and the colour part (preprocess_color_kernel) later, so that the colour
// evaluation overlaps the host round trip that fetches the instance count.
// One Gaussian of the geometry pass; returns the number of tiles it touches (0: culled).
template <bool STAGE_SH, bool WITH_SH>
__device__ __forceinline__ uint32_t preprocess_one(const PreParams& p, const int idx, const float* my_sh) {
    if (idx >= p.N) return 0u;

    // defaults for a culled Gaussian
    p.radii[idx] = 0;
    p.tiles[idx] = 0;
    if (p.rect) p.rect[idx] = make_uint2(0u, 0u);
    p.dkey[idx] = 0xFFFFFFFFu;

    const float* V = p.view;
    const float* P = p.proj;
    const float px = p.means[3 * idx + 0], py = p.means[3 * idx + 1], pz = p.means[3 * idx + 2];
    const float vx = px * V[0] + py * V[4] + pz * V[8] + V[12];
    const float vy = px * V[1] + py * V[5] + pz * V[9] + V[13];
    const float vz = px * V[2] + py * V[6] + pz * V[10] + V[14];
    if (!(vz > GSR_NEAR_N)) return 0u;

    float Tu[3], Tv[3], Tw[3], nrm[3];
    if (p.tprecomp == nullptr) {
        float qw = p.rots[4 * idx + 0], qx = p.rots[4 * idx + 1], qy = p.rots[4 * idx + 2],
              qz = p.rots[4 * idx + 3];
        const float s = rsqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
        qw *= s; qx *= s; qy *= s; qz *= s;
        // columns of R (utils/general_utils.py:90-98)
        const float r00 = 1.f - 2.f * (qy * qy + qz * qz), r10 = 2.f * (qx * qy + qw * qz), r20 = 2.f * (qx * qz - qw * qy);
        const float r01 = 2.f * (qx * qy - qw * qz), r11 = 1.f - 2.f * (qx * qx + qz * qz), r21 = 2.f * (qy * qz + qw * qx);
        const float r02 = 2.f * (qx * qz + qw * qy), r12 = 2.f * (qy * qz - qw * qx), r22 = 1.f - 2.f * (qx * qx + qy * qy);
        float sx = p.scales[2 * idx + 0], sy = p.scales[2 * idx + 1];
        if (p.raw) { sx = expf(sx); sy = expf(sy); }          // scaling_activation = exp
        sx *= p.mod; sy *= p.mod;
        const float rows[3][4] = {{r00 * sx, r10 * sx, r20 * sx, 0.f},
                                  {r01 * sy, r11 * sy, r21 * sy, 0.f},
                                  {px, py, pz, 1.f}};
        const float hw = 0.5f * (float)p.W, hh = 0.5f * (float)p.H;
        const float cw = 0.5f * (float)(p.W - 1), ch = 0.5f * (float)(p.H - 1);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float a = rows[i][0], b = rows[i][1], c = rows[i][2], e = rows[i][3];
            const float h0 = a * P[0] + b * P[4] + c * P[8] + e * P[12];
            const float h1 = a * P[1] + b * P[5] + c * P[9] + e * P[13];
            const float h3 = a * P[3] + b * P[7] + c * P[11] + e * P[15];
            Tu[i] = h0 * hw + h3 * cw;
            Tv[i] = h1 * hh + h3 * ch;
            Tw[i] = h3;
        }
        nrm[0] = r02 * V[0] + r12 * V[4] + r22 * V[8];
        nrm[1] = r02 * V[1] + r12 * V[5] + r22 * V[9];
        nrm[2] = r02 * V[2] + r12 * V[6] + r22 * V[10];
    } else {
        const float* t = p.tprecomp + 9 * (size_t)idx;
#pragma unroll
        for (int i = 0; i < 3; ++i) { Tu[i] = t[i]; Tv[i] = t[3 + i]; Tw[i] = t[6 + i]; }
        nrm[0] = 0.f; nrm[1] = 0.f; nrm[2] = 1.f;
    }

    const float cosv = -(vx * nrm[0] + vy * nrm[1] + vz * nrm[2]);
    if (cosv == 0.0f) return 0u;
    const float sgn = cosv > 0.0f ? 1.0f : -1.0f;
    nrm[0] *= sgn; nrm[1] *= sgn; nrm[2] *= sgn;

    // AABB of the 3-sigma ellipse under the homography
    const float t0 = GSR_CUTOFF * GSR_CUTOFF, t1 = GSR_CUTOFF * GSR_CUTOFF, t2 = -1.0f;
    const float d = t0 * Tw[0] * Tw[0] + t1 * Tw[1] * Tw[1] + t2 * Tw[2] * Tw[2];
    if (d == 0.0f) return 0u;
    const float inv_d = 1.0f / d;
    const float f0 = t0 * inv_d, f1 = t1 * inv_d, f2 = t2 * inv_d;
    const float cx = f0 * Tu[0] * Tw[0] + f1 * Tu[1] * Tw[1] + f2 * Tu[2] * Tw[2];
    const float cy = f0 * Tv[0] * Tw[0] + f1 * Tv[1] * Tw[1] + f2 * Tv[2] * Tw[2];
    const float h0x = cx * cx - (f0 * Tu[0] * Tu[0] + f1 * Tu[1] * Tu[1] + f2 * Tu[2] * Tu[2]);
    const float h0y = cy * cy - (f0 * Tv[0] * Tv[0] + f1 * Tv[1] * Tv[1] + f2 * Tv[2] * Tv[2]);
    const float ex = sqrtf(fmaxf(GSR_AABB_MIN_EXT2, h0x));
    const float ey = sqrtf(fmaxf(GSR_AABB_MIN_EXT2, h0y));
    const float radius = ceilf(fmaxf(fmaxf(ex, ey), GSR_CUTOFF * GSR_FILTER_SIZE));
    if (!(isfinite(cx) && isfinite(cy) && isfinite(radius))) return 0u;

    int x0, y0, x1, y1;
    gsr_tile_rect(cx, cy, (int)radius, p.gx, p.gy, x0, y0, x1, y1);
    const int ntiles = (x1 - x0) * (y1 - y0);
    if (ntiles == 0) return 0u;

    uint32_t clamp_bits = 0;
    float3 rgb = make_float3(0.f, 0.f, 0.f);
    if (p.colors == nullptr && WITH_SH) {
        float dx = px - p.campos[0], dy = py - p.campos[1], dz = pz - p.campos[2];
        const float il = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
        const float3 dir = make_float3(dx * il, dy * il, dz * il);
        if (STAGE_SH) rgb = sh_to_rgb(p.deg, my_sh, 3, dir, clamp_bits);
        else          rgb = sh_to_rgb(p.deg, p.shs + (size_t)idx * p.M * 3, 3, dir, clamp_bits);
    } else if (p.colors != nullptr) {
        rgb = make_float3(p.colors[3 * idx], p.colors[3 * idx + 1], p.colors[3 * idx + 2]);
    }

    const float opacity = p.raw ? 1.0f / (1.0f + expf(-p.opac[idx])) : p.opac[idx];   // opacity_activation = sigmoid
    // conservative pixel box of {pixels where alpha can reach 1/255}:
    //   alpha = opa * exp(-rho/2) >= 1/255  <=>  rho = min(rho3d, rho2d) <= rho_max = 2 ln(255 opa)
    //   rho2d <= rho_max : disc of radius sqrt(rho_max / 2) around the AABB centre
    //   rho3d <= rho_max : projection of the splat-space disc of radius sqrt(rho_max), bounded with the
    //                      same conic formula as the 3-sigma AABB (valid while the disc stays in front
    //                      of the camera plane, d < 0; otherwise no culling)
    uint32_t rect_x = 0x7FFF8000u, rect_y = 0x7FFF8000u;   // [-32768, 32767]: never culled
    {
        const float opa = opacity;
        const float c2 = (fmaxf(2.0f * logf(255.0f * opa), 0.0f) + 0.05f) * 1.02f;
        if (255.0f * opa < 0.999f) {
            rect_x = 0x80007FFFu; rect_y = 0x80007FFFu;     // x0 = 32767 > x1 = -32768: cannot reach 1/255 anywhere
        } else {
            const float dd = c2 * (Tw[0] * Tw[0] + Tw[1] * Tw[1]) - Tw[2] * Tw[2];
            if (dd < 0.0f) {
                const float idd = 1.0f / dd;
                const float g0 = c2 * idd, g2 = -idd;
                const float qx = g0 * (Tu[0] * Tw[0] + Tu[1] * Tw[1]) + g2 * Tu[2] * Tw[2];
                const float qy = g0 * (Tv[0] * Tw[0] + Tv[1] * Tw[1]) + g2 * Tv[2] * Tw[2];
                const float hx = sqrtf(fmaxf(qx * qx - (g0 * (Tu[0] * Tu[0] + Tu[1] * Tu[1]) + g2 * Tu[2] * Tu[2]), 0.0f));
                const float hy = sqrtf(fmaxf(qy * qy - (g0 * (Tv[0] * Tv[0] + Tv[1] * Tv[1]) + g2 * Tv[2] * Tv[2]), 0.0f));
                const float r2 = sqrtf(0.5f * c2);
                float lox = fminf(qx - hx, cx - r2), hix = fmaxf(qx + hx, cx + r2);
                float loy = fminf(qy - hy, cy - r2), hiy = fmaxf(qy + hy, cy + r2);
                const float mx = 1.0f + 0.01f * (hix - lox), my = 1.0f + 0.01f * (hiy - loy);
                lox -= mx; hix += mx; loy -= my; hiy += my;
                if (isfinite(lox) && isfinite(hix) && isfinite(loy) && isfinite(hiy)) {
                    const int x0 = (int)fminf(fmaxf(floorf(lox), -32768.f), 32767.f);
                    const int x1 = (int)fminf(fmaxf(ceilf(hix), -32768.f), 32767.f);
                    const int y0 = (int)fminf(fmaxf(floorf(loy), -32768.f), 32767.f);
                    const int y1 = (int)fminf(fmaxf(ceilf(hiy), -32768.f), 32767.f);
                    rect_x = ((uint32_t)x0 & 0xFFFFu) | ((uint32_t)x1 << 16);
                    rect_y = ((uint32_t)y0 & 0xFFFFu) | ((uint32_t)y1 << 16);
                }
            }
        }
    }

    float4* rec = reinterpret_cast<float4*>(p.splat + (size_t)idx * GSR_SPLAT_FLOATS);
    rec[0] = make_float4(Tu[0], Tu[1], Tu[2], Tv[0]);
    rec[1] = make_float4(Tv[1], Tv[2], Tw[0], Tw[1]);
    rec[2] = make_float4(Tw[2], cx, cy, nrm[0]);
    rec[3] = make_float4(nrm[1], nrm[2], opacity, rgb.x);
    rec[4] = make_float4(rgb.y, rgb.z, __uint_as_float(rect_x), __uint_as_float(rect_y));
    p.clamped[idx] = clamp_bits;
    p.radii[idx] = (int)radius;
    p.tiles[idx] = (uint32_t)ntiles;
    if (p.rect) p.rect[idx] = make_uint2((uint32_t)x0 | ((uint32_t)y0 << 16), (uint32_t)(x1 - x0) | ((uint32_t)(y1 - y0) << 16));
    p.dkey[idx] = __float_as_uint(vz);
    return (uint32_t)ntiles;
}

template <bool STAGE_SH, bool WITH_SH>
__global__ void __launch_bounds__(PRE_BLOCK, 8) preprocess_fwd_kernel(PreParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ unsigned long long s_cnt[PRE_BLOCK / 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int idx = blockIdx.x * PRE_BLOCK + tid;
    const int wave_first = blockIdx.x * PRE_BLOCK + wave * 64;
    for (int i = blockIdx.x * PRE_BLOCK + tid; i < p.zero_words; i += gridDim.x * PRE_BLOCK) p.zero[i] = 0;

    float* my_sh = nullptr;
    if (STAGE_SH && WITH_SH) {
        // cooperative, fully coalesced fetch of this wave's 64 SH blocks (wave-private LDS region:
        // LDS ops of one wave complete in order, no workgroup barrier needed)
        float* wl = lds + wave * 64 * SH_ROW_FLOATS;
        sh_stage<true>(wl, const_cast<float*>(p.shs), const_cast<float*>(p.shs_rest), p.M, wave_first,
                       min(64, p.N - wave_first), lane);
        my_sh = wl + lane * SH_ROW_FLOATS;
    }
    const uint32_t ntiles = preprocess_one<STAGE_SH, WITH_SH>(p, idx, my_sh);
    // instance count of the frame: one 64-bit partial sum per workgroup, added up by the host (gsr_forward reads it back
    // while the depth sort runs) -- no separate counting launch
    if (p.count_partial) {
        unsigned long long sum = ntiles;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) sum += __shfl_down(sum, d, 64);
        if (lane == 0) s_cnt[wave] = sum;
        __syncthreads();
        if (tid == 0) p.count_partial[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    }
}


// Colour part of K1: SH -> RGB (+0.5, clamp at 0, clamp mask) for the Gaussians that survived the
// culls, written into their splat record.  Runs after the binning front end has been enqueued.
template <bool STAGE_SH>
__global__ void __launch_bounds__(PRE_BLOCK) preprocess_color_kernel(PreParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int idx = blockIdx.x * PRE_BLOCK + tid;
    const int wave_first = blockIdx.x * PRE_BLOCK + wave * 64;
    float* my_sh = nullptr;
    if (STAGE_SH) {
        float* wl = lds + wave * 64 * SH_ROW_FLOATS;
        sh_stage<true>(wl, const_cast<float*>(p.shs), const_cast<float*>(p.shs_rest), p.M, wave_first,
                       min(64, p.N - wave_first), lane);
        my_sh = wl + lane * SH_ROW_FLOATS;
    }
    if (idx >= p.N || p.radii[idx] <= 0) return;
    const float dx = p.means[3 * idx + 0] - p.campos[0], dy = p.means[3 * idx + 1] - p.campos[1],
                dz = p.means[3 * idx + 2] - p.campos[2];
    const float il = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    const float3 dir = make_float3(dx * il, dy * il, dz * il);
    uint32_t clamp_bits = 0;
    const float3 rgb = STAGE_SH ? sh_to_rgb(p.deg, my_sh, 3, dir, clamp_bits)
                                : sh_to_rgb(p.deg, p.shs + (size_t)idx * p.M * 3, 3, dir, clamp_bits);
    float* rec = p.splat + (size_t)idx * GSR_SPLAT_FLOATS;
    rec[GSR_SP_RGB] = rgb.x; rec[GSR_SP_RGB + 1] = rgb.y; rec[GSR_SP_RGB + 2] = rgb.z;
    p.clamped[idx] = clamp_bits;
}

// Colour part of K1, register-only form: 16 lanes (one DPP row) per Gaussian, lane k owns SH coefficient k.
//   * lane k loads ITS coefficient's three channels (12 contiguous bytes; a row reads the Gaussian's 192 contiguous
//     bytes), for four consecutive Gaussians per row before anything is consumed (3 KB per wave in flight);
//   * every basis function of utils/sh_utils.py:57-112 has the form  C_k x^a y^b z^c (al xx + be yy + ga zz + de)
//     with a, b, c in {0, 1}: the eight per-k constants sit in a table read once per lane, so the evaluation is
//     branch-free and identical for all lanes (13 VALU);
//   * the 16 products per channel are summed inside the row with four DPP steps; lane 0 stores.
// No LDS (the staged form is limited to 12 waves per CU by its 13 KB tile per wave and spends ~300 VALU per lane
// re-packing 45-float rows into it), no alignment requirement.  Latency-bound: per step at 1M Gaussians, geometry +
// colour launches together, staged 0.129 ms; this form with 1 / 2 / 3 / 4 / 8 Gaussians per row and pass:
// 0.140 / 0.119 / 0.122 / 0.124 / 0.202 ms.
struct ShBasisRow { float C, fx, fy, fz, al, be, ga, de; };
__constant__ ShBasisRow k_sh_basis_rows[16] = {
    {GSR_SH_C0, 0, 0, 0, 0, 0, 0, 1},       {-GSR_SH_C1, 0, 1, 0, 0, 0, 0, 1},      {GSR_SH_C1, 0, 0, 1, 0, 0, 0, 1},
    {-GSR_SH_C1, 1, 0, 0, 0, 0, 0, 1},      {GSR_SH_C2_0, 1, 1, 0, 0, 0, 0, 1},     {GSR_SH_C2_1, 0, 1, 1, 0, 0, 0, 1},
    {GSR_SH_C2_2, 0, 0, 0, -1, -1, 2, 0},   {GSR_SH_C2_3, 1, 0, 1, 0, 0, 0, 1},     {GSR_SH_C2_4, 0, 0, 0, 1, -1, 0, 0},
    {GSR_SH_C3_0, 0, 1, 0, 3, -1, 0, 0},    {GSR_SH_C3_1, 1, 1, 1, 0, 0, 0, 1},     {GSR_SH_C3_2, 0, 1, 0, -1, -1, 4, 0},
    {GSR_SH_C3_3, 0, 0, 1, -3, -3, 2, 0},   {GSR_SH_C3_4, 1, 0, 0, -1, -1, 4, 0},   {GSR_SH_C3_5, 0, 0, 1, 1, -1, 0, 0},
    {GSR_SH_C3_6, 1, 0, 0, 1, -3, 0, 0}};

#ifndef PC_PER_ROW
#define PC_PER_ROW 2     // Gaussians per 16-lane row and pass
#endif
__global__ void __launch_bounds__(PRE_BLOCK) preprocess_color16_kernel(PreParams p) {
    const int l16 = threadIdx.x & 15;
    const long long row = ((long long)blockIdx.x * PRE_BLOCK + threadIdx.x) >> 4;
    const int first = (int)(row * PC_PER_ROW);
    if (first >= p.N) return;                       // whole rows leave together (DPP stays inside a row)
    ShBasisRow b = k_sh_basis_rows[l16];
    const int n_active = (p.deg + 1) * (p.deg + 1);
    if (l16 >= n_active || l16 >= p.M) b.C = 0.f;   // above the active degree: no contribution
    const float ofx = 1.f - b.fx, ofy = 1.f - b.fy, ofz = 1.f - b.fz;
    const float cpx = p.campos[0], cpy = p.campos[1], cpz = p.campos[2];

    int radius[PC_PER_ROW];
    float mx[PC_PER_ROW], my[PC_PER_ROW], mz[PC_PER_ROW], s0[PC_PER_ROW], s1[PC_PER_ROW], s2[PC_PER_ROW];
#pragma unroll
    for (int u = 0; u < PC_PER_ROW; ++u) {
        const int idx = first + u;
        radius[u] = 0; mx[u] = my[u] = mz[u] = 0.f; s0[u] = s1[u] = s2[u] = 0.f;
        if (idx < p.N) {
            radius[u] = p.radii[idx];
            mx[u] = p.means[3 * (size_t)idx + 0]; my[u] = p.means[3 * (size_t)idx + 1]; mz[u] = p.means[3 * (size_t)idx + 2];
            if (l16 < p.M) {
                const float* src = p.shs_rest == nullptr ? p.shs + ((size_t)idx * p.M + l16) * 3
                                   : (l16 == 0 ? p.shs + (size_t)idx * 3 : p.shs_rest + ((size_t)idx * (p.M - 1) + (l16 - 1)) * 3);
                s0[u] = src[0]; s1[u] = src[1]; s2[u] = src[2];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < PC_PER_ROW; ++u) {
        const int idx = first + u;
        const float dx = mx[u] - cpx, dy = my[u] - cpy, dz = mz[u] - cpz;
        const float il = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
        const float x = dx * il, y = dy * il, z = dz * il;
        const float poly = fmaf(b.al, x * x, fmaf(b.be, y * y, fmaf(b.ga, z * z, b.de)));
        const float ux = fmaf(b.fx, x, ofx), uy = fmaf(b.fy, y, ofy), uz = fmaf(b.fz, z, ofz);
        const float mono = ux * uy * uz;
        const bool vis = radius[u] > 0;
        const float bk = vis ? b.C * mono * poly : 0.f;     // (culled: il may be anything)
        float v[3] = {bk * s0[u], bk * s1[u], bk * s2[u]};
        if (p.jac != nullptr) {
            // d(basis_k)/d(dir) of the same product form, and with it J[c][a] = sum_k d(basis_k)/d(dir_a) * sh[k][c]: what
            // preprocess_bwd needs for the view-direction term of dL/dmean, so that it does not read the SH coefficients again
            const float cm = b.C * mono, cp_ = b.C * poly;
            const float dbx = vis ? fmaf(cp_, b.fx * uy * uz, cm * 2.f * b.al * x) : 0.f;
            const float dby = vis ? fmaf(cp_, b.fy * ux * uz, cm * 2.f * b.be * y) : 0.f;
            const float dbz = vis ? fmaf(cp_, b.fz * ux * uy, cm * 2.f * b.ga * z) : 0.f;
            float j[9] = {dbx * s0[u], dby * s0[u], dbz * s0[u], dbx * s1[u], dby * s1[u], dbz * s1[u],
                          dbx * s2[u], dby * s2[u], dbz * s2[u]};
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                j[q] += dpp_move<0xB1, 0xf>(j[q]);
                j[q] += dpp_move<0x4E, 0xf>(j[q]);
                j[q] += dpp_move<0x141, 0xf>(j[q]);
                j[q] += dpp_move<0x140, 0xf>(j[q]);
            }
            // every lane of the row holds the nine sums: lane q stores J[q] (36 contiguous bytes per Gaussian)
            float mine = j[0];
#pragma unroll
            for (int q = 1; q < 9; ++q) mine = l16 == q ? j[q] : mine;
            if (l16 < 9 && idx < p.N && vis) p.jac[(size_t)idx * 9 + l16] = mine;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            v[c] += dpp_move<0xB1, 0xf>(v[c]);      // quad_perm [1,0,3,2]
            v[c] += dpp_move<0x4E, 0xf>(v[c]);      // quad_perm [2,3,0,1]
            v[c] += dpp_move<0x141, 0xf>(v[c]);     // row_half_mirror
            v[c] += dpp_move<0x140, 0xf>(v[c]);     // row_mirror
        }
        if (l16 == 0 && idx < p.N && radius[u] > 0) {
            uint32_t clamp_bits = 0;
            float* rec = p.splat + (size_t)idx * GSR_SPLAT_FLOATS;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float t = v[c] + 0.5f;
                if (t < 0.0f) clamp_bits |= (1u << c);
                rec[GSR_SP_RGB + c] = fmaxf(t, 0.0f);
            }
            p.clamped[idx] = clamp_bits;
        }
    }
}

static void fill_pre_params(PreParams& p, const GsrView& v, const GsrGaussians& g, float* splat, uint32_t* clamped,
                            uint32_t* tiles_touched, uint2* tile_rect, uint32_t* depth_key, int32_t* radii) {
    p.N = g.count; p.W = v.width; p.H = v.height;
    p.gx = (v.width + GSR_TILE - 1) / GSR_TILE; p.gy = (v.height + GSR_TILE - 1) / GSR_TILE;
    p.deg = v.sh_degree; p.M = v.sh_coeffs; p.mod = v.scale_modifier;
    p.view = v.viewmatrix; p.proj = v.projmatrix; p.campos = v.campos;
    p.raw = (v.flags & (uint32_t)GSR_FLAG_RAW_PARAMS) != 0;
    p.means = g.means3D; p.shs = g.shs; p.shs_rest = g.shs_rest; p.colors = (v.channels == 3 && !(v.flags & (uint32_t)GSR_FLAG_COLOR_CACHED)) ? g.colors_precomp : nullptr /* wide payloads are read by id in K6/K7; a colour cache is applied where the colour pass would run */; p.opac = g.opacities;
    p.scales = g.scales; p.rots = g.rotations; p.tprecomp = g.transmat_precomp;
    p.splat = splat; p.clamped = clamped; p.tiles = tiles_touched; p.rect = tile_rect; p.dkey = depth_key; p.radii = radii;
    p.jac = nullptr;
    p.count_partial = nullptr; p.zero = nullptr; p.zero_words = 0;
}

bool gsr_color_jac_available(const GsrView& v, const GsrGaussians& g) {
    // only the 16-lane register form of the colour pass computes it (a colour cache always carries it)
    if (v.flags & (uint32_t)GSR_FLAG_COLOR_CACHED) return true;
    return g.shs != nullptr && v.sh_coeffs <= 16 && !getenv("GSR_COLOR_STAGED");
}

// GSR_FLAG_COLOR_CACHED: the colour of this view was computed by gsr_adam_sh_factored_next while it updated the
// coefficients; copy rgb and clamp bits of the visible Gaussians into their records (16 B read + 16 B written each).
__global__ void __launch_bounds__(PRE_BLOCK) color_apply_kernel(int N, const int32_t* __restrict__ radii,
                                                                const float* __restrict__ cache, float* __restrict__ splat,
                                                                uint32_t* __restrict__ clamped) {
    const int idx = blockIdx.x * PRE_BLOCK + threadIdx.x;
    if (idx >= N || radii[idx] <= 0) return;
    float* rec = splat + (size_t)idx * GSR_SPLAT_FLOATS;
    rec[GSR_SP_RGB] = cache[3 * (size_t)idx]; rec[GSR_SP_RGB + 1] = cache[3 * (size_t)idx + 1]; rec[GSR_SP_RGB + 2] = cache[3 * (size_t)idx + 2];
    clamped[idx] = reinterpret_cast<const uint32_t*>(cache)[3 * (size_t)N + idx];
}

int gsr_launch_preprocess_color(const GsrView& v, const GsrGaussians& g, float* splat, uint32_t* clamped,
                                int32_t* radii, float* color_jac, hipStream_t s) {
    if (g.count <= 0 || g.shs == nullptr) return GSR_OK;
    if (v.flags & (uint32_t)GSR_FLAG_COLOR_CACHED) {
        GsrProfileScope prof(GSR_K_PREPROCESS_FWD, s);
        hipLaunchKernelGGL(color_apply_kernel, dim3((g.count + PRE_BLOCK - 1) / PRE_BLOCK), dim3(PRE_BLOCK), 0, s, g.count, radii,
                           g.colors_precomp, splat, clamped);
        GSR_LAUNCH_CHECK();
        return GSR_OK;
    }
    PreParams p;
    fill_pre_params(p, v, g, splat, clamped, nullptr, nullptr, nullptr, radii);
    p.jac = gsr_color_jac_available(v, g) ? color_jac : nullptr;
    const int blocks = (g.count + PRE_BLOCK - 1) / PRE_BLOCK;
    GsrProfileScope prof(GSR_K_PREPROCESS_FWD, s);
    const bool stage = sh_can_stage(g.shs, g.shs_rest, v.sh_coeffs);
    if (g.shs_rest && !stage) { gsr_set_error("split SH storage needs 16-byte aligned pointers and <= 16 coefficients"); return GSR_E_UNSUPPORTED; }
    if (v.sh_coeffs <= 16 && !getenv("GSR_COLOR_STAGED")) {      // (the staged form stays selectable for A/B runs)
        const long long rows = ((long long)g.count + PC_PER_ROW - 1) / PC_PER_ROW;
        const unsigned blocks16 = (unsigned)((rows * 16 + PRE_BLOCK - 1) / PRE_BLOCK);
        hipLaunchKernelGGL(preprocess_color16_kernel, dim3(blocks16), dim3(PRE_BLOCK), 0, s, p);
    } else if (stage) {
        const size_t lds_bytes = (size_t)(PRE_BLOCK / 64) * 64 * SH_ROW_FLOATS * sizeof(float);
        hipLaunchKernelGGL(preprocess_color_kernel<true>, dim3(blocks), dim3(PRE_BLOCK), lds_bytes, s, p);
    } else {
        hipLaunchKernelGGL(preprocess_color_kernel<false>, dim3(blocks), dim3(PRE_BLOCK), 0, s, p);
    }
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}

// Geometry part of K1 (everything except the SH colour; precomputed colours are copied here).
int gsr_preprocess_fwd_blocks(int N) { return (N + PRE_BLOCK - 1) / PRE_BLOCK; }

int gsr_launch_preprocess_fwd(const GsrView& v, const GsrGaussians& g, float* splat,
                              uint32_t* clamped, uint32_t* tiles_touched, uint2* tile_rect, uint32_t* depth_key,
                              int32_t* radii, unsigned long long* count_partial, uint32_t* zero, size_t zero_words,
                              hipStream_t s) {
    if (g.count <= 0) return GSR_OK;
    PreParams p;
    fill_pre_params(p, v, g, splat, clamped, tiles_touched, tile_rect, depth_key, radii);
    p.count_partial = count_partial; p.zero = zero; p.zero_words = (int)zero_words;
    const int blocks = (g.count + PRE_BLOCK - 1) / PRE_BLOCK;
    GsrProfileScope prof(GSR_K_PREPROCESS_FWD, s);
    hipLaunchKernelGGL((preprocess_fwd_kernel<false, false>), dim3(blocks), dim3(PRE_BLOCK), 0, s, p);
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}
