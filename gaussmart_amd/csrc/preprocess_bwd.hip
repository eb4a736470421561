// K8 preprocess_bwd, two launches:
//   reduce_rows : 16 lanes per Gaussian IN DEPTH-RANK ORDER add its per-(instance, 4x4 block) gradient
//      rows -- one contiguous run of 80-byte rows -- in a fixed order and scatter the 80-byte sum to the
//      Gaussian's id;
//   preprocess_bwd : one thread per Gaussian, id order:
//   1. read its 80-byte row sum,
//   2. densification statistic for means2D.grad (consumer scene/gaussian_model.py:551-553),
//   3. chain rule  AABB centre -> T,  T -> (mean3D, scale, quaternion),  normal -> quaternion,
//   4. SH backward (clamp mask, view-direction term into mean3D).
// Restates the [U] preprocess backward; differentiates preprocess_fwd.hip exactly (same anchors).
// HBM-bound: reads 80 B x instances + 232 B, writes 232 B (192 of them dSH) per Gaussian; the SH
// block is staged through LDS both ways so global traffic is coalesced 16-byte accesses.
// GSR_FLAG_FACTORED_SH_GRAD: the 192 B of dSH are replaced by the 12 B masked colour gradient (the optimiser
// rebuilds basis x g itself, adam.hip), followed by the camera position in the 4 floats after the [N,3] block.
#include "gsr_common.h"
#include "sh_stage.h"
#include "wave_reduce.h"

#define PB_BLOCK 256

struct PreBwdParams {
    int N, W, H;
    int deg, M;
    float mod;
    const float* view; const float* proj; const float* campos;
    bool raw;
    bool aabb_grad_cutoff1;   // GSR_FLAG_AABB_GRAD_CUTOFF1
    bool factored;      // GSR_FLAG_FACTORED_SH_GRAD: masked colour gradient out, no SH gradient arrays
    const float* means; const float* shs; const float* shs_rest; const float* opac; const float* scales; const float* rots;
    const float* tprecomp;
    const int32_t* radii; const float* splat; const uint32_t* clamped;
    const float* row_sums;
    const float* jac;   // d(rgb)/d(dir) [N,9] left by the colour pass (factored mode: the SH coefficients are then not read at all)
    GsrGrads out;
};

// 16 lanes (one DPP row) per Gaussian, in depth-rank order.  Instance e owns the dense gradient rows
// [slot_off[e], slot_off[e + 1]) and a Gaussian's instances are consecutive in emission order, so ALL rows of
// the Gaussian of depth rank r are one contiguous run starting at row_begin[r] = slot_off[offs[r]] (written by the
// render_bwd launch, render_bwd_shared.h) -- a run of 64-byte rows (and one run of 8-byte xy pairs in the second array).  The 16 lanes
// read four rows (256 contiguous, 64-byte aligned bytes) per round, four rounds in flight: lane s always handles 16-byte
// column group s % 4 of row 4k + s / 4.  Three DPP shifts then add the four lanes that share a column group; the xy
// pairs are read one row per lane and summed over the 16 lanes.  Fixed order => bitwise reproducible gradients.
//
// Load balance: a splat that fills the screen owns tens of thousands of rows (one per 4x4 block it was blended into),
// and 16 lanes walking them 12 at a time would hold the whole kernel hostage (measured: 1 M splats of 24 px mean radius,
// after 400 training steps reduce_rows 0.35 -> 5.6 ms per launch).  A Gaussian with more than RR_BIG rows is therefore
// only NOTED by its 16 lanes (LDS list, at most 16 per workgroup) and summed afterwards by the WHOLE 256-thread
// workgroup: group g takes rounds g, g + 16, ... of four rows, the 16 partial sums are combined through LDS in group
// order.  The assignment of rows to lanes is fixed, so gradients stay bitwise reproducible.
#define RR_BIG 192
#define RR_GROUPS 16
// sum over the 16 lanes of a DPP row, in a fixed order; the total is valid in lane 15 of the row
__device__ __forceinline__ float rr_row_total(float v) {
    v += dpp_move<0x111, 0xf>(v);   // row_shr:1
    v += dpp_move<0x112, 0xf>(v);   // row_shr:2
    v += dpp_move<0x114, 0xf>(v);   // row_shr:4
    v += dpp_move<0x118, 0xf>(v);   // row_shr:8
    return v;
}
// lanes s, s + 4, s + 8, s + 12 of a DPP row hold the same 16-byte column group of four different rows: their sum, valid
// in lanes 12..15
__device__ __forceinline__ float4 rr_fold_phases(const float4 acc) {
    float o[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float a4 = dpp_move<0x114, 0xf>(o[c]);    // row_shr:4  (lane s receives lane s - 4)
        const float a8 = dpp_move<0x118, 0xf>(o[c]);    // row_shr:8
        const float a12 = dpp_move<0x11C, 0xf>(o[c]);   // row_shr:12
        o[c] = ((o[c] + a4) + a8) + a12;
    }
    return make_float4(o[0], o[1], o[2], o[3]);
}
// XY_IN_RGB (wide payloads, render_bwd_wide.hip): dxy arrives in colour columns 0 and 1 of the rows, there is no xy array.
template <bool XY_IN_RGB>
__global__ void __launch_bounds__(256, 8) reduce_rows_kernel(int N, const uint32_t* __restrict__ order,
                                                             const uint32_t* __restrict__ row_begin,
                                                          const float4* __restrict__ rows,
                                                          const float2* __restrict__ rows_xy,
                                                          float4* __restrict__ sums) {
    __shared__ uint32_t s_big[RR_GROUPS];
    __shared__ int s_nbig;
    __shared__ float4 s_part[RR_GROUPS][5];
    if (threadIdx.x == 0) s_nbig = 0;
    __syncthreads();
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int r = (int)(t >> 4), s16 = (int)(t & 15), group = (int)(threadIdx.x >> 4);
    bool valid = r < N;
    uint32_t s0 = 0, s1 = 0;
    uint32_t dst = 0;                 // read here, beside the row range: at the store it would be one more dependent trip
    if (r < N) { s0 = row_begin[r]; s1 = row_begin[r + 1]; dst = order[r]; }
    int n_rows = (int)(s1 - s0);
    if (n_rows > RR_BIG) {            // (uniform over the 16 lanes of the Gaussian)
        if (s16 == 0) s_big[atomicAdd(&s_nbig, 1)] = (uint32_t)r;
        valid = false;
    }
    if (!valid) n_rows = 0;
    const int sub0 = s16 >> 2;                      // 0..3: row inside the round
    {
        const float4* src = rows + (size_t)s0 * 4 + s16;
        const float2* src_xy = rows_xy + (size_t)s0 + s16;
        float2 xy = make_float2(0.f, 0.f);
        if (!XY_IN_RGB && s16 < n_rows) xy = src_xy[0];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k0 = 0; 4 * k0 < n_rows; k0 += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (4 * (k0 + u) + sub0 < n_rows) v[u] = src[16 * (k0 + u)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
        if (!XY_IN_RGB)
            for (int k = 16 + s16; k < n_rows; k += 16) { const float2 w = src_xy[k - s16]; xy.x += w.x; xy.y += w.y; }
        // all 64 lanes reach this point
        float4 o = rr_fold_phases(acc);
        float x_tot, y_tot;
        if (XY_IN_RGB) {
            x_tot = o.y; y_tot = o.z;                                  // (meaningful in lane 15: columns 12..15)
            if (s16 == 15) o = make_float4(o.x, 0.f, 0.f, 0.f);
        } else {
            x_tot = rr_row_total(xy.x); y_tot = rr_row_total(xy.y);
        }
        if (valid && s16 >= 12) sums[(size_t)dst * 5 + (s16 - 12)] = o;
        if (valid && s16 == 15) sums[(size_t)dst * 5 + 4] = make_float4(x_tot, y_tot, 0.f, 0.f);
    }
    __syncthreads();
    const int nbig = s_nbig;          // uniform over the workgroup; 0 for almost every workgroup of an ordinary frame
    for (int b = 0; b < nbig; ++b) {
        const uint32_t rb = s_big[b];
        const uint32_t b0 = row_begin[rb], b1 = row_begin[rb + 1];
        const int nr = (int)(b1 - b0);
        const float4* src = rows + (size_t)b0 * 4 + s16;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k0 = group; 4 * k0 < nr; k0 += 4 * RR_GROUPS) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                const int k = k0 + u * RR_GROUPS;
                if (4 * k + sub0 < nr) v[u] = src[(size_t)16 * k];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
        float2 xy = make_float2(0.f, 0.f);
        if (!XY_IN_RGB)
            for (int k = (int)threadIdx.x; k < nr; k += 256) { const float2 w = rows_xy[(size_t)b0 + k]; xy.x += w.x; xy.y += w.y; }
        float4 o = rr_fold_phases(acc);
        float x_tot, y_tot;
        if (XY_IN_RGB) {
            x_tot = o.y; y_tot = o.z;
            if (s16 == 15) o = make_float4(o.x, 0.f, 0.f, 0.f);
        } else {
            x_tot = rr_row_total(xy.x); y_tot = rr_row_total(xy.y);
        }
        if (s16 >= 12) s_part[group][s16 - 12] = o;
        if (s16 == 15) s_part[group][4] = make_float4(x_tot, y_tot, 0.f, 0.f);
        __syncthreads();
        if (threadIdx.x < 5) {
            float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int g = 0; g < RR_GROUPS; ++g) {
                const float4 v = s_part[g][threadIdx.x];
                tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w;
            }
            sums[(size_t)order[rb] * 5 + threadIdx.x] = tot;
        }
        __syncthreads();
    }
}

int gsr_launch_reduce_rows(int N, const uint32_t* order, const uint32_t* row_begin,
                           const float* grad_rows, const float* grad_xy, float* row_sums, hipStream_t s) {
    if (N <= 0) return GSR_OK;
    GsrProfileScope prof(GSR_K_PREPROCESS_BWD, s);
    const long long n_threads = (long long)N * 16;
    const dim3 grid((unsigned)((n_threads + 255) / 256)), block(256);
    const float4* rows = reinterpret_cast<const float4*>(grad_rows);
    float4* sums = reinterpret_cast<float4*>(row_sums);
    if (grad_xy) hipLaunchKernelGGL(reduce_rows_kernel<false>, grid, block, 0, s, N, order, row_begin, rows,
                                    reinterpret_cast<const float2*>(grad_xy), sums);
    else hipLaunchKernelGGL(reduce_rows_kernel<true>, grid, block, 0, s, N, order, row_begin, rows,
                            static_cast<const float2*>(nullptr), sums);
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}

template <bool STAGE_SH>
__global__ void __launch_bounds__(PB_BLOCK) preprocess_bwd_kernel(PreBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int idx = blockIdx.x * PB_BLOCK + tid;
    const int wave_first = blockIdx.x * PB_BLOCK + wave * 64;
    const int row_f = p.M * 3;
    const int n_here = min(64, p.N - wave_first);
    const bool has_sh = p.shs != nullptr;

    float* wl = lds + wave * 64 * SH_ROW_FLOATS;
    float* my_sh = wl + lane * SH_ROW_FLOATS;
    if (STAGE_SH)
        sh_stage<true>(wl, const_cast<float*>(p.shs), const_cast<float*>(p.shs_rest), p.M, wave_first, n_here, lane);
    const bool use_jac = !STAGE_SH && p.jac != nullptr;

    const bool valid = idx < p.N;
    const bool visible = valid && p.radii[idx] > 0;

    float dmean[3] = {0.f, 0.f, 0.f};
    float dT[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dxy0 = 0.f, dxy1 = 0.f, dn[3] = {0.f, 0.f, 0.f}, dopa = 0.f, drgb[3] = {0.f, 0.f, 0.f};
    float d2d0 = 0.f, d2d1 = 0.f;
    float dscale0 = 0.f, dscale1 = 0.f, dq[4] = {0.f, 0.f, 0.f, 0.f};

    float Tu[3], Tv[3], Tw[3];
    if (visible) {
        // ---- 1. instance rows ---------------------------------------------------------------
        float acc[20];
        const float4* rs = reinterpret_cast<const float4*>(p.row_sums + (size_t)idx * GSR_GROW_FLOATS);
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const float4 v = rs[q];
            acc[4 * q + 0] = v.x; acc[4 * q + 1] = v.y; acc[4 * q + 2] = v.z; acc[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) dT[k] = acc[GSR_GR_T + k];
        dxy0 = acc[GSR_GR_XY]; dxy1 = acc[GSR_GR_XY + 1];
        dn[0] = acc[GSR_GR_NRM]; dn[1] = acc[GSR_GR_NRM + 1]; dn[2] = acc[GSR_GR_NRM + 2];
        dopa = acc[GSR_GR_OPA];
        drgb[0] = acc[GSR_GR_RGB]; drgb[1] = acc[GSR_GR_RGB + 1]; drgb[2] = acc[GSR_GR_RGB + 2];

        const float* rec = p.splat + (size_t)idx * GSR_SPLAT_FLOATS;
#pragma unroll
        for (int i = 0; i < 3; ++i) { Tu[i] = rec[GSR_SP_TU + i]; Tv[i] = rec[GSR_SP_TV + i]; Tw[i] = rec[GSR_SP_TW + i]; }

        // ---- 2. densification statistic from the RAW dL/dT ---------------------------------
        d2d0 = dT[2] * Tw[2] * 0.5f * (float)p.W;
        d2d1 = dT[5] * Tw[2] * 0.5f * (float)p.H;

        // ---- 3a. AABB centre -> T -----------------------------------------------------------
        if (dxy0 != 0.f || dxy1 != 0.f) {
            // (GSR_FLAG_AABB_GRAD_CUTOFF1: the recalled variant that chains this gradient with weights (1, 1, -1))
            const float cc = p.aabb_grad_cutoff1 ? 1.0f : GSR_CUTOFF * GSR_CUTOFF;
            const float t[3] = {cc, cc, -1.0f};
            const float d = t[0] * Tw[0] * Tw[0] + t[1] * Tw[1] * Tw[1] + t[2] * Tw[2] * Tw[2];
            const float inv_d = d != 0.0f ? 1.0f / d : 0.0f;
            float dL_dd = 0.f;
            float dTw_add[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float f = t[i] * inv_d;
                dT[0 + i] += dxy0 * f * Tw[i];
                dT[3 + i] += dxy1 * f * Tw[i];
                dTw_add[i] = dxy0 * f * Tu[i] + dxy1 * f * Tv[i];
                const float dL_df = dxy0 * Tu[i] * Tw[i] + dxy1 * Tv[i] * Tw[i];
                dL_dd += dL_df * f;
            }
            dL_dd *= -inv_d;
#pragma unroll
            for (int i = 0; i < 3; ++i) dT[6 + i] += dTw_add[i] + dL_dd * t[i] * Tw[i] * 2.0f;
        }

        if (p.tprecomp == nullptr) {
            // ---- 3b. T -> rows of H = [tu,0; tv,0; p,1] ------------------------------------
            const float* P = p.proj;
            const float hw = 0.5f * (float)p.W, hh = 0.5f * (float)p.H;
            const float cw = 0.5f * (float)(p.W - 1), ch = 0.5f * (float)(p.H - 1);
            float dtu[3], dtv[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                // M3[r][c]: column c in (x, y, w) of projmatrix @ ndc2pix
                const float m0 = P[4 * r + 0] * hw + P[4 * r + 3] * cw;
                const float m1 = P[4 * r + 1] * hh + P[4 * r + 3] * ch;
                const float m2 = P[4 * r + 3];
                dtu[r] = dT[0] * m0 + dT[3] * m1 + dT[6] * m2;
                dtv[r] = dT[1] * m0 + dT[4] * m1 + dT[7] * m2;
                dmean[r] = dT[2] * m0 + dT[5] * m1 + dT[8] * m2;
            }
            float qw = p.rots[4 * idx + 0], qx = p.rots[4 * idx + 1], qy = p.rots[4 * idx + 2], qz = p.rots[4 * idx + 3];
            const float qn2 = qw * qw + qx * qx + qy * qy + qz * qz;
            const float s = rsqrtf(qn2);
            qw *= s; qx *= s; qy *= s; qz *= s;
            const float r00 = 1.f - 2.f * (qy * qy + qz * qz), r10 = 2.f * (qx * qy + qw * qz), r20 = 2.f * (qx * qz - qw * qy);
            const float r01 = 2.f * (qx * qy - qw * qz), r11 = 1.f - 2.f * (qx * qx + qz * qz), r21 = 2.f * (qy * qz + qw * qx);
            const float r02 = 2.f * (qx * qz + qw * qy), r12 = 2.f * (qy * qz - qw * qx), r22 = 1.f - 2.f * (qx * qx + qy * qy);
            float sx = p.scales[2 * idx + 0], sy = p.scales[2 * idx + 1];
            if (p.raw) { sx = expf(sx); sy = expf(sy); }
            sx *= p.mod; sy *= p.mod;

            // normal = sign * (R[:,2] @ V3); the sign is recomputed exactly like the forward
            const float* V = p.view;
            const float px = p.means[3 * idx + 0], py = p.means[3 * idx + 1], pz = p.means[3 * idx + 2];
            const float vx = px * V[0] + py * V[4] + pz * V[8] + V[12];
            const float vy = px * V[1] + py * V[5] + pz * V[9] + V[13];
            const float vz = px * V[2] + py * V[6] + pz * V[10] + V[14];
            const float nv0 = r02 * V[0] + r12 * V[4] + r22 * V[8];
            const float nv1 = r02 * V[1] + r12 * V[5] + r22 * V[9];
            const float nv2 = r02 * V[2] + r12 * V[6] + r22 * V[10];
            const float sgn = (-(vx * nv0 + vy * nv1 + vz * nv2)) > 0.f ? 1.f : -1.f;
            float dtn[3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
                dtn[r] = sgn * (dn[0] * V[4 * r + 0] + dn[1] * V[4 * r + 1] + dn[2] * V[4 * r + 2]);

            dscale0 = (dtu[0] * r00 + dtu[1] * r10 + dtu[2] * r20) * p.mod;
            dscale1 = (dtv[0] * r01 + dtv[1] * r11 + dtv[2] * r21) * p.mod;
            // dL/dR, column-wise
            const float d00 = dtu[0] * sx, d10 = dtu[1] * sx, d20 = dtu[2] * sx;
            const float d01 = dtv[0] * sy, d11 = dtv[1] * sy, d21 = dtv[2] * sy;
            const float d02 = dtn[0], d12 = dtn[1], d22 = dtn[2];
            const float gr = 2.f * (-qz * d01 + qy * d02 + qz * d10 - qx * d12 - qy * d20 + qx * d21);
            const float gx = 2.f * (qy * d01 + qz * d02 + qy * d10 - 2.f * qx * d11 - qw * d12 + qz * d20 + qw * d21 - 2.f * qx * d22);
            const float gy = 2.f * (-2.f * qy * d00 + qx * d01 + qw * d02 + qx * d10 + qz * d12 - qw * d20 + qz * d21 - 2.f * qy * d22);
            const float gz = 2.f * (-2.f * qz * d00 - qw * d01 + qx * d02 + qw * d10 - 2.f * qz * d11 + qy * d12 + qx * d20 + qy * d21);
            dq[0] = gr * s; dq[1] = gx * s; dq[2] = gy * s; dq[3] = gz * s;
            if (p.raw) {
                // through the activations: exp for the scales, normalize for the quaternion
                dscale0 *= sx / p.mod; dscale1 *= sy / p.mod;      // d exp(x)/dx = exp(x)
                // torch.nn.functional.normalize divides by max(|q|, 1e-12) (scene/gaussian_model.py:109): below that norm the
                // activation is a plain scaling, its derivative has no projection, and the operator's own (detached)
                // normalisation supplies the rest of 1 / |q| -- the row above is already that gradient
                if (!(qn2 < 1e-24f)) {
                    const float qd = qw * gr + qx * gx + qy * gy + qz * gz;
                    dq[0] = (gr - qw * qd) * s; dq[1] = (gx - qx * qd) * s;
                    dq[2] = (gy - qy * qd) * s; dq[3] = (gz - qz * qd) * s;
                }
            }
        }
    }

    // ---- 4. SH backward ---------------------------------------------------------------------
    if (has_sh) {
        float basis[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) basis[k] = 0.f;
        float g[3] = {0.f, 0.f, 0.f};
        if (visible) {
            const uint32_t cb = p.clamped[idx];
            g[0] = (cb & 1u) ? 0.f : drgb[0];
            g[1] = (cb & 2u) ? 0.f : drgb[1];
            g[2] = (cb & 4u) ? 0.f : drgb[2];
            const float ox = p.means[3 * idx + 0] - p.campos[0];
            const float oy = p.means[3 * idx + 1] - p.campos[1];
            const float oz = p.means[3 * idx + 2] - p.campos[2];
            const float il = 1.0f / sqrtf(ox * ox + oy * oy + oz * oz);
            const float x = ox * il, y = oy * il, z = oz * il;
            float ddx = 0.f, ddy = 0.f, ddz = 0.f;
            if (use_jac) {
                // dL/ddir = g^T J with J = d(rgb)/d(dir) from the colour pass: 36 bytes instead of the 192 of the coefficients
                const float* J = p.jac + (size_t)idx * 9;
                ddx = g[0] * J[0] + g[1] * J[3] + g[2] * J[6];
                ddy = g[0] * J[1] + g[1] * J[4] + g[2] * J[7];
                ddz = g[0] * J[2] + g[1] * J[5] + g[2] * J[8];
            } else {
                const int deg = p.deg;
                float dbx[16], dby[16], dbz[16];
    #pragma unroll
                for (int k = 0; k < 16; ++k) { dbx[k] = 0.f; dby[k] = 0.f; dbz[k] = 0.f; }
                basis[0] = GSR_SH_C0;
                if (deg > 0) {
                    basis[1] = -GSR_SH_C1 * y; basis[2] = GSR_SH_C1 * z; basis[3] = -GSR_SH_C1 * x;
                    dby[1] = -GSR_SH_C1; dbz[2] = GSR_SH_C1; dbx[3] = -GSR_SH_C1;
                    if (deg > 1) {
                        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        basis[4] = GSR_SH_C2_0 * xy; basis[5] = GSR_SH_C2_1 * yz;
                        basis[6] = GSR_SH_C2_2 * (2.f * zz - xx - yy);
                        basis[7] = GSR_SH_C2_3 * xz; basis[8] = GSR_SH_C2_4 * (xx - yy);
                        dbx[4] = GSR_SH_C2_0 * y; dby[4] = GSR_SH_C2_0 * x;
                        dby[5] = GSR_SH_C2_1 * z; dbz[5] = GSR_SH_C2_1 * y;
                        dbx[6] = GSR_SH_C2_2 * -2.f * x; dby[6] = GSR_SH_C2_2 * -2.f * y; dbz[6] = GSR_SH_C2_2 * 4.f * z;
                        dbx[7] = GSR_SH_C2_3 * z; dbz[7] = GSR_SH_C2_3 * x;
                        dbx[8] = GSR_SH_C2_4 * 2.f * x; dby[8] = GSR_SH_C2_4 * -2.f * y;
                        if (deg > 2) {
                            basis[9] = GSR_SH_C3_0 * y * (3.f * xx - yy);
                            basis[10] = GSR_SH_C3_1 * xy * z;
                            basis[11] = GSR_SH_C3_2 * y * (4.f * zz - xx - yy);
                            basis[12] = GSR_SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
                            basis[13] = GSR_SH_C3_4 * x * (4.f * zz - xx - yy);
                            basis[14] = GSR_SH_C3_5 * z * (xx - yy);
                            basis[15] = GSR_SH_C3_6 * x * (xx - 3.f * yy);
                            dbx[9] = GSR_SH_C3_0 * 6.f * xy; dby[9] = GSR_SH_C3_0 * (3.f * xx - 3.f * yy);
                            dbx[10] = GSR_SH_C3_1 * yz; dby[10] = GSR_SH_C3_1 * xz; dbz[10] = GSR_SH_C3_1 * xy;
                            dbx[11] = GSR_SH_C3_2 * -2.f * xy; dby[11] = GSR_SH_C3_2 * (4.f * zz - xx - 3.f * yy); dbz[11] = GSR_SH_C3_2 * 8.f * yz;
                            dbx[12] = GSR_SH_C3_3 * -6.f * xz; dby[12] = GSR_SH_C3_3 * -6.f * yz; dbz[12] = GSR_SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy);
                            dbx[13] = GSR_SH_C3_4 * (4.f * zz - 3.f * xx - yy); dby[13] = GSR_SH_C3_4 * -2.f * xy; dbz[13] = GSR_SH_C3_4 * 8.f * xz;
                            dbx[14] = GSR_SH_C3_5 * 2.f * xz; dby[14] = GSR_SH_C3_5 * -2.f * yz; dbz[14] = GSR_SH_C3_5 * (xx - yy);
                            dbx[15] = GSR_SH_C3_6 * (3.f * xx - 3.f * yy); dby[15] = GSR_SH_C3_6 * -6.f * xy;
                        }
                    }
                }
                // dL/ddir = sum_c g_c * sum_k dbasis_k * sh[k][c]
                const int nk = (deg + 1) * (deg + 1);
                const float* shp = STAGE_SH ? my_sh : p.shs + (size_t)idx * row_f;
    #pragma unroll
                for (int k = 1; k < 16; ++k) {
                    if (k < nk && k < p.M) {
                        const float w = g[0] * shp[3 * k] + g[1] * shp[3 * k + 1] + g[2] * shp[3 * k + 2];
                        ddx += dbx[k] * w; ddy += dby[k] * w; ddz += dbz[k] * w;
                    }
                }
            }
            // through the normalisation dir = o / |o|
            const float dotp = x * ddx + y * ddy + z * ddz;
            dmean[0] += (ddx - x * dotp) * il;
            dmean[1] += (ddy - y * dotp) * il;
            dmean[2] += (ddz - z * dotp) * il;
        }
        if (p.factored) {
            // the optimiser rebuilds basis_k x g itself (gsr_adam_sh_factored): 12 bytes out instead of 192
            if (valid) {
                p.out.dL_dcolors[3 * idx + 0] = g[0]; p.out.dL_dcolors[3 * idx + 1] = g[1]; p.out.dL_dcolors[3 * idx + 2] = g[2];
            }
            if (idx == 0) {      // the record of a view is self-contained: [N,3] colour gradient + camera position
                float* tail = p.out.dL_dcolors + 3 * (size_t)p.N;
                tail[0] = p.campos[0]; tail[1] = p.campos[1]; tail[2] = p.campos[2]; tail[3] = 0.f;
            }
        } else if (STAGE_SH) {
            // overwrite the staged coefficients with their gradients, then stream the tile out
            __builtin_amdgcn_wave_barrier();
            if (valid) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    if (k < p.M) {
                        my_sh[3 * k + 0] = basis[k] * g[0];
                        my_sh[3 * k + 1] = basis[k] * g[1];
                        my_sh[3 * k + 2] = basis[k] * g[2];
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            sh_stage<false>(wl, p.out.dL_dshs, p.out.dL_dshs_rest, p.M, wave_first, n_here, lane);
        } else if (valid) {
            float* o = p.out.dL_dshs + (size_t)idx * row_f;
            for (int k = 0; k < p.M; ++k) {
                const float bk = k < 16 ? basis[k] : 0.f;
                o[3 * k + 0] = bk * g[0]; o[3 * k + 1] = bk * g[1]; o[3 * k + 2] = bk * g[2];
            }
        }
    }

    if (!valid) return;
    p.out.dL_dmeans3D[3 * idx + 0] = dmean[0];
    p.out.dL_dmeans3D[3 * idx + 1] = dmean[1];
    p.out.dL_dmeans3D[3 * idx + 2] = dmean[2];
    p.out.dL_dmeans2D[3 * idx + 0] = d2d0;
    p.out.dL_dmeans2D[3 * idx + 1] = d2d1;
    p.out.dL_dmeans2D[3 * idx + 2] = 0.f;
    if (p.raw && visible) {
        const float o = 1.0f / (1.0f + expf(-p.opac[idx]));
        dopa *= o * (1.0f - o);                                 // d sigmoid
    }
    p.out.dL_dopacity[idx] = dopa;
    if (p.out.dL_dcolors && !p.factored) {
        p.out.dL_dcolors[3 * idx + 0] = drgb[0]; p.out.dL_dcolors[3 * idx + 1] = drgb[1]; p.out.dL_dcolors[3 * idx + 2] = drgb[2];
    }
    if (p.out.dL_dscales) { p.out.dL_dscales[2 * idx + 0] = dscale0; p.out.dL_dscales[2 * idx + 1] = dscale1; }
    if (p.out.dL_drotations) {
        p.out.dL_drotations[4 * idx + 0] = dq[0]; p.out.dL_drotations[4 * idx + 1] = dq[1];
        p.out.dL_drotations[4 * idx + 2] = dq[2]; p.out.dL_drotations[4 * idx + 3] = dq[3];
    }
    if (p.out.dL_dtransmat) {
#pragma unroll
        for (int k = 0; k < 9; ++k) p.out.dL_dtransmat[9 * (size_t)idx + k] = dT[k];
    }
}

int gsr_launch_preprocess_bwd(const GsrView& v, const GsrGaussians& g, const int32_t* radii,
                              const float* splat, const uint32_t* clamped,
                              const float* row_sums, const float* color_jac, const GsrGrads& out, hipStream_t s) {
    if (g.count <= 0) return GSR_OK;
    PreBwdParams p;
    p.N = g.count; p.W = v.width; p.H = v.height; p.deg = v.sh_degree; p.M = v.sh_coeffs;
    p.mod = v.scale_modifier; p.view = v.viewmatrix; p.proj = v.projmatrix; p.campos = v.campos;
    p.raw = (v.flags & (uint32_t)GSR_FLAG_RAW_PARAMS) != 0;
    p.aabb_grad_cutoff1 = (v.flags & (uint32_t)GSR_FLAG_AABB_GRAD_CUTOFF1) != 0;
    p.factored = (v.flags & (uint32_t)GSR_FLAG_FACTORED_SH_GRAD) != 0 && g.shs != nullptr;
    if (p.factored && !out.dL_dcolors) { gsr_set_error("GSR_FLAG_FACTORED_SH_GRAD needs dL_dcolors"); return GSR_E_INVALID; }
    p.means = g.means3D; p.shs = g.shs; p.shs_rest = g.shs_rest; p.opac = g.opacities; p.scales = g.scales; p.rots = g.rotations;
    p.tprecomp = g.transmat_precomp; p.radii = radii; p.splat = splat; p.clamped = clamped;
    p.row_sums = row_sums; p.out = out;
    // factored mode + Jacobian from the colour pass: no SH read, no SH write, no LDS
    p.jac = (p.factored && gsr_color_jac_available(v, g)) ? color_jac : nullptr;
    const int blocks = (g.count + PB_BLOCK - 1) / PB_BLOCK;
    GsrProfileScope prof(GSR_K_PREPROCESS_BWD, s);
    const bool stage = sh_can_stage(g.shs, g.shs_rest, v.sh_coeffs) &&
                       (p.factored || sh_can_stage(out.dL_dshs, out.dL_dshs_rest, v.sh_coeffs));
    if (g.shs_rest && (!stage || (!p.factored && !out.dL_dshs_rest))) { gsr_set_error("split SH storage needs 16-byte aligned pointers, <= 16 coefficients and dL_dshs_rest"); return GSR_E_UNSUPPORTED; }
    if (stage && p.jac == nullptr) {
        const size_t lds_bytes = (size_t)(PB_BLOCK / 64) * 64 * SH_ROW_FLOATS * sizeof(float);
        hipLaunchKernelGGL(preprocess_bwd_kernel<true>, dim3(blocks), dim3(PB_BLOCK), lds_bytes, s, p);
    } else {
        hipLaunchKernelGGL(preprocess_bwd_kernel<false>, dim3(blocks), dim3(PB_BLOCK), 0, s, p);
    }
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}
