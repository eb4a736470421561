// K7 render_bwd: back-to-front replay of every tile -- no floating-point atomics anywhere.
//
// Restates the [U]/[P] backward of the surfel compositing (colour, expected depth, alpha, normal,
// median depth, distortion), see DESIGN.md section "render_bwd" for the recursion; it is the exact
// derivative of render_fwd except for the two flagged quirks (GSR_FLAG_*).
//
// MI355X mapping: WAVE-INDEPENDENT, like the forward.  Each wave64 of a tile's workgroup owns an 8x8
// pixel quad and never synchronises with the other three:
//   * it starts at the deepest list entry ANY OF ITS 64 PIXELS reached (quad-level, not tile-level)
//     and streams the (tile, depth)-ordered splat records backwards, 64 per batch, coalesced, into
//     its private LDS slice while the next batch is prefetched into registers;
//   * a 64-bit ballot of the forward's "blended into >= 1 pixel of this quad" bytes selects EXACTLY
//     the splats that carry gradient here; nothing else is even evaluated;
//   * the suffix recursions of colour, depth, alpha and normal are collapsed into ONE scalar
//     recursion (they are linear: q_i = c_i.dL/dC + z_i dL/dD + dL/dA + n_i.dL/dN);
//   * the 18 partial derivatives are summed over the 64 pixels with the transposed butterfly below
//     (gfx950 v_permlane32_swap / v_permlane16_swap + DPP, ~50 VALU) and land distributed over the
//     lanes, which store them straight into the wave's OWN 80-byte sub-row of that instance
//     (4 sub-rows per instance, one per quad) plus a 1-byte "written" flag;
//   * preprocess_bwd adds the flagged sub-rows in fixed order => bitwise reproducible gradients, and
//     HBM sees plain streaming stores instead of ~18 atomics per pixel-splat pair.
#include "gsr_common.h"
#include "pair_eval.h"
#include "wave_reduce.h"

// five coalesced 16-byte loads per lane = 64 records of 80 bytes; pieces beyond `lim` read as zero
#define GSR_LOAD5(ptr, lim)                                            \
    do {                                                               \
        const int lim_ = (lim);                                        \
        pf0 = zero4; pf1 = zero4; pf2 = zero4; pf3 = zero4; pf4 = zero4; \
        if (lane < lim_) pf0 = (ptr)[lane];                            \
        if (64 + lane < lim_) pf1 = (ptr)[64 + lane];                  \
        if (128 + lane < lim_) pf2 = (ptr)[128 + lane];                \
        if (192 + lane < lim_) pf3 = (ptr)[192 + lane];                \
        if (256 + lane < lim_) pf4 = (ptr)[256 + lane];                \
    } while (0)

#define RB_BLOCK 256
#define RB_WAVES 4
#define RB_ROW GSR_GROW_FLOATS   // 20 floats

struct RenderBwdParams {
    int W, H, gx;
    uint32_t flags;
    const uint32_t* ranges; const uint32_t* inst_row;
    const float4* stream; const uint8_t* touch; const float* bg;
    const float* final_T; const uint32_t* n_contrib;
    const float* dL_dcolor; const float* dL_dallmap;
    float* grad_rows; uint8_t* row_flags;
    // wide payload (FEAT16 > 0): features by Gaussian id, their gradient sub-rows [(instance*4+quad)*C + ch]
    const float* feat; const uint32_t* point_list; float* feat_rows; int C;
};

#ifndef RB_MIN_WAVES
#define RB_MIN_WAVES 5   // <= 96 VGPRs: measured 1.45 -> 1.35 ms at 1M/1080p
#endif
// FEAT16: see render_fwd.hip -- 0 = RGB from the record, 1..4 = up to 16*FEAT16 feature channels by id.
template <int FEAT16>
__global__ void __launch_bounds__(RB_BLOCK, FEAT16 == 0 ? RB_MIN_WAVES : 2) render_bwd_kernel(RenderBwdParams p) {
    __shared__ float4 s_rec_all[RB_WAVES][64 * 5];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    float4* s_rec = s_rec_all[wave];
    const int tile_x = blockIdx.x, tile_y = blockIdx.y;
    const int qx0 = tile_x * GSR_TILE + (wave & 1) * 8, qy0 = tile_y * GSR_TILE + (wave >> 1) * 8;
    const int pxi = qx0 + (lane & 7), pyi = qy0 + (lane >> 3);
    const bool inside = pxi < p.W && pyi < p.H;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const int pix_id = pyi * p.W + pxi;
    const int HW = p.W * p.H;

    const uint32_t tile = (uint32_t)(tile_y * p.gx + tile_x);
    const uint32_t r0 = p.ranges[2 * tile];

    // the deepest list entry any pixel of THIS QUAD reached
    const int last_contributor = inside ? (int)p.n_contrib[pix_id] : 0;
    int max_contrib = last_contributor;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) max_contrib = max(max_contrib, __shfl_xor(max_contrib, d, 64));
    max_contrib = __builtin_amdgcn_readfirstlane(max_contrib);
    if (max_contrib == 0) return;

    const bool clamp_pass = (p.flags & GSR_FLAG_CLAMP_PASSTHROUGH) != 0;
    const bool filter_depth_quirk = (p.flags & GSR_FLAG_FILTER_DEPTH_GRAD) != 0;
    const bool no_cull = (p.flags & (uint32_t)GSR_FLAG_DEBUG_NO_CULL) != 0;

    // per-pixel state saved by the forward
    const float T_final = inside ? p.final_T[pix_id] : 0.f;
    const float final_D = inside ? p.final_T[pix_id + HW] : 0.f;       // sum m w
    const float final_D2 = inside ? p.final_T[pix_id + 2 * HW] : 0.f;  // sum m^2 w
    const float final_A = 1.0f - T_final;
    const int median_contributor = inside ? (int)p.n_contrib[pix_id + HW] : 0;

    float dL_dpix0 = 0.f, dL_dpix1 = 0.f, dL_dpix2 = 0.f;
    constexpr int NF = FEAT16 > 0 ? 16 * FEAT16 : 1;
    float dL_dpixf[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) dL_dpixf[k] = 0.f;
    float dL_ddepth = 0.f, dL_daccum = 0.f, dL_dreg = 0.f, dL_dmedian = 0.f;
    float dL_dn0 = 0.f, dL_dn1 = 0.f, dL_dn2 = 0.f;
    if (inside) {
        if (FEAT16 == 0) {
            dL_dpix0 = p.dL_dcolor[pix_id]; dL_dpix1 = p.dL_dcolor[pix_id + HW]; dL_dpix2 = p.dL_dcolor[pix_id + 2 * HW];
        } else {
#pragma unroll
            for (int k = 0; k < NF; ++k)
                if (k < p.C) dL_dpixf[k] = p.dL_dcolor[pix_id + (size_t)k * HW];
        }
        dL_ddepth = p.dL_dallmap[pix_id + 0 * HW];
        dL_daccum = p.dL_dallmap[pix_id + 1 * HW];
        dL_dn0 = p.dL_dallmap[pix_id + 2 * HW];
        dL_dn1 = p.dL_dallmap[pix_id + 3 * HW];
        dL_dn2 = p.dL_dallmap[pix_id + 4 * HW];
        dL_dmedian = p.dL_dallmap[pix_id + 5 * HW];
        dL_dreg = p.dL_dallmap[pix_id + 6 * HW];
    }
    float bg_dot_dpixel = 0.f;
    if (FEAT16 == 0) {
        bg_dot_dpixel = p.bg[0] * dL_dpix0 + p.bg[1] * dL_dpix1 + p.bg[2] * dL_dpix2;
    } else {
#pragma unroll
        for (int k = 0; k < NF; ++k)
            if (k < p.C) bg_dot_dpixel += p.bg[k] * dL_dpixf[k];
    }

    // running state of the back-to-front recursion
    float T = T_final;
    float last_alpha = 0.f, last_q = 0.f, acc_q = 0.f, last_dL_dT = 0.f;

    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 pf0, pf1, pf2, pf3, pf4;
    uint32_t pf_touch = 0;   // named (not an array): keeps the prefetch in VGPRs, not scratch
    uint32_t pf_row = 0;
    uint32_t pf_id = 0;
    int hi = max_contrib;
    {
        const int lo = max(0, hi - 64), cnt = hi - lo;
        const float4* src = p.stream + (size_t)(r0 + lo) * 5;
GSR_LOAD5(src, cnt * 5);
        pf_row = lane < cnt ? p.inst_row[r0 + lo + lane] : 0u;
        pf_touch = lane < cnt ? p.touch[((size_t)r0 + lo + lane) * 4 + wave] : 0u;
        if (FEAT16 > 0) pf_id = lane < cnt ? p.point_list[r0 + lo + lane] : 0u;
    }

    while (hi > 0) {
        const int lo = max(0, hi - 64), nb = hi - lo;
s_rec[lane] = pf0; s_rec[64 + lane] = pf1; s_rec[128 + lane] = pf2; s_rec[192 + lane] = pf3; s_rec[256 + lane] = pf4;
        const uint32_t row_of_lane = pf_row;          // emission index of staged entry `lane`
        const uint32_t touch_of_lane = pf_touch;      // did the forward blend staged entry `lane` in this quad?
        const uint32_t id_of_lane = pf_id;            // Gaussian id of staged entry `lane` (wide payload only)
        {   // prefetch the next (shallower) batch
            const int hi2 = lo, lo2 = max(0, hi2 - 64), cnt = hi2 - lo2;
            const float4* src = p.stream + (size_t)(r0 + lo2) * 5;
GSR_LOAD5(src, cnt * 5);
            pf_row = lane < cnt ? p.inst_row[r0 + lo2 + lane] : 0u;
            pf_touch = lane < cnt ? p.touch[((size_t)r0 + lo2 + lane) * 4 + wave] : 0u;
            if (FEAT16 > 0) pf_id = lane < cnt ? p.point_list[r0 + lo2 + lane] : 0u;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // only the splats the forward blended into >= 1 pixel of this quad carry any gradient
        const bool ov = lane < nb && (no_cull || touch_of_lane != 0u);
        unsigned long long todo_mask = __ballot(ov);
        while (todo_mask) {
            const int j = 63 - __builtin_clzll(todo_mask);      // deepest first
            todo_mask &= ~(1ull << j);
            const int cidx = lo + j;                            // 0-based position in the tile list
            const float4 a0 = s_rec[j * 5 + 0], a1 = s_rec[j * 5 + 1], a2 = s_rec[j * 5 + 2];
            const float4 a3 = s_rec[j * 5 + 3];
            GsrPair pr;
            bool active = cidx < last_contributor;
            if (active) active = gsr_pair_eval(pxf, pyf, a0, a1, a2, a3.z, pr);
            if (!__any(active)) continue;           // whole wave untouched by this splat

            float gT[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            float gxy0 = 0.f, gxy1 = 0.f, gn0 = 0.f, gn1 = 0.f, gn2 = 0.f, gopa = 0.f;
            float gc0 = 0.f, gc1 = 0.f, gc2 = 0.f;
            float w_pair = 0.f;   // blending weight of this pair (wide payload: d feature = w * dL/dpixel)
            if (active) {
                const float4 a4 = s_rec[j * 5 + 4];
                const float alpha = pr.alpha, G = pr.G, c_d = pr.depth;
                const float one_m_alpha = 1.0f - alpha;
                const float inv_oma = gsr_rcp(one_m_alpha);
                T = T * inv_oma;
                const float w = alpha * T;

                // colour, expected depth, alpha and normal share one suffix recursion:
                //   q_i = c_i . dL/dC + z_i dL/dD + 1 dL/dA + n_i . dL/dN
                const float c0 = a3.w, c1 = a4.x, c2 = a4.y;
                const float n0 = a2.w, n1 = a3.x, n2 = a3.y;
                float q = c_d * dL_ddepth + dL_daccum + n0 * dL_dn0 + n1 * dL_dn1 + n2 * dL_dn2;
                if (FEAT16 == 0) {
                    q = c0 * dL_dpix0 + c1 * dL_dpix1 + c2 * dL_dpix2 + c_d * dL_ddepth + dL_daccum
                      + n0 * dL_dn0 + n1 * dL_dn1 + n2 * dL_dn2;
                } else {
                    const uint32_t gid = (uint32_t)__builtin_amdgcn_readlane((int)id_of_lane, j);   // wave-uniform
                    const float4* f = reinterpret_cast<const float4*>(p.feat + (size_t)gid * p.C);
                    float qc = 0.f;
#pragma unroll
                    for (int k = 0; k < NF / 4; ++k) {
                        if (4 * k < p.C) {
                            const float4 v = f[k];
                            qc += v.x * dL_dpixf[4 * k] + v.y * dL_dpixf[4 * k + 1] + v.z * dL_dpixf[4 * k + 2] + v.w * dL_dpixf[4 * k + 3];
                        }
                    }
                    q += qc;
                    w_pair = w;
                }
                acc_q = last_alpha * last_q + (1.f - last_alpha) * acc_q;
                last_q = q;
                float dL_dalpha = q - acc_q;
                gc0 = w * dL_dpix0; gc1 = w * dL_dpix1; gc2 = w * dL_dpix2;
                gn0 = w * dL_dn0; gn1 = w * dL_dn1; gn2 = w * dL_dn2;

                // distortion, median depth
                float dmd_dd;
                const float m_d = gsr_depth_map(c_d, dmd_dd);
                float dL_dz = w * dL_ddepth;
                if (cidx == median_contributor - 1) dL_dz += dL_dmedian;
                const float dL_dweight = (final_D2 + m_d * m_d * final_A - 2.f * m_d * final_D) * dL_dreg;
                dL_dalpha += dL_dweight - last_dL_dT;
                last_dL_dT = dL_dweight * alpha + one_m_alpha * last_dL_dT;
                dL_dz += 2.0f * w * (m_d * final_A - final_D) * dL_dreg * dmd_dd;

                dL_dalpha *= T;
                last_alpha = alpha;
                // alpha also scales how much background shows through
                dL_dalpha -= T_final * inv_oma * bg_dot_dpixel;

                // alpha = min(0.99, opa * G)
                const float dL_daraw = (clamp_pass || pr.araw <= GSR_ALPHA_MAX) ? dL_dalpha : 0.f;
                const float dL_dG = a3.z * dL_daraw;
                gopa = G * dL_daraw;

                const float Twx = a1.z, Twy = a1.w;
                if (pr.use3d) {
                    const float dL_dsx = dL_dG * (-G * pr.sx) + dL_dz * Twx;
                    const float dL_dsy = dL_dG * (-G * pr.sy) + dL_dz * Twy;
                    const float dpx = dL_dsx * pr.inv_pz, dpy = dL_dsy * pr.inv_pz;
                    const float dpz = -(dpx * pr.sx + dpy * pr.sy);
                    // dL/dk = l x dL/dp ; dL/dl = dL/dp x k
                    const float dkx = pr.ly * dpz - pr.lz * dpy, dky = pr.lz * dpx - pr.lx * dpz, dkz = pr.lx * dpy - pr.ly * dpx;
                    const float dlx = dpy * pr.kz - dpz * pr.ky, dly = dpz * pr.kx - dpx * pr.kz, dlz = dpx * pr.ky - dpy * pr.kx;
                    gT[0] = -dkx; gT[1] = -dky; gT[2] = -dkz;
                    gT[3] = -dlx; gT[4] = -dly; gT[5] = -dlz;
                    gT[6] = pxf * dkx + pyf * dlx + dL_dz * pr.sx;
                    gT[7] = pxf * dky + pyf * dly + dL_dz * pr.sy;
                    gT[8] = pxf * dkz + pyf * dlz + dL_dz;
                } else {
                    gxy0 = dL_dG * (-G * GSR_FILTER_INV_SQUARE * pr.dx);
                    gxy1 = dL_dG * (-G * GSR_FILTER_INV_SQUARE * pr.dy);
                    if (filter_depth_quirk) { gT[6] = pr.sx * dL_dz; gT[7] = pr.sy * dL_dz; }
                    gT[8] = dL_dz;
                }
            }

            // wave-level sums, stored by the lanes that end up holding them (row layout GSR_GR_*)
            {
                const float v16[16] = {gT[0], gT[1], gT[2], gT[3], gT[4], gT[5], gT[6], gT[7], gT[8],
                                       gn0, gn1, gn2, gopa, gc0, gc1, gc2};
                const float tot = wave_sum16_transposed(v16, lane);
                const float xy = wave_sum2(gxy0, gxy1);
                const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)row_of_lane, j);
                float* row = p.grad_rows + ((size_t)e * 4 + wave) * RB_ROW;
                const int vi = lane >> 2;
                if ((lane & 3) == 0) row[vi < 9 ? vi : vi + 2] = tot;      // skip the two xy columns
                if (lane == 16) row[GSR_GR_XY] = xy;
                if (lane == 48) row[GSR_GR_XY + 1] = xy;
                if (lane == 0) p.row_flags[(size_t)e * 4 + wave] = 1;
                if (FEAT16 > 0) {   // 16 channels per butterfly; lane 4*c ends up holding channel c of the group
                    float* frow = p.feat_rows + ((size_t)e * 4 + wave) * p.C;
#pragma unroll
                    for (int grp = 0; grp < FEAT16; ++grp) {
                        if (16 * grp < p.C) {
                            float f16[16];
#pragma unroll
                            for (int k = 0; k < 16; ++k) f16[k] = w_pair * dL_dpixf[16 * grp + k];
                            const float ft = wave_sum16_transposed(f16, lane);
                            const int ch = 16 * grp + vi;
                            if ((lane & 3) == 0 && ch < p.C) frow[ch] = ft;
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();   // all reads of this batch precede the next batch's LDS writes
        hi = lo;
    }
}

// Wide payload: add a Gaussian's flagged per-(instance, quad) feature sub-rows in fixed order.  One thread per
// (depth rank, 4-channel piece); writes dL_dcolors [N,C] by Gaussian id (zeros for Gaussians with no instance).
__global__ void __launch_bounds__(256) reduce_feat_rows_kernel(long long n_threads, int C4,
                                                               const uint32_t* __restrict__ order,
                                                               const uint32_t* __restrict__ offs,
                                                               const float4* __restrict__ rows,
                                                               const uint32_t* __restrict__ flags,
                                                               float4* __restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_threads) return;
    const int r = (int)(t / C4), q = (int)(t - (long long)r * C4);
    const uint32_t e0 = offs[r], e1 = offs[r + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (uint32_t e = e0; e < e1; ++e) {
        const uint32_t f = flags[e];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            if ((f >> (8 * sub)) & 0xFFu) {
                const float4 v = rows[((size_t)e * 4 + sub) * C4 + q];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
    }
    out[(size_t)order[r] * C4 + q] = acc;
}

int gsr_launch_reduce_feat_rows(int N, int C, const uint32_t* order, const uint32_t* offs, const float* feat_rows,
                                const uint32_t* row_flags, float* dL_dcolors, hipStream_t s) {
    if (N <= 0) return GSR_OK;
    GsrProfileScope prof(GSR_K_PREPROCESS_BWD, s);
    const int C4 = C / 4;
    const long long n_threads = (long long)N * C4;
    hipLaunchKernelGGL(reduce_feat_rows_kernel, dim3((unsigned)((n_threads + 255) / 256)), dim3(256), 0, s, n_threads, C4,
                       order, offs, reinterpret_cast<const float4*>(feat_rows), row_flags,
                       reinterpret_cast<float4*>(dL_dcolors));
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}

int gsr_launch_render_bwd(const GsrView& v, const uint32_t* ranges, const uint32_t* inst_row,
                          const float* stream, const uint8_t* touch, const float* final_T, const uint32_t* n_contrib,
                          const float* dL_dcolor, const float* dL_dallmap, float* grad_rows,
                          uint8_t* row_flags, const float* feat, const uint32_t* point_list, float* feat_rows,
                          hipStream_t s) {
    RenderBwdParams p;
    p.W = v.width; p.H = v.height; p.gx = (v.width + GSR_TILE - 1) / GSR_TILE;
    const int gy = (v.height + GSR_TILE - 1) / GSR_TILE;
    p.flags = v.flags;
    p.ranges = ranges; p.inst_row = inst_row; p.stream = reinterpret_cast<const float4*>(stream); p.touch = touch; p.bg = v.bg;
    p.final_T = final_T; p.n_contrib = n_contrib; p.dL_dcolor = dL_dcolor; p.dL_dallmap = dL_dallmap;
    p.grad_rows = grad_rows; p.row_flags = row_flags;
    p.feat = feat; p.point_list = point_list; p.feat_rows = feat_rows; p.C = v.channels;
    if (p.gx <= 0 || gy <= 0) return GSR_OK;
    GsrProfileScope prof(GSR_K_RENDER_BWD, s);
    const dim3 grid(p.gx, gy), block(RB_BLOCK);
    if (feat == nullptr) {
        hipLaunchKernelGGL(render_bwd_kernel<0>, grid, block, 0, s, p);
    } else {
        switch ((v.channels + 15) / 16) {
            case 1: hipLaunchKernelGGL(render_bwd_kernel<1>, grid, block, 0, s, p); break;
            case 2: hipLaunchKernelGGL(render_bwd_kernel<2>, grid, block, 0, s, p); break;
            case 3: hipLaunchKernelGGL(render_bwd_kernel<3>, grid, block, 0, s, p); break;
            case 4: hipLaunchKernelGGL(render_bwd_kernel<4>, grid, block, 0, s, p); break;
            default: gsr_set_error("wide payload supports at most 64 channels"); return GSR_E_UNSUPPORTED;
        }
    }
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}
