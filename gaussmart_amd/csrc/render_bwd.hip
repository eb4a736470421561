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
};

// ---- wave64 reductions -----------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float v) {
    // lanes whose source is disabled/out of range receive 0 (bound_ctrl)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
typedef unsigned int gsr_u2 __attribute__((ext_vector_type(2)));

// "Transposed butterfly": sums 16 per-lane values over the 64 lanes in 6 stages while HALVING the
// number of live registers at each of the first four (gfx950 v_permlane32_swap / v_permlane16_swap,
// then DPP row_ror:8 and row_half_mirror with a select).  ~50 VALU instead of 16 x 6 DPP steps.
// On return every lane l holds the wave total of v[l >> 2].
__device__ __forceinline__ float wave_sum16_transposed(const float (&v)[16], int lane) {
    float r[8], q[4], p[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {      // bit 5: lanes < 32 keep v[i], lanes >= 32 keep v[i+8]
        const gsr_u2 t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 8]), false, false);
        r[i] = __uint_as_float(t.x) + __uint_as_float(t.y);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {      // bit 4: even rows keep r[i], odd rows keep r[i+4]
        const gsr_u2 t = __builtin_amdgcn_permlane16_swap(__float_as_uint(r[i]), __float_as_uint(r[i + 4]), false, false);
        q[i] = __uint_as_float(t.x) + __uint_as_float(t.y);
    }
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {      // bit 3: partner is lane ^ 8 (row_ror:8)
        const float keep = b3 ? q[i + 2] : q[i], send = b3 ? q[i] : q[i + 2];
        p[i] = keep + dpp_move<0x128, 0xf>(send);
    }
    {                                   // bit 2: partner is lane ^ 7 inside each group of 8 (row_half_mirror)
        const float keep = b2 ? p[1] : p[0], send = b2 ? p[0] : p[1];
        p[0] = keep + dpp_move<0x141, 0xf>(send);
    }
    p[0] += dpp_move<0xB1, 0xf>(p[0]);  // quad_perm [1,0,3,2]
    p[0] += dpp_move<0x4E, 0xf>(p[0]);  // quad_perm [2,3,0,1]
    return p[0];
}
// Two values: on return lanes 16..31 hold the wave total of a, lanes 48..63 that of b.
__device__ __forceinline__ float wave_sum2(float a, float b) {
    const gsr_u2 t = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    float v = __uint_as_float(t.x) + __uint_as_float(t.y);
    v += dpp_move<0xB1, 0xf>(v);
    v += dpp_move<0x4E, 0xf>(v);
    v += dpp_move<0x141, 0xf>(v);   // row_half_mirror
    v += dpp_move<0x140, 0xf>(v);   // row_mirror: every lane holds its row's sum
    v += dpp_move<0x142, 0xa>(v);   // row_bcast:15: rows 1 and 3 add the previous row
    return v;
}

#ifndef RB_MIN_WAVES
#define RB_MIN_WAVES 5   // <= 96 VGPRs: measured 1.45 -> 1.35 ms at 1M/1080p
#endif
__global__ void __launch_bounds__(RB_BLOCK, RB_MIN_WAVES) render_bwd_kernel(RenderBwdParams p) {
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
    float dL_ddepth = 0.f, dL_daccum = 0.f, dL_dreg = 0.f, dL_dmedian = 0.f;
    float dL_dn0 = 0.f, dL_dn1 = 0.f, dL_dn2 = 0.f;
    if (inside) {
        dL_dpix0 = p.dL_dcolor[pix_id]; dL_dpix1 = p.dL_dcolor[pix_id + HW]; dL_dpix2 = p.dL_dcolor[pix_id + 2 * HW];
        dL_ddepth = p.dL_dallmap[pix_id + 0 * HW];
        dL_daccum = p.dL_dallmap[pix_id + 1 * HW];
        dL_dn0 = p.dL_dallmap[pix_id + 2 * HW];
        dL_dn1 = p.dL_dallmap[pix_id + 3 * HW];
        dL_dn2 = p.dL_dallmap[pix_id + 4 * HW];
        dL_dmedian = p.dL_dallmap[pix_id + 5 * HW];
        dL_dreg = p.dL_dallmap[pix_id + 6 * HW];
    }
    const float bg_dot_dpixel = p.bg[0] * dL_dpix0 + p.bg[1] * dL_dpix1 + p.bg[2] * dL_dpix2;

    // running state of the back-to-front recursion
    float T = T_final;
    float last_alpha = 0.f, last_q = 0.f, acc_q = 0.f, last_dL_dT = 0.f;

    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 pf0, pf1, pf2, pf3, pf4;
    uint32_t pf_touch = 0;   // named (not an array): keeps the prefetch in VGPRs, not scratch
    uint32_t pf_row = 0;
    int hi = max_contrib;
    {
        const int lo = max(0, hi - 64), cnt = hi - lo;
        const float4* src = p.stream + (size_t)(r0 + lo) * 5;
GSR_LOAD5(src, cnt * 5);
        pf_row = lane < cnt ? p.inst_row[r0 + lo + lane] : 0u;
        pf_touch = lane < cnt ? p.touch[((size_t)r0 + lo + lane) * 4 + wave] : 0u;
    }

    while (hi > 0) {
        const int lo = max(0, hi - 64), nb = hi - lo;
s_rec[lane] = pf0; s_rec[64 + lane] = pf1; s_rec[128 + lane] = pf2; s_rec[192 + lane] = pf3; s_rec[256 + lane] = pf4;
        const uint32_t row_of_lane = pf_row;          // emission index of staged entry `lane`
        const uint32_t touch_of_lane = pf_touch;      // did the forward blend staged entry `lane` in this quad?
        {   // prefetch the next (shallower) batch
            const int hi2 = lo, lo2 = max(0, hi2 - 64), cnt = hi2 - lo2;
            const float4* src = p.stream + (size_t)(r0 + lo2) * 5;
GSR_LOAD5(src, cnt * 5);
            pf_row = lane < cnt ? p.inst_row[r0 + lo2 + lane] : 0u;
            pf_touch = lane < cnt ? p.touch[((size_t)r0 + lo2 + lane) * 4 + wave] : 0u;
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
                const float q = c0 * dL_dpix0 + c1 * dL_dpix1 + c2 * dL_dpix2 + c_d * dL_ddepth + dL_daccum
                              + n0 * dL_dn0 + n1 * dL_dn1 + n2 * dL_dn2;
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
            }
        }
        __builtin_amdgcn_wave_barrier();   // all reads of this batch precede the next batch's LDS writes
        hi = lo;
    }
}

int gsr_launch_render_bwd(const GsrView& v, const uint32_t* ranges, const uint32_t* inst_row,
                          const float* stream, const uint8_t* touch, const float* final_T, const uint32_t* n_contrib,
                          const float* dL_dcolor, const float* dL_dallmap, float* grad_rows,
                          uint8_t* row_flags, hipStream_t s) {
    RenderBwdParams p;
    p.W = v.width; p.H = v.height; p.gx = (v.width + GSR_TILE - 1) / GSR_TILE;
    const int gy = (v.height + GSR_TILE - 1) / GSR_TILE;
    p.flags = v.flags;
    p.ranges = ranges; p.inst_row = inst_row; p.stream = reinterpret_cast<const float4*>(stream); p.touch = touch; p.bg = v.bg;
    p.final_T = final_T; p.n_contrib = n_contrib; p.dL_dcolor = dL_dcolor; p.dL_dallmap = dL_dallmap;
    p.grad_rows = grad_rows; p.row_flags = row_flags;
    if (p.gx <= 0 || gy <= 0) return GSR_OK;
    GsrProfileScope prof(GSR_K_RENDER_BWD, s);
    hipLaunchKernelGGL(render_bwd_kernel, dim3(p.gx, gy), dim3(RB_BLOCK), 0, s, p);
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}
