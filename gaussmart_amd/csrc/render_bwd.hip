// K7 render_bwd: back-to-front replay of every tile, producing ONE gradient row per
// (Gaussian, tile) instance -- no floating-point atomics anywhere.
//
// Restates the [U]/[P] backward of the surfel compositing (colour, expected depth, alpha, normal,
// median depth, distortion), see DESIGN.md section "render_bwd" for the recursion; it is the exact
// derivative of render_fwd except for the two flagged quirks (GSR_FLAG_*).
//
// MI355X mapping
//   * same pixel mapping as the forward: 4 wave64s x (8x8 pixel quad);
//   * per (wave, splat) the 18 partial derivatives are summed over the wave's 64 pixels with DPP
//     row operations (4 in-row butterflies + row_bcast15 + row_bcast31: 6 VALU per value) and only
//     if some lane of the wave actually touched the splat;
//   * each wave deposits its sums in its OWN LDS slot; after a batch the workgroup adds the (at
//     most 4) slots in fixed wave order and stores the 80-byte row of that instance with plain
//     16-byte stores.  Summation order is fixed => bitwise reproducible gradients, and the HBM
//     side sees streaming stores instead of ~18 atomics per pixel-splat pair.
//   * rows are indexed by emission order (inst_row), so preprocess_bwd reads each Gaussian's rows
//     as one contiguous segment.
#include "gsr_common.h"
#include "pair_eval.h"

#define RB_BLOCK 256
#define RB_WAVES 4
#define RB_BATCH 64
#define RB_ROW GSR_GROW_FLOATS   // 20 floats

struct RenderBwdParams {
    int W, H, gx;
    uint32_t flags;
    const uint32_t* ranges; const uint32_t* point_list; const uint32_t* inst_row;
    const float* splat; const float* bg;
    const float* final_T; const uint32_t* n_contrib;
    const float* dL_dcolor; const float* dL_dallmap;
    float* grad_rows;
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

__global__ void __launch_bounds__(RB_BLOCK) render_bwd_kernel(RenderBwdParams p) {
    __shared__ float4 s_rec[RB_BATCH * 5];
    __shared__ __attribute__((aligned(16))) float s_acc[RB_WAVES][RB_BATCH][RB_ROW];
    __shared__ unsigned long long s_touched[RB_WAVES];
    __shared__ uint32_t s_row[RB_BATCH];
    __shared__ uint32_t s_max_contrib;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int tile_x = blockIdx.x, tile_y = blockIdx.y;
    const int pxi = tile_x * GSR_TILE + (wave & 1) * 8 + (lane & 7);
    const int pyi = tile_y * GSR_TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = pxi < p.W && pyi < p.H;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const int pix_id = pyi * p.W + pxi;
    const int HW = p.W * p.H;

    const uint32_t tile = (uint32_t)(tile_y * p.gx + tile_x);
    const uint32_t r0 = p.ranges[2 * tile], r1 = p.ranges[2 * tile + 1];
    const int n_list = (int)(r1 - r0);
    if (n_list == 0) return;

    const bool clamp_pass = (p.flags & GSR_FLAG_CLAMP_PASSTHROUGH) != 0;
    const bool filter_depth_quirk = (p.flags & GSR_FLAG_FILTER_DEPTH_GRAD) != 0;
    const bool no_cull = (p.flags & (uint32_t)GSR_FLAG_DEBUG_NO_CULL) != 0;
    const int qx0 = tile_x * GSR_TILE + (wave & 1) * 8, qy0 = tile_y * GSR_TILE + (wave >> 1) * 8;

    // per-pixel state saved by the forward
    const float T_final = inside ? p.final_T[pix_id] : 0.f;
    const float final_D = inside ? p.final_T[pix_id + HW] : 0.f;       // sum m w
    const float final_D2 = inside ? p.final_T[pix_id + 2 * HW] : 0.f;  // sum m^2 w
    const float final_A = 1.0f - T_final;
    const int last_contributor = inside ? (int)p.n_contrib[pix_id] : 0;
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

    // the deepest list entry any pixel of the tile reached
    if (tid == 0) s_max_contrib = 0;
    __syncthreads();
    {
        int m = last_contributor;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
        if (lane == 0) atomicMax(&s_max_contrib, (uint32_t)m);
    }
    __syncthreads();
    const int max_contrib = (int)s_max_contrib;

    // rows of instances nobody reached are zero
    for (int i = max_contrib + tid; i < n_list; i += RB_BLOCK) {
        float4* row = reinterpret_cast<float4*>(p.grad_rows + (size_t)p.inst_row[r0 + i] * RB_ROW);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < RB_ROW / 4; ++q) row[q] = z;
    }

    // running state of the back-to-front recursion
    float T = T_final;
    float last_alpha = 0.f;
    float last_c0 = 0.f, last_c1 = 0.f, last_c2 = 0.f, acc_c0 = 0.f, acc_c1 = 0.f, acc_c2 = 0.f;
    float last_depth = 0.f, acc_depth = 0.f, acc_alpha = 0.f;
    float last_n0 = 0.f, last_n1 = 0.f, last_n2 = 0.f, acc_n0 = 0.f, acc_n1 = 0.f, acc_n2 = 0.f;
    float last_dL_dT = 0.f;

    for (int hi = max_contrib; hi > 0; hi -= RB_BATCH) {
        const int nb = min(RB_BATCH, hi);
        __syncthreads();   // previous batch fully flushed before LDS is reused
        if (tid < nb) {
            const int li = hi - 1 - tid;            // list index of staged entry `tid`
            const uint32_t gid = p.point_list[r0 + li];
            s_row[tid] = p.inst_row[r0 + li];
            const float4* src = reinterpret_cast<const float4*>(p.splat + (size_t)gid * GSR_SPLAT_FLOATS);
#pragma unroll
            for (int q = 0; q < 5; ++q) s_rec[tid * 5 + q] = src[q];
        }
        __syncthreads();

        unsigned long long touched = 0ull;
        // which staged splats can reach alpha >= 1/255 inside this wave's 8x8 quad at all?
        bool ov = false;
        if (lane < nb) {
            const float4 r4 = s_rec[lane * 5 + 4];
            ov = no_cull || gsr_rect_overlaps_quad(__float_as_uint(r4.z), __float_as_uint(r4.w), qx0, qy0);
        }
        unsigned long long todo_mask = __ballot(ov);
        while (todo_mask) {
            const int j = __builtin_ctzll(todo_mask);
            todo_mask &= todo_mask - 1;
            const int cidx = hi - 1 - j;            // 0-based position in the tile list
            const float4 a0 = s_rec[j * 5 + 0], a1 = s_rec[j * 5 + 1], a2 = s_rec[j * 5 + 2];
            const float4 a3 = s_rec[j * 5 + 3];
            GsrPair pr;
            bool active = cidx < last_contributor;
            if (active) active = gsr_pair_eval(pxf, pyf, a0, a1, a2, a3.z, pr);
            if (!__any(active)) continue;           // whole wave untouched by this splat
            touched |= 1ull << j;

            float gT[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            float gxy0 = 0.f, gxy1 = 0.f, gn0 = 0.f, gn1 = 0.f, gn2 = 0.f, gopa = 0.f;
            float gc0 = 0.f, gc1 = 0.f, gc2 = 0.f;
            if (active) {
                const float4 a4 = s_rec[j * 5 + 4];
                const float alpha = pr.alpha, G = pr.G, c_d = pr.depth;
                const float one_m_alpha = 1.0f - alpha;
                T = T * gsr_rcp(one_m_alpha);
                const float w = alpha * T;

                float dL_dalpha = 0.f;
                // colour
                const float c0 = a3.w, c1 = a4.x, c2 = a4.y;
                acc_c0 = last_alpha * last_c0 + (1.f - last_alpha) * acc_c0; last_c0 = c0;
                acc_c1 = last_alpha * last_c1 + (1.f - last_alpha) * acc_c1; last_c1 = c1;
                acc_c2 = last_alpha * last_c2 + (1.f - last_alpha) * acc_c2; last_c2 = c2;
                dL_dalpha += (c0 - acc_c0) * dL_dpix0 + (c1 - acc_c1) * dL_dpix1 + (c2 - acc_c2) * dL_dpix2;
                gc0 = w * dL_dpix0; gc1 = w * dL_dpix1; gc2 = w * dL_dpix2;

                // distortion, median depth
                float dmd_dd;
                const float m_d = gsr_depth_map(c_d, dmd_dd);
                float dL_dz = 0.f;
                if (cidx == median_contributor - 1) dL_dz += dL_dmedian;
                const float dL_dweight = (final_D2 + m_d * m_d * final_A - 2.f * m_d * final_D) * dL_dreg;
                dL_dalpha += dL_dweight - last_dL_dT;
                last_dL_dT = dL_dweight * alpha + one_m_alpha * last_dL_dT;
                const float dL_dmd = 2.0f * w * (m_d * final_A - final_D) * dL_dreg;
                dL_dz += dL_dmd * dmd_dd;

                // expected depth, alpha
                acc_depth = last_alpha * last_depth + (1.f - last_alpha) * acc_depth; last_depth = c_d;
                dL_dalpha += (c_d - acc_depth) * dL_ddepth;
                acc_alpha = last_alpha + (1.f - last_alpha) * acc_alpha;
                dL_dalpha += (1.f - acc_alpha) * dL_daccum;

                // normal
                const float n0 = a2.w, n1 = a3.x, n2 = a3.y;
                acc_n0 = last_alpha * last_n0 + (1.f - last_alpha) * acc_n0; last_n0 = n0;
                acc_n1 = last_alpha * last_n1 + (1.f - last_alpha) * acc_n1; last_n1 = n1;
                acc_n2 = last_alpha * last_n2 + (1.f - last_alpha) * acc_n2; last_n2 = n2;
                dL_dalpha += (n0 - acc_n0) * dL_dn0 + (n1 - acc_n1) * dL_dn1 + (n2 - acc_n2) * dL_dn2;
                gn0 = w * dL_dn0; gn1 = w * dL_dn1; gn2 = w * dL_dn2;

                dL_dalpha *= T;
                last_alpha = alpha;
                // alpha also scales how much background shows through
                dL_dalpha += (-T_final * gsr_rcp(one_m_alpha)) * bg_dot_dpixel;

                // alpha = min(0.99, opa * G)
                const float dL_daraw = (clamp_pass || pr.araw <= GSR_ALPHA_MAX) ? dL_dalpha : 0.f;
                const float dL_dG = a3.z * dL_daraw;
                gopa = G * dL_daraw;
                dL_dz += w * dL_ddepth;

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

            // wave-level sums -> this wave's LDS slot (row layout GSR_GR_*)
            {
                const float v16[16] = {gT[0], gT[1], gT[2], gT[3], gT[4], gT[5], gT[6], gT[7], gT[8],
                                       gn0, gn1, gn2, gopa, gc0, gc1, gc2};
                const float tot = wave_sum16_transposed(v16, lane);
                const float xy = wave_sum2(gxy0, gxy1);
                float* slot = &s_acc[wave][j][0];
                const int vi = lane >> 2;
                if ((lane & 3) == 0) slot[vi < 9 ? vi : vi + 2] = tot;      // skip the two xy columns
                if (lane == 16) slot[GSR_GR_XY] = xy;
                if (lane == 48) slot[GSR_GR_XY + 1] = xy;
            }
        }
        if (lane == 0) s_touched[wave] = touched;
        __syncthreads();

        // flush: 4 threads per instance, 5 floats each... one float4-sized piece (+1) per thread
        // layout: thread t -> instance j = t >> 2, piece q = t & 3 handles float4 q, and q==0 also float4 4
        {
            const int j = tid >> 2, q = tid & 3;
            if (j < nb) {
                float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
#pragma unroll
                for (int w = 0; w < RB_WAVES; ++w) {
                    if ((s_touched[w] >> j) & 1ull) {
                        const float4 v = *reinterpret_cast<const float4*>(&s_acc[w][j][4 * q]);
                        s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
                        if (q == 0) {
                            const float4 u = *reinterpret_cast<const float4*>(&s_acc[w][j][16]);
                            s1.x += u.x; s1.y += u.y; s1.z += u.z; s1.w += u.w;
                        }
                    }
                }
                float4* row = reinterpret_cast<float4*>(p.grad_rows + (size_t)s_row[j] * RB_ROW);
                row[q] = s0;
                if (q == 0) row[4] = s1;
            }
        }
    }
}

int gsr_launch_render_bwd(const GsrView& v, const uint32_t* ranges, const uint32_t* point_list,
                          const uint32_t* inst_row, const float* splat, const float* final_T,
                          const uint32_t* n_contrib, const float* dL_dcolor,
                          const float* dL_dallmap, float* grad_rows, hipStream_t s) {
    RenderBwdParams p;
    p.W = v.width; p.H = v.height; p.gx = (v.width + GSR_TILE - 1) / GSR_TILE;
    const int gy = (v.height + GSR_TILE - 1) / GSR_TILE;
    p.flags = v.flags;
    p.ranges = ranges; p.point_list = point_list; p.inst_row = inst_row; p.splat = splat; p.bg = v.bg;
    p.final_T = final_T; p.n_contrib = n_contrib; p.dL_dcolor = dL_dcolor; p.dL_dallmap = dL_dallmap;
    p.grad_rows = grad_rows;
    if (p.gx <= 0 || gy <= 0) return GSR_OK;
    GsrProfileScope prof(GSR_K_RENDER_BWD, s);
    hipLaunchKernelGGL(render_bwd_kernel, dim3(p.gx, gy), dim3(RB_BLOCK), 0, s, p);
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}
