// K7 for wide per-pixel payloads (colors_precomp [N,C], C = 4..64; SURVEY 8(f) N4, BASELINE.json config 5): the backward of
// render_fwd's FEAT16 > 0 kernels.  Same replay as render_bwd.hip -- wave = 8x8 quad, DPP row = 4x4 pixel block walking its
// own list back to front, geometry gradients summed over the block with the transposed butterfly into dense rows -- but the
// two C-wide pieces run on the MATRIX pipe instead of the vector unit (round 2 ran them as C fmas + one 16-lane butterfly
// per 16 channels and iteration: 1.38 ms at C = 16, 4.2 ms at C = 64 against 0.65 ms for RGB; this: 0.99 / 2.3 ms):
//   q[pixel][entry]   = sum_ch f[entry][ch] * dL/dpixel[pixel][ch]          (the colour term of the suffix recursion)
//   frow[entry][ch]   = sum_{pixels of the block} w[pixel][entry] * dL/dpixel[pixel][ch]    (the feature gradient rows)
// Both are products with the per-pixel gradient, which is constant for the wave's whole life.  A block's pending entries are
// taken in WINDOWS of 16 (deepest first, the order the recursion needs); per window:
//   1. Q: v_mfma_f32_16x16x1_4b_f32 (4 blocks = the 4 DPP rows, K = 1) once per channel: A = feature ch of "my window entry"
//      (lane (b, t) owns entry t of block b's window and loads its feature row itself), B = dL/dpixel[ch] of "my pixel" (lane
//      (b, px) -- the natural lane = pixel layout), D[t][px] goes to a 4 KiB LDS tile from which iteration t reads its q;
//   2. the 16 iterations: the RGB kernel's body with q taken from the tile; every lane parks its blending weight w in the
//      word it has just read q from (tile [block][t][pixel]);
//   3. rows: once per pixel index p and 16-channel tile: A = w[t][p] (lane (b, t) reads its row of the weight tile), B =
//      dL/dpixel of pixel p with the CHANNELS on the lanes (a second, transposed register copy of the gradient made once per
//      wave through LDS), D[t][ch] leaves straight for the feature rows.
// Exact f32 (one fma per product, fixed order): deterministic, parity tests as for the scalar form
// (tests/test_gpu_wide_payload.py).  Lane maps of the instruction: scripts/microbench/mfma_layout.hip.
#include <stdlib.h>
#include "gsr_common.h"
#include "pair_eval.h"
#include "wave_reduce.h"
#include "render_bwd_shared.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NM>   // 16-channel tiles: C <= 16 NM
__device__ __forceinline__ void render_bwd_wide_tile(const RenderBwdParams& p) {
    __shared__ float4 s_rec_all[RB_WAVES][64 * 5];
    __shared__ int s_win_all[RB_WAVES][4][16];          // [block][t] -> staged entry of the batch, -1: none
    __shared__ uint32_t s_slot_all[RB_WAVES][4][16];    // [block][t] -> gradient row of (entry, block), 0xFFFFFFFF: none
    // [block][t][pixel of the block]: the colour term of q until iteration t has read it, that pixel's blending weight
    // afterwards (same lane, same word: one 4 KiB tile per wave instead of two keeps four workgroups on a CU)
    __shared__ float s_q_all[RB_WAVES][4][16][16];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    float4* s_rec = s_rec_all[wave];
    int (*s_win)[16] = s_win_all[wave];
    uint32_t (*s_slot)[16] = s_slot_all[wave];
    float (*s_q)[16][16] = s_q_all[wave];
    const int tile_lin = (int)(blockIdx.x & 7u) * p.per_xcd + (int)(blockIdx.x >> 3);   // XCD-aware tile order (render_bwd.hip)
    if (tile_lin >= p.n_tiles) return;
    const int tile_y = tile_lin / p.gx, tile_x = tile_lin - tile_y * p.gx;
    const int qx0 = tile_x * GSR_TILE + (wave & 1) * 8, qy0 = tile_y * GSR_TILE + (wave >> 1) * 8;
    const int grp = lane >> 4, l16 = lane & 15;   // DPP row = 4x4 pixel block
    const uint32_t below_mask = ((1u << (8 * wave + grp)) - 1u) & 0x0F0F0F0Fu;   // touch bits of the blocks before mine
    const int pxi = qx0 + (grp & 1) * 4 + (l16 & 3), pyi = qy0 + (grp >> 1) * 4 + (l16 >> 2);
    const bool inside = pxi < p.W && pyi < p.H;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const int pix_id = pyi * p.W + pxi;
    const int HW = p.W * p.H;

    const uint32_t tile = (uint32_t)(tile_y * p.gx + tile_x);
    const uint32_t r0 = p.ranges[2 * tile];
    const uint4 cov4 = *reinterpret_cast<const uint4*>(p.covered + 4 * tile);

    const int last_contributor = inside ? (int)p.n_contrib[pix_id] : 0;
    int max_contrib = last_contributor;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) max_contrib = max(max_contrib, __shfl_xor(max_contrib, d, 64));
    max_contrib = __builtin_amdgcn_readfirstlane(max_contrib);
    if (max_contrib == 0) return;

    const bool clamp_pass = (p.flags & GSR_FLAG_CLAMP_PASSTHROUGH) != 0;
    const bool filter_depth_quirk = (p.flags & GSR_FLAG_FILTER_DEPTH_GRAD) != 0;

    const float T_final = inside ? p.final_T[pix_id] : 0.f;
    const float final_D = inside ? p.final_T[pix_id + HW] : 0.f;
    const float final_D2 = inside ? p.final_T[pix_id + 2 * HW] : 0.f;
    const float final_A = 1.0f - T_final;
    const int median_contributor = inside ? (int)p.n_contrib[pix_id + HW] : 0;

    // (a pixel nothing was blended into takes no part, whatever gradient arrives for it: render_bwd.hip)
    const bool lit = inside && last_contributor > 0;
    // dL/dpixel of MY pixel, all channels (lane = pixel): the B operand of the Q products
    constexpr int NF = 16 * NM;
    float g[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) g[k] = (lit && k < p.C) ? p.dL_dcolor[pix_id + (size_t)k * HW] : 0.f;
    float dL_ddepth = 0.f, dL_daccum = 0.f, dL_dreg = 0.f, dL_dmedian = 0.f;
    float dL_dn0 = 0.f, dL_dn1 = 0.f, dL_dn2 = 0.f;
    if (lit) {
        dL_ddepth = p.dL_dallmap[pix_id + 0 * HW];
        dL_daccum = p.dL_dallmap[pix_id + 1 * HW];
        dL_dn0 = p.dL_dallmap[pix_id + 2 * HW];
        dL_dn1 = p.dL_dallmap[pix_id + 3 * HW];
        dL_dn2 = p.dL_dallmap[pix_id + 4 * HW];
        dL_dmedian = p.dL_dallmap[pix_id + 5 * HW];
        dL_dreg = p.dL_dallmap[pix_id + 6 * HW];
    }
    float bg_dot_dpixel = 0.f;
#pragma unroll
    for (int k = 0; k < NF; ++k)
        if (k < p.C) bg_dot_dpixel += p.bg[k] * g[k];

    // the same gradient with the CHANNELS on the lanes: gt[p_][m] = dL/dpixel[pixel p_ of MY block][channel 16 m + l16] --
    // the B operand of the row products.  Transposed once per wave through the (not yet used) q tile, 16 channels at a time.
    float gt[16][NM];
    {
        float* tmp = &s_q[0][0][0];      // 64 x 16 floats
#pragma unroll
        for (int m = 0; m < NM; ++m) {
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4)
                *reinterpret_cast<float4*>(tmp + lane * 16 + 4 * c4) =
                    make_float4(g[16 * m + 4 * c4], g[16 * m + 4 * c4 + 1], g[16 * m + 4 * c4 + 2], g[16 * m + 4 * c4 + 3]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int pp = 0; pp < 16; ++pp) gt[pp][m] = tmp[(16 * grp + pp) * 16 + l16];
            __builtin_amdgcn_wave_barrier();
        }
    }

    const bool quad_has_dist = __any(dL_dreg != 0.f), quad_has_median = __any(dL_dmedian != 0.f);   // wave-uniform
    const bool quad_has_surf = __any(dL_ddepth != 0.f || dL_daccum != 0.f || dL_dn0 != 0.f || dL_dn1 != 0.f || dL_dn2 != 0.f);

    float T = T_final;
    float last_alpha = 0.f, last_q = 0.f, acc_q = 0.f, last_dL_dT = 0.f;

    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 pf0, pf1, pf2, pf3, pf4;
    uint32_t pf_touch = 0, pf_slot = 0;
    int hi = max_contrib;
    uint32_t ids_cur, ids_nxt, rows_nxt;
    {
        const int lo = max(0, hi - 64), cnt = hi - lo;
        ids_cur = lane < cnt ? p.point_list[r0 + lo + lane] : 0u;
        pf_slot = lane < cnt ? p.slot_off[p.inst_row[r0 + lo + lane]] : 0u;
        pf_touch = lane < cnt ? rb_defined_touch(p.touch[(size_t)r0 + lo + lane], (uint32_t)(lo + lane), cov4) : 0u;
        const int lo2 = max(0, lo - 64), cnt2 = lo - lo2;
        ids_nxt = lane < cnt2 ? p.point_list[r0 + lo2 + lane] : 0u;
        rows_nxt = lane < cnt2 ? p.inst_row[r0 + lo2 + lane] : 0u;
    }

    while (hi > 0) {
        const int lo = max(0, hi - 64), nb = hi - lo;
        GSR_GATHER5(ids_cur, nb);
        const uint32_t touch_of_lane = pf_touch;
        const uint32_t slot_of_lane = pf_slot;
        s_rec[lane * 5] = pf0; s_rec[lane * 5 + 1] = pf1; s_rec[lane * 5 + 2] = pf2; s_rec[lane * 5 + 3] = pf3;
        s_rec[lane * 5 + 4] = make_float4(pf4.x, pf4.y, __uint_as_float(slot_of_lane), __uint_as_float(touch_of_lane));
        const uint32_t id_of_lane = ids_cur;          // Gaussian id of staged entry `lane`
        {   // prefetch the next (shallower) batch
            const int hi2 = lo, lo2 = max(0, hi2 - 64), cnt = hi2 - lo2;
            pf_slot = lane < cnt ? p.slot_off[rows_nxt] : 0u;
            pf_touch = lane < cnt ? rb_defined_touch(p.touch[(size_t)r0 + lo2 + lane], (uint32_t)(lo2 + lane), cov4) : 0u;
            ids_cur = ids_nxt;
            const int lo3 = max(0, lo2 - 64), cnt3 = lo2 - lo3;
            ids_nxt = lane < cnt3 ? p.point_list[r0 + lo3 + lane] : 0u;
            rows_nxt = lane < cnt3 ? p.inst_row[r0 + lo3 + lane] : 0u;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // per block: which staged entries did the forward blend into >= 1 of its pixels (plain numbering: bit j = entry j)
        const uint32_t nib = lane < nb ? (touch_of_lane >> (8 * wave)) & 0xFu : 0u;
        const unsigned long long mk0 = __ballot((nib & 1u) != 0), mk1 = __ballot((nib & 2u) != 0);
        const unsigned long long mk2 = __ballot((nib & 4u) != 0), mk3 = __ballot((nib & 8u) != 0);
        const int n_max = max(max(__popcll(mk0), __popcll(mk1)), max(__popcll(mk2), __popcll(mk3)));
        // rank of staged entry `lane` in block b's walk (deepest first) = how many entries of the mask lie above it
        const unsigned long long above = ((~0ull << lane) << 1);
        const int rk0 = __popcll(mk0 & above), rk1 = __popcll(mk1 & above), rk2 = __popcll(mk2 & above), rk3 = __popcll(mk3 & above);

        for (int w0 = 0; w0 < n_max; w0 += 16) {
            // ---- the window: entries of rank w0 .. w0 + 15 of every block
            s_win[grp][l16] = -1;
            s_slot[grp][l16] = 0xFFFFFFFFu;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if ((nib & 1u) && (unsigned)(rk0 - w0) < 16u) s_win[0][rk0 - w0] = lane;
            if ((nib & 2u) && (unsigned)(rk1 - w0) < 16u) s_win[1][rk1 - w0] = lane;
            if ((nib & 4u) && (unsigned)(rk2 - w0) < 16u) s_win[2][rk2 - w0] = lane;
            if ((nib & 8u) && (unsigned)(rk3 - w0) < 16u) s_win[3][rk3 - w0] = lane;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();

            // ---- 1. Q[t][pixel] of the window: lane (b, t) owns entry t of block b
            {
                const int my_j = s_win[grp][l16];
                const uint32_t my_gid = (uint32_t)__shfl((int)id_of_lane, my_j & 63, 64);
                const float4* fsrc = reinterpret_cast<const float4*>(p.feat + (size_t)my_gid * p.C);
                f32x16 qa;
#pragma unroll
                for (int v = 0; v < 16; ++v) qa[v] = 0.f;
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    float4 f4[4];
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        f4[c4] = zero4;
                        if (my_j >= 0 && 16 * m + 4 * c4 < p.C) f4[c4] = fsrc[4 * m + c4];
                    }
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
                        qa = __builtin_amdgcn_mfma_f32_16x16x1f32(f4[c4].x, g[16 * m + 4 * c4 + 0], qa, 0, 0, 0);
                        qa = __builtin_amdgcn_mfma_f32_16x16x1f32(f4[c4].y, g[16 * m + 4 * c4 + 1], qa, 0, 0, 0);
                        qa = __builtin_amdgcn_mfma_f32_16x16x1f32(f4[c4].z, g[16 * m + 4 * c4 + 2], qa, 0, 0, 0);
                        qa = __builtin_amdgcn_mfma_f32_16x16x1f32(f4[c4].w, g[16 * m + 4 * c4 + 3], qa, 0, 0, 0);
                    }
                }
                // register 4 b + r of lane (k', pixel) = Q of (block b, t = 4 k' + r, that pixel)
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s_q[b][4 * grp + r][l16] = qa[4 * b + r];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();

            // ---- 2. the iterations of the window
            for (int t = 0; t < 16; ++t) {
                const int jw = s_win[grp][t];
                const bool has = jw >= 0;
                if (!__any(has)) break;                     // (the blocks' entries fill their windows from t = 0)
                const int j = jw & 63;
                const int cidx = lo + j;
                const float4 a0 = s_rec[j * 5 + 0], a1 = s_rec[j * 5 + 1], a2 = s_rec[j * 5 + 2];
                const float4 a3 = s_rec[j * 5 + 3];
                GsrPair pr;
                const bool ok = gsr_pair_eval(pxf, pyf, a0, a1, a2, a3.z, pr);
                const bool active = has && cidx < last_contributor && ok;
                float gT[9];
                float gxy0, gxy1, gn0, gn1, gn2, gopa;
                uint32_t rec_slot, rec_touch;
                {
                    const float4 a4 = s_rec[j * 5 + 4];
                    rec_slot = __float_as_uint(a4.z); rec_touch = __float_as_uint(a4.w);
                    const float alpha = active ? pr.alpha : 0.f, G = active ? pr.G : 0.f, c_d = active ? pr.depth : 1.f;
                    const float sx = active ? pr.sx : 0.f, sy = active ? pr.sy : 0.f, inv_pz = active ? pr.inv_pz : 0.f;
                    const float one_m_alpha = 1.0f - alpha;
                    const float inv_oma = gsr_rcp(one_m_alpha);
                    T = T * inv_oma;
                    const float w = alpha * T;
                    const float n0 = a2.w, n1 = a3.x, n2 = a3.y;
                    float q = s_q[grp][t][l16];                 // colour term: sum_ch f[ch] dL/dpixel[ch]
                    s_q[grp][t][l16] = w;                       // ... replaced by the A operand of the row products
                    if (quad_has_surf) q += c_d * dL_ddepth + dL_daccum + n0 * dL_dn0 + n1 * dL_dn1 + n2 * dL_dn2;
                    acc_q = last_alpha * last_q + (1.f - last_alpha) * acc_q;
                    last_q = q;
                    float dL_dalpha = q - acc_q;
                    gn0 = 0.f; gn1 = 0.f; gn2 = 0.f;
                    if (quad_has_surf) { gn0 = w * dL_dn0; gn1 = w * dL_dn1; gn2 = w * dL_dn2; }

                    float dL_dz = w * dL_ddepth;
                    if (quad_has_median && active && cidx == median_contributor - 1) dL_dz += dL_dmedian;
                    if (quad_has_dist) {
                        float dmd_dd;
                        const float m_d = gsr_depth_map(c_d, dmd_dd);
                        const float dL_dweight = (final_D2 + m_d * m_d * final_A - 2.f * m_d * final_D) * dL_dreg;
                        dL_dalpha += dL_dweight - last_dL_dT;
                        last_dL_dT = dL_dweight * alpha + one_m_alpha * last_dL_dT;
                        dL_dz += 2.0f * w * (m_d * final_A - final_D) * dL_dreg * dmd_dd;
                    }

                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha -= T_final * inv_oma * bg_dot_dpixel;

                    const float dL_daraw = (clamp_pass || pr.araw <= GSR_ALPHA_MAX) ? dL_dalpha : 0.f;
                    const float dL_dG = a3.z * dL_daraw;
                    gopa = G * dL_daraw;

                    const float Twx = a1.z, Twy = a1.w;
                    if (pr.use3d) {
                        const float dL_dsx = dL_dG * (-G * sx) + dL_dz * Twx;
                        const float dL_dsy = dL_dG * (-G * sy) + dL_dz * Twy;
                        float dpx = dL_dsx * inv_pz, dpy = dL_dsy * inv_pz;
                        if (__builtin_expect(pr.tiny_any, 0)) {       // (pair_eval.h: a denormal p.z; the empty asm keeps this a BRANCH -- if-converted it cost
                            asm volatile("");                         //  three vector instructions on every pair, +3.7 % of K7's issue)
                            const float zs = pr.tiny ? GSR_TINY_PZ_SCALE : 1.f; dpx *= zs; dpy *= zs;
                        }
                        const float dpz = -(dpx * sx + dpy * sy);
                        const float ux = dpy * pr.lz - dpz * pr.ly, uy = dpz * pr.lx - dpx * pr.lz, uz = dpx * pr.ly - dpy * pr.lx;
                        const float vx = pr.ky * dpz - pr.kz * dpy, vy = pr.kz * dpx - pr.kx * dpz, vz = pr.kx * dpy - pr.ky * dpx;
                        gT[0] = ux; gT[1] = uy; gT[2] = uz;
                        gT[3] = vx; gT[4] = vy; gT[5] = vz;
                        gT[6] = dL_dz * sx - pxf * ux - pyf * vx;
                        gT[7] = dL_dz * sy - pxf * uy - pyf * vy;
                        gT[8] = dL_dz - pxf * uz - pyf * vz;
                        gxy0 = 0.f; gxy1 = 0.f;
                    } else {
                        gxy0 = dL_dG * (-G * GSR_FILTER_INV_SQUARE * pr.dx);
                        gxy1 = dL_dG * (-G * GSR_FILTER_INV_SQUARE * pr.dy);
                        gT[0] = 0.f; gT[1] = 0.f; gT[2] = 0.f; gT[3] = 0.f; gT[4] = 0.f; gT[5] = 0.f;
                        gT[6] = filter_depth_quirk ? sx * dL_dz : 0.f;
                        gT[7] = filter_depth_quirk ? sy * dL_dz : 0.f;
                        gT[8] = dL_dz;
                    }
                }
                // block-level sums of the geometry partials (16 lanes), row layout GSR_GR_*.  A wide payload's colour
                // gradient lives in the feature rows, so the three colour columns of the row are free: two of them carry
                // the centre gradient dxy of the low-pass branch (reduce_rows<true> moves them to their place), and this
                // kernel writes no second array
                {
                    const float v16[16] = {gT[0], gT[1], gT[2], gT[3], gT[4], gT[5], gT[6], gT[7], gT[8],
                                           gn0, gn1, gn2, gopa, gxy0, gxy1, 0.f};
                    const float tot = row_sum16_transposed(v16, l16);
                    const uint32_t slot = rec_slot + (uint32_t)__popc(rec_touch & below_mask);
                    if (has) {
                        p.grad_rows[(size_t)slot * RB_ROW + l16] = tot;
                        if (l16 == 0) s_slot[grp][t] = slot;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();

            // ---- 3. feature rows of the window: frow[t][ch] = sum over the block's 16 pixels of w[t][pixel] dL/dpixel[pixel][ch]
            {
                float wa[16];      // my entry's weights at the 16 pixels of my block (0 for iterations that did not run)
                const bool ran = s_slot[grp][l16] != 0xFFFFFFFFu;
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const float4 v = *reinterpret_cast<const float4*>(&s_q[grp][l16][4 * c4]);
                    wa[4 * c4] = ran ? v.x : 0.f; wa[4 * c4 + 1] = ran ? v.y : 0.f; wa[4 * c4 + 2] = ran ? v.z : 0.f; wa[4 * c4 + 3] = ran ? v.w : 0.f;
                }
                uint32_t sl[4][4];  // rows of (block b, t = 4 grp + r)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint4 v = *reinterpret_cast<const uint4*>(&s_slot[b][4 * grp]);
                    sl[b][0] = v.x; sl[b][1] = v.y; sl[b][2] = v.z; sl[b][3] = v.w;
                }
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    if (16 * m < p.C) {
                        f32x16 fa;
#pragma unroll
                        for (int v = 0; v < 16; ++v) fa[v] = 0.f;
#pragma unroll
                        for (int pp = 0; pp < 16; ++pp) fa = __builtin_amdgcn_mfma_f32_16x16x1f32(wa[pp], gt[pp][m], fa, 0, 0, 0);
                        const int ch = 16 * m + l16;
#pragma unroll
                        for (int b = 0; b < 4; ++b)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (sl[b][r] != 0xFFFFFFFFu && ch < p.C) p.feat_rows[(size_t)sl[b][r] * p.C + ch] = fa[4 * b + r];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();   // the tables are rewritten by the next window
        }
        __builtin_amdgcn_wave_barrier();       // all reads of this batch precede the next batch's LDS writes
        hi = lo;
    }
}

template <int NM>
__global__ void __launch_bounds__(RB_BLOCK, NM == 1 ? 4 : (NM == 2 ? 3 : 2)) render_bwd_wide_kernel(RenderBwdParams p) {
    render_bwd_wide_tile<NM>(p);
    rb_row_begin_job(p);   // behind the tile's work, where none of its registers is live (render_bwd.hip)
}

int gsr_launch_render_bwd_wide(const RenderBwdParams& p, int channels, dim3 grid, hipStream_t s) {
    switch ((channels + 15) / 16) {
        case 1: hipLaunchKernelGGL(render_bwd_wide_kernel<1>, grid, dim3(RB_BLOCK), 0, s, p); break;
        case 2: hipLaunchKernelGGL(render_bwd_wide_kernel<2>, grid, dim3(RB_BLOCK), 0, s, p); break;
        case 3: hipLaunchKernelGGL(render_bwd_wide_kernel<3>, grid, dim3(RB_BLOCK), 0, s, p); break;
        case 4: hipLaunchKernelGGL(render_bwd_wide_kernel<4>, grid, dim3(RB_BLOCK), 0, s, p); break;
        default: gsr_set_error("wide payload supports at most 64 channels"); return GSR_E_UNSUPPORTED;
    }
    return GSR_OK;
}
