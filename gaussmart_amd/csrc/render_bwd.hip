// K7 render_bwd: back-to-front replay of every tile -- no floating-point atomics anywhere.
//
// Restates the [U]/[P] backward of the surfel compositing (colour, expected depth, alpha, normal,
// median depth, distortion), see DESIGN.md section "render_bwd" for the recursion; it is the exact
// derivative of render_fwd except for the two flagged quirks (GSR_FLAG_*).
//
// MI355X mapping: WAVE-INDEPENDENT, like the forward, and BLOCK-INDEPENDENT inside the wave.  Each wave64 of
// a tile's workgroup owns an 8x8 pixel quad and never synchronises with the other three; each DPP row of
// 16 lanes owns one 4x4 pixel block of that quad and walks ITS OWN list:
//   * the wave starts at the deepest list entry ANY OF ITS 64 PIXELS reached (quad-level, not tile-level)
//     and walks the (tile, depth)-ordered list backwards, 64 entries per batch: ids two batches ahead, touch
//     words and row slots one batch ahead, the 80-byte records gathered by id at the start of their batch (lane l
//     fetches entry l) and staged in the wave's private LDS slice;
//   * the forward left 4 bits per (instance, quad): "blended into >= 1 pixel of block g".  Four ballots
//     turn them into one 64-bit to-do mask PER BLOCK; every iteration each row takes the deepest entry
//     of its own mask, so the wave needs max_g |list_g| iterations instead of |union of the lists|
//     (measured at 1M/1080p: 0.67x the iterations; lane utilisation 30 % -> 45 %);
//   * the suffix recursions of colour, depth, alpha and normal are collapsed into ONE scalar
//     recursion (they are linear: q_i = c_i.dL/dC + z_i dL/dD + dL/dA + n_i.dL/dN);
//   * the 18 partial derivatives are summed over the 16 pixels of the block with the transposed butterfly
//     of wave_reduce.h (DPP only, ~50 VALU) and land one per lane, which store them straight into the
//     block's OWN 80-byte sub-row of that instance (16 sub-rows per instance) plus a 1-byte flag;
//   * reduce_rows adds the flagged sub-rows in fixed order => bitwise reproducible gradients, and
//     HBM sees plain streaming stores instead of ~18 atomics per pixel-splat pair.
#include <stdlib.h>
#include "gsr_common.h"
#include "pair_eval.h"
#include "wave_reduce.h"

#include "render_bwd_shared.h"

#ifndef RB_MIN_WAVES
#define RB_MIN_WAVES 6   // <= 80 VGPRs.  Round 2's kernel needed 72 (seven waves per SIMD); with the two-array rows and the
                         // row_begin side job it takes the 80 (six waves).  A 72-register variant -- xy address derived from
                         // the row address, side job behind the tile's work -- runs seven waves again and is 1 % SLOWER
                         // (K7 0.669 vs 0.662 ms, profiles/r03_notes/ab_k7_seven_waves_again.log): behind the tile the job's
                         // two memory trips are a tail nothing overlaps, in front they hide behind the tile's own first loads
#endif
// (Wide per-pixel payloads have their own kernel: render_bwd_wide.hip.)
// PROBE (developer builds only: make PROBES=1, scripts/dev_probe.py): 1 = eight more dependent VALU per iteration,
// 4 = eight more dependent SALU, 3 = no row store (WRONG gradients: it exists to time the loop), 6 = 20 KB more LDS per
// workgroup (fewer waves per SIMD), 8 = four LDS reads of the record instead of five (WRONG gradients).
#ifdef GSR_DEV_PROBES
// PROBE 7 (round 4): every wave leaves its lifetime on the shader clock (s_memtime) and on the 100 MHz s_memrealtime
// clock: their ratio is the clock the chip holds under this kernel (scripts/dev_clock_probe.py).
#define RB_STAMP_WAVES 65536
__device__ unsigned long long g_rb_stamps[2 * RB_STAMP_WAVES];
extern "C" int gsr_probe_read_stamps_bwd(void* dst, size_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_rb_stamps), bytes < sizeof(g_rb_stamps) ? bytes : sizeof(g_rb_stamps)) == hipSuccess ? 0 : -1;
}
#endif
// NOSURF (round 4; GSR_FLAG_NO_SURFACE_GRAD): the caller promises that dL/dallmap is identically zero -- the reference's own
// evaluation flags (scripts/dtu_eval.py:45: --lambda_normal 0 --lambda_dist 0) and the first 7,000 iterations of EVERY run
// (train.py:132-133: lambda_normal from 7,000, lambda_dist from 3,000 and 0 by default).  Then no gradient reaches depth,
// alpha, normal, median depth or distortion: dL/dz = 0 and dL/dn = 0 identically, the per-pixel state loses five loads and
// ten registers, and the staged record needs neither the normal nor the cull rect: 16 floats instead of 20, FOUR per-lane
// ds_read_b128 per iteration instead of five (the fifth was worth 4 % of the kernel: DESIGN.md section 4, probe 8).  The
// entry's first gradient row and its touch nibble share the sixteenth word: row + popcount of the touch bits of the quads
// before this one (28 bits; the launcher falls back to the general kernel beyond 2^28 rows) | this quad's nibble << 28.
// Same arithmetic in the same order for everything that is not multiplied by one of those zeros, so the gradients equal the
// general kernel's fed with a zero dL/dallmap bit for bit (tests/test_gpu_rasterizer.py).
template <int PROBE = 0, bool NOSURF = false>
__global__ void __launch_bounds__(RB_BLOCK, NOSURF ? 8 : RB_MIN_WAVES) render_bwd_kernel(RenderBwdParams p) {
#ifdef GSR_DEV_PROBES
    unsigned long long stamp_t0 = 0, stamp_c0 = 0;
    if (PROBE == 7) { stamp_t0 = __builtin_amdgcn_s_memrealtime(); stamp_c0 = __builtin_amdgcn_s_memtime(); }
#endif
    __shared__ float s_probe_pad[PROBE == 6 ? 5120 : 1];
    if (PROBE == 6 && p.W < 0) s_probe_pad[threadIdx.x] = 1.f;
    constexpr int RS = NOSURF ? 4 : 5;          // float4 parts of a staged record
    __shared__ float4 s_rec_all[RB_WAVES][64 * RS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    float4* s_rec = s_rec_all[wave];
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so workgroup b
    // takes tile (b % 8) * per_xcd + b / 8 -- every XCD owns one contiguous band of tiles, and the records shared by
    // neighbouring tiles are fetched into ONE L2 instead of several
    const int tile_lin = (int)(blockIdx.x & 7u) * p.per_xcd + (int)(blockIdx.x >> 3);
    rb_row_begin_job(p);
    if (tile_lin >= p.n_tiles) return;
    const int tile_y = tile_lin / p.gx, tile_x = tile_lin - tile_y * p.gx;
    const int qx0 = tile_x * GSR_TILE + (wave & 1) * 8, qy0 = tile_y * GSR_TILE + (wave >> 1) * 8;
    const int grp = lane >> 4, l16 = lane & 15;   // DPP row = 4x4 pixel block, same mapping as render_fwd
    const uint32_t below_mask = ((1u << (8 * wave + grp)) - 1u) & 0x0F0F0F0Fu;   // touch bits of the blocks before mine
    const uint32_t quads_below_mask = ((1u << (8 * wave)) - 1u) & 0x0F0F0F0Fu;   // ... of the quads before mine (NOSURF)
    const uint32_t pick_shift = (uint32_t)grp * 16u;   // where this row's pick sits in the packed 64-bit scalar
    const int pxi = qx0 + (grp & 1) * 4 + (l16 & 3), pyi = qy0 + (grp >> 1) * 4 + (l16 >> 2);
    const bool inside = pxi < p.W && pyi < p.H;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const int pix_id = pyi * p.W + pxi;
    const int HW = p.W * p.H;

    const uint32_t tile = (uint32_t)(tile_y * p.gx + tile_x);
    const uint32_t r0 = p.ranges[2 * tile];
    // the forward wrote quad w's touch byte of a list entry only below covered[w] (wave-uniform)
    const uint4 cov4 = *reinterpret_cast<const uint4*>(p.covered + 4 * tile);

    // the deepest list entry any pixel of THIS QUAD reached
    const int last_contributor = inside ? (int)p.n_contrib[pix_id] : 0;
    int max_contrib = last_contributor;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) max_contrib = max(max_contrib, __shfl_xor(max_contrib, d, 64));
    max_contrib = __builtin_amdgcn_readfirstlane(max_contrib);
    if (max_contrib == 0) return;

    const bool clamp_pass = (p.flags & GSR_FLAG_CLAMP_PASSTHROUGH) != 0;
    const bool filter_depth_quirk = (p.flags & GSR_FLAG_FILTER_DEPTH_GRAD) != 0;

    // per-pixel state saved by the forward
    const float T_final = inside ? p.final_T[pix_id] : 0.f;
    const float final_D = (!NOSURF && inside) ? p.final_T[pix_id + HW] : 0.f;       // sum m w
    const float final_D2 = (!NOSURF && inside) ? p.final_T[pix_id + 2 * HW] : 0.f;  // sum m^2 w
    const float final_A = 1.0f - T_final;
    const int median_contributor = (!NOSURF && inside) ? (int)p.n_contrib[pix_id + HW] : 0;

    // A pixel nothing was blended into takes no part in the reference's backward (its loop over contributors is empty), so
    // whatever gradient arrives for it must not be read: the replay below is branch-free -- an idle lane contributes
    // 0 * (its pixel's gradient) to the 16-lane sums -- and the reference's OWN objective sends NaN to exactly these pixels
    // (gaussian_renderer/__init__.py:131-132: depth / alpha with alpha = 0, nan_to_num on the value only).
    const bool lit = inside && last_contributor > 0;
    float dL_dpix0 = 0.f, dL_dpix1 = 0.f, dL_dpix2 = 0.f;
    float dL_ddepth = 0.f, dL_daccum = 0.f, dL_dreg = 0.f, dL_dmedian = 0.f;
    float dL_dn0 = 0.f, dL_dn1 = 0.f, dL_dn2 = 0.f;
    if (lit) { dL_dpix0 = p.dL_dcolor[pix_id]; dL_dpix1 = p.dL_dcolor[pix_id + HW]; dL_dpix2 = p.dL_dcolor[pix_id + 2 * HW]; }
    if (!NOSURF && lit) {
        dL_ddepth = p.dL_dallmap[pix_id + 0 * HW];
        dL_daccum = p.dL_dallmap[pix_id + 1 * HW];
        dL_dn0 = p.dL_dallmap[pix_id + 2 * HW];
        dL_dn1 = p.dL_dallmap[pix_id + 3 * HW];
        dL_dn2 = p.dL_dallmap[pix_id + 4 * HW];
        dL_dmedian = p.dL_dallmap[pix_id + 5 * HW];
        dL_dreg = p.dL_dallmap[pix_id + 6 * HW];
    }
    const float bg_dot_dpixel = p.bg[0] * dL_dpix0 + p.bg[1] * dL_dpix1 + p.bg[2] * dL_dpix2;

    // (GSR_FLAG_NO_DIST_MEDIAN: the forward returned channels 5 and 6 as constants -- gradients sent to them are ignored)
    const bool dm_live = !NOSURF && (p.flags & (uint32_t)GSR_FLAG_NO_DIST_MEDIAN) == 0;
    const bool quad_has_dist = dm_live && __any(dL_dreg != 0.f), quad_has_median = dm_live && __any(dL_dmedian != 0.f);   // wave-uniform
    const bool quad_has_surf = !NOSURF && __any(dL_ddepth != 0.f || dL_daccum != 0.f || dL_dn0 != 0.f || dL_dn1 != 0.f || dL_dn2 != 0.f);

    // running state of the back-to-front recursion
    float T = T_final;
    float last_alpha = 0.f, last_q = 0.f, acc_q = 0.f, last_dL_dT = 0.f;

    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 pf0, pf1, pf2, pf3, pf4;
    uint32_t pf_touch = 0;   // named (not an array): keeps the prefetch in VGPRs, not scratch
    uint32_t pf_slot = 0;    // first gradient row of the prefetched entry (slot_off[emission index]: a dependent load, so
                             // the emission indices run two batches ahead like the ids)
    int hi = max_contrib;
    // ids and emission indices run two batches ahead of the replay, the per-entry words one batch ahead
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
        // The records of THIS batch are gathered here, not a batch ahead: holding the next batch's 20 registers per lane
        // through the replay cost two waves per SIMD (92 VGPRs, 5 waves -> 72, 7 waves), and seven waves hide the gather's
        // latency better than the prefetch did (K7 0.660 -> 0.640 ms same-box).  The ids, the touch
        // words and the row slots -- five registers -- still run ahead.
        GSR_GATHER5(ids_cur, nb);
        const uint32_t touch_of_lane = pf_touch;      // 4 bytes (one per quad) x 4 bits (one per 4x4 block): blended there?
        // first gradient row of staged entry `lane` (its rows are dense, in (quad, block) order of the set bits)
        const uint32_t slot_of_lane = pf_slot;
        // the record's two cull-rect words mean nothing to the backward: the staged copy carries the entry's first row
        // and its touch word there instead, so a block that picks entry j reads them with the record (they used to come
        // through two ds_bpermute per iteration)
        if (NOSURF) {   // [Tu Tv.x | Tv.yz Tw.xy | Tw.z xy opacity | rgb (first row up to this quad | this quad's nibble << 28)]
            const uint32_t packed = (slot_of_lane + (uint32_t)__popc(touch_of_lane & quads_below_mask)) |
                                    (((touch_of_lane >> (8 * wave)) & 0xFu) << 28);
            s_rec[lane * RS] = pf0; s_rec[lane * RS + 1] = pf1;
            s_rec[lane * RS + 2] = make_float4(pf2.x, pf2.y, pf2.z, pf3.z);
            s_rec[lane * RS + 3] = make_float4(pf3.w, pf4.x, pf4.y, __uint_as_float(packed));
        } else {
            s_rec[lane * 5] = pf0; s_rec[lane * 5 + 1] = pf1; s_rec[lane * 5 + 2] = pf2; s_rec[lane * 5 + 3] = pf3;
            s_rec[lane * 5 + 4] = make_float4(pf4.x, pf4.y, __uint_as_float(slot_of_lane), __uint_as_float(touch_of_lane));
        }
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

        // only the splats the forward blended into >= 1 pixel of a block carry any gradient there: one to-do
        // mask per 4x4 block (= DPP row), held by all 16 lanes of the row
        // (the four masks live in SGPRs: picking and clearing bits is scalar work, the vector unit only selects)
        // (the masks are kept BIT-REVERSED: the deepest entry of a block is then the lowest set bit, which s_ff1 finds
        // and reports as -1 on an empty mask by itself -- two scalar instructions per block and iteration)
        unsigned long long m0, m1, m2, m3;
        {
            const uint32_t t = lane < nb ? (touch_of_lane >> (8 * wave)) & 0xFu : 0u;
            m0 = __builtin_bitreverse64(__ballot((t & 1u) != 0)); m1 = __builtin_bitreverse64(__ballot((t & 2u) != 0));
            m2 = __builtin_bitreverse64(__ballot((t & 4u) != 0)); m3 = __builtin_bitreverse64(__ballot((t & 8u) != 0));
        }
        float probe_v = pxf; uint32_t probe_s = (uint32_t)__builtin_amdgcn_readfirstlane(nb);
        for (;;) {
            // deepest entry of each block's mask first; the four picks reach the lanes packed in one 64-bit scalar
            const int r0_ = rb_take_first(m0), r1_ = rb_take_first(m1), r2_ = rb_take_first(m2), r3_ = rb_take_first(m3);
            const uint32_t p01 = rb_pack16(r0_, r1_), p23 = rb_pack16(r2_, r3_);
            if ((p01 & p23) == 0xFFFFFFFFu) break;                  // all four masks were empty
            const unsigned long long picks = ((unsigned long long)p23 << 32) | p01;
            const int jrev = (int)(short)(picks >> pick_shift);     // this row's pick in reversed numbering; -1: none
            const bool has = jrev >= 0;
            const int j = (63 - jrev) & 63;
            const int cidx = lo + j;                            // 0-based position in the tile list
            const float4 a0 = s_rec[j * RS + 0], a1 = PROBE == 8 ? a0 : s_rec[j * RS + 1], a2 = s_rec[j * RS + 2];
            const float4 a3 = s_rec[j * RS + 3];     // general: [n.y n.z opacity r]; NOSURF: [r g b packed row | nibble]
            const float opa = NOSURF ? a2.w : a3.z;
            // Branch-free: EVERY lane runs the gradient math (masked-off lanes would cost the same issue slots),
            // and a lane that does not blend this splat gets alpha = G = 0 and harmless finite geometry, which
            // makes all 18 of its partial derivatives exact zeros and leaves its recursion state untouched
            // (T / (1 - 0) = T; the suffix sums advance by a zero-weight term).
            if (PROBE == 1) {
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(probe_v));
            }
            if (PROBE == 4) {
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("s_add_u32 %0, %0, 1" : "+s"(probe_s) : : "scc");
            }
            GsrPair pr;
            const bool ok = gsr_pair_eval(pxf, pyf, a0, a1, a2, opa, pr);
            const bool active = has && cidx < last_contributor && ok;
            float gT[9];
            float gxy0, gxy1, gn0, gn1, gn2, gopa, gc0, gc1, gc2;
            uint32_t rec_slot, rec_touch;   // first gradient row and touch word of this row's entry (staged with the record)
            {
                float4 a4;
                if (NOSURF) {
                    const uint32_t packed = __float_as_uint(a3.w);
                    rec_slot = packed & 0x0FFFFFFFu;               // (already counts the quads before this one)
                    rec_touch = (packed >> 28) << (8 * wave);      // this quad's nibble, where below_mask looks for it
                    a4 = make_float4(a3.y, a3.z, 0.f, 0.f);
                } else {
                    a4 = s_rec[j * 5 + 4];
                    rec_slot = __float_as_uint(a4.z); rec_touch = __float_as_uint(a4.w);
                }
                const float alpha = active ? pr.alpha : 0.f, G = active ? pr.G : 0.f, c_d = active ? pr.depth : 1.f;
                const float sx = active ? pr.sx : 0.f, sy = active ? pr.sy : 0.f, inv_pz = active ? pr.inv_pz : 0.f;
                const float one_m_alpha = 1.0f - alpha;
                const float inv_oma = gsr_rcp(one_m_alpha);
                T = T * inv_oma;
                const float w = alpha * T;

                // colour, expected depth, alpha and normal share one suffix recursion:
                //   q_i = c_i . dL/dC + z_i dL/dD + 1 dL/dA + n_i . dL/dN
                const float c0 = NOSURF ? a3.x : a3.w, c1 = a4.x, c2 = a4.y;
                const float n0 = a2.w, n1 = a3.x, n2 = a3.y;    // (only read when quad_has_surf)
                // (the surface channels -- depth, alpha, normal -- carry no gradient before the regularizers switch on)
                float q = 0.f;
                if (quad_has_surf) q = c_d * dL_ddepth + dL_daccum + n0 * dL_dn0 + n1 * dL_dn1 + n2 * dL_dn2;
                q += c0 * dL_dpix0 + c1 * dL_dpix1 + c2 * dL_dpix2;
                acc_q = last_alpha * last_q + (1.f - last_alpha) * acc_q;
                last_q = q;
                float dL_dalpha = q - acc_q;
                gc0 = w * dL_dpix0; gc1 = w * dL_dpix1; gc2 = w * dL_dpix2;
                gn0 = 0.f; gn1 = 0.f; gn2 = 0.f;
                if (quad_has_surf) { gn0 = w * dL_dn0; gn1 = w * dL_dn1; gn2 = w * dL_dn2; }

                // median depth and distortion: skipped (wave-uniformly) when the whole quad receives no gradient on that
                // channel -- the reference's defaults (depth_ratio = 0, lambda_dist = 0) make both identically zero, and
                // every term below is a multiple of it
                float dL_dz = NOSURF ? 0.f : w * dL_ddepth;
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
                // alpha also scales how much background shows through
                dL_dalpha -= T_final * inv_oma * bg_dot_dpixel;

                // alpha = min(0.99, opa * G)
                const float dL_daraw = (clamp_pass || pr.araw <= GSR_ALPHA_MAX) ? dL_dalpha : 0.f;
                const float dL_dG = opa * dL_daraw;
                gopa = G * dL_daraw;

                const float Twx = a1.z, Twy = a1.w;
                if (pr.use3d) {
                    const float dL_dsx = NOSURF ? dL_dG * (-G * sx) : dL_dG * (-G * sx) + dL_dz * Twx;
                    const float dL_dsy = NOSURF ? dL_dG * (-G * sy) : dL_dG * (-G * sy) + dL_dz * Twy;
                    float dpx = dL_dsx * inv_pz, dpy = dL_dsy * inv_pz;
                    if (__builtin_expect(pr.tiny_any, 0)) {       // (pair_eval.h: a denormal p.z; the empty asm keeps this a BRANCH -- if-converted it cost
                        asm volatile("");                         //  three vector instructions on every pair, +3.7 % of K7's issue)
                        const float zs = pr.tiny ? GSR_TINY_PZ_SCALE : 1.f; dpx *= zs; dpy *= zs;
                    }
                    const float dpz = -(dpx * sx + dpy * sy);
                    // dL/dTu = -dL/dk = dL/dp x l ;  dL/dTv = -dL/dl = k x dL/dp
                    const float ux = dpy * pr.lz - dpz * pr.ly, uy = dpz * pr.lx - dpx * pr.lz, uz = dpx * pr.ly - dpy * pr.lx;
                    const float vx = pr.ky * dpz - pr.kz * dpy, vy = pr.kz * dpx - pr.kx * dpz, vz = pr.kx * dpy - pr.ky * dpx;
                    gT[0] = ux; gT[1] = uy; gT[2] = uz;
                    gT[3] = vx; gT[4] = vy; gT[5] = vz;
                    if (NOSURF) {      // (the general form with dL_dz = +0: 0 * s - a - b == -a - b up to the sign of a zero)
                        gT[6] = 0.f - pxf * ux - pyf * vx;
                        gT[7] = 0.f - pxf * uy - pyf * vy;
                        gT[8] = 0.f - pxf * uz - pyf * vz;
                    } else {
                        gT[6] = dL_dz * sx - pxf * ux - pyf * vx;
                        gT[7] = dL_dz * sy - pxf * uy - pyf * vy;
                        gT[8] = dL_dz - pxf * uz - pyf * vz;
                    }
                    gxy0 = 0.f; gxy1 = 0.f;
                } else {
                    gxy0 = dL_dG * (-G * GSR_FILTER_INV_SQUARE * pr.dx);
                    gxy1 = dL_dG * (-G * GSR_FILTER_INV_SQUARE * pr.dy);
                    gT[0] = 0.f; gT[1] = 0.f; gT[2] = 0.f; gT[3] = 0.f; gT[4] = 0.f; gT[5] = 0.f;
                    gT[6] = (!NOSURF && filter_depth_quirk) ? sx * dL_dz : 0.f;
                    gT[7] = (!NOSURF && filter_depth_quirk) ? sy * dL_dz : 0.f;
                    gT[8] = dL_dz;
                }
            }

            // block-level sums (16 lanes), stored by the lanes that end up holding them (row layout GSR_GR_*)
            {
                const float v16[16] = {gT[0], gT[1], gT[2], gT[3], gT[4], gT[5], gT[6], gT[7], gT[8],
                                       gn0, gn1, gn2, gopa, gc0, gc1, gc2};
                const float tot = row_sum16_transposed(v16, l16);
                const float xy = row_sum2(gxy0, gxy1, l16);
                // every row whose to-do bit was set writes its slot (zeros if no pixel turned out active), so
                // the reduction never reads a row that was not written
                // (NOSURF: rec_slot already counts the quads before this one and rec_touch holds this quad's nibble only)
                const size_t slot = (size_t)rec_slot + __popc(rec_touch & below_mask);
                if (has) {
                    p.grad_rows[slot * RB_ROW + l16] = tot;                   // one aligned 64-byte store per row
                    if ((l16 & 7) == 0) p.grad_xy[slot * GSR_GROW_XY + (l16 >> 3)] = xy;
                }
            }
        }
        if (PROBE == 1 && probe_v == 1.2345f) T += 1.f;
        if (PROBE == 4 && probe_s == 0x7fffffffu) T += 1.f;
        __builtin_amdgcn_wave_barrier();   // all reads of this batch precede the next batch's LDS writes
        hi = lo;
    }
#ifdef GSR_DEV_PROBES
    if (PROBE == 7) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
        const uint32_t w = blockIdx.x * 4u + (uint32_t)wave;
        if (lane == 0 && w < RB_STAMP_WAVES) { g_rb_stamps[2 * w] = c1 - stamp_c0; g_rb_stamps[2 * w + 1] = t1 - stamp_t0; }
    }
#endif
}

// Wide payload: add a Gaussian's feature rows (same dense slots as the geometry rows) in fixed order.  One thread
// per (depth rank, 4-channel piece); writes dL_dcolors [N,C] by Gaussian id (zeros for Gaussians with no instance).
// As in reduce_rows, a Gaussian with more than RF_BIG rows is noted (by its piece-0 thread) and summed afterwards by the
// whole workgroup -- 256 / C4 row lanes per piece, partials combined through LDS in lane order -- instead of one thread
// walking tens of thousands of rows.
#define RF_BIG 192
__global__ void __launch_bounds__(256) reduce_feat_rows_kernel(long long n_threads, int C4,
                                                               const uint32_t* __restrict__ order,
                                                               const uint32_t* __restrict__ row_begin,
                                                               const float4* __restrict__ rows,
                                                               float4* __restrict__ out) {
    __shared__ uint32_t s_big[256];
    __shared__ int s_nbig;
    __shared__ float4 s_part[256];
    if (threadIdx.x == 0) s_nbig = 0;
    __syncthreads();
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_threads) {
        const int r = (int)(t / C4), q = (int)(t - (long long)r * C4);
        // (row_begin[r] = slot_off[offs[r]] comes from the render_bwd launch; the destination is read here, beside the row
        // range, and four rows are in flight per thread: the kernel is bound by its chain of dependent loads, not by bytes)
        const uint32_t s0 = row_begin[r], s1 = row_begin[r + 1], dst = order[r];
        if (s1 - s0 > (uint32_t)RF_BIG) {
            if (q == 0) s_big[atomicAdd(&s_nbig, 1)] = (uint32_t)r;
        } else {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (uint32_t sl = s0; sl < s1; sl += 4) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (sl + u < s1) v[u] = rows[(size_t)(sl + u) * C4 + q];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
            out[(size_t)dst * C4 + q] = acc;
        }
    }
    __syncthreads();
    const int nbig = s_nbig;                       // uniform over the workgroup
    const int L = 256 / C4;                        // row lanes per piece
    const int q = (int)threadIdx.x % C4, rl = (int)threadIdx.x / C4;
    for (int b = 0; b < nbig; ++b) {
        const uint32_t rb = s_big[b];
        const uint32_t s0 = row_begin[rb], s1 = row_begin[rb + 1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rl < L)
            for (uint32_t sl = s0 + (uint32_t)rl; sl < s1; sl += (uint32_t)L) {
                const float4 v = rows[(size_t)sl * C4 + q];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        s_part[threadIdx.x] = acc;
        __syncthreads();
        if ((int)threadIdx.x < C4) {
            float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int l = 0; l < L; ++l) {
                const float4 v = s_part[l * C4 + threadIdx.x];
                tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w;
            }
            out[(size_t)order[rb] * C4 + threadIdx.x] = tot;
        }
        __syncthreads();
    }
}

int gsr_launch_reduce_feat_rows(int N, int C, const uint32_t* order, const uint32_t* row_begin,
                                const float* feat_rows, float* dL_dcolors, hipStream_t s) {
    if (N <= 0) return GSR_OK;
    GsrProfileScope prof(GSR_K_PREPROCESS_BWD, s);
    const int C4 = C / 4;
    const long long n_threads = (long long)N * C4;
    hipLaunchKernelGGL(reduce_feat_rows_kernel, dim3((unsigned)((n_threads + 255) / 256)), dim3(256), 0, s, n_threads, C4,
                       order, row_begin, reinterpret_cast<const float4*>(feat_rows),
                       reinterpret_cast<float4*>(dL_dcolors));
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}

int gsr_launch_render_bwd(const GsrView& v, const uint32_t* ranges, const uint32_t* covered, const uint32_t* inst_row,
                          const float* splat, const uint32_t* touch, const uint32_t* slot_off, const float* final_T,
                          const uint32_t* n_contrib, const float* dL_dcolor, const float* dL_dallmap, float* grad_rows,
                          float* grad_xy, int N, const uint32_t* offs, uint32_t* row_begin, const float* feat,
                          const uint32_t* point_list, float* feat_rows, size_t max_rows, hipStream_t s) {
    RenderBwdParams p;
    p.W = v.width; p.H = v.height; p.gx = (v.width + GSR_TILE - 1) / GSR_TILE;
    const int gy = (v.height + GSR_TILE - 1) / GSR_TILE;
    p.flags = v.flags;
    p.ranges = ranges; p.covered = covered; p.inst_row = inst_row; p.splat = reinterpret_cast<const float4*>(splat); p.touch = touch;
    p.slot_off = slot_off; p.bg = v.bg;
    p.final_T = final_T; p.n_contrib = n_contrib; p.dL_dcolor = dL_dcolor; p.dL_dallmap = dL_dallmap;
    p.grad_rows = grad_rows; p.grad_xy = grad_xy;
    p.N = N; p.offs = offs; p.row_begin = row_begin;
    p.feat = feat; p.point_list = point_list; p.feat_rows = feat_rows; p.C = v.channels;
    if (p.gx <= 0 || gy <= 0) return GSR_OK;
    GsrProfileScope prof(GSR_K_RENDER_BWD, s);
    p.n_tiles = p.gx * gy; p.per_xcd = (p.n_tiles + 7) / 8;
    const dim3 grid(8 * p.per_xcd), block(RB_BLOCK);
    // no gradient on any allmap channel (the caller's promise) and row indices that fit 28 bits: the 4-part-record kernel
    const bool nosurf = feat == nullptr && (v.flags & (uint32_t)GSR_FLAG_NO_SURFACE_GRAD) != 0 && max_rows < (size_t(1) << 28);
    if (nosurf) {
        hipLaunchKernelGGL((render_bwd_kernel<0, true>), grid, block, 0, s, p);
    } else if (feat == nullptr) {
#ifdef GSR_DEV_PROBES
        const char* e = getenv("GSR_K7_PROBE");   // re-read per launch
        switch (e ? atoi(e) : 0) {
            case 1: hipLaunchKernelGGL((render_bwd_kernel<1>), grid, block, 0, s, p); break;
            case 3: hipLaunchKernelGGL((render_bwd_kernel<3>), grid, block, 0, s, p); break;
            case 4: hipLaunchKernelGGL((render_bwd_kernel<4>), grid, block, 0, s, p); break;
            case 6: hipLaunchKernelGGL((render_bwd_kernel<6>), grid, block, 0, s, p); break;
            case 7: hipLaunchKernelGGL((render_bwd_kernel<7>), grid, block, 0, s, p); break;
            case 8: hipLaunchKernelGGL((render_bwd_kernel<8>), grid, block, 0, s, p); break;
            default: hipLaunchKernelGGL((render_bwd_kernel<>), grid, block, 0, s, p);
        }
#else
        hipLaunchKernelGGL((render_bwd_kernel<>), grid, block, 0, s, p);
#endif
    } else {
        const int rc = gsr_launch_render_bwd_wide(p, v.channels, grid, s);   // render_bwd_wide.hip
        if (rc != GSR_OK) return rc;
    }
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}
