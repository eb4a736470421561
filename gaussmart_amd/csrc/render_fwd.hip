// K6 render_fwd: front-to-back alpha compositing of 2D-Gaussian surfels with the depth / normal /
// median-depth / distortion side outputs.  Restates the [U] forward render; output channel order
// pinned by gaussian_renderer/__init__.py:117-141 (allmap = depth, alpha, normal xyz, median, dist).
//
// MI355X mapping: WAVE-INDEPENDENT.  A 16x16 tile is still one 256-thread workgroup (tile keys stay
// those of the reference), but its four wave64s each own an 8x8 pixel quad and never synchronise
// with each other -- no workgroup barrier in the compositing (one at the very start clears an LDS counter):
//   * a wave stages the tile's splat records 64 at a time into its PRIVATE 5 KiB LDS slice: the Gaussian
//     ids of the list (point_list) are read two batches ahead, the 80-byte records are gathered by id
//     one batch ahead (five 16-byte loads per lane, consecutive lanes = consecutive pieces of a record)
//     into registers while the current batch is composited -- no copy of the records in list order;
//   * each lane tests ONE staged splat's cull rect against the wave's quad; the 64-bit ballot is the
//     list of splats the wave has to look at at all (iterated with s_ff1), the rest cost nothing;
//   * per surviving splat the record is read with wave-uniform (broadcast) ds_read_b128;
//   * the wave leaves as soon as ITS 64 pixels are done (quad-level early termination);
//   * per (instance, quad) it records one byte: "blended into at least one pixel" -- the exact set of
//     pairs the backward has to evaluate;
//   * the last wave of a tile to finish counts the tile's gradient rows per instance for the backward (see the
//     end of the kernel).
#include <stdlib.h>
#include "gsr_common.h"
#include "pair_eval.h"

// Records are gathered by Gaussian id straight from the splat table (80-byte records, 16-byte aligned): lane l
// fetches the five 16-byte parts of staged entry l, whose id it already holds.  (A cooperative mapping --
// consecutive lanes = consecutive parts -- touches fewer lines per instruction but needs five ds_bpermute and ten
// more live registers; measured slower.)  No copy of the records in list order exists any more: the former
// "splat stream" cost a 140 us kernel and 240 MB per frame to save the render kernels nothing they can feel.
#define GSR_GATHER5(ids_, cnt_)                                                                      \
    do {                                                                                             \
        pf0 = zero4; pf1 = zero4; pf2 = zero4; pf3 = zero4; pf4 = zero4;                             \
        if (lane < (cnt_)) {                                                                         \
            const float4* rec_ = p.splat + (size_t)(ids_) * 5;                                       \
            pf0 = rec_[0]; pf1 = rec_[1]; pf2 = rec_[2]; pf3 = rec_[3]; pf4 = rec_[4];               \
        }                                                                                            \
    } while (0)

#ifdef GSR_DEV_PROBES
// PROBE 7: every wave leaves (start, end) on the 100 MHz s_memrealtime clock, its hardware id and (round 4) its start and
// end on the shader clock (s_memtime) -- the occupancy of a launch over time (scripts/dev_wave_timeline.py) and the clock
// the chip holds while the kernel runs (scripts/dev_clock_probe.py).  Developer builds only.
#define RF_STAMP_WAVES 65536
#define RF_STAMP_WORDS 5
__device__ unsigned long long g_rf_stamps[RF_STAMP_WORDS * RF_STAMP_WAVES];
extern "C" int gsr_probe_read_stamps(void* dst, size_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_rf_stamps), bytes < sizeof(g_rf_stamps) ? bytes : sizeof(g_rf_stamps)) == hipSuccess ? 0 : -1;
}
#endif
#define RF_BLOCK 256
#define RF_WAVES 4
#ifndef RF_PFD
#define RF_PFD 2   // wide payload: how many surviving entries ahead their features are fetched
#endif

struct RenderFwdParams {
    int W, H, gx, n_tiles, per_xcd;
    uint32_t flags;
    const uint32_t* ranges; const float4* splat;   // splat table [N] x 5 float4, gathered by id
    const float* bg;
    float* final_T; uint32_t* n_contrib; float* out_color; float* out_allmap;
    uint8_t* touch; uint32_t* covered;
    // SAVE: the last wave of a tile to finish counts the tile's gradient rows per instance (slot_cnt[inst_row[...]])
    const uint32_t* inst_row; uint8_t* slot_cnt;
    // wide payload (FEAT16 > 0): C = 4..64 feature channels per Gaussian instead of the RGB of the record
    const float* feat; const uint32_t* point_list; int C;
};

#ifndef RF_MIN_WAVES
#define RF_MIN_WAVES 7   // <= 72 VGPRs.  With the two-stage quad cull the kernel no longer fits 64 registers without
                         // spilling 13 dwords per lane around every batch: 8 waves 0.450 ms, 7 waves 0.424 ms, 6 waves
                         // (79 VGPRs, no scratch at all) 0.437 ms
#endif
template <int CTRL>
__device__ __forceinline__ uint32_t row_or_step(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

// FEAT16 = 0: three colour channels taken from the splat record (the reference's configuration).
// FEAT16 = 1..4: up to 16*FEAT16 feature channels read from `feat` [N,C] by Gaussian id (wide per-pixel
// payload); everything else is identical.  The feature accumulate C[pixel][ch] += w[pixel] * f[ch] is an outer product
// per surviving entry, and it runs on the MATRIX pipe: v_mfma_f32_16x16x1_4b_f32 (exact f32, one fma per product, k-ordered:
// bit-identical to the scalar fmaf chain it replaces) takes B = the lane's own blending weight -- block b of the
// instruction is DPP row b = 4x4 pixel block b, column j = lane & 15 = the lane's own pixel, so the weights need no
// cross-lane movement at all -- and A = the entry's features, 16 lanes x FEAT16 consecutive floats read straight from the
// feature row with a scalar base (row i of M-tile m is channel FEAT16 * i + m).  One instruction per 16 channels and
// entry, issued for every entry that survives the quad cull with weight 0 on the lanes that do not blend it; the VALU only
// sees FEAT16 more issue slots per pair instead of C fused multiply-adds behind a per-splat feature fetch (round 2:
// C = 64 ran at 3.8 ms load-latency-bound, C = 16 at 0.86 ms).  Features are prefetched two surviving entries ahead.
// SAVE = false: forward-only rendering (render.py / view.py under torch.no_grad(); utils/mesh_utils.py:100-123): no
// touch bytes, no per-pixel state for a backward that will never run -- the mask bookkeeping, four DPP OR steps and
// eight readlanes per batch and 20 of the 60 bytes written per pixel disappear.
// PROBE (developer builds only: make PROBES=1, scripts/dev_probe.py): 1 = eight more dependent VALU per splat, 4 = eight
// more dependent SALU, 2 = no distortion arithmetic, 3 = four LDS reads instead of five, 5 = no reciprocal in the depth
// mapping.  Probes 2, 3 and 5 render WRONG images: they exist to time the loop (DESIGN.md section 4 holds what they showed).
// leaving a pair early: the RGB kernel goes straight to the next entry; the wide kernel still owes the entry its MFMA
// (a plain block, NOT do { } while (0): `continue` must reach the loop over the surviving entries)
#define RF_NEXT_PAIR { if (FEAT16 > 0) goto pair_done; else continue; }
// MAPS (round 4): which allmap channels the caller consumes.
//   2  all seven (the reference operator).
//   1  GSR_FLAG_NO_DIST_MEDIAN: not the distortion and the median depth -- the reference's DEFAULT training configuration
//      (lambda_dist = 0, arguments/__init__.py:87; depth_ratio = 0, :72: surf_depth is the expected depth alone,
//      gaussian_renderer/__init__.py:141).  The distortion arithmetic (one reciprocal included) is 10 % of the kernel
//      (DESIGN_history.md, probe 2); channels 5 and 6 and the M1 / M2 / median-contributor state come back as zeros / -1, and
//      the backward ignores gradients sent to those two channels (they are constants).
//   0  GSR_FLAG_COLOR_ONLY: none -- the fused trainer while no regularizer is active (the first 7,000 iterations of every
//      run, train.py:132-133, and the reference's evaluation flags, scripts/dtu_eval.py:45): allmap is not written, the
//      image state keeps only T and the last contributor, the backward runs under GSR_FLAG_NO_SURFACE_GRAD.
// Colour, T, the remaining channels, radii, touch words and row counts are those of the general kernel bit for bit.
template <int FEAT16, bool SAVE, int PROBE = 0, int MAPS = 2>
__global__ void __launch_bounds__(RF_BLOCK, FEAT16 == 0 ? RF_MIN_WAVES : (FEAT16 == 1 ? 5 : FEAT16 == 2 ? 4 : 3)) render_fwd_kernel(RenderFwdParams p) {
    __shared__ float4 s_rec_all[RF_WAVES][64 * 5];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    float4* s_rec = s_rec_all[wave];
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so workgroup b
    // takes tile (b % 8) * per_xcd + b / 8 -- every XCD owns one contiguous band of tiles, and the records shared by
    // neighbouring tiles are fetched into ONE L2 instead of several
    const int tile_lin = (int)(blockIdx.x & 7u) * p.per_xcd + (int)(blockIdx.x >> 3);
#ifdef GSR_DEV_PROBES
    unsigned long long stamp_t0 = 0, stamp_c0 = 0;
    if (PROBE == 7) { stamp_t0 = __builtin_amdgcn_s_memrealtime(); stamp_c0 = __builtin_amdgcn_s_memtime(); }
#endif
    if (tile_lin >= p.n_tiles) return;
    __shared__ uint32_t s_waves_done, s_cov[RF_WAVES];
    if (SAVE) {   // the only workgroup barrier of the kernel, before anything has started
        if (tid == 0) s_waves_done = 0;
        __syncthreads();
    }
    const int tile_y = tile_lin / p.gx, tile_x = tile_lin - tile_y * p.gx;
    const int qx0 = tile_x * GSR_TILE + (wave & 1) * 8, qy0 = tile_y * GSR_TILE + (wave >> 1) * 8;
    // lanes 16g..16g+15 (one DPP row) own the 4x4 pixel block g of the quad: the backward walks per-block lists
    const int grp = lane >> 4, l16 = lane & 15;
    const int pxi = qx0 + (grp & 1) * 4 + (l16 & 3), pyi = qy0 + (grp >> 1) * 4 + (l16 >> 2);
    const bool inside = pxi < p.W && pyi < p.H;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const int pix_id = pyi * p.W + pxi;
    const int HW = p.W * p.H;
    const bool no_cull = (p.flags & (uint32_t)GSR_FLAG_DEBUG_NO_CULL) != 0;
    const bool tight_cull = (p.flags & (uint32_t)GSR_FLAG_DEBUG_RECT_CULL_ONLY) == 0;

    const uint32_t tile = (uint32_t)(tile_y * p.gx + tile_x);
    const uint32_t r0 = p.ranges[2 * tile], r1 = p.ranges[2 * tile + 1];
    const int n_list = (int)(r1 - r0);

    bool done = !inside;
    float T = 1.0f;
    uint32_t last_contributor = 0;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f;
    // wide payload: FEAT16 accumulator tiles of the 4-block outer-product MFMA; register v of lane l holds
    // (pixel 16 (v >> 2) + (l & 15), channel FEAT16 * (4 (l >> 4) + (v & 3)) + m) of tile m
    constexpr int NM = FEAT16 > 0 ? FEAT16 : 1;
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc[NM];
#pragma unroll
    for (int k = 0; k < NM; ++k)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[k][v] = 0.f;
    float N0 = 0.f, N1 = 0.f, N2 = 0.f;
    float Dacc = 0.f, M1 = 0.f, M2 = 0.f, dist = 0.f, med_depth = 0.f;
    uint32_t med_contrib = 0xFFFFFFFFu;   // "-1" stored in the u32 plane, as recalled

    float4 pf0, pf1, pf2, pf3, pf4;   // named (not an array): keeps the prefetch in VGPRs, not scratch
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // ids run two batches ahead of the compositing, records one batch ahead
    uint32_t ids_cur = lane < min(64, n_list) ? p.point_list[r0 + lane] : 0u;
    GSR_GATHER5(ids_cur, min(64, n_list));
    uint32_t ids_nxt = 64 + lane < n_list ? p.point_list[r0 + 64 + lane] : 0u;

    int covered = 0;   // list entries whose touch byte this wave has written
    for (int base = 0; base < n_list; base += 64) {
        if (__all(done)) break;
        const int nb = min(64, n_list - base);
        // lane l still holds the record of staged entry l: cull against this quad before the registers are recycled
        // (rect first; the ellipse test only tightens it)
        bool ov = false;
        if (lane < nb)
            ov = no_cull || (gsr_rect_overlaps_quad(__float_as_uint(pf4.z), __float_as_uint(pf4.w), qx0, qy0) &&
                             (!tight_cull || gsr_tight_overlaps_quad(pf0, pf1, pf2, pf3.z, qx0, qy0)));
s_rec[lane * 5] = pf0; s_rec[lane * 5 + 1] = pf1; s_rec[lane * 5 + 2] = pf2; s_rec[lane * 5 + 3] = pf3; s_rec[lane * 5 + 4] = pf4;
        const uint32_t id_of_lane = ids_cur;
        {   // prefetch the next batch while this one is composited
            const int nxt = base + 64;
            const int cnt_nxt = nxt < n_list ? min(64, n_list - nxt) : 0;
            GSR_GATHER5(ids_nxt, cnt_nxt);
            ids_cur = ids_nxt;
            ids_nxt = nxt + 64 + lane < n_list ? p.point_list[r0 + nxt + 64 + lane] : 0u;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        unsigned long long m = __ballot(ov);
        // wide payload: A operands (features) of the next RF_PFD surviving entries, fetched ahead of their use
        float fq[RF_PFD][NM];
        unsigned long long mq = m;
        auto fetch_features = [&](float (&f)[NM]) {
#pragma unroll
            for (int e = 0; e < NM; ++e) f[e] = 0.f;
            if (FEAT16 > 0 && mq) {                          // wave-uniform
                const int jn = __builtin_ctzll(mq);
                mq &= mq - 1;
                const uint32_t gid = (uint32_t)__builtin_amdgcn_readlane((int)id_of_lane, jn);   // scalar
                const float* src = p.feat + (size_t)gid * p.C + NM * l16;
#pragma unroll
                for (int e = 0; e < NM; ++e)
                    if (NM * l16 + e < p.C) f[e] = src[e];
            }
        };
        if (FEAT16 > 0) {
#pragma unroll
            for (int d = 0; d < RF_PFD; ++d) fetch_features(fq[d]);
        }
        uint32_t mine_lo = 0u, mine_hi = 0u;   // staged splats THIS pixel blends (bit j)
        float probe_v = pxf; uint32_t probe_s = (uint32_t)__builtin_amdgcn_readfirstlane(nb);
        while (m) {
            // every lane is active here (the loop is wave-uniform), so the vote sees the whole wave
            if (__all(done)) break;
            const int j = __builtin_ctzll(m);
            m &= m - 1;
            if (PROBE == 4) {
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("s_add_u32 %0, %0, 1" : "+s"(probe_s) : : "scc");
            }
            float w_pair = 0.f;     // this lane's blending weight of entry j (0: not blended) -- the MFMA's B operand
            float fc[NM];
            if (FEAT16 > 0) {
#pragma unroll
                for (int e = 0; e < NM; ++e) {
                    fc[e] = fq[0][e];
#pragma unroll
                    for (int d = 0; d + 1 < RF_PFD; ++d) fq[d][e] = fq[d + 1][e];
                }
                fetch_features(fq[RF_PFD - 1]);
            }
            {
            if (done) RF_NEXT_PAIR;
            const uint32_t contributor = (uint32_t)(base + j + 1);   // 1-based position in the tile list
            const float4 a0 = s_rec[j * 5 + 0], a1 = s_rec[j * 5 + 1], a2 = s_rec[j * 5 + 2];
            const float4 a3 = s_rec[j * 5 + 3];
            if (PROBE == 1) {
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(probe_v));
            }
            GsrPair pr;
            if (!gsr_pair_eval(pxf, pyf, a0, a1, a2, a3.z, pr)) RF_NEXT_PAIR;
            const float alpha = pr.alpha, depth = pr.depth;
            const float test_T = T * (1.0f - alpha);
            if (test_T < GSR_T_EPS) { done = true; RF_NEXT_PAIR; }
            const float4 a4 = PROBE == 3 ? a3 : s_rec[j * 5 + 4];
            const float w = alpha * T;
            if (MAPS == 2) {
                const float A = 1.0f - T;
                float dm_dz_unused;
                const float m_d = PROBE == 5 ? depth : gsr_depth_map(depth, dm_dz_unused);
                if (PROBE != 2) {
                    dist += (m_d * m_d * A + M2 - 2.0f * m_d * M1) * w;
                    M1 += m_d * w;
                    M2 += m_d * m_d * w;
                }
                if (T > 0.5f) { med_depth = depth; med_contrib = contributor; }
            }
            if (MAPS >= 1) {
                Dacc += depth * w;
                N0 += a2.w * w; N1 += a3.x * w; N2 += a3.y * w;
            }
            if (FEAT16 == 0) { C0 += a3.w * w; C1 += a4.x * w; C2 += a4.y * w; }
            else w_pair = w;
            T = test_T;
            if (SAVE) {
                last_contributor = contributor;
                const unsigned long long bit = 1ull << j;          // wave-uniform (scalar shift)
                mine_lo |= (uint32_t)bit; mine_hi |= (uint32_t)(bit >> 32);
            }
            }
        pair_done:
            if (FEAT16 > 0) {   // every lane is active again: outer product of this entry's features with the 64 weights
#pragma unroll
                for (int k = 0; k < NM; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x1f32(fc[k], w_pair, acc[k], 0, 0, 0);
            }
        }
        if (PROBE == 1) C0 += probe_v * 1e-30f;
        if (PROBE == 4) C0 += (float)probe_s * 1e-30f;
        if (!SAVE) {
            __builtin_amdgcn_wave_barrier();   // all reads of this batch precede the next batch's LDS writes
            continue;
        }
        // OR over the 16 pixels of each 4x4 block (DPP inside a row), once per batch: which staged splats did
        // this block blend at all?
        mine_lo |= row_or_step<0xB1>(mine_lo); mine_hi |= row_or_step<0xB1>(mine_hi);     // quad_perm [1,0,3,2]
        mine_lo |= row_or_step<0x4E>(mine_lo); mine_hi |= row_or_step<0x4E>(mine_hi);     // quad_perm [2,3,0,1]
        mine_lo |= row_or_step<0x141>(mine_lo); mine_hi |= row_or_step<0x141>(mine_hi);   // row_half_mirror
        mine_lo |= row_or_step<0x140>(mine_lo); mine_hi |= row_or_step<0x140>(mine_hi);   // row_mirror
        // one byte per (instance, quad) for the backward, bit g = "block g blended it": lane j reports staged splat j
        // (the readlanes stay outside the `lane < nb` branch: lanes 16g must be active when they are read)
        uint32_t byte = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)mine_lo, 16 * g);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)mine_hi, 16 * g);
            byte |= (((lane < 32 ? lo : hi) >> (lane & 31)) & 1u) << g;
        }
        if (lane < nb) p.touch[((size_t)r0 + base + lane) * 4 + wave] = (uint8_t)byte;
        covered = base + nb;
        __builtin_amdgcn_wave_barrier();   // all reads of this batch precede the next batch's LDS writes
    }

    // entries behind the point where every pixel saturated were never staged: nothing was blended there and their touch
    // bytes stay UNWRITTEN -- the backward reads this quad's byte only below `covered` (with thousands of occluded
    // entries per tile, zero-filling them cost more store traffic than the compositing itself)
    if (SAVE && lane == 0) p.covered[tile * 4 + wave] = (uint32_t)covered;

    if (inside) {
        if (SAVE) {
            p.final_T[pix_id] = T;
            p.n_contrib[pix_id] = last_contributor;
            if (MAPS >= 1) {      // (MAPS == 1: zeros and "no median contributor" -- the accumulators were never touched)
                p.final_T[pix_id + HW] = M1;
                p.final_T[pix_id + 2 * HW] = M2;
                p.n_contrib[pix_id + HW] = med_contrib;
            }
        }
        if (FEAT16 == 0) {
            p.out_color[pix_id] = C0 + T * p.bg[0];
            p.out_color[pix_id + HW] = C1 + T * p.bg[1];
            p.out_color[pix_id + 2 * HW] = C2 + T * p.bg[2];
        }
        if (MAPS >= 1) {
            p.out_allmap[pix_id + 0 * HW] = Dacc;
            p.out_allmap[pix_id + 1 * HW] = 1.0f - T;
            p.out_allmap[pix_id + 2 * HW] = N0;
            p.out_allmap[pix_id + 3 * HW] = N1;
            p.out_allmap[pix_id + 4 * HW] = N2;
            p.out_allmap[pix_id + 5 * HW] = med_depth;
            p.out_allmap[pix_id + 6 * HW] = dist;
        }
    }
    if (FEAT16 > 0) {   // the accumulator tiles hold (pixel of block b, channel) pairs of OTHER lanes' pixels: see `acc`
        // Round 4: the [C, H, W] image leaves through the wave's LDS slice (the list walk is over), 16 channels at a time, so
        // that ONE store instruction writes ONE channel plane: 8 rows of 32 bytes of the quad, lane = pixel.  Straight from
        // the accumulator layout an instruction wrote 4 channels x 4 rows x 16 bytes -- sixteen 16-byte pieces, each charged a
        // whole 64-byte sector (0.3 ms of the 1.6 ms at C = 64).
        float* tile = reinterpret_cast<float*>(s_rec);                 // 16 channels x 65 floats (padded: rows hit distinct banks)
        const int rqx = lane & 7, rqy = lane >> 3;                     // my pixel when the tile is read back: row-major in the quad
        const int src_lane = 16 * ((rqy >> 2) * 2 + (rqx >> 2)) + (rqy & 3) * 4 + (rqx & 3);      // ... and the lane that composited it
        const float Tq = __shfl(T, src_lane, 64);
        const int opx = qx0 + rqx, opy = qy0 + rqy;
        const bool oin = opx < p.W && opy < p.H;
        const size_t opid = (size_t)opy * p.W + opx;
#pragma unroll
        for (int k = 0; k < NM; ++k) {
            __builtin_amdgcn_wave_barrier();                           // the previous tile has been read
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int bq = v >> 2;
                const int q = ((bq >> 1) * 4 + (l16 >> 2)) * 8 + (bq & 1) * 4 + (l16 & 3);
                tile[(4 * grp + (v & 3)) * 65 + q] = acc[k][v];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int ch = NM * c + k;                             // (wave-uniform)
                if (ch < p.C && oin) p.out_color[opid + (size_t)ch * HW] = tile[c * 65 + lane] + Tq * p.bg[ch];
            }
        }
    }
    // Gradient rows per instance for the backward (it used to be a launch of its own, 40 us between the loss kernels and
    // K7): a row exists per set bit of an entry's touch word, and the word is complete once all four quads of the tile
    // are done.  Each wave announces its end in LDS; the LAST one counts the bits of the prefix of the list that any quad
    // staged and scatters the counts (bytes) to the emission indices -- memory work of one wave beside a chip that is
    // busy issuing vector instructions.  Entries behind `covered` of every quad keep the zero finalize_bins wrote.
    if (SAVE && p.slot_cnt) {
        if (lane == 0) s_cov[wave] = (uint32_t)covered;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // touch bytes and s_cov before the announcement
        uint32_t before = 0;
        if (lane == 0) before = atomicAdd(&s_waves_done, 1u);
        before = (uint32_t)__builtin_amdgcn_readfirstlane((int)before);
        if (before == RF_WAVES - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const uint32_t c0 = s_cov[0], c1 = s_cov[1], c2 = s_cov[2], c3 = s_cov[3];
            const uint32_t walked = max(max(c0, c1), max(c2, c3));
            const uint32_t* words = reinterpret_cast<const uint32_t*>(p.touch) + r0;
            const uint32_t* rows = p.inst_row + r0;
            for (uint32_t pos0 = 0; pos0 < walked; pos0 += 256) {
                uint32_t w[4], e[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t pos = pos0 + 64 * u + lane;
                    w[u] = 0; e[u] = 0;
                    if (pos < walked) { w[u] = words[pos]; e[u] = rows[pos]; }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t pos = pos0 + 64 * u + lane;
                    // a quad's byte means something only below that quad's `covered`
                    const uint32_t m = (pos < c0 ? 0x0000000Fu : 0u) | (pos < c1 ? 0x00000F00u : 0u) |
                                       (pos < c2 ? 0x000F0000u : 0u) | (pos < c3 ? 0x0F000000u : 0u);
                    if (pos < walked) p.slot_cnt[e[u]] = (uint8_t)__popc(w[u] & m);
                }
            }
        }
    }
#ifdef GSR_DEV_PROBES
    if (PROBE == 7) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), t1 = __builtin_amdgcn_s_memrealtime();
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const uint32_t w = blockIdx.x * 4u + (uint32_t)wave;
        if (lane == 0 && w < RF_STAMP_WAVES) {
            g_rf_stamps[RF_STAMP_WORDS * w] = stamp_t0; g_rf_stamps[RF_STAMP_WORDS * w + 1] = t1;
            g_rf_stamps[RF_STAMP_WORDS * w + 2] = ((unsigned long long)xcc << 32) | hw;
            g_rf_stamps[RF_STAMP_WORDS * w + 3] = stamp_c0; g_rf_stamps[RF_STAMP_WORDS * w + 4] = c1;
        }
    }
#endif
}

int gsr_launch_render_fwd(const GsrView& v, const uint32_t* ranges, const float* splat,
                          float* final_T, uint32_t* n_contrib, float* out_color,
                          float* out_allmap, uint8_t* touch, uint32_t* covered, const uint32_t* inst_row, uint8_t* slot_cnt,
                          const float* feat, const uint32_t* point_list, hipStream_t s) {
    RenderFwdParams p;
    p.W = v.width; p.H = v.height; p.gx = (v.width + GSR_TILE - 1) / GSR_TILE; p.flags = v.flags;
    const int gy = (v.height + GSR_TILE - 1) / GSR_TILE;
    p.ranges = ranges; p.splat = reinterpret_cast<const float4*>(splat); p.bg = v.bg;
    p.final_T = final_T; p.n_contrib = n_contrib; p.out_color = out_color; p.out_allmap = out_allmap;
    p.touch = touch; p.covered = covered; p.inst_row = inst_row; p.slot_cnt = slot_cnt; p.feat = feat; p.point_list = point_list; p.C = v.channels;
    if (p.gx <= 0 || gy <= 0) return GSR_OK;
    GsrProfileScope prof(GSR_K_RENDER_FWD, s);
    p.n_tiles = p.gx * gy; p.per_xcd = (p.n_tiles + 7) / 8;
    const dim3 grid(8 * p.per_xcd), block(RF_BLOCK);
    if (feat == nullptr) {
        if (v.flags & (uint32_t)GSR_FLAG_FORWARD_ONLY) hipLaunchKernelGGL((render_fwd_kernel<0, false>), grid, block, 0, s, p);
        else if (v.flags & (uint32_t)GSR_FLAG_COLOR_ONLY) hipLaunchKernelGGL((render_fwd_kernel<0, true, 0, 0>), grid, block, 0, s, p);
        else if (v.flags & (uint32_t)GSR_FLAG_NO_DIST_MEDIAN) hipLaunchKernelGGL((render_fwd_kernel<0, true, 0, 1>), grid, block, 0, s, p);
        else {
#ifdef GSR_DEV_PROBES
            const char* e = getenv("GSR_K6_PROBE");   // re-read per launch
            switch (e ? atoi(e) : 0) {
                case 1: hipLaunchKernelGGL((render_fwd_kernel<0, true, 1>), grid, block, 0, s, p); break;
                case 2: hipLaunchKernelGGL((render_fwd_kernel<0, true, 2>), grid, block, 0, s, p); break;
                case 3: hipLaunchKernelGGL((render_fwd_kernel<0, true, 3>), grid, block, 0, s, p); break;
                case 4: hipLaunchKernelGGL((render_fwd_kernel<0, true, 4>), grid, block, 0, s, p); break;
                case 5: hipLaunchKernelGGL((render_fwd_kernel<0, true, 5>), grid, block, 0, s, p); break;
                case 7: hipLaunchKernelGGL((render_fwd_kernel<0, true, 7>), grid, block, 0, s, p); break;
                default: hipLaunchKernelGGL((render_fwd_kernel<0, true>), grid, block, 0, s, p);
            }
#else
            hipLaunchKernelGGL((render_fwd_kernel<0, true>), grid, block, 0, s, p);
#endif
        }
    } else {
        switch ((v.channels + 15) / 16) {   // (wide payloads always keep the backward state)
            case 1: hipLaunchKernelGGL((render_fwd_kernel<1, true>), grid, block, 0, s, p); break;
            case 2: hipLaunchKernelGGL((render_fwd_kernel<2, true>), grid, block, 0, s, p); break;
            case 3: hipLaunchKernelGGL((render_fwd_kernel<3, true>), grid, block, 0, s, p); break;
            case 4: hipLaunchKernelGGL((render_fwd_kernel<4, true>), grid, block, 0, s, p); break;
            default: gsr_set_error("wide payload supports at most 64 channels"); return GSR_E_UNSUPPORTED;
        }
    }
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}
