// K6 render_fwd: one 256-thread workgroup per 16x16 tile, front-to-back alpha compositing of
// 2D-Gaussian surfels with the depth / normal / median-depth / distortion side outputs.
// Restates the [U] forward render; output channel order pinned by
// gaussian_renderer/__init__.py:117-141 (allmap = depth, alpha, normal xyz, median depth, dist).
//
// MI355X mapping: the tile's 4 wave64s each own an 8x8 pixel quad (compact footprint => a small
// splat touches few waves and whole-wave skips are common).  Per round the workgroup gathers up to
// 256 splat records (80 B, five 16-byte loads per thread) into LDS; the inner loop reads them with
// wave-uniform (broadcast) ds_read_b128.  Per 64 staged splats each lane tests ONE splat's cull rect
// against the wave's quad and a 64-bit ballot becomes the list of splats the wave has to look at at
// all (s_ff1 iteration): splats that cannot reach alpha >= 1/255 in the quad cost nothing.  A wave
// whose 64 pixels are all done leaves the round; the workgroup leaves once every pixel is done.
#include "gsr_common.h"
#include "pair_eval.h"

#define RF_BLOCK 256
#define RF_BATCH 256

struct RenderFwdParams {
    int W, H, gx;
    uint32_t flags;
    const uint32_t* ranges; const uint32_t* point_list; const float* splat;
    const float* bg;
    float* final_T; uint32_t* n_contrib; float* out_color; float* out_allmap;
};

__global__ void __launch_bounds__(RF_BLOCK) render_fwd_kernel(RenderFwdParams p) {
    __shared__ float4 s_rec[RF_BATCH * 5];

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
    int todo = (int)(r1 - r0);
    const int rounds = (todo + RF_BATCH - 1) / RF_BATCH;

    bool done = !inside;
    float T = 1.0f;
    uint32_t last_contributor = 0;
    const bool no_cull = (p.flags & (uint32_t)GSR_FLAG_DEBUG_NO_CULL) != 0;
    const int qx0 = tile_x * GSR_TILE + (wave & 1) * 8, qy0 = tile_y * GSR_TILE + (wave >> 1) * 8;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f;
    float N0 = 0.f, N1 = 0.f, N2 = 0.f;
    float Dacc = 0.f, M1 = 0.f, M2 = 0.f, dist = 0.f, med_depth = 0.f;
    uint32_t med_contrib = 0xFFFFFFFFu;   // "-1" stored in the u32 plane, as recalled

    for (int rd = 0; rd < rounds; ++rd, todo -= RF_BATCH) {
        if (__syncthreads_and(done)) break;
        const int progress = rd * RF_BATCH + tid;
        if (r0 + progress < r1) {
            const uint32_t gid = p.point_list[r0 + progress];
            const float4* src = reinterpret_cast<const float4*>(p.splat + (size_t)gid * GSR_SPLAT_FLOATS);
#pragma unroll
            for (int q = 0; q < 5; ++q) s_rec[tid * 5 + q] = src[q];
        }
        __syncthreads();
        const int nb = min(RF_BATCH, todo);
        bool wave_done = false;
        for (int jbase = 0; jbase < nb && !wave_done; jbase += 64) {
            // one ballot per 64 staged splats: which of them can reach alpha >= 1/255 inside this
            // wave's 8x8 pixel quad at all (conservative cull rect from preprocess_fwd)?
            const int jj = jbase + lane;
            bool ov = false;
            if (jj < nb) {
                const float4 r4 = s_rec[jj * 5 + 4];
                ov = no_cull || gsr_rect_overlaps_quad(__float_as_uint(r4.z), __float_as_uint(r4.w), qx0, qy0);
            }
            unsigned long long m = __ballot(ov);
            while (m) {
                // every lane is active here (the loop is wave-uniform), so the vote sees the whole wave
                if (__all(done)) { wave_done = true; break; }
                const int j = jbase + __builtin_ctzll(m);
                m &= m - 1;
                if (done) continue;
                const uint32_t contributor = (uint32_t)(rd * RF_BATCH + j + 1);   // 1-based list position
                const float4 a0 = s_rec[j * 5 + 0], a1 = s_rec[j * 5 + 1], a2 = s_rec[j * 5 + 2];
                const float4 a3 = s_rec[j * 5 + 3];
                GsrPair pr;
                if (!gsr_pair_eval(pxf, pyf, a0, a1, a2, a3.z, pr)) continue;
                const float alpha = pr.alpha, depth = pr.depth;
                const float test_T = T * (1.0f - alpha);
                if (test_T < GSR_T_EPS) { done = true; continue; }
                const float4 a4 = s_rec[j * 5 + 4];
                const float w = alpha * T;
                const float A = 1.0f - T;
                float dm_dz_unused;
                const float m_d = gsr_depth_map(depth, dm_dz_unused);
                dist += (m_d * m_d * A + M2 - 2.0f * m_d * M1) * w;
                Dacc += depth * w;
                M1 += m_d * w;
                M2 += m_d * m_d * w;
                if (T > 0.5f) { med_depth = depth; med_contrib = contributor; }
                N0 += a2.w * w; N1 += a3.x * w; N2 += a3.y * w;
                C0 += a3.w * w; C1 += a4.x * w; C2 += a4.y * w;
                T = test_T;
                last_contributor = contributor;
            }
        }
    }

    if (inside) {
        p.final_T[pix_id] = T;
        p.final_T[pix_id + HW] = M1;
        p.final_T[pix_id + 2 * HW] = M2;
        p.n_contrib[pix_id] = last_contributor;
        p.n_contrib[pix_id + HW] = med_contrib;
        p.out_color[pix_id] = C0 + T * p.bg[0];
        p.out_color[pix_id + HW] = C1 + T * p.bg[1];
        p.out_color[pix_id + 2 * HW] = C2 + T * p.bg[2];
        p.out_allmap[pix_id + 0 * HW] = Dacc;
        p.out_allmap[pix_id + 1 * HW] = 1.0f - T;
        p.out_allmap[pix_id + 2 * HW] = N0;
        p.out_allmap[pix_id + 3 * HW] = N1;
        p.out_allmap[pix_id + 4 * HW] = N2;
        p.out_allmap[pix_id + 5 * HW] = med_depth;
        p.out_allmap[pix_id + 6 * HW] = dist;
    }
}

int gsr_launch_render_fwd(const GsrView& v, const uint32_t* ranges, const uint32_t* point_list,
                          const float* splat, float* final_T, uint32_t* n_contrib,
                          float* out_color, float* out_allmap, hipStream_t s) {
    RenderFwdParams p;
    p.W = v.width; p.H = v.height; p.gx = (v.width + GSR_TILE - 1) / GSR_TILE; p.flags = v.flags;
    const int gy = (v.height + GSR_TILE - 1) / GSR_TILE;
    p.ranges = ranges; p.point_list = point_list; p.splat = splat; p.bg = v.bg;
    p.final_T = final_T; p.n_contrib = n_contrib; p.out_color = out_color; p.out_allmap = out_allmap;
    if (p.gx <= 0 || gy <= 0) return GSR_OK;
    GsrProfileScope prof(GSR_K_RENDER_FWD, s);
    hipLaunchKernelGGL(render_fwd_kernel, dim3(p.gx, gy), dim3(RF_BLOCK), 0, s, p);
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}
