// Declarations shared by render_bwd.hip (RGB payload) and render_bwd_wide.hip (wide payloads).
#pragma once
#include "gsr_common.h"

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

#define RF_BLOCK 256
#define RF_WAVES 4

#define RB_BLOCK 256
#define RB_WAVES 4
#define RB_ROW GSR_GROW_MAIN   // 16 floats: one aligned 64-byte store per block and entry

struct RenderBwdParams {
    int W, H, gx, n_tiles, per_xcd;
    uint32_t flags;
    const uint32_t* ranges; const uint32_t* covered; const uint32_t* inst_row;
    const float4* splat; const uint32_t* touch; const uint32_t* slot_off; const float* bg;
    const float* final_T; const uint32_t* n_contrib;
    const float* dL_dcolor; const float* dL_dallmap;
    float* grad_rows; float* grad_xy;   // [R][16] and [R][2]
    // side job of the launch (rb_row_begin_job): row_begin[r] = slot_off[offs[r]], r = 0..N, for reduce_rows
    const uint32_t* offs; uint32_t* row_begin; int N;
    // wide payload (FEAT16 > 0): features by Gaussian id, their gradient sub-rows [(instance*4+quad)*C + ch]
    const float* feat; const uint32_t* point_list; float* feat_rows; int C;
};

// lowest set bit of a 64-bit scalar mask (-1 if empty: s_ff1 says so itself) and its removal -- one scalar
// instruction each (s_bitset0 with index -1 clears bit 63 of a mask that is empty anyway)
__device__ __forceinline__ int rb_take_first(unsigned long long& m) {
    int j;
    asm("s_ff1_i32_b64 %0, %1\n\ts_bitset0_b64 %1, %0" : "=&s"(j), "+s"(m));
    return j;
}
__device__ __forceinline__ uint32_t rb_pack16(int lo, int hi) {
    uint32_t r;
    asm("s_pack_ll_b32_b16 %0, %1, %2" : "=s"(r) : "s"(lo), "s"(hi));
    return r;
}

// First gradient row of every depth rank, row_begin[r] = slot_off[offs[r]] (r = 0..N): reduce_rows, the next kernel on the
// stream, then starts its row reads after ONE dependent load instead of two (it is bound by that chain, not by bytes).
// The render_bwd launch is the first kernel that sees the finished slot_off; every thread takes a few ranks before its
// own work.
__device__ __forceinline__ void rb_row_begin_job(const RenderBwdParams& p) {
    const int n_threads = (int)(gridDim.x * blockDim.x);
    // four ranks per thread and round with their two dependent loads side by side: at N = 5 M / 4,056 tiles a thread
    // has five ranks, and one at a time they held every workgroup's start back by ten memory round trips
    constexpr int U = 4;
    for (long long r0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; r0 <= p.N; r0 += (long long)n_threads * U) {   // (64-bit: r0 + U n_threads may pass 2^31)
        uint32_t o[U], v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const long long r = r0 + (long long)u * n_threads; o[u] = r <= p.N ? p.offs[r] : 0u; }
#pragma unroll
        for (int u = 0; u < U; ++u) { const long long r = r0 + (long long)u * n_threads; v[u] = r <= p.N ? p.slot_off[o[u]] : 0u; }
#pragma unroll
        for (int u = 0; u < U; ++u) { const long long r = r0 + (long long)u * n_threads; if (r <= p.N) p.row_begin[r] = v[u]; }
    }
}

// touch word of list position `pos` with the bytes of the quads that never staged that entry forced to zero
__device__ __forceinline__ uint32_t rb_defined_touch(uint32_t word, uint32_t pos, const uint4 cov) {
    const uint32_t m = (pos < cov.x ? 0x0000000Fu : 0u) | (pos < cov.y ? 0x00000F00u : 0u) |
                       (pos < cov.z ? 0x000F0000u : 0u) | (pos < cov.w ? 0x0F000000u : 0u);
    return word & m;
}


// wide payloads (render_bwd_wide.hip): launches the kernel for ceil(channels / 16) 16-channel tiles on `grid`
int gsr_launch_render_bwd_wide(const RenderBwdParams& p, int channels, dim3 grid, hipStream_t s);
