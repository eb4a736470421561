// DPP-row (16-lane) reductions shared by the backward kernels (gfx950 only).
#pragma once
#include "gsr_common.h"

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float v) {
    // lanes whose source is disabled/out of range receive 0 (bound_ctrl)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
// ---- 16-lane (one DPP row) reductions: no cross-row traffic at all ------------------------------------
// Transposed butterfly over the 16 lanes of a row: 8 + 4 + 2 + 1 adds, each fused with its DPP move.
// On return lane l (0..15 inside its row) holds the ROW total of v[l].
//
// Stages "bit 3" and "bit 2" pair lanes that sit in different DPP banks (groups of 4 lanes), so the usual
// select-then-add (2 v_cndmask + 1 v_add per output) becomes two bank-masked v_add_f32_dpp writing
// complementary lanes of the same register: dst[lanes of banks M] = src(partner) + src(self).  The compiler has
// no builtin for a bank-masked fused add, hence the asm; the s_nop of the first stage covers the "VALU write -> DPP read"
// hazard (2 wait states) that the hazard recogniser cannot see across an asm boundary.
#define GSR_DPP_PAIR(dst, lo, hi, CTRL, M_LO, M_HI, NOP)                                        \
    asm volatile(NOP "v_add_f32_dpp %0, %1, %1 " CTRL " row_mask:0xf bank_mask:" M_LO "\n\t"   \
                 "v_add_f32_dpp %0, %2, %2 " CTRL " row_mask:0xf bank_mask:" M_HI               \
                 : "=&v"(dst) : "v"(lo), "v"(hi))
__device__ __forceinline__ float row_sum16_transposed(const float (&v)[16], int l16) {
    const bool b1 = (l16 & 2) != 0, b0 = (l16 & 1) != 0;
    float r[8], q[4], p[2];
    // bit 3: partner is lane ^ 8 (row_ror:8); lanes 0..7 = banks 0,1 keep v[i].  ONE asm statement for the eight pairs,
    // so the "VALU write -> DPP read" hazard (2 wait states, invisible to the hazard recogniser across an asm boundary)
    // is paid with one s_nop instead of eight.
#define GSR_P8(i) "v_add_f32_dpp %" #i ", %" #i "+8, %" #i "+8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %1, %9, %9 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %2, %10, %10 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %3, %11, %11 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %4, %12, %12 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %5, %13, %13 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %6, %14, %14 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %7, %15, %15 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %0, %16, %16 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %1, %17, %17 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %2, %18, %18 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %3, %19, %19 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %4, %20, %20 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %5, %21, %21 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %6, %22, %22 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_add_f32_dpp %7, %23, %23 row_ror:8 row_mask:0xf bank_mask:0xc"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                 : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
                   "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
#undef GSR_P8
#pragma unroll
    for (int i = 0; i < 4; ++i)        // bit 2: partner is lane ^ 7 (row_half_mirror); banks 0,2 keep r[i]
        GSR_DPP_PAIR(q[i], r[i], r[i + 4], "row_half_mirror", "0x5", "0xa", "");   // inputs written >= 6 instructions ago
#pragma unroll
    for (int i = 0; i < 2; ++i) {      // bit 1: partner is lane ^ 2 (quad_perm [2,3,0,1])
        const float keep = b1 ? q[i + 2] : q[i], send = b1 ? q[i] : q[i + 2];
        p[i] = keep + dpp_move<0x4E, 0xf>(send);
    }
    const float keep = b0 ? p[1] : p[0], send = b0 ? p[0] : p[1];
    return keep + dpp_move<0xB1, 0xf>(send);   // bit 0: partner is lane ^ 1 (quad_perm [1,0,3,2])
}
// Two values: on return lanes 0..7 of the row hold the row total of a, lanes 8..15 that of b.
__device__ __forceinline__ float row_sum2(float a, float b, int l16) {
    const bool b3 = (l16 & 8) != 0;
    const float keep = b3 ? b : a, send = b3 ? a : b;
    float v = keep + dpp_move<0x128, 0xf>(send);
    v += dpp_move<0xB1, 0xf>(v);
    v += dpp_move<0x4E, 0xf>(v);
    v += dpp_move<0x141, 0xf>(v);   // row_half_mirror: every lane of the 8 holds their sum
    return v;
}
