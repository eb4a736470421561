// The (pixel, splat) evaluation shared by render_fwd and render_bwd.  Both kernels must take the
// SAME skip decisions for every pair (the backward replays the forward's traversal), so every
// multiply-add is an explicit fmaf: no contraction choice is left to the compiler.
// Math: 2DGS eq. 8-11 (ray-splat intersection through two homogeneous planes, low-pass filter).
#pragma once
#include "gsr_common.h"

struct GsrPair {
    float kx, ky, kz, lx, ly, lz;   // planes k = px*Tw - Tu, l = py*Tw - Tv
    float inv_pz;                   // 1 / (k x l).z  (times 2^-64 on a lane with `tiny` set)
    float sx, sy;                   // intersection in splat coordinates
    float dx, dy;                   // AABB centre - pixel
    float depth, G, araw, alpha;
    bool use3d;
    bool tiny;                      // |(k x l).z| below the smallest normal number: inv_pz carries the factor 2^-64
    bool tiny_any;                  // ... on any lane of the wave (wave-uniform: a scalar branch for the callers)
};

// v_rcp_f32 flushes a denormal operand: 1 / pz reads inf, s = p / pz reads inf or NaN, and the one term of the backward that
// uses s although the low-pass filter won (the reference's `s.x * dL_dz`, GSR_FLAG_FILTER_DEPTH_GRAD) turns the gradient
// row of that Gaussian into NaN, where the reference's IEEE division p.x / p.z is finite (~1e20).  A surfel whose two scales
// have collapsed to ~1e-21 passes through this band on its way to pz == 0 (profiles/r04_notes/collapsed_surfel.md: the
// 30,000-iteration schedule at the headline shape met one at iteration 24,079).  Such a lane divides through the scaled
// reciprocal instead: pz 2^64 is a normal number, the factor is exact, and nothing changes for any other pair.
#define GSR_TINY_PZ_SCALE 0x1p64f
#define GSR_FLT_MIN_NORMAL 0x1p-126f

// Returns false when the pair is skipped before the transmittance test.
// The rare rejects (degenerate intersection, depth behind the near plane) are
// folded into ONE predicate instead of early exits: on a wave that saves three exec-mask
// save/branch/restore sequences per splat, and a lane that fails an early test is invalid whatever
// garbage its later values hold, so the decisions are exactly those of the sequential tests.
__device__ __forceinline__ bool gsr_pair_eval(float pxf, float pyf, const float4 a0, const float4 a1,
                                              const float4 a2, float opa, GsrPair& o) {
    const float Tux = a0.x, Tuy = a0.y, Tuz = a0.z, Tvx = a0.w, Tvy = a1.x, Tvz = a1.y;
    const float Twx = a1.z, Twy = a1.w, Twz = a2.x, cx = a2.y, cy = a2.z;
    o.kx = fmaf(pxf, Twx, -Tux); o.ky = fmaf(pxf, Twy, -Tuy); o.kz = fmaf(pxf, Twz, -Tuz);
    o.lx = fmaf(pyf, Twx, -Tvx); o.ly = fmaf(pyf, Twy, -Tvy); o.lz = fmaf(pyf, Twz, -Tvz);
    const float ppx = fmaf(o.ky, o.lz, -(o.kz * o.ly));
    const float ppy = fmaf(o.kz, o.lx, -(o.kx * o.lz));
    const float ppz = fmaf(o.kx, o.ly, -(o.ky * o.lx));
    bool valid = true;
    o.inv_pz = gsr_rcp(ppz);
    o.sx = ppx * o.inv_pz; o.sy = ppy * o.inv_pz;
    o.tiny = !(fabsf(ppz) >= GSR_FLT_MIN_NORMAL);             // zero, denormal (or NaN): the same one compare `pz != 0` cost
    o.tiny_any = __builtin_amdgcn_ballot_w64(o.tiny) != 0ull;
    if (__builtin_expect(o.tiny_any, 0)) {
        const float r = gsr_rcp(ppz * GSR_TINY_PZ_SCALE);
        valid = !(ppz == 0.0f);
        o.inv_pz = o.tiny ? r : o.inv_pz;
        o.sx = o.tiny ? (ppx * r) * GSR_TINY_PZ_SCALE : o.sx;
        o.sy = o.tiny ? (ppy * r) * GSR_TINY_PZ_SCALE : o.sy;
    }
    const float rho3d = fmaf(o.sx, o.sx, o.sy * o.sy);
    o.dx = cx - pxf; o.dy = cy - pyf;
    const float rho2d = GSR_FILTER_INV_SQUARE * fmaf(o.dx, o.dx, o.dy * o.dy);
    o.use3d = rho3d <= rho2d;
    const float rho = fminf(rho3d, rho2d);
    o.depth = o.use3d ? fmaf(o.sx, Twx, fmaf(o.sy, Twy, Twz)) : Twz;
    valid = valid && !(o.depth < GSR_NEAR_N);
    // (the reference's `power > 0` reject cannot fire here: rho is the smaller of two sums of squares -- fminf drops a NaN --
    // so the power is never positive, and the test would only cost a compare per pair)
    const float power = -0.5f * rho;
    o.G = __expf(power);
    o.araw = opa * o.G;
    o.alpha = fminf(GSR_ALPHA_MAX, o.araw);
    return valid && o.alpha >= GSR_ALPHA_MIN;
}

// distortion depth mapping m(z) and its derivative
__device__ __forceinline__ float gsr_depth_map(float z, float& dm_dz) {
    const float iz = gsr_rcp(z);
    dm_dz = (GSR_FAR_N * GSR_NEAR_N) / (GSR_FAR_N - GSR_NEAR_N) * iz * iz;
    return GSR_FAR_N / (GSR_FAR_N - GSR_NEAR_N) * (1.0f - GSR_NEAR_N * iz);
}
