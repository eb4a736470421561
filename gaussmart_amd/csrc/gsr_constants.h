// Every [U] constant of the rasterizer in ONE place (SURVEY.md provenance tag [U]: recalled
// behaviour of the un-vendored upstream submodule; correct here if upstream is ever observed).
// Mirrored by oracle/surfel_ref.py.
#pragma once

#define GSR_TILE 16                    // BLOCK_X = BLOCK_Y
#define GSR_TILE_PIXELS 256
#define GSR_NEAR_N 0.2f                // frustum cull (z <= near) and per-pixel depth reject
#define GSR_FAR_N 100.0f               // far plane of the distortion depth mapping
#define GSR_CUTOFF 3.0f                // AABB cutoff in sigmas
#define GSR_FILTER_SIZE 0.707106f      // low-pass filter radius literal
#define GSR_FILTER_INV_SQUARE 2.0f
#define GSR_ALPHA_MAX 0.99f
#define GSR_ALPHA_MIN (1.0f / 255.0f)
#define GSR_T_EPS 0.0001f
#define GSR_AABB_MIN_EXT2 1e-4f

#define GSR_SH_C0 0.28209479177387814f
#define GSR_SH_C1 0.4886025119029199f
#define GSR_SH_C2_0 1.0925484305920792f
#define GSR_SH_C2_1 -1.0925484305920792f
#define GSR_SH_C2_2 0.31539156525252005f
#define GSR_SH_C2_3 -1.0925484305920792f
#define GSR_SH_C2_4 0.5462742152960396f
#define GSR_SH_C3_0 -0.5900435899266435f
#define GSR_SH_C3_1 2.890611442640554f
#define GSR_SH_C3_2 -0.4570457994644658f
#define GSR_SH_C3_3 0.3731763325901154f
#define GSR_SH_C3_4 -0.4570457994644658f
#define GSR_SH_C3_5 1.445305721320277f
#define GSR_SH_C3_6 -0.5900435899266435f
