// Dense Adam step over all parameter tensors of the Gaussian model in ONE launch.
// Same update as torch.optim.Adam (the reference's optimiser: scene/gaussian_model.py:282-295,
// eps = 1e-15, betas (0.9, 0.999), no weight decay, no amsgrad):
//     m <- m + (g - m) (1 - b1);  v <- b2 v + (1 - b2) g^2
//     p <- p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// Dense on purpose: Gaussians outside the current view still decay their moments and move by
// momentum, exactly as with the reference's optimiser.
// Pure HBM streaming: 16 B read + 12 B written per parameter (1.62 GB at 1M Gaussians).
#include "gsr_common.h"
#include "sh_basis.h"

#define AD_BLOCK 256
typedef float ad_f4 __attribute__((ext_vector_type(4)));   // the non-temporal builtins want a native vector type
__device__ __forceinline__ float4 nt_load4(const float* base, long long i) {
    const ad_f4 t = __builtin_nontemporal_load(reinterpret_cast<const ad_f4*>(base) + i);
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void nt_store4(float* base, long long i, const float4 v) {
    ad_f4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<ad_f4*>(base) + i);
}
#define AD_MAX_TENSORS 8

struct AdamBatch {
    int count;
    float beta1, beta2, omb1, omb2, eps;   // 1 - beta is formed in double on the host, like torch does
    float* p[AD_MAX_TENSORS]; const float* g[AD_MAX_TENSORS]; float* m[AD_MAX_TENSORS]; float* v[AD_MAX_TENSORS];
    long long n[AD_MAX_TENSORS];
    float step_size[AD_MAX_TENSORS];     // lr / (1 - beta1^t)
    float inv_bc2_sqrt[AD_MAX_TENSORS];  // 1 / sqrt(1 - beta2^t)
    float* old[AD_MAX_TENSORS];          // NULL, or where the parameter's value BEFORE the update is kept (gsr_adam_step_keep)
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float omb1, float b2, float omb2,
                                         float eps, float step_size, float inv_bc2_sqrt) {
    m = m + (g - m) * omb1;
    v = v * b2 + omb2 * g * g;
    const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

__global__ void __launch_bounds__(AD_BLOCK) adam_kernel(AdamBatch b) {
    const int t = blockIdx.y;
    if (t >= b.count) return;
    float* __restrict__ p = b.p[t]; const float* __restrict__ g = b.g[t];
    float* __restrict__ m = b.m[t]; float* __restrict__ v = b.v[t];
    float* __restrict__ old = b.old[t];
    const long long n = b.n[t];
    const float ss = b.step_size[t], ib = b.inv_bc2_sqrt[t];
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) |
                       reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    const long long n4 = vec ? n / 4 : 0;
    const long long stride = (long long)gridDim.x * AD_BLOCK;
    for (long long i = (long long)blockIdx.x * AD_BLOCK + threadIdx.x; i < n4; i += stride) {
        // gradients and moments are touched once per step: non-temporal, so that they do not evict the parameters
        // (which the next forward reads) from the 256 MB Infinity Cache
        float4 pp = reinterpret_cast<float4*>(p)[i];
        if (old) reinterpret_cast<float4*>(old)[i] = pp;
        const float4 gg = nt_load4(g, i);
        float4 mm = nt_load4(m, i), vv = nt_load4(v, i);
        adam_one(pp.x, gg.x, mm.x, vv.x, b.omb1, b.beta2, b.omb2, b.eps, ss, ib);
        adam_one(pp.y, gg.y, mm.y, vv.y, b.omb1, b.beta2, b.omb2, b.eps, ss, ib);
        adam_one(pp.z, gg.z, mm.z, vv.z, b.omb1, b.beta2, b.omb2, b.eps, ss, ib);
        adam_one(pp.w, gg.w, mm.w, vv.w, b.omb1, b.beta2, b.omb2, b.eps, ss, ib);
        reinterpret_cast<float4*>(p)[i] = pp;
        nt_store4(m, i, mm);
        nt_store4(v, i, vv);
    }
    for (long long i = n4 * 4 + (long long)blockIdx.x * AD_BLOCK + threadIdx.x; i < n; i += stride) {
        float pp = p[i], mm = m[i], vv = v[i];
        if (old) old[i] = pp;
        adam_one(pp, g[i], mm, vv, b.omb1, b.beta2, b.omb2, b.eps, ss, ib);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

extern "C" int32_t gsr_adam_step(int32_t count, float* const* params, const float* const* grads,
                                 float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel,
                                 const float* step_size, const float* inv_bc2_sqrt, double beta1, double beta2,
                                 double eps, gsr_stream_t stream_) {
    return gsr_adam_step_keep(count, params, grads, exp_avg, exp_avg_sq, numel, step_size, inv_bc2_sqrt, beta1, beta2, eps,
                              nullptr, stream_);
}

extern "C" int32_t gsr_adam_step_keep(int32_t count, float* const* params, const float* const* grads,
                                      float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel,
                                      const float* step_size, const float* inv_bc2_sqrt, double beta1, double beta2,
                                      double eps, float* const* keep_old, gsr_stream_t stream_) {
    if (count < 0 || count > AD_MAX_TENSORS) { gsr_set_error("adam: at most %d tensors per call", AD_MAX_TENSORS); return GSR_E_INVALID; }
    if (count == 0) return GSR_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !step_size || !inv_bc2_sqrt) { gsr_set_error("adam: null argument"); return GSR_E_INVALID; }
    AdamBatch b;
    b.count = count; b.beta1 = (float)beta1; b.beta2 = (float)beta2; b.eps = (float)eps;
    b.omb1 = (float)(1.0 - beta1); b.omb2 = (float)(1.0 - beta2);
    long long max_n = 0;
    for (int i = 0; i < count; ++i) {
        if (numel[i] < 0 || (numel[i] > 0 && (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i]))) {
            gsr_set_error("adam: bad tensor %d", i);
            return GSR_E_INVALID;
        }
        b.p[i] = params[i]; b.g[i] = grads[i]; b.m[i] = exp_avg[i]; b.v[i] = exp_avg_sq[i]; b.n[i] = numel[i];
        b.step_size[i] = step_size[i]; b.inv_bc2_sqrt[i] = inv_bc2_sqrt[i];
        b.old[i] = keep_old ? keep_old[i] : nullptr;
        if (numel[i] > max_n) max_n = numel[i];
    }
    if (max_n == 0) return GSR_OK;
    hipStream_t s = static_cast<hipStream_t>(stream_);
    GsrProfileScope prof(GSR_K_ADAM, s);
    long long blocks = (max_n / 4 + AD_BLOCK - 1) / AD_BLOCK;
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;      // grid-stride beyond 16 blocks per CU
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks, (unsigned)count), dim3(AD_BLOCK), 0, s, b);
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Adam step of the SH tensors from FACTORED gradients (include/gsr.h: gsr_adam_sh_factored).
// One wave64 owns 64 consecutive Gaussians:
//   1. lane = Gaussian: grad[k][c] = scale * sum_views basis_k(dir_view) * g_view[c] in 48 registers, written to a
//      wave-private LDS tile (rows of 49 floats: lane-per-row writes and element-order reads are conflict free);
//   2. the wave streams its contiguous 64 x (M-1) x 3 block of features_rest (parameter, two moments) with 16-byte
//      accesses, three vectors of each array in flight per lane, taking the gradient of every element from the tile;
//   3. the same for the 64 x 3 block of features_dc.
// No workgroup barrier (the tile is wave-private).  HBM traffic: 28 B per parameter minus the 4 B gradient read,
// plus 12 B x (1 + n_views) per Gaussian.
#define AS_BLOCK 256
#define AS_ROW 49
#define AS_MAX_VIEWS 16

struct AdamShParams {
    int first, count, M, deg, n_views, campos_stride;
    long long view_stride;
    float grad_scale;
    const float* xyz; const float* cg; const float* campos;
    float* p_dc; float* m_dc; float* v_dc; float* p_rest; float* m_rest; float* v_rest;
    float ss_dc, ib_dc, ss_rest, ib_rest;
    float beta2, omb1, omb2, eps;
    // gsr_adam_sh_factored_next: colour of the NEXT view from the coefficients just updated (NULL: plain step)
    const float* xyz_next; const float* campos_next; int deg_next; int n_total; float* cache;
};

// Adam over the wave's n_f contiguous floats of one tensor starting at float offset `off`; element e belongs to tile
// row e / row_f, column col0 + e % row_f.
// KEEP: the updated parameter replaces the gradient in the tile, which then holds the wave's new coefficients.
template <bool KEEP>
__device__ __forceinline__ void adam_sh_segment(float* wl, int col0, int row_f, float* __restrict__ p,
                                                float* __restrict__ m, float* __restrict__ v, long long off, int n_f,
                                                float ss, float ib, const AdamShParams& a, int lane) {
    p += off; m += off; v += off;
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    const int n4 = vec ? n_f >> 2 : 0;
    constexpr int U = 3;
    for (int i0 = lane; i0 < n4; i0 += 64 * U) {
        float4 pp[U], mm[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + 64 * u;
            if (i < n4) { pp[u] = reinterpret_cast<float4*>(p)[i]; mm[u] = nt_load4(m, i); vv[u] = nt_load4(v, i); }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + 64 * u;
            if (i < n4) {
                const int e = i << 2;
                int r = e / row_f, c = e - r * row_f;
                float* P = reinterpret_cast<float*>(&pp[u]);
                float* Mo = reinterpret_cast<float*>(&mm[u]);
                float* V = reinterpret_cast<float*>(&vv[u]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float g = wl[r * AS_ROW + col0 + c];
                    adam_one(P[k], g, Mo[k], V[k], a.omb1, a.beta2, a.omb2, a.eps, ss, ib);
                    if (KEEP) wl[r * AS_ROW + col0 + c] = P[k];
                    if (++c == row_f) { c = 0; ++r; }
                }
                reinterpret_cast<float4*>(p)[i] = pp[u];
                nt_store4(m, i, mm[u]);
                nt_store4(v, i, vv[u]);
            }
        }
    }
    for (int e = (n4 << 2) + lane; e < n_f; e += 64) {
        const int r = e / row_f, c = e - r * row_f;
        float pp = p[e], mm = m[e], vv = v[e];
        adam_one(pp, wl[r * AS_ROW + col0 + c], mm, vv, a.omb1, a.beta2, a.omb2, a.eps, ss, ib);
        if (KEEP) wl[r * AS_ROW + col0 + c] = pp;
        p[e] = pp; m[e] = mm; v[e] = vv;
    }
}

// NEXT: after the update the wave evaluates, from the new coefficients still in its tile, the SH colour of the NEXT view
// for its 64 Gaussians -- rgb (+0.5, clamped at 0), the clamp mask and d(rgb)/d(dir) -- into the colour cache the next
// forward / backward read (GSR_FLAG_COLOR_CACHED): the 192 bytes per Gaussian the colour pass would read again are already
// here.  Same formulas as preprocess_color16_kernel / utils/sh_utils.py:57-112.
template <bool NEXT>
__global__ void __launch_bounds__(AS_BLOCK) adam_sh_factored_kernel(AdamShParams a) {
    __shared__ float tile[AS_BLOCK / 64][64 * AS_ROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rel = (blockIdx.x * (AS_BLOCK / 64) + wave) * 64;      // first Gaussian of this wave, relative to a.first
    const int n_here = min(64, a.count - rel);
    if (n_here <= 0) return;                                          // wave-uniform; no workgroup barrier below
    const int wave_first = a.first + rel;
    float* wl = tile[wave];

    if (lane < n_here) {
        const int idx = wave_first + lane;
        const float px = a.xyz[3 * (size_t)idx + 0], py = a.xyz[3 * (size_t)idx + 1], pz = a.xyz[3 * (size_t)idx + 2];
        float acc[48];
#pragma unroll
        for (int j = 0; j < 48; ++j) acc[j] = 0.f;
        for (int r = 0; r < a.n_views; ++r) {
            const float* gp = a.cg + (size_t)r * a.view_stride + 3 * (size_t)idx;
            const float g0 = gp[0], g1 = gp[1], g2 = gp[2];
            if (g0 != 0.f || g1 != 0.f || g2 != 0.f) {
                const float* cp = a.campos + r * a.campos_stride;
                const float ox = px - cp[0], oy = py - cp[1], oz = pz - cp[2];
                const float il = 1.0f / sqrtf(ox * ox + oy * oy + oz * oz);
                float basis[16];
                sh_basis16(a.deg, ox * il, oy * il, oz * il, basis);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    acc[3 * k + 0] = fmaf(basis[k], g0, acc[3 * k + 0]);
                    acc[3 * k + 1] = fmaf(basis[k], g1, acc[3 * k + 1]);
                    acc[3 * k + 2] = fmaf(basis[k], g2, acc[3 * k + 2]);
                }
            }
        }
        const float sc = a.grad_scale;
#pragma unroll
        for (int j = 0; j < 48; ++j)
            if (j < 3 * a.M) wl[lane * AS_ROW + j] = acc[j] * sc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const int rest_f = (a.M - 1) * 3;
    if (rest_f > 0)
        adam_sh_segment<NEXT>(wl, 3, rest_f, a.p_rest, a.m_rest, a.v_rest, (long long)wave_first * rest_f, n_here * rest_f,
                              a.ss_rest, a.ib_rest, a, lane);
    adam_sh_segment<NEXT>(wl, 0, 3, a.p_dc, a.m_dc, a.v_dc, (long long)wave_first * 3, n_here * 3, a.ss_dc, a.ib_dc, a, lane);
    if (!NEXT) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < n_here) {
        const int idx = wave_first + lane;
        const float ox = a.xyz_next[3 * (size_t)idx + 0] - a.campos_next[0];
        const float oy = a.xyz_next[3 * (size_t)idx + 1] - a.campos_next[1];
        const float oz = a.xyz_next[3 * (size_t)idx + 2] - a.campos_next[2];
        const float il = 1.0f / sqrtf(ox * ox + oy * oy + oz * oz);
        float basis[16], dbx[16], dby[16], dbz[16];
        sh_basis16_grad(a.deg_next, ox * il, oy * il, oz * il, basis, dbx, dby, dbz);
        const float* sh = wl + lane * AS_ROW;          // [k][c] of this Gaussian, k < M
        float rgb[3] = {0.f, 0.f, 0.f}, J[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) J[q] = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (k < a.M) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float v = sh[3 * k + c];
                    rgb[c] = fmaf(basis[k], v, rgb[c]);
                    J[3 * c + 0] = fmaf(dbx[k], v, J[3 * c + 0]);
                    J[3 * c + 1] = fmaf(dby[k], v, J[3 * c + 1]);
                    J[3 * c + 2] = fmaf(dbz[k], v, J[3 * c + 2]);
                }
            }
        }
        uint32_t clamp_bits = 0;
        float* out_rgb = a.cache + 3 * (size_t)idx;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float t = rgb[c] + 0.5f;
            if (t < 0.0f) clamp_bits |= (1u << c);
            out_rgb[c] = fmaxf(t, 0.0f);
        }
        reinterpret_cast<uint32_t*>(a.cache)[3 * (size_t)a.n_total + idx] = clamp_bits;
        float* out_j = a.cache + 4 * (size_t)a.n_total + 9 * (size_t)idx;
#pragma unroll
        for (int q = 0; q < 9; ++q) out_j[q] = J[q];
    }
}

static int adam_sh_factored_impl(int32_t first, int32_t count, int32_t sh_coeffs, int32_t sh_degree, const float* xyz,
                                 int32_t n_views, const float* color_grad, int64_t view_stride, const float* campos,
                                 int32_t campos_stride, float grad_scale,
                                 float* p_dc, float* m_dc, float* v_dc, float step_size_dc, float inv_bc2_sqrt_dc,
                                 float* p_rest, float* m_rest, float* v_rest, float step_size_rest,
                                 float inv_bc2_sqrt_rest, double beta1, double beta2, double eps,
                                 const float* xyz_next, const float* campos_next, int32_t sh_degree_next, int32_t n_total,
                                 float* color_cache, gsr_stream_t stream_) {
    if (first < 0 || count < 0 || sh_coeffs < 1 || sh_coeffs > 16 || sh_degree < 0 || sh_degree > 3 ||
        n_views < 1 || n_views > AS_MAX_VIEWS || campos_stride < 3 || view_stride < 0) {
        gsr_set_error("adam_sh_factored: bad sizes (coeffs 1..16, degree 0..3, views 1..%d)", AS_MAX_VIEWS);
        return GSR_E_INVALID;
    }
    if (color_cache && (!xyz_next || !campos_next || sh_degree_next < 0 || sh_degree_next > 3 || n_total < first + count)) {
        gsr_set_error("adam_sh_factored_next: xyz_next / campos_next missing, degree not in 0..3 or n_total too small");
        return GSR_E_INVALID;
    }
    if (count == 0) return GSR_OK;
    if (!xyz || !color_grad || !campos || !p_dc || !m_dc || !v_dc || (sh_coeffs > 1 && (!p_rest || !m_rest || !v_rest))) {
        gsr_set_error("adam_sh_factored: null argument");
        return GSR_E_INVALID;
    }
    AdamShParams a;
    a.first = first; a.count = count; a.M = sh_coeffs; a.deg = sh_degree; a.n_views = n_views;
    a.campos_stride = campos_stride; a.view_stride = view_stride; a.grad_scale = grad_scale;
    a.xyz = xyz; a.cg = color_grad; a.campos = campos;
    a.p_dc = p_dc; a.m_dc = m_dc; a.v_dc = v_dc; a.p_rest = p_rest; a.m_rest = m_rest; a.v_rest = v_rest;
    a.ss_dc = step_size_dc; a.ib_dc = inv_bc2_sqrt_dc; a.ss_rest = step_size_rest; a.ib_rest = inv_bc2_sqrt_rest;
    a.beta2 = (float)beta2; a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2); a.eps = (float)eps;
    a.xyz_next = xyz_next; a.campos_next = campos_next; a.deg_next = sh_degree_next; a.n_total = n_total; a.cache = color_cache;
    hipStream_t s = static_cast<hipStream_t>(stream_);
    GsrProfileScope prof(GSR_K_ADAM, s);
    const unsigned blocks = (unsigned)((count + AS_BLOCK - 1) / AS_BLOCK);
    if (color_cache) hipLaunchKernelGGL(adam_sh_factored_kernel<true>, dim3(blocks), dim3(AS_BLOCK), 0, s, a);
    else hipLaunchKernelGGL(adam_sh_factored_kernel<false>, dim3(blocks), dim3(AS_BLOCK), 0, s, a);
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}

extern "C" int32_t gsr_adam_sh_factored(int32_t first, int32_t count, int32_t sh_coeffs, int32_t sh_degree, const float* xyz,
                                        int32_t n_views, const float* color_grad, int64_t view_stride, const float* campos,
                                        int32_t campos_stride, float grad_scale,
                                        float* p_dc, float* m_dc, float* v_dc, float step_size_dc, float inv_bc2_sqrt_dc,
                                        float* p_rest, float* m_rest, float* v_rest, float step_size_rest,
                                        float inv_bc2_sqrt_rest, double beta1, double beta2, double eps,
                                        gsr_stream_t stream_) {
    return adam_sh_factored_impl(first, count, sh_coeffs, sh_degree, xyz, n_views, color_grad, view_stride, campos, campos_stride,
                                 grad_scale, p_dc, m_dc, v_dc, step_size_dc, inv_bc2_sqrt_dc, p_rest, m_rest, v_rest,
                                 step_size_rest, inv_bc2_sqrt_rest, beta1, beta2, eps, nullptr, nullptr, 0, 0, nullptr, stream_);
}

extern "C" int32_t gsr_adam_sh_factored_next(int32_t first, int32_t count, int32_t sh_coeffs, int32_t sh_degree, const float* xyz,
                                             int32_t n_views, const float* color_grad, int64_t view_stride, const float* campos,
                                             int32_t campos_stride, float grad_scale,
                                             float* p_dc, float* m_dc, float* v_dc, float step_size_dc, float inv_bc2_sqrt_dc,
                                             float* p_rest, float* m_rest, float* v_rest, float step_size_rest,
                                             float inv_bc2_sqrt_rest, double beta1, double beta2, double eps,
                                             const float* xyz_next, const float* campos_next, int32_t sh_degree_next,
                                             int32_t n_total, float* color_cache, gsr_stream_t stream_) {
    if (!color_cache) { gsr_set_error("adam_sh_factored_next: color_cache missing"); return GSR_E_INVALID; }
    return adam_sh_factored_impl(first, count, sh_coeffs, sh_degree, xyz, n_views, color_grad, view_stride, campos, campos_stride,
                                 grad_scale, p_dc, m_dc, v_dc, step_size_dc, inv_bc2_sqrt_dc, p_rest, m_rest, v_rest,
                                 step_size_rest, inv_bc2_sqrt_rest, beta1, beta2, eps, xyz_next, campos_next, sh_degree_next,
                                 n_total, color_cache, stream_);
}
