// K9 knn3_mean_sqdist: exact mean squared distance to the 3 nearest OTHER points.
// Replaces simple_knn._C.distCUDA2 (call site scene/gaussian_model.py:261; the caller clamps to
// >= 1e-7 and takes log(sqrt(.)) as the initial log-scale).  [U] algorithm outline: Morton-order
// the points, bound 1024-point boxes, seed the 3 best with the +-3 Morton neighbours, then visit
// only boxes closer than the current 3rd best.  The result is the exact 3-NN, so it is checked
// against scipy.spatial.cKDTree in tests/test_knn.py.
#include "gsr_common.h"
#include <cfloat>

#define KNN_BOX 1024
#define KNN_BLOCK 256

__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

__global__ void __launch_bounds__(KNN_BLOCK) knn_bounds_kernel(const float* __restrict__ xyz, int n,
                                                               uint32_t* __restrict__ mm /*[6]*/) {
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * KNN_BLOCK + threadIdx.x; i < n; i += gridDim.x * KNN_BLOCK) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = xyz[3 * (size_t)i + a];
            lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], d, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d, 64));
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            atomicMin(&mm[a], f2ord(lo[a]));
            atomicMax(&mm[3 + a], f2ord(hi[a]));
        }
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t x) {
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ void __launch_bounds__(KNN_BLOCK) knn_morton_kernel(const float* __restrict__ xyz, int n,
                                                               const uint32_t* __restrict__ mm,
                                                               uint32_t* __restrict__ codes) {
    const int i = blockIdx.x * KNN_BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t c = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = ord2f(mm[a]), hi = ord2f(mm[3 + a]);
        const float ext = hi - lo;
        const float rel = ext > 0.f ? (xyz[3 * (size_t)i + a] - lo) / ext : 0.f;
        const uint32_t q = (uint32_t)fminf(fmaxf(rel * 1023.0f, 0.f), 1023.f);
        c |= spread10(q) << a;
    }
    codes[i] = c;
}

// gather the points into Morton order (float4 for 16-byte loads) and bound each 1024-point box
__global__ void __launch_bounds__(KNN_BLOCK) knn_boxes_kernel(const float* __restrict__ xyz, int n,
                                                              const uint32_t* __restrict__ order,
                                                              float4* __restrict__ sorted,
                                                              float* __restrict__ boxes /*[nbox][6]*/) {
    __shared__ float s_lo[3][KNN_BLOCK / 64], s_hi[3][KNN_BLOCK / 64];
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    const int base = blockIdx.x * KNN_BOX;
    for (int k = threadIdx.x; k < KNN_BOX; k += KNN_BLOCK) {
        const int i = base + k;
        if (i < n) {
            const uint32_t src = order[i];
            const float x = xyz[3 * (size_t)src], y = xyz[3 * (size_t)src + 1], z = xyz[3 * (size_t)src + 2];
            sorted[i] = make_float4(x, y, z, 0.f);
            lo[0] = fminf(lo[0], x); lo[1] = fminf(lo[1], y); lo[2] = fminf(lo[2], z);
            hi[0] = fmaxf(hi[0], x); hi[1] = fmaxf(hi[1], y); hi[2] = fmaxf(hi[2], z);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], d, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d, 64));
        }
        if ((threadIdx.x & 63) == 0) { s_lo[a][threadIdx.x >> 6] = lo[a]; s_hi[a][threadIdx.x >> 6] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        float l = s_lo[a][0], h = s_hi[a][0];
        for (int w = 1; w < KNN_BLOCK / 64; ++w) { l = fminf(l, s_lo[a][w]); h = fmaxf(h, s_hi[a][w]); }
        boxes[6 * (size_t)blockIdx.x + a] = l;
        boxes[6 * (size_t)blockIdx.x + 3 + a] = h;
    }
}

__device__ __forceinline__ void keep3(float d, float* best) {
    // best[0] <= best[1] <= best[2]
    if (d < best[2]) {
        if (d < best[1]) {
            best[2] = best[1];
            if (d < best[0]) { best[1] = best[0]; best[0] = d; } else best[1] = d;
        } else best[2] = d;
    }
}
__device__ __forceinline__ float sqdist(const float4 a, const float4 b) {
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return dx * dx + dy * dy + dz * dz;
}

__global__ void __launch_bounds__(KNN_BLOCK) knn_search_kernel(int n, const float4* __restrict__ sorted,
                                                               const uint32_t* __restrict__ order,
                                                               const float* __restrict__ boxes, int nbox,
                                                               float* __restrict__ out) {
    const int i = blockIdx.x * KNN_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float4 p = sorted[i];
    float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    for (int k = max(0, i - 3); k <= min(n - 1, i + 3); ++k)
        if (k != i) keep3(sqdist(p, sorted[k]), best);
    const float reject = best[2];
    best[0] = FLT_MAX; best[1] = FLT_MAX; best[2] = FLT_MAX;
    for (int b = 0; b < nbox; ++b) {
        const float* bx = boxes + 6 * (size_t)b;
        float dx = 0.f, dy = 0.f, dz = 0.f;
        if (p.x < bx[0] || p.x > bx[3]) dx = fminf(fabsf(p.x - bx[0]), fabsf(p.x - bx[3]));
        if (p.y < bx[1] || p.y > bx[4]) dy = fminf(fabsf(p.y - bx[1]), fabsf(p.y - bx[4]));
        if (p.z < bx[2] || p.z > bx[5]) dz = fminf(fabsf(p.z - bx[2]), fabsf(p.z - bx[5]));
        const float dist = dx * dx + dy * dy + dz * dz;
        if (dist > reject || dist > best[2]) continue;
        const int e = min(n, (b + 1) * KNN_BOX);
        for (int k = b * KNN_BOX; k < e; ++k)
            if (k != i) keep3(sqdist(p, sorted[k]), best);
    }
    out[order[i]] = (best[0] + best[1] + best[2]) / 3.0f;
}

static inline size_t knn_ws(int64_t n, size_t* o_codes, size_t* o_sorted_codes, size_t* o_order, size_t* o_kt,
                            size_t* o_vt, size_t* o_pts, size_t* o_boxes, size_t* o_mm, size_t* o_sortws) {
    const size_t nb = gsr_align(size_t(n > 0 ? n : 1) * 4);
    const int64_t nbox = (n + KNN_BOX - 1) / KNN_BOX;
    size_t o = 0;
    *o_codes = o; o += nb;
    *o_sorted_codes = o; o += nb;
    *o_order = o; o += nb;
    *o_kt = o; o += nb;
    *o_vt = o; o += nb;
    *o_pts = o; o += gsr_align(size_t(n > 0 ? n : 1) * 16);
    *o_boxes = o; o += gsr_align(size_t(nbox > 0 ? nbox : 1) * 24);
    *o_mm = o; o += 256;
    *o_sortws = o; o += gsr_sort_ws_bytes(n);
    return o;
}

extern "C" size_t gsr_knn3_workspace_bytes(int32_t n) {
    size_t a, b, c, d, e, f, g, h, i;
    return knn_ws(n, &a, &b, &c, &d, &e, &f, &g, &h, &i);
}

extern "C" int32_t gsr_knn3(const float* xyz, int32_t n, float* out, void* ws, size_t ws_bytes,
                            gsr_stream_t stream_) {
    if (n < 0) { gsr_set_error("negative point count"); return GSR_E_INVALID; }
    if (n == 0) return GSR_OK;
    size_t oc, osc, oo, okt, ovt, op, ob, om, osw;
    const size_t need = knn_ws(n, &oc, &osc, &oo, &okt, &ovt, &op, &ob, &om, &osw);
    if (!xyz || !out || !ws || ws_bytes < need) { gsr_set_error("knn buffers / workspace too small"); return GSR_E_INVALID; }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    char* w = static_cast<char*>(ws);
    uint32_t* codes = reinterpret_cast<uint32_t*>(w + oc);
    uint32_t* sorted_codes = reinterpret_cast<uint32_t*>(w + osc);
    uint32_t* order = reinterpret_cast<uint32_t*>(w + oo);
    uint32_t* mm = reinterpret_cast<uint32_t*>(w + om);
    float4* pts = reinterpret_cast<float4*>(w + op);
    float* boxes = reinterpret_cast<float*>(w + ob);
    const int nbox = (n + KNN_BOX - 1) / KNN_BOX;

    GsrProfileScope prof(GSR_K_KNN, s);
    GSR_HIP_CHECK(hipMemsetAsync(mm, 0xFF, 12, s));
    GSR_HIP_CHECK(hipMemsetAsync(mm + 3, 0x00, 12, s));
    const int rb = min(1024, (n + KNN_BLOCK - 1) / KNN_BLOCK);
    hipLaunchKernelGGL(knn_bounds_kernel, dim3(rb), dim3(KNN_BLOCK), 0, s, xyz, n, mm);
    hipLaunchKernelGGL(knn_morton_kernel, dim3((n + KNN_BLOCK - 1) / KNN_BLOCK), dim3(KNN_BLOCK), 0, s, xyz, n, mm, codes);
    GSR_LAUNCH_CHECK();
    int rc = gsr_radix_sort_pairs(codes, nullptr, sorted_codes, order, reinterpret_cast<uint32_t*>(w + okt),
                                  reinterpret_cast<uint32_t*>(w + ovt), n, 0, 30, w + osw, s);
    if (rc != GSR_OK) return rc;
    hipLaunchKernelGGL(knn_boxes_kernel, dim3(nbox), dim3(KNN_BLOCK), 0, s, xyz, n, order, pts, boxes);
    hipLaunchKernelGGL(knn_search_kernel, dim3((n + KNN_BLOCK - 1) / KNN_BLOCK), dim3(KNN_BLOCK), 0, s, n, pts,
                       order, boxes, nbox, out);
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}
