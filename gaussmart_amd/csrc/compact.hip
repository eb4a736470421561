// Row compaction of the Gaussian SoA: pruning keeps the rows of a boolean mask in ALL per-Gaussian
// tensors -- six parameters, their twelve Adam moments and the densification statistics
// (scene/gaussian_model.py:398-470 does `tensor[mask]` tensor by tensor: a nonzero() with a host synchronisation and
// an index kernel for each of ~21 tensors).  Here: one exclusive scan of the mask, ONE host read of the kept count,
// one launch that moves every tensor.
#include "gsr_common.h"

#define CP_MAX_TENSORS 24

struct CompactBatch {
    int count;
    long long n_rows;
    const uint32_t* src[CP_MAX_TENSORS];
    uint32_t* dst[CP_MAX_TENSORS];
    int words[CP_MAX_TENSORS];          // 4-byte words per row
};

__global__ void __launch_bounds__(256) mask_to_u32_kernel(const uint8_t* __restrict__ keep, long long n, uint32_t* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = keep[i] ? 1u : 0u;
}

// one thread per 4-byte word of the source tensor: coalesced reads, writes coalesced inside every kept row
__global__ void __launch_bounds__(256) compact_rows_kernel(CompactBatch b, const uint8_t* __restrict__ keep,
                                                           const uint32_t* __restrict__ offsets) {
    const int t = blockIdx.y;
    if (t >= b.count) return;
    const int wpr = b.words[t];
    const long long total = b.n_rows * wpr;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += stride) {
        const long long row = w / wpr;
        if (keep[row]) b.dst[t][(long long)offsets[row] * wpr + (w - row * wpr)] = b.src[t][w];
    }
}

extern "C" size_t gsr_compact_workspace_bytes(int64_t n_rows) {
    const size_t n = size_t(n_rows > 0 ? n_rows : 1);
    return gsr_align(n * 4) + gsr_align((n + 1) * 4) + gsr_scan_workspace_bytes((int64_t)n);
}

extern "C" int32_t gsr_compact_plan(const uint8_t* keep, int64_t n_rows, void* ws, size_t ws_bytes,
                                    const uint32_t** offsets_out, gsr_stream_t stream_) {
    if (n_rows < 0 || (n_rows > 0 && !keep) || !ws || !offsets_out || ws_bytes < gsr_compact_workspace_bytes(n_rows)) {
        gsr_set_error("bad compact_plan arguments");
        return GSR_E_INVALID;
    }
    hipStream_t s = static_cast<hipStream_t>(stream_);
    const size_t n = size_t(n_rows > 0 ? n_rows : 1);
    char* w = static_cast<char*>(ws);
    uint32_t* flags = reinterpret_cast<uint32_t*>(w);
    uint32_t* offsets = reinterpret_cast<uint32_t*>(w + gsr_align(n * 4));
    void* scan_ws = w + gsr_align(n * 4) + gsr_align((n + 1) * 4);
    *offsets_out = offsets;
    if (n_rows == 0) { GSR_HIP_CHECK(hipMemsetAsync(offsets, 0, 4, s)); return GSR_OK; }
    hipLaunchKernelGGL(mask_to_u32_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, s, keep, (long long)n_rows, flags);
    GSR_LAUNCH_CHECK();
    return gsr_exclusive_scan_u32(flags, nullptr, offsets, n_rows, scan_ws, s);   // offsets[n_rows] = rows kept
}

extern "C" int32_t gsr_compact_apply(int32_t count, const void* const* src, void* const* dst, const int32_t* row_bytes,
                                     int64_t n_rows, const uint8_t* keep, const uint32_t* offsets, gsr_stream_t stream_) {
    if (count < 0 || count > CP_MAX_TENSORS) { gsr_set_error("compact: at most %d tensors per call", CP_MAX_TENSORS); return GSR_E_INVALID; }
    if (count == 0 || n_rows <= 0) return GSR_OK;
    if (!src || !dst || !row_bytes || !keep || !offsets) { gsr_set_error("compact: null argument"); return GSR_E_INVALID; }
    CompactBatch b;
    b.count = count; b.n_rows = n_rows;
    long long max_words = 0;
    for (int i = 0; i < count; ++i) {
        if (row_bytes[i] <= 0 || (row_bytes[i] & 3) || !src[i] || !dst[i]) {
            gsr_set_error("compact: tensor %d needs a positive row size that is a multiple of 4 bytes", i);
            return GSR_E_INVALID;
        }
        b.src[i] = static_cast<const uint32_t*>(src[i]); b.dst[i] = static_cast<uint32_t*>(dst[i]);
        b.words[i] = row_bytes[i] / 4;
        if ((long long)b.words[i] * n_rows > max_words) max_words = (long long)b.words[i] * n_rows;
    }
    long long blocks = (max_words + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipStream_t s = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(compact_rows_kernel, dim3((unsigned)blocks, (unsigned)count), dim3(256), 0, s, b, keep, offsets);
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}

// Densification statistics of one iteration (train.py:199-203, scene/gaussian_model.py:551-553) in one launch and
// without the host synchronisation of the reference's boolean-mask indexing:
//   visible = radii > 0;  max_radii2D[visible] = max(max_radii2D, radii);
//   xyz_gradient_accum[visible] += ||means2D.grad||;  denom[visible] += 1
__global__ void __launch_bounds__(256) densify_stats_kernel(int n, const int32_t* __restrict__ radii,
                                                            const float* __restrict__ grad2d, float* __restrict__ max_radii,
                                                            float* __restrict__ accum, float* __restrict__ denom) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = radii[i];
    if (r <= 0) return;
    max_radii[i] = fmaxf(max_radii[i], (float)r);
    const float gx = grad2d[3 * i], gy = grad2d[3 * i + 1], gz = grad2d[3 * i + 2];
    accum[i] += sqrtf(gx * gx + gy * gy + gz * gz);
    denom[i] += 1.0f;
}

extern "C" int32_t gsr_densify_stats(int32_t n, const int32_t* radii, const float* grad2d, float* max_radii2D,
                                     float* xyz_gradient_accum, float* denom, gsr_stream_t stream_) {
    if (n < 0 || (n > 0 && (!radii || !grad2d || !max_radii2D || !xyz_gradient_accum || !denom))) {
        gsr_set_error("bad densify_stats arguments");
        return GSR_E_INVALID;
    }
    if (n == 0) return GSR_OK;
    hipStream_t s = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(densify_stats_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, radii, grad2d, max_radii2D,
                       xyz_gradient_accum, denom);
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}
