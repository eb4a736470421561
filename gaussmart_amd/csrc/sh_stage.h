// Wave-private LDS staging of the spherical-harmonics block of 64 consecutive Gaussians, shared by
// preprocess_fwd and preprocess_bwd.  LDS image: row r (Gaussian wave_first + r) holds its
// coefficients as [k][c] floats at wl[r * SH_ROW_FLOATS + 3 k + c]; rows are 52 floats (208 B) so
// that row reads with ds_read_b128 are bank-conflict free.
// Two storage layouts of the operator input:
//   unified : shs       f32 [N, M, 3]                       (GaussianModel.get_features, the cat)
//   split   : shs (dc)  f32 [N, 1, 3] + shs_rest f32 [N, M-1, 3]   (the two parameters themselves,
//             GSR raw-parameter entry point: saves the 192 MB concatenation and its backward)
// Global traffic is coalesced 16-byte accesses in both directions; the wave's 64 rows are contiguous
// in memory, only the row boundaries differ from the LDS rows.
#pragma once
#include "gsr_common.h"

#define SH_ROW_FLOATS 52

// Copies n_floats contiguous floats (16-byte aligned base) between global memory and LDS rows of
// `row_f` payload floats starting at column `col0` of each LDS row.  TO_LDS: global -> LDS.
template <bool TO_LDS>
__device__ __forceinline__ void sh_copy_rows(float* wl, float* gptr, int n_floats, int row_f, int col0, int lane) {
    if (row_f <= 0 || n_floats <= 0) return;
    const int n_vec = n_floats >> 2;
    // element index of this lane's first float, tracked incrementally as (row, col)
    int e = lane << 2;
    int row = e / row_f, col = e - row * row_f;
    const int q_step = 256 / row_f, r_step = 256 - q_step * row_f;
    float4* g4 = reinterpret_cast<float4*>(gptr);
    // four 16-byte vectors per lane and round: all four global loads are in flight before the first LDS write
    // (one load per round left the wave waiting ~12 memory latencies in a row for a 64 x 45-float block)
    constexpr int U = 12;
    for (int v0 = lane; v0 < n_vec; v0 += 64 * U) {
        float4 d[U];
        if (TO_LDS) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (v0 + 64 * u < n_vec) d[u] = g4[v0 + 64 * u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (v0 + 64 * u < n_vec) {
                int r = row, c = col;
                float* vals = reinterpret_cast<float*>(&d[u]);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (TO_LDS) wl[r * SH_ROW_FLOATS + col0 + c] = vals[k];
                    else vals[k] = wl[r * SH_ROW_FLOATS + col0 + c];
                    if (++c == row_f) { c = 0; ++r; }
                }
                if (!TO_LDS) g4[v0 + 64 * u] = d[u];
            }
            row += q_step; col += r_step;
            if (col >= row_f) { col -= row_f; ++row; }
        }
    }
    for (int t = (n_vec << 2) + lane; t < n_floats; t += 64) {      // < 4 trailing floats
        const int r = t / row_f, c = t - r * row_f;
        if (TO_LDS) wl[r * SH_ROW_FLOATS + col0 + c] = gptr[t];
        else gptr[t] = wl[r * SH_ROW_FLOATS + col0 + c];
    }
}

template <bool TO_LDS>
__device__ __forceinline__ void sh_stage(float* wl, float* shs, float* shs_rest, int M, int wave_first, int n_here, int lane) {
    if (n_here <= 0) return;
    if (shs_rest == nullptr) {
        sh_copy_rows<TO_LDS>(wl, shs + (size_t)wave_first * M * 3, n_here * M * 3, M * 3, 0, lane);
    } else {
        sh_copy_rows<TO_LDS>(wl, shs + (size_t)wave_first * 3, n_here * 3, 3, 0, lane);
        sh_copy_rows<TO_LDS>(wl, shs_rest + (size_t)wave_first * (M - 1) * 3, n_here * (M - 1) * 3, (M - 1) * 3, 3, lane);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// staging needs 16-byte aligned bases (wave blocks are multiples of 16 bytes) and M <= 16
static inline bool sh_can_stage(const float* shs, const float* shs_rest, int M) {
    if (!shs || M < 1 || M > 16) return false;
    if (reinterpret_cast<uintptr_t>(shs) & 15) return false;
    if (shs_rest && (reinterpret_cast<uintptr_t>(shs_rest) & 15)) return false;
    return true;
}
