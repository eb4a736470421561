// Block-level exclusive scan pieces shared by the binning kernels (binning.hip) and by the kernels that carry the scan of the
// backward's row counts as a side job (loss.hip): ONE definition of the tile shape and of both halves of the scan.
#pragma once
#include "gsr_common.h"

#define SCAN_BLOCK 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_BLOCK * SCAN_ITEMS)

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t n = __shfl_up(v, d, 64);
        if (lane >= d) v += n;
    }
    return v;
}

// exclusive scan of one value per thread across a 256-thread block; returns block total
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t& total, uint32_t* wave_tot /*[4+]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_scan(v, lane);
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
        const uint32_t t = wave_tot[w];
        if (w < wave) base += t;
        tot += t;
    }
    total = tot;
    __syncthreads();
    return base + inc - v;
}

// first half: the total of scan tile `block` -> partial[block]   (one 256-thread workgroup; wt: shared, >= 4 words)
template <typename T>
__device__ __forceinline__ void scan_reduce_body(const T* __restrict__ in, const uint32_t* __restrict__ gather,
                                                 uint32_t* __restrict__ partial, int64_t n, int block, uint32_t* wt) {
    const int64_t base = (int64_t)block * SCAN_TILE;
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int64_t j = base + (int64_t)i * SCAN_BLOCK + threadIdx.x;
        if (j < n) sum += gather ? in[gather[j]] : in[j];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_down(sum, d, 64);
    if ((threadIdx.x & 63) == 0) wt[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) partial[block] = wt[0] + wt[1] + wt[2] + wt[3];
}

// second half: out[j] for the elements of scan tile `block` (and out[n] = grand total from the tile that holds n - 1)
template <typename T>
__device__ __forceinline__ void scan_apply_body(const T* __restrict__ in, const uint32_t* __restrict__ gather,
                                                const uint32_t* __restrict__ partial, uint32_t* __restrict__ out, int64_t n,
                                                int block, uint32_t* wt) {
    // thread owns SCAN_ITEMS consecutive elements so the block tile is scanned in index order
    const int64_t first = (int64_t)block * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t tsum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int64_t j = first + i;
        v[i] = j < n ? (gather ? in[gather[j]] : in[j]) : 0;
        tsum += v[i];
    }
    // offset of this tile = sum of the preceding tiles' totals (<= a few thousand values: cheaper than a
    // third launch that scans them)
    uint32_t pre = 0;
    for (int j = threadIdx.x; j < block; j += SCAN_BLOCK) pre += partial[j];
    uint32_t tile_offset, total;
    (void)block_excl_scan(pre, tile_offset, wt);
    uint32_t run = tile_offset + block_excl_scan(tsum, total, wt);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        const int64_t j = first + i;
        if (j < n) out[j] = run;
        run += v[i];
    }
    // grand total lands in out[n]
    if (first <= n - 1 && n - 1 < first + SCAN_ITEMS) out[n] = run;
}
