// K2-K5 binning: exclusive scan, stable LSD radix sort of (u32 key, u32 value) pairs, instance
// emission and per-tile ranges.
//
// The recalled upstream sorts D (Gaussian, tile) pairs by the 64-bit key (tile << 32 | depth bits)
// in one ~6-pass radix sort.  The same final order is obtained here with far less traffic:
//   1. stable sort of the N Gaussians by depth bits (culled ones get key 0xFFFFFFFF),
//   2. instances are emitted in that order (tiles row-major inside a rect),
//   3. a stable sort of the D instances by tile id only (ceil(log2 tiles) bits, 2 passes).
// Stability makes ties fall back to emission order = (depth, Gaussian index), which is exactly the
// order a stable sort of the 64-bit keys emitted in Gaussian-index order produces, so
// `point_list` is bit-identical (checked against NumPy's stable argsort in tests/test_gpu_rasterizer.py::test_binning_bit_exact
// and tests/test_gpu_sort_knn.py).
//
// All integer work; HBM-bound.  Wave64 throughout: digit matching uses 64-bit ballots.
#include "gsr_common.h"
#include "scan_bodies.h"
#include <type_traits>

// ============================================================================ scan
template <typename T>
__global__ void __launch_bounds__(SCAN_BLOCK) scan_reduce_kernel(const T* __restrict__ in,
                                                                 const uint32_t* __restrict__ gather,
                                                                 uint32_t* __restrict__ partial, int64_t n) {
    __shared__ uint32_t wt[SCAN_BLOCK / 64];
    scan_reduce_body<T>(in, gather, partial, n, (int)blockIdx.x, wt);
}
template <typename T>
__global__ void __launch_bounds__(SCAN_BLOCK) scan_apply_kernel(const T* __restrict__ in,
                                                                const uint32_t* __restrict__ gather,
                                                                const uint32_t* __restrict__ partial,
                                                                uint32_t* __restrict__ out, int64_t n) {
    __shared__ uint32_t wt[SCAN_BLOCK / 64];
    scan_apply_body<T>(in, gather, partial, out, n, (int)blockIdx.x, wt);
}

size_t gsr_scan_workspace_bytes(int64_t n) {
    const int64_t blocks = (n + SCAN_TILE - 1) / SCAN_TILE;
    // (tile totals + eight in-tile segment prefixes per tile for the rank scan; also holds gsr_forward's count partials)
    const int64_t words = blocks * (1 + SCAN_ITEMS);
    return gsr_align(size_t(words > 2 * GSR_COUNT_PARTIALS ? words : 2 * GSR_COUNT_PARTIALS) * 4);
}

template <typename T>
static int exclusive_scan_any(const T* in, const uint32_t* gather, uint32_t* out, int64_t n, void* ws, hipStream_t s) {
    if (n <= 0) {
        GSR_HIP_CHECK(hipMemsetAsync(out, 0, 4, s));
        return GSR_OK;
    }
    GsrProfileScope prof(GSR_K_SCAN, s);
    uint32_t* partial = static_cast<uint32_t*>(ws);
    const int blocks = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
    hipLaunchKernelGGL(scan_reduce_kernel<T>, dim3(blocks), dim3(SCAN_BLOCK), 0, s, in, gather, partial, n);
    hipLaunchKernelGGL(scan_apply_kernel<T>, dim3(blocks), dim3(SCAN_BLOCK), 0, s, in, gather, partial, out, n);
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}

int gsr_exclusive_scan_u32(const uint32_t* in, const uint32_t* gather, uint32_t* out, int64_t n,
                           void* ws, hipStream_t s) {
    return exclusive_scan_any<uint32_t>(in, gather, out, n, ws, s);
}

// same, over byte-sized inputs (the per-instance gradient-row counts, <= 16 each)
int gsr_exclusive_scan_u8(const uint8_t* in, uint32_t* out, int64_t n, void* ws, hipStream_t s) {
    return exclusive_scan_any<uint8_t>(in, nullptr, out, n, ws, s);
}

// ============================================================================ radix sort
// One pass = TWO launches (round 3; rounds 1-2 ran histogram -> per-digit table scan -> scatter, and the scan was 64-256
// workgroups walking `nblocks` counters serially: 10 us per pass at 1 M keys, 54 us at 22 M):
//   rs_hist_kernel     one workgroup per GROUP of G consecutive tiles: per tile the digit counts of the group's earlier
//                      tiles (`tile_pre`, exclusive, in-group), per group the totals (`group_tot`), and -- with one
//                      256-byte-contiguous integer atomic wave-instruction per 64 digits -- the totals of the group's
//                      SUPERGROUP of 32 groups (`super_tot`; <= 32 adders per row, the shape that runs at full rate);
//   rs_scatter_kernel  every workgroup serves itself: digit base = scan over the digits of sum(super_tot), plus the
//                      supergroups before its own (<= 64 rows), the groups before its own inside its supergroup (<= 31
//                      rows), plus its tile_pre row.  All rows are block-major (one coalesced row of NB counters per load)
//                      and L2-resident; the loads are issued behind the tile's own key loads.
// G = smallest power of two with nblocks / G <= 512 up to ~4 M pairs (at most 16 supergroups: a scatter workgroup fetches
// all its rows in one round trip), <= 2048 beyond (at most 32 tiles per group: the group's histograms live in LDS side by
// side).
#define RS_BLOCK 256
#define RS_WAVES (RS_BLOCK / 64)
// 2048 pairs per workgroup: the depth sort of 1M Gaussians then has 489 workgroups (two per CU) instead of 245 (fewer
// than CUs) -- measured per step at 1M / 1080p, same box: histograms 0.055 -> 0.048 ms, scatters 0.120 -> 0.103 ms, the
// longer digit rows cost the scans 0.003 ms (16 items: base; 12: -0.008 ms; 6: -0.016; 4: -0.004)
#ifndef RS_ITEMS
#define RS_ITEMS 8
#endif
#define RS_TILE (RS_BLOCK * RS_ITEMS)
#define RS_MAX_BINS 256
#define RS_SUPER 32          // groups per supergroup
#ifndef RS_RB
#define RS_RB 16             // table rows a scatter workgroup has in flight at a time (registers against round trips)
#endif
#define RS_MAX_G 32          // tiles per group at most (one LDS histogram per tile of the group: 32 KB)
#define RS_GROUPS_TARGET 512 // => at most 16 supergroups (ONE batch of row loads per scatter workgroup) up to 33 M pairs

struct RsGeom {
    int nblocks, G, ngroups, nsuper;
    explicit RsGeom(int64_t n) {
        nblocks = (int)((n + RS_TILE - 1) / RS_TILE);
        if (nblocks < 1) nblocks = 1;
        // up to ~4 M pairs the sort is latency-bound and a scatter workgroup should fetch its rows in one round trip
        // (<= 512 groups = 16 supergroups); beyond that it is throughput-bound, the histogram kernel needs its parallelism
        // back (345 workgroups walking 32 tiles each took 60 us per launch at 22.6 M pairs inside the step) and a few more
        // row batches per scatter workgroup disappear behind the other resident workgroups: <= 2048 groups
        const int target = nblocks <= 2048 ? RS_GROUPS_TARGET : 4 * RS_GROUPS_TARGET;
        G = 1;
        while ((nblocks + G - 1) / G > target && G < RS_MAX_G) G <<= 1;
        ngroups = (nblocks + G - 1) / G;
        nsuper = (ngroups + RS_SUPER - 1) / RS_SUPER;
    }
};

__global__ void __launch_bounds__(RS_BLOCK) rs_hist_kernel(const uint32_t* __restrict__ keys, int64_t n, int shift,
                                                           uint32_t mask, int nbins, int nblocks, int G,
                                                           uint32_t* __restrict__ tile_pre,
                                                           uint32_t* __restrict__ group_tot,
                                                           uint32_t* __restrict__ super_tot) {
    extern __shared__ uint32_t hist[];   // [tiles of this group][nbins]
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * G, t1 = min(nblocks, t0 + G), nt = t1 - t0;
    for (int i = tid; i < nt * nbins; i += RS_BLOCK) hist[i] = 0;
    __syncthreads();
    // no barrier between the tiles of the group: their loads are independent, U tiles (8 U keys per thread) are in
    // flight at a time
    auto count_tiles = [&](auto U_) {
        constexpr int U = decltype(U_)::value;
        for (int t = 0; t < nt; t += U) {
            uint32_t k[U][RS_ITEMS];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t base = (int64_t)(t0 + t + u) * RS_TILE + tid;
#pragma unroll
                for (int i = 0; i < RS_ITEMS; ++i) {
                    const int64_t j = base + (int64_t)i * RS_BLOCK;
                    k[u][i] = (t + u < nt && j < n) ? keys[j] : 0u;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t base = (int64_t)(t0 + t + u) * RS_TILE + tid;
                uint32_t* h = hist + (t + u) * nbins;
#pragma unroll
                for (int i = 0; i < RS_ITEMS; ++i)
                    if (t + u < nt && base + (int64_t)i * RS_BLOCK < n) atomicAdd(&h[(k[u][i] >> shift) & mask], 1u);
            }
        }
    };
    if (G >= 4) count_tiles(std::integral_constant<int, 4>{});
    else if (G == 2) count_tiles(std::integral_constant<int, 2>{});
    else count_tiles(std::integral_constant<int, 1>{});
    __syncthreads();
    if (tid < nbins) {
        uint32_t run = 0;
        for (int t = 0; t < nt; ++t) {
            tile_pre[(size_t)(t0 + t) * nbins + tid] = run;
            run += hist[t * nbins + tid];
        }
        group_tot[(size_t)blockIdx.x * nbins + tid] = run;
        atomicAdd(&super_tot[(size_t)(blockIdx.x / RS_SUPER) * nbins + tid], run);
    }
}

// V2: a SECOND value array travels with the pairs (the tile sort carries the Gaussian id next to the emission index, so
// that no pass has to gather one through the other afterwards: at D = 20 M that gather cost 0.4 ms per frame).
template <int BITS, bool V2>
__global__ void __launch_bounds__(RS_BLOCK) rs_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                              const uint32_t* __restrict__ vals_in,
                                                              uint32_t* __restrict__ keys_out,
                                                              uint32_t* __restrict__ vals_out, int64_t n,
                                                              int shift, int G, int nsuper,
                                                              const uint32_t* __restrict__ tile_pre,
                                                              const uint32_t* __restrict__ group_tot,
                                                              const uint32_t* __restrict__ super_tot,
                                                              uint32_t* __restrict__ super_next,
                                                              const uint32_t* __restrict__ vals2_in,
                                                              uint32_t* __restrict__ vals2_out) {
    constexpr int NB = 1 << BITS;
    constexpr uint32_t MASK = NB - 1;
    __shared__ uint32_t wave_hist[RS_WAVES][NB];
    __shared__ uint32_t digit_start[NB];   // start of each digit in the block-local sorted order
    __shared__ uint32_t gbase[NB];         // global base of (digit, this block)
    __shared__ uint32_t s_keys[RS_TILE];
    __shared__ uint32_t s_vals[RS_TILE];
    __shared__ uint32_t s_vals2[V2 ? RS_TILE : 1];
    __shared__ uint32_t wt[RS_WAVES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t block_base = (int64_t)blockIdx.x * RS_TILE;
    const int block_n = (int)min((int64_t)RS_TILE, n - block_base);
    const int wave_base_idx = wave * (RS_TILE / RS_WAVES);

    // the tile's own loads go out first; the table rows below arrive while they are in flight
    uint32_t key[RS_ITEMS], val[RS_ITEMS], val2[V2 ? RS_ITEMS : 1], rank[RS_ITEMS];
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const int local = wave_base_idx + r * 64 + lane;   // index order = (wave, round, lane)
        const bool valid = local < block_n;
        key[r] = valid ? keys_in[block_base + local] : 0xFFFFFFFFu;
        // vals_in == NULL means "values are the element indices" (first pass of an argsort)
        val[r] = valid ? (vals_in ? vals_in[block_base + local] : (uint32_t)(block_base + local)) : 0u;
        if (V2) val2[r] = valid ? vals2_in[block_base + local] : 0u;
    }
    for (int i = tid; i < RS_WAVES * NB; i += RS_BLOCK) (&wave_hist[0][0])[i] = 0;
    // the next pass's histogram kernel accumulates into the other supergroup table: left zeroed here
    if (blockIdx.x == 0 && super_next)
        for (int i = tid; i < nsuper * RS_MAX_BINS; i += RS_BLOCK) super_next[i] = 0;
    {   // global base of (digit, this tile) = first output index of the digit + the digit's count in earlier tiles.
        // Every row load is independent of the others and issued before any is consumed (fixed trip counts, clamped row
        // index + select): one L2 round trip for the lot, behind the tile's own loads.  (Keeping the rows in registers
        // until after the ranking instead costs 50 VGPRs = half the occupancy: measured slower from 3 M pairs up.)
        uint32_t tot = 0, before = 0;
        if (tid < NB) {
            const int g = (int)blockIdx.x / G, sg = g / RS_SUPER, g0 = sg * RS_SUPER;
            const uint32_t pre = tile_pre[(size_t)blockIdx.x * NB + tid];
            for (int q0 = 0; q0 < nsuper; q0 += RS_RB) {
                uint32_t sv[RS_RB];
#pragma unroll
                for (int k = 0; k < RS_RB; ++k) sv[k] = super_tot[(size_t)min(q0 + k, nsuper - 1) * NB + tid];
#pragma unroll
                for (int k = 0; k < RS_RB; ++k) {
                    const uint32_t v = q0 + k < nsuper ? sv[k] : 0u;
                    tot += v;
                    before += q0 + k < sg ? v : 0u;
                }
            }
            for (int k0 = 0; g0 + k0 < g; k0 += RS_RB) {      // (wave-uniform trip count)
                uint32_t gv[RS_RB];
#pragma unroll
                for (int k = 0; k < RS_RB; ++k) gv[k] = group_tot[(size_t)min(g0 + k0 + k, g) * NB + tid];
#pragma unroll
                for (int k = 0; k < RS_RB; ++k) before += g0 + k0 + k < g ? gv[k] : 0u;
            }
            before += pre;
        }
        uint32_t tot_unused;
        const uint32_t dbase = block_excl_scan(tot, tot_unused, wt);   // (two barriers: wave_hist is zeroed behind them)
        if (tid < NB) gbase[tid] = dbase + before;
    }

    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const int local = wave_base_idx + r * 64 + lane;
        const bool valid = local < block_n;
        const uint32_t d = (key[r] >> shift) & MASK;
        // lanes holding the same digit (among valid lanes)
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < BITS; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            same &= bit ? bal : ~bal;
        }
        const uint32_t before = __popcll(same & lt_mask);
        const uint32_t count = __popcll(same);
        const bool last = valid && ((same >> lane) >> 1) == 0ull;
        uint32_t pre = 0;
        if (valid) pre = wave_hist[wave][d];
        rank[r] = pre + before;
        __builtin_amdgcn_wave_barrier();
        if (last) wave_hist[wave][d] = pre + count;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();

    // per digit: exclusive prefix over waves, block totals, then exclusive scan over digits
    uint32_t my_total = 0;
    if (tid < NB) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) {
            const uint32_t c = wave_hist[w][tid];
            wave_hist[w][tid] = run;
            run += c;
        }
        my_total = run;
    }
    uint32_t tot_unused;
    const uint32_t ex = block_excl_scan(my_total, tot_unused, wt);
    if (tid < NB) digit_start[tid] = ex;
    __syncthreads();

    // block-local reorder through LDS so the global writes are runs of consecutive addresses
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const int local = wave_base_idx + r * 64 + lane;
        if (local < block_n) {
            const uint32_t d = (key[r] >> shift) & MASK;
            const uint32_t pos = digit_start[d] + wave_hist[wave][d] + rank[r];
            s_keys[pos] = key[r];
            s_vals[pos] = val[r];
            if (V2) s_vals2[pos] = val2[r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const int pos = i * RS_BLOCK + tid;
        if (pos < block_n) {
            const uint32_t k = s_keys[pos];
            const uint32_t d = (k >> shift) & MASK;
            const size_t g = (size_t)gbase[d] + (uint32_t)(pos - digit_start[d]);
            keys_out[g] = k;
            vals_out[g] = s_vals[pos];
            if (V2) vals2_out[g] = s_vals2[pos];
        }
    }
}

__global__ void iota_kernel(uint32_t* out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint32_t)i;
}

size_t gsr_sort_ws_bytes(int64_t n) {
    // tile_pre [nblocks][256] + group_tot [ngroups][256] + two supergroup tables (this pass / next pass)
    const RsGeom geo(n);
    return gsr_align(size_t(geo.nblocks) * RS_MAX_BINS * 4) + gsr_align(size_t(geo.ngroups) * RS_MAX_BINS * 4) +
           2 * gsr_align(size_t(geo.nsuper) * RS_MAX_BINS * 4);
}

// The words of the workspace that must be zero when the first pass starts (its supergroup table).  gsr_radix_sort_pairs
// clears them itself (one memset launch, ~5 us) unless the caller says a kernel of its own already did.
void gsr_sort_zero_region(void* ws, int64_t n, uint32_t** ptr, size_t* words) {
    const RsGeom geo(n > 0 ? n : 1);
    char* w8 = static_cast<char*>(ws);
    *ptr = reinterpret_cast<uint32_t*>(w8 + gsr_align(size_t(geo.nblocks) * RS_MAX_BINS * 4) +
                                       gsr_align(size_t(geo.ngroups) * RS_MAX_BINS * 4));
    *words = size_t(geo.nsuper) * RS_MAX_BINS;
}

int gsr_radix_sort_pairs(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out,
                         uint32_t* vals_out, uint32_t* keys_tmp, uint32_t* vals_tmp, int64_t n,
                         int begin_bit, int end_bit, void* ws, hipStream_t s,
                         const uint32_t* vals2_in, uint32_t* vals2_out, uint32_t* vals2_tmp, bool table_zeroed) {
    if (n <= 0) return GSR_OK;
    const int bits = end_bit - begin_bit;
    const bool v2 = vals2_in != nullptr;
    if (bits <= 0) {   // nothing to sort on: stable sort is the identity
        GSR_HIP_CHECK(hipMemcpyAsync(keys_out, keys_in, size_t(n) * 4, hipMemcpyDeviceToDevice, s));
        if (v2) GSR_HIP_CHECK(hipMemcpyAsync(vals2_out, vals2_in, size_t(n) * 4, hipMemcpyDeviceToDevice, s));
        if (vals_in) GSR_HIP_CHECK(hipMemcpyAsync(vals_out, vals_in, size_t(n) * 4, hipMemcpyDeviceToDevice, s));
        else hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, vals_out, n);
        GSR_LAUNCH_CHECK();
        return GSR_OK;
    }
    const int passes = (bits + 7) / 8;
    const RsGeom geo(n);
    char* w8 = static_cast<char*>(ws);
    uint32_t* tile_pre = reinterpret_cast<uint32_t*>(w8);
    uint32_t* group_tot = reinterpret_cast<uint32_t*>(w8 + gsr_align(size_t(geo.nblocks) * RS_MAX_BINS * 4));
    const size_t super_bytes = gsr_align(size_t(geo.nsuper) * RS_MAX_BINS * 4);
    uint32_t* super_tab[2];
    super_tab[0] = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(group_tot) + gsr_align(size_t(geo.ngroups) * RS_MAX_BINS * 4));
    super_tab[1] = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(super_tab[0]) + super_bytes);
    // the first pass's supergroup table is zeroed here, every later one by the scatter kernel of the pass before it
    if (!table_zeroed) GSR_HIP_CHECK(hipMemsetAsync(super_tab[0], 0, size_t(geo.nsuper) * RS_MAX_BINS * 4, s));

    // ping-pong so that the last pass lands in *_out
    const uint32_t* src_k = keys_in; const uint32_t* src_v = vals_in; const uint32_t* src_v2 = vals2_in;
    int bit = begin_bit;
    for (int p = 0; p < passes; ++p) {
        const int remaining_passes = passes - p;
        const int remaining_bits = end_bit - bit;
        const int w = (remaining_bits + remaining_passes - 1) / remaining_passes;  // even split, <= 8
        const int nbins = 1 << w;
        const bool to_out = ((passes - 1 - p) % 2) == 0;
        uint32_t* dst_k = to_out ? keys_out : keys_tmp;
        uint32_t* dst_v = to_out ? vals_out : vals_tmp;
        uint32_t* dst_v2 = to_out ? vals2_out : vals2_tmp;
        uint32_t* sup = super_tab[p & 1];
        uint32_t* sup_next = p + 1 < passes ? super_tab[(p + 1) & 1] : nullptr;
        {
            GsrProfileScope prof(GSR_K_SORT_HIST, s);
            hipLaunchKernelGGL(rs_hist_kernel, dim3(geo.ngroups), dim3(RS_BLOCK), size_t(geo.G) * nbins * 4, s, src_k, n, bit,
                               (uint32_t)(nbins - 1), nbins, geo.nblocks, geo.G, tile_pre, group_tot, sup);
        }
        {
            GsrProfileScope prof(GSR_K_SORT_SCATTER, s);
#define RS_CASE(B) case B: if (v2) hipLaunchKernelGGL((rs_scatter_kernel<B, true>), dim3(geo.nblocks), dim3(RS_BLOCK), 0, s, \
                                              src_k, src_v, dst_k, dst_v, n, bit, geo.G, geo.nsuper, tile_pre, group_tot, sup, sup_next, src_v2, dst_v2); \
                   else hipLaunchKernelGGL((rs_scatter_kernel<B, false>), dim3(geo.nblocks), dim3(RS_BLOCK), 0, s, \
                                              src_k, src_v, dst_k, dst_v, n, bit, geo.G, geo.nsuper, tile_pre, group_tot, sup, sup_next, nullptr, nullptr); break;
            switch (w) { RS_CASE(1) RS_CASE(2) RS_CASE(3) RS_CASE(4) RS_CASE(5) RS_CASE(6) RS_CASE(7) RS_CASE(8)
                default: gsr_set_error("radix digit width %d", w); return GSR_E_INVALID; }
#undef RS_CASE
        }
        GSR_LAUNCH_CHECK();
        src_k = dst_k; src_v = dst_v; src_v2 = dst_v2;
        bit += w;
    }
    return GSR_OK;
}

// ============================================================================ emit / finalize
// Instance count ahead of the depth sort: partial sums of tiles_touched (the host adds <= 256 words), so that the one
// host round trip of the forward overlaps the depth sort instead of following the scan.
__global__ void __launch_bounds__(256) count_partials_kernel(const uint32_t* __restrict__ counts, int N, int chunk,
                                                             unsigned long long* __restrict__ partial,
                                                             uint32_t* __restrict__ zero, int zero_words) {
    __shared__ unsigned long long wt[4];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < zero_words; i += gridDim.x * 256) zero[i] = 0;
    const int begin = blockIdx.x * chunk, end = min(N, begin + chunk);
    unsigned long long sum = 0;     // 64-bit: a scene that overflows the 32-bit instance indices must be SEEN to
    for (int i = begin + threadIdx.x; i < end; i += 256) sum += counts[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_down(sum, d, 64);
    if ((threadIdx.x & 63) == 0) wt[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = wt[0] + wt[1] + wt[2] + wt[3];
}

int gsr_launch_count_partials(const uint32_t* counts, int N, unsigned long long* partial, int* n_partial,
                              uint32_t* zero, size_t zero_words, hipStream_t s) {
    int blocks = (N + 4095) / 4096;
    if (blocks > GSR_COUNT_PARTIALS) blocks = GSR_COUNT_PARTIALS;
    if (blocks < 1) blocks = 1;
    const int chunk = ((N + blocks - 1) / blocks + 255) & ~255;
    blocks = (N + chunk - 1) / chunk;
    if (blocks < 1) blocks = 1;
    *n_partial = blocks;
    GsrProfileScope prof(GSR_K_SCAN, s);
    hipLaunchKernelGGL(count_partials_kernel, dim3(blocks), dim3(256), 0, s, counts, N, chunk, partial, zero, (int)zero_words);
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}

// Tile rects in depth-rank order: ONE gather by Gaussian id; the emission below then reads everything coalesced.
// The same kernel prepares the exclusive scan of the instance counts (rect width x height) that the emission finishes
// itself: a workgroup owns the SCAN_TILE ranks of one scan tile and leaves the tile's total and, for each of its
// SCAN_ITEMS segments of SCAN_BLOCK consecutive ranks (= one emission workgroup), the count of the segments before it
// inside the tile.  (Round 3: this used to be followed by a scan launch that wrote `offs` and a count array.)
__global__ void __launch_bounds__(SCAN_BLOCK) rank_gather_kernel(int N, const uint32_t* __restrict__ order,
                                                                 const uint2* __restrict__ tile_rect,
                                                                 uint2* __restrict__ rank_rect,
                                                                 uint32_t* __restrict__ partial,
                                                                 uint32_t* __restrict__ seg_before) {
    __shared__ uint32_t wt[SCAN_ITEMS][SCAN_BLOCK / 64];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x;
    uint32_t g[SCAN_ITEMS];
    uint2 R[SCAN_ITEMS];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) g[i] = base + i * SCAN_BLOCK < N ? order[base + i * SCAN_BLOCK] : 0u;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) R[i] = base + i * SCAN_BLOCK < N ? tile_rect[g[i]] : make_uint2(0u, 0u);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {      // item i of every thread = segment i of the tile
        const int r = base + i * SCAN_BLOCK;
        uint32_t c = 0;
        if (r < N) {
            c = (R[i].y & 0xFFFFu) * (R[i].y >> 16);
            rank_rect[r] = R[i];
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
        if ((threadIdx.x & 63) == 0) wt[i][threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; ++i) {
            seg_before[blockIdx.x * SCAN_ITEMS + i] = run;
            run += (wt[i][0] + wt[i][1]) + (wt[i][2] + wt[i][3]);
        }
        partial[blockIdx.x] = run;
    }
}

// rank_rect in depth-rank order + the scan partials of the instance counts (scan_ws: gsr_scan_workspace_bytes(N));
// gsr_launch_emit finishes the scan and writes offs
int gsr_launch_rank_gather_scan(int N, const uint32_t* order, const uint2* tile_rect, uint2* rank_rect, void* scan_ws,
                                hipStream_t s) {
    if (N <= 0) return GSR_OK;
    uint32_t* partial = static_cast<uint32_t*>(scan_ws);
    const int blocks = (N + SCAN_TILE - 1) / SCAN_TILE;
    GsrProfileScope prof(GSR_K_EMIT, s);
    hipLaunchKernelGGL(rank_gather_kernel, dim3(blocks), dim3(SCAN_BLOCK), 0, s, N, order, tile_rect, rank_rect, partial,
                       partial + blocks);
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}

// Load-balanced expansion: a wave owns 64 consecutive depth ranks, whose instances are one contiguous output
// range.  Lane k parks rank k's (first output index, Gaussian id, tile rect) in LDS; then the 64 lanes fill the
// range 64 outputs at a time, each finding its owner by binary search over the 64 first-indices -- the two
// output streams are written fully coalesced however unequal the rects are.
// The first output index of a rank is the exclusive scan of the instance counts, finished here: totals of the scan
// tiles before this workgroup's (a few hundred words, summed by the workgroup) + the segments before it inside its
// tile (rank_gather_kernel) + a scan over the workgroup's own 256 counts; offs[0..N] is written for the backward.
// (Also clears two small arrays later kernels of the frame need zeroed -- the tile ranges and the tile sort's supergroup
// table -- instead of a memset launch each.)
__global__ void __launch_bounds__(SCAN_BLOCK) emit_instances_kernel(int N, int gx,
                                                             const uint32_t* __restrict__ order,
                                                             const uint32_t* __restrict__ partial,
                                                             const uint32_t* __restrict__ seg_before,
                                                             uint32_t* __restrict__ offs,
                                                             const uint2* __restrict__ rank_rect,
                                                             uint32_t* __restrict__ tile_keys,
                                                             uint32_t* __restrict__ emit_gid,
                                                             uint32_t* __restrict__ zero_a, int words_a,
                                                             uint32_t* __restrict__ zero_b, int words_b) {
    __shared__ uint32_t s_off[4][64], s_gid[4][64], s_xy[4][64], s_w[4][64];
    __shared__ uint32_t wt[SCAN_BLOCK / 64];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < words_a; i += gridDim.x * 256) zero_a[i] = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < words_b; i += gridDim.x * 256) zero_b[i] = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;   // depth rank
    uint32_t g = 0, xy = 0, w = 1, cnt = 0;
    if (r < N) {
        g = order[r];
        const uint2 R = rank_rect[r];
        xy = R.x;
        w = max(1u, R.y & 0xFFFFu);
        cnt = (R.y & 0xFFFFu) * (R.y >> 16);
    }
    // exclusive scan of the counts: scan tiles before mine + segments before mine in my tile + ranks before mine here
    const int my_tile = blockIdx.x / SCAN_ITEMS;
    uint32_t pre = 0;
    for (int j = threadIdx.x; j < my_tile; j += SCAN_BLOCK) pre += partial[j];
    uint32_t tiles_before, total_unused;
    (void)block_excl_scan(pre, tiles_before, wt);
    const uint32_t first = tiles_before + seg_before[blockIdx.x];
    const uint32_t begin_of_rank = first + block_excl_scan(cnt, total_unused, wt);
    uint32_t off = 0xFFFFFFFFu;
    if (r < N) {
        off = begin_of_rank;
        offs[r] = off;
        if (r == N - 1) offs[N] = off + cnt;
    }
    s_off[wave][lane] = off; s_gid[wave][lane] = g; s_xy[wave][lane] = xy; s_w[wave][lane] = w;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int wave_first = blockIdx.x * blockDim.x + wave * 64;
    if (wave_first >= N) return;
    const int n_here = min(64, N - wave_first);
    const uint32_t begin = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
    const uint32_t end = (uint32_t)__builtin_amdgcn_readlane((int)(off + cnt), n_here - 1);
    for (uint32_t o = begin + lane; o < end; o += 64) {
        // owner = last rank k of the wave with s_off[k] <= o (ranks without instances share their successor's
        // first index, so "last" skips them; lanes beyond N hold 0xFFFFFFFF)
        int k = 0;
#pragma unroll
        for (int step = 32; step > 0; step >>= 1)
            if (s_off[wave][k + step] <= o) k += step;
        const uint32_t i = o - s_off[wave][k], ww = s_w[wave][k], c = s_xy[wave][k];
        const uint32_t row = i / ww, col = i - row * ww;
        tile_keys[o] = ((c >> 16) + row) * (uint32_t)gx + (c & 0xFFFFu) + col;
        emit_gid[o] = s_gid[wave][k];
    }
}

int gsr_launch_emit(int N, int grid_x, int grid_y, const uint32_t* order, const void* scan_ws, uint32_t* offs,
                    const uint2* rank_rect, uint32_t* tile_keys, uint32_t* emit_gid,
                    uint32_t* zero_a, size_t words_a, uint32_t* zero_b, size_t words_b, hipStream_t s) {
    if (N <= 0) return GSR_OK;
    (void)grid_y;
    GsrProfileScope prof(GSR_K_EMIT, s);
    const uint32_t* partial = static_cast<const uint32_t*>(scan_ws);
    const int tiles = (N + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(emit_instances_kernel, dim3((N + SCAN_BLOCK - 1) / SCAN_BLOCK), dim3(SCAN_BLOCK), 0, s, N, grid_x,
                       order, partial, partial + tiles, offs, rank_rect, tile_keys, emit_gid, zero_a, (int)words_a, zero_b,
                       (int)words_b);
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}

// Tile ranges from the sorted tile keys (Gaussian id and emission index of every list entry are the tile sort's own two
// value outputs; the 80-byte records themselves are NOT copied into list order: the render kernels gather them by id).
// Also clears the per-instance row-count bytes the compositing kernel fills in (render_fwd.hip, last wave of a tile;
// instances nobody walked keep 0).
__global__ void __launch_bounds__(256) finalize_bins_kernel(int D, const uint32_t* __restrict__ tile_sorted,
                                                            uint32_t* __restrict__ ranges, uint8_t* __restrict__ slot_cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    // 16 bytes per store from every 16th thread (the region is padded to 256 bytes): one byte per thread cost 0.14 ms at
    // D = 22.6 M, a byte-masked store being charged its whole sector
    if (slot_cnt && (i & 15) == 0) *reinterpret_cast<uint4*>(slot_cnt + i) = make_uint4(0u, 0u, 0u, 0u);
    const uint32_t t = tile_sorted[i];
    const uint32_t tp = i > 0 ? tile_sorted[i - 1] : 0xFFFFFFFFu;
    if (i == 0) ranges[2 * t] = 0;
    else if (tp != t) { ranges[2 * tp + 1] = i; ranges[2 * t] = i; }
    if (i == D - 1) ranges[2 * t + 1] = D;
}

// `ranges` must already be zero (gsr_launch_emit clears it); `slot_cnt` (may be NULL): D bytes to clear
int gsr_launch_finalize_bins(int D, int n_tiles, const uint32_t* tile_keys_sorted, uint32_t* ranges, uint8_t* slot_cnt,
                             hipStream_t s) {
    (void)n_tiles;
    if (D <= 0) return GSR_OK;
    GsrProfileScope prof(GSR_K_FINALIZE, s);
    hipLaunchKernelGGL(finalize_bins_kernel, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, s, D, tile_keys_sorted, ranges, slot_cnt);
    GSR_LAUNCH_CHECK();
    return GSR_OK;
}
