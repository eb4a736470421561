// Developer aid: what do rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for THIS library's access patterns?
// MI355X_MICROARCH.md calibrates only wide coalesced streaming reads (FETCH_SIZE = half the bytes) and 16-B-per-lane
// streaming stores / float atomics (WRITE_SIZE exact) and says "other access widths are uncalibrated: calibrate on a
// known byte count in your own access pattern".  Every kernel below moves a KNOWN number of bytes out of tables far
// larger than the 256 MiB Infinity Cache, each byte touched exactly once:
//   stream16   : 16 B per lane, coalesced                       (the guide's calibrated case; Adam, preprocess)
//   stream4    : 4 B per lane, coalesced                        (id lists, touch words, per-pixel planes)
//   gather80   : one 80-byte record per lane, five 16-B loads   (render_fwd / render_bwd record gather by id)
//   gather4    : one random dword per lane                      (finalize_bins / slot_count style gathers)
//   store80    : 16 lanes write one 80-byte row (dense rows)    (render_bwd gradient rows)
//   store1     : one byte per lane, stride 4                    (render_fwd touch bytes)
// run:  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- ./fetch_calib   (then again with --pmc WRITE_SIZE)
// and divide the counter of each kernel by the byte count this program prints.
// build: hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

__global__ void __launch_bounds__(256) stream16(const float4* __restrict__ in, size_t n, float* out) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void __launch_bounds__(256) stream4(const float* __restrict__ in, size_t n, float* out) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += in[i];
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void __launch_bounds__(256) gather80(const float4* __restrict__ table, const uint32_t* __restrict__ ids, size_t n, float* out) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4* r = table + (size_t)ids[i] * 5;
        const float4 a = r[0], b = r[1], c = r[2], d = r[3], e = r[4];
        acc += a.x + b.y + c.z + d.w + e.x;
    }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void __launch_bounds__(256) gather4(const float* __restrict__ table, const uint32_t* __restrict__ ids, size_t n, float* out) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += table[(size_t)ids[i] * 32];   // one dword per 128-B line
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void __launch_bounds__(256) store80(float* __restrict__ rows, size_t n_rows) {
    const int l16 = threadIdx.x & 15;
    for (size_t r = (blockIdx.x * 256ull + threadIdx.x) >> 4; r < n_rows; r += ((size_t)gridDim.x * 256) >> 4) {
        float* row = rows + r * 20;
        row[l16 < 9 ? l16 : l16 + 2] = (float)r;
        if ((l16 & 7) == 0) row[9 + (l16 >> 3)] = 1.f;
        if (l16 < 2) row[18 + l16] = 0.f;      // (the library leaves these two pad floats unwritten; here every byte is written)
    }
}
__global__ void __launch_bounds__(256) store1(uint8_t* __restrict__ bytes, size_t n_words, int wave_byte) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n_words; i += (size_t)gridDim.x * 256) bytes[i * 4 + wave_byte] = (uint8_t)i;
}

int main() {
    const size_t GiB = 1ull << 30;
    const size_t table_bytes = 2 * GiB;               // far beyond the Infinity Cache
    float* table; hipMalloc(&table, table_bytes);
    hipMemset(table, 0, table_bytes);
    float* out; hipMalloc(&out, 256);
    const int grid = 256 * 8;
    // ids: a random permutation, so every record / line is read exactly once
    const size_t n_rec = table_bytes / 80, n_line = table_bytes / 128;
    std::vector<uint32_t> perm(n_rec);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937 rng(1);
    std::shuffle(perm.begin(), perm.end(), rng);
    uint32_t* ids; hipMalloc(&ids, n_rec * 4);
    hipMemcpy(ids, perm.data(), n_rec * 4, hipMemcpyHostToDevice);
    std::vector<uint32_t> perm2(n_line);
    std::iota(perm2.begin(), perm2.end(), 0u);
    std::shuffle(perm2.begin(), perm2.end(), rng);
    uint32_t* ids2; hipMalloc(&ids2, n_line * 4);
    hipMemcpy(ids2, perm2.data(), n_line * 4, hipMemcpyHostToDevice);
    hipDeviceSynchronize();

    hipLaunchKernelGGL(stream16, dim3(grid), dim3(256), 0, 0, (const float4*)table, table_bytes / 16, out);
    hipLaunchKernelGGL(stream4, dim3(grid), dim3(256), 0, 0, (const float*)table, table_bytes / 4, out);
    hipLaunchKernelGGL(gather80, dim3(grid), dim3(256), 0, 0, (const float4*)table, ids, n_rec, out);
    hipLaunchKernelGGL(gather4, dim3(grid), dim3(256), 0, 0, (const float*)table, ids2, n_line, out);
    hipLaunchKernelGGL(store80, dim3(grid), dim3(256), 0, 0, table, table_bytes / 80);
    hipLaunchKernelGGL(store1, dim3(grid), dim3(256), 0, 0, (uint8_t*)table, table_bytes / 4, 1);
    hipDeviceSynchronize();
    printf("{\"stream16\": {\"read\": %zu}, \"stream4\": {\"read\": %zu}, \"gather80\": {\"read_payload\": %zu, \"read_ids\": %zu, "
           "\"read_64B_sectors\": %zu, \"read_128B_lines_avg\": %zu}, \"gather4\": {\"read_payload\": %zu, \"read_ids\": %zu, "
           "\"read_64B_sectors\": %zu, \"read_128B_lines\": %zu}, \"store80\": {\"written\": %zu}, \"store1\": {\"written_payload\": %zu, "
           "\"lines_touched_bytes\": %zu}}\n",
           table_bytes, table_bytes, n_rec * 80, n_rec * 4, n_rec * 128, n_rec * 192, n_line * 4, n_line * 4, n_line * 64, n_line * 128,
           (table_bytes / 80) * 80, table_bytes / 4, table_bytes);
    return 0;
}
