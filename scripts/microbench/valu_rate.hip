// Developer aid: measured wave64 VALU issue rate of the MI355X for the instruction kinds the compositing kernels are
// made of, and a KNOWN vector-instruction count per kernel to calibrate the SQ_INSTS_VALU counter on.
// Every timed instruction is inline assembly: round 2's version left the loop to the compiler, which SLP-packed the
// "eight independent v_fma_f32" into v_pk_fma_f32 (64 packed, 0 scalar in the disassembly) and reported the packed rate
// counted twice.  scripts/valu_calib.sh builds this file, checks the disassembly, runs it, and runs
// ONE rocprofv3 --pmc SQ_INSTS_VALU pass over it.
// Round 4: every loop runs >= 5 ms (ITER 65536) and every wave brackets its loop with s_memtime
// (shader cycles) and s_memrealtime (100 MHz): the ratio is the clock the chip HOLDS under that load, and the rate is
// quoted in cycles per wave64 instruction per SIMD at THAT clock next to the nominal 2.4 GHz figure.
// build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define ITER 65536
#define REP8(S) S S S S S S S S
template <int KIND>
__global__ void __launch_bounds__(256) valu_kernel(float* out, float a, float b, unsigned long long* stamps) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
#pragma unroll 1
    for (int i = 0; i < ITER; ++i) {
        if (KIND == 0) {          // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (KIND == 1) {   // 8 independent v_pk_fma_f32 (two lanes' worth of FMAs each): 4 registers pairs, twice
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
        } else if (KIND == 2) {   // 8 dependent-pair (compare, select) = 16 VALU
            asm volatile(REP8("v_cmp_gt_f32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %1, vcc\n")
                         : "+v"(x0), "+v"(x1) : "v"(a) : "vcc");
        } else if (KIND == 3) {   // 8 independent v_rcp_f32
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                         "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (KIND == 5) {   // 8 independent v_cndmask_b32 on a constant vcc
            asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (KIND == 6) {   // 8 v_cmp_gt_f32 (each writes vcc; nothing reads it)
            asm volatile(REP8("v_cmp_gt_f32 vcc, %0, %1\n") : : "v"(x0), "v"(a) : "vcc");
        } else if (KIND == 7) {   // 8 independent v_mul_f32
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (KIND == 8) {   // 4 v_fma_f32 + 4 v_add_f32 dpp, interleaved (does the mix pay the sum of its parts?)
            asm volatile("s_nop 1\n"
                         "v_fma_f32 %0, %0, %8, %9\n v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_fma_f32 %1, %1, %8, %9\n v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                         "v_fma_f32 %2, %2, %8, %9\n v_add_f32_dpp %6, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xf\n"
                         "v_fma_f32 %3, %3, %8, %9\n v_add_f32_dpp %7, %7, %7 row_mirror row_mask:0xf bank_mask:0xf"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (KIND == 9) {   // 8 independent v_exp_f32
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                         "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (KIND == 10) {  // 8 independent v_mov_b32 (register to register)
            asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                         "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        } else if (KIND == 11) {  // 4 v_fma_f32 + 4 v_cndmask_b32 (constant vcc), interleaved
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_cndmask_b32 %4, %4, %8, vcc\n v_fma_f32 %1, %1, %8, %9\n v_cndmask_b32 %5, %5, %8, vcc\n"
                         "v_fma_f32 %2, %2, %8, %9\n v_cndmask_b32 %6, %6, %8, vcc\n v_fma_f32 %3, %3, %8, %9\n v_cndmask_b32 %7, %7, %8, vcc"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (KIND == 12) {  // 8 v_cndmask_b32_e64 on ONE constant SGPR pair (not vcc)
            asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[4:5]\n v_cndmask_b32_e64 %1, %1, %8, s[4:5]\n v_cndmask_b32_e64 %2, %2, %8, s[4:5]\n v_cndmask_b32_e64 %3, %3, %8, s[4:5]\n"
                         "v_cndmask_b32_e64 %4, %4, %8, s[4:5]\n v_cndmask_b32_e64 %5, %5, %8, s[4:5]\n v_cndmask_b32_e64 %6, %6, %8, s[4:5]\n v_cndmask_b32_e64 %7, %7, %8, s[4:5]"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (KIND == 13) {  // 4 v_cndmask in a row, then 4 v_fma in a row (the shape the compiler emits: runs of selects)
            asm volatile("v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                         "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (KIND == 14) {  // pairs: 2 v_cndmask, 2 v_fma, 2 v_cndmask, 2 v_fma
            asm volatile("v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n"
                         "v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (KIND == 15) {  // 8 v_cndmask_b32 whose two data sources differ from the destination (no read of its own result)
            asm volatile("v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n"
                         "v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (KIND == 16) {  // 4 v_mul_f32 by a 0/1 mask register + 4 v_fma_f32 (the arithmetic alternative to a select)
            asm volatile("v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (KIND == 17) {  // 8 v_fma_f32 with an SGPR operand (is it the SGPR read, not the select?)
            asm volatile("v_fma_f32 %0, %0, s4, %8\n v_fma_f32 %1, %1, s4, %8\n v_fma_f32 %2, %2, s4, %8\n v_fma_f32 %3, %3, s4, %8\n"
                         "v_fma_f32 %4, %4, s4, %8\n v_fma_f32 %5, %5, s4, %8\n v_fma_f32 %6, %6, s4, %8\n v_fma_f32 %7, %7, s4, %8"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else {                  // 8 independent v_add_f32 with a DPP operand (quad_perm / row mirror)
            asm volatile("s_nop 1\n"
                         "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %4, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %5, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %6, %6, %6 row_mirror row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %7, %7, %7 row_mirror row_mask:0xf bank_mask:0xf"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* st = stamps + 2 * (blockIdx.x * 4 + (threadIdx.x >> 6));
        st[0] = c1 - c0; st[1] = r1 - r0;
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int KIND> void run(const char* name, float* out, unsigned long long* stamps, int per_iter) {
    const int blocks = 256 * 8;   // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(valu_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, stamps);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(valu_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, stamps);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks * 4);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (size_t w = 0; w < h.size() / 2; ++w) { cyc += (double)h[2 * w]; real += (double)h[2 * w + 1]; }
    const double clock_mhz = cyc / real * 100.0;                 // shader cycles per 10 ns tick
    const double wave_instr = double(blocks) * 4 * ITER * per_iter;
    const double rate = wave_instr / (ms * 1e-3);
    // per wave: its own cycles / its own instructions, times the 8 waves that share the SIMD
    const double cyc_per_instr_inwave = (cyc / (h.size() / 2)) / (double(ITER) * per_iter) / 8.0;
    printf("%-16s %8.3f ms  %7.2f G wave-instr/s  clock %6.0f MHz  = %.2f cycles per wave64 instruction per SIMD at that clock "
           "(in-wave stamps: %.2f; %.2f at the nominal 2.4 GHz)   loop VALU per launch %.0f\n",
           name, ms, rate / 1e9, clock_mhz, 1024.0 * clock_mhz * 1e6 / rate, cyc_per_instr_inwave, 1024.0 * 2.4e9 / rate, wave_instr);
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    unsigned long long* stamps; hipMalloc(&stamps, 256 * 8 * 4 * 2 * 8);
    run<0>("v_fma_f32", out, stamps, 8); run<1>("v_pk_fma_f32", out, stamps, 8); run<2>("v_cmp+v_cndmask", out, stamps, 16);
    run<3>("v_rcp_f32", out, stamps, 8); run<4>("v_add_f32 dpp", out, stamps, 8);
    if (getenv("VALU_RATE_ALL")) {   // round 4: the other instruction kinds of the compositing loops (not part of the PMC calibration)
        run<5>("v_cndmask_b32", out, stamps, 8); run<6>("v_cmp_gt_f32", out, stamps, 8); run<7>("v_mul_f32", out, stamps, 8);
        run<8>("4 fma + 4 dpp", out, stamps, 8); run<9>("v_exp_f32", out, stamps, 8); run<10>("v_mov_b32", out, stamps, 8);
        run<11>("4 fma + 4 cndmask", out, stamps, 8);
        run<12>("cndmask e64 sgpr", out, stamps, 8); run<13>("4 cndmask, 4 fma", out, stamps, 8); run<14>("2cnd 2fma 2cnd 2fma", out, stamps, 8);
        run<15>("cndmask dst!=src", out, stamps, 8); run<16>("4 mul + 4 fma", out, stamps, 8); run<17>("v_fma_f32 sgpr src", out, stamps, 8);
    }
    return 0;
}
