// Developer aid: measured wave64 VALU issue rate of the MI355X for the instruction kinds the compositing kernels are
// made of, and a KNOWN vector-instruction count per kernel to calibrate the SQ_INSTS_VALU counter on.
// Every timed instruction is inline assembly: round 2's version left the loop to the compiler, which SLP-packed the
// "eight independent v_fma_f32" into v_pk_fma_f32 (64 packed, 0 scalar in the disassembly) and reported the packed rate
// counted twice.  scripts/valu_calib.sh builds this file, checks the disassembly, runs it, and runs
// ONE rocprofv3 --pmc SQ_INSTS_VALU pass over it.
// build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
#define REP8(S) S S S S S S S S
template <int KIND>
__global__ void __launch_bounds__(256) valu_kernel(float* out, float a, float b) {
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
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int KIND> void run(const char* name, float* out, int per_iter) {
    const int blocks = 256 * 8;   // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(valu_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(valu_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = double(blocks) * 4 * ITER * per_iter;
    printf("%-16s %8.3f ms  %7.2f G wave-instr/s  = %.2f cycles per wave64 instruction per SIMD at 2.4 GHz   loop VALU per launch %.0f\n",
           name, ms, wave_instr / ms / 1e6, 1024.0 * 2.4e9 / (wave_instr / (ms * 1e-3)), wave_instr);
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32", out, 8); run<1>("v_pk_fma_f32", out, 8); run<2>("v_cmp+v_cndmask", out, 16); run<3>("v_rcp_f32", out, 8);
    run<4>("v_add_f32 dpp", out, 8);
    return 0;
}
