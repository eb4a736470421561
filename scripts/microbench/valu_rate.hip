// Developer aid: measured wave64 VALU issue rate of one MI355X SIMD for the instruction kinds the compositing kernels
// are made of (fp32 FMA, v_cndmask, DPP add, v_rcp / v_exp).  build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4096
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITER; ++i) {
        if (KIND == 0) {   // 8 independent FMAs
            x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
            x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
        } else if (KIND == 1) {   // 8 selects
            x0 = x0 > a ? x1 : x0; x1 = x1 > a ? x2 : x1; x2 = x2 > a ? x3 : x2; x3 = x3 > a ? x4 : x3;
            x4 = x4 > a ? x5 : x4; x5 = x5 > a ? x6 : x5; x6 = x6 > a ? x7 : x6; x7 = x7 > a ? x0 : x7;
        } else if (KIND == 2) {   // 8 transcendentals
            x0 = __builtin_amdgcn_rcpf(x0); x1 = __builtin_amdgcn_rcpf(x1); x2 = __builtin_amdgcn_rcpf(x2); x3 = __builtin_amdgcn_rcpf(x3);
            x4 = __builtin_amdgcn_rcpf(x4); x5 = __builtin_amdgcn_rcpf(x5); x6 = __builtin_amdgcn_rcpf(x6); x7 = __builtin_amdgcn_rcpf(x7);
        } else {   // 8 DPP adds
            x0 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x0), 0xB1, 0xf, 0xf, true));
            x1 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x1), 0xB1, 0xf, 0xf, true));
            x2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x2), 0x4E, 0xf, 0xf, true));
            x3 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x3), 0x4E, 0xf, 0xf, true));
            x4 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x4), 0x141, 0xf, 0xf, true));
            x5 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x5), 0x141, 0xf, 0xf, true));
            x6 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x6), 0x140, 0xf, 0xf, true));
            x7 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x7), 0x140, 0xf, 0xf, true));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int KIND> void run(const char* name, float* out) {
    const int blocks = 256 * 8;   // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = double(blocks) * 4 * ITER * 8;
    printf("%-14s %8.3f ms  %7.2f G wave-instr/s  = %.2f cycles per wave64 instruction per SIMD at 2.4 GHz\n", name, ms,
           wave_instr / ms / 1e6, 1024.0 * 2.4e9 / (wave_instr / (ms * 1e-3)));
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32", out); run<1>("v_cndmask", out); run<2>("v_rcp_f32", out); run<3>("v_add_f32 dpp", out);
    return 0;
}
