// Developer aid: scalar-ALU issue rate of an MI355X CU, alone and mixed with vector work -- the compositing kernels
// keep their per-block to-do masks in SGPRs, so they issue almost as many SALU as VALU instructions.
// build: hipcc --offload-arch=gfx950 -O3 salu_rate.hip -o salu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 2048
// KIND 0: 16 SALU per iteration; 1: 16 VALU; 2: 16 SALU + 16 VALU interleaved; 3: 8 SALU + 16 VALU
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, float a, float b, unsigned s0) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    unsigned y0 = s0, y1 = s0 + 1, y2 = s0 + 2, y3 = s0 + 3;
    for (int i = 0; i < ITER; ++i) {
#define SA(y) asm volatile("s_add_u32 %0, %0, %1" : "+s"(y) : "s"(s0) : "scc");
#define VA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
        if (KIND == 0) { SA(y0) SA(y1) SA(y2) SA(y3) SA(y0) SA(y1) SA(y2) SA(y3) SA(y0) SA(y1) SA(y2) SA(y3) SA(y0) SA(y1) SA(y2) SA(y3) }
        if (KIND == 1) { VA(x0) VA(x1) VA(x2) VA(x3) VA(x0) VA(x1) VA(x2) VA(x3) VA(x0) VA(x1) VA(x2) VA(x3) VA(x0) VA(x1) VA(x2) VA(x3) }
        if (KIND == 2) { VA(x0) SA(y0) VA(x1) SA(y1) VA(x2) SA(y2) VA(x3) SA(y3) VA(x0) SA(y0) VA(x1) SA(y1) VA(x2) SA(y2) VA(x3) SA(y3)
                         VA(x0) SA(y0) VA(x1) SA(y1) VA(x2) SA(y2) VA(x3) SA(y3) VA(x0) SA(y0) VA(x1) SA(y1) VA(x2) SA(y2) VA(x3) SA(y3) }
        if (KIND == 3) { VA(x0) SA(y0) VA(x1) VA(x2) SA(y1) VA(x3) VA(x0) SA(y2) VA(x1) VA(x2) SA(y3) VA(x3)
                         VA(x0) SA(y0) VA(x1) VA(x2) SA(y1) VA(x3) VA(x0) SA(y2) VA(x1) VA(x2) SA(y3) VA(x3) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + (float)(y0 + y1 + y2 + y3);
}
template <int KIND> void run(const char* name, float* out, int wg_per_cu, double n_s, double n_v) {
    const int blocks = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, 3u);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, 3u);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = double(blocks) * 4, cu_cycles = ms * 1e-3 * 2.4e9 * 256;
    printf("%-28s %d waves/SIMD %8.3f ms   SALU %.2f /cycle/CU   VALU %.2f /cycle/CU\n", name, wg_per_cu, ms,
           waves * ITER * n_s / cu_cycles, waves * ITER * n_v / cu_cycles);
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {2, 4, 8}) {
        run<0>("SALU only", out, w, 16, 0); run<1>("VALU only", out, w, 0, 16);
        run<2>("SALU:VALU 1:1", out, w, 16, 16); run<3>("SALU:VALU 1:2", out, w, 8, 16);
    }
    return 0;
}
