// Developer aid: is v_mfma_f32_16x16x1_4b_f32 bit-for-bit an fmaf per product?  Random floats, 64 k-steps accumulated,
// compared with the fmaf chain and with the (round(a*b) + c) chain on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define KS 64
__global__ void k(const float* A, const float* B, float* D) {
    const int l = threadIdx.x;
    f32x16 acc = {0};
    for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x1f32(A[s * 64 + l], B[s * 64 + l], acc, 0, 0, 0);
    for (int v = 0; v < 16; ++v) D[v * 64 + l] = acc[v];
}
int main() {
    static float hA[KS * 64], hB[KS * 64], hD[1024];
    float *dA, *dB, *dD;
    srand(1);
    for (int i = 0; i < KS * 64; ++i) { hA[i] = rand() / (float)RAND_MAX; hB[i] = (rand() % 3 == 0) ? 0.f : rand() / (float)RAND_MAX * 0.1f; }
    (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dD, sizeof hD);
    (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad_fma = 0, bad_mul_add = 0;
    for (int v = 0; v < 16; ++v)
        for (int l = 0; l < 64; ++l) {
            const int b = v >> 2, i = 4 * (l >> 4) + (v & 3), j = l & 15;
            float f = 0.f, g = 0.f;
            for (int s = 0; s < KS; ++s) {
                const float a = hA[s * 64 + 16 * b + i], bb = hB[s * 64 + 16 * b + j];
                f = fmaf(a, bb, f);
                volatile float pr = a * bb; g = pr + g;
            }
            if (memcmp(&f, &hD[v * 64 + l], 4)) ++bad_fma;
            if (memcmp(&g, &hD[v * 64 + l], 4)) ++bad_mul_add;
        }
    printf("mfma_f32_16x16x1: %d of 1024 differ from the fmaf chain, %d differ from the mul-then-add chain\n", bad_fma, bad_mul_add);
    return 0;
}
