// Developer aid: operand / result lane maps of v_mfma_f32_16x16x1_4b_f32 (__builtin_amdgcn_mfma_f32_16x16x1f32), checked
// with asymmetric integer data -- the wide-payload forward accumulates features with it (render_fwd.hip).
// Expected: A: lane l -> A[block l >> 4][i = l & 15]; B: lane l -> B[block l >> 4][j = l & 15];
//           D: register v of lane l -> D[block v >> 2][i = 4 (l >> 4) + (v & 3)][j = l & 15].
// build: hipcc --offload-arch=gfx950 -O2 mfma_layout.hip -o mfma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, float* D) {
    const int l = threadIdx.x;
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_16x16x1f32(A[l], B[l], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x1f32(A[64 + l], B[64 + l], acc, 0, 0, 0);   // second k step accumulates
    for (int v = 0; v < 16; ++v) D[v * 64 + l] = acc[v];
}
int main() {
    float hA[128], hB[128], hD[1024], *dA, *dB, *dD;
    for (int s = 0; s < 2; ++s)
        for (int l = 0; l < 64; ++l) { hA[s * 64 + l] = float(1 + l + 100 * s); hB[s * 64 + l] = float(3 + 2 * l + 7 * s); }
    (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dD, sizeof hD);
    (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int v = 0; v < 16; ++v)
        for (int l = 0; l < 64; ++l) {
            const int b = v >> 2, i = 4 * (l >> 4) + (v & 3), j = l & 15;
            float want = 0;
            for (int s = 0; s < 2; ++s) want += hA[s * 64 + 16 * b + i] * hB[s * 64 + 16 * b + j];
            if (hD[v * 64 + l] != want) { if (bad < 8) printf("mismatch v %d lane %d: got %g want %g\n", v, l, hD[v * 64 + l], want); ++bad; }
        }
    printf("mfma_f32_16x16x1 (4 blocks) lane map: %s (%d mismatches)\n", bad ? "DIFFERENT FROM THE ASSUMED MAP" : "as assumed", bad);
    return bad != 0;
}
