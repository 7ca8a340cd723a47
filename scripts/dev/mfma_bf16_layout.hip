// Dev check of the operand layout of v_mfma_f32_32x32x16_bf16 (gfx950): C[32x32] = A[32x16] * B[16x32] with
// lane l holding A[i = l%32][k = 8*(l/32) .. +7] and B[k = 8*(l/32) .. +7][j = l%32]; C as for 32x32x2 f32.
// build: hipcc --offload-arch=gfx950 -O2 mfma_bf16_layout.hip -o /tmp/mfma_layout && /tmp/mfma_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ unsigned short f2bf(float x) { unsigned u = __float_as_uint(x); return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); }
__global__ void k(const float* A, const float* B, float* C) {   // A[32][16], B[16][32] row-major, C[32][32]
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    union { unsigned short s[8]; bf16x8 v; } a, b;
    for (int e = 0; e < 8; ++e) { a.s[e] = f2bf(A[i * 16 + 8 * h + e]); b.s[e] = f2bf(B[(8 * h + e) * 32 + i]); }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[(4 * h + (r & 3) + 8 * (r >> 2)) * 32 + i] = acc[r];
}
int main() {
    float hA[512], hB[512], hC[1024], ref[1024];
    for (int x = 0; x < 512; ++x) { hA[x] = (float)((x * 7) % 13 - 6) * 0.25f; hB[x] = (float)((x * 5) % 11 - 5) * 0.5f; }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int kk = 0; kk < 16; ++kk) s += hA[i * 16 + kk] * hB[kk * 32 + j]; ref[i * 32 + j] = s; }
    float *dA, *dB, *dC;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
    double err = 0; for (int x = 0; x < 1024; ++x) err = fmax(err, fabs(hC[x] - ref[x]));
    printf("mfma_f32_32x32x16_bf16 layout check: max abs err %.3g (%s)\n", err, err < 1e-4 ? "OK" : "MISMATCH");
    return err < 1e-4 ? 0 : 1;
}
