// mma_tile.h -- LDS -> v_mfma_f32_32x32x2_f32 inner step shared by the conv and dense GEMM kernels.
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;   // K depth of one LDS stage

// Operand fragments of v_mfma_f32_32x32x2_f32: lane l holds A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; with k-major LDS tiles both are 32 consecutive floats per half-wave.
// LDS -> MFMA: one 16-deep step of the wave's TM x TN tiles
template <int TM, int TN, int LDA, int LDB, int KB = BK>
__device__ __forceinline__ void mma_step(const float (*As)[LDA], const float (*Bs)[LDB], f32x16 (&acc)[TM][TN],
                                         int a_col, int b_col, int half) {
#pragma unroll
    for (int kk = 0; kk < KB; kk += 2) {
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As[kk + half][a_col + 32 * i];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Bs[kk + half][b_col + 32 * j];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}

