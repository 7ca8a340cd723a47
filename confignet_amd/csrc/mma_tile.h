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
    // operands of k-pair kk+2 are fetched from LDS BEFORE the MFMAs of k-pair kk are issued (two register sets), so
    // the wave never sits in s_waitcnt lgkmcnt(0) between two MFMA groups
    float a[2][TM], b[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) a[0][i] = As[half][a_col + 32 * i];
#pragma unroll
    for (int j = 0; j < TN; ++j) b[0][j] = Bs[half][b_col + 32 * j];
#pragma unroll
    for (int kk = 0; kk < KB; kk += 2) {
        const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
        if (kk + 2 < KB) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[nxt][i] = As[kk + 2 + half][a_col + 32 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[nxt][j] = Bs[kk + 2 + half][b_col + 32 * j];
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the fetch of the next operands ahead of this group's MFMAs
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
    }
}
