// Dev lab: fp32 MFMA GEMM main-loop variants, C[M][N] = A[M][K] B[K][N] (row-major, M % BM == N % BN == K % KB == 0).
// v2: K-permuted 16-byte LDS reads of A (K-contiguous rows), N-interleaved 16-byte reads of B: one ds_read_b128 feeds 4 MFMAs.
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int N> struct VecT;
template <> struct VecT<4> { typedef f4 t; };
template <> struct VecT<2> { typedef f2 t; };
template <> struct VecT<1> { typedef float t; };
__device__ __forceinline__ void unpack(const f4& v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
__device__ __forceinline__ void unpack(const f2& v, float (&o)[2]) { o[0] = v.x; o[1] = v.y; }
__device__ __forceinline__ void unpack(const float& v, float (&o)[1]) { o[0] = v; }

template <int WM, int WN, int TM, int TN, int KB, int NACC = 1>
__global__ __launch_bounds__(256) void gemm_v2(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                               int M, int N, int K) {
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN, LDA = KB + 4, LDB = BN + 4;
    constexpr int KQ = KB / 4, AP = BM * KQ / 256, BP = KB * BN / 4 / 256;
    typedef typename VecT<TN>::t bvec;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float (*As)[BM][LDA] = reinterpret_cast<float (*)[BM][LDA]>(smem);
    float (*Bs)[KB][LDB] = reinterpret_cast<float (*)[KB][LDB]>(smem + 2 * BM * LDA);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int ntn = N / BN;
    const int bx = blockIdx.x / ntn, by = blockIdx.x % ntn;
    const int m0 = bx * BM, n0 = by * BN;
    f32x16 acc[NACC][TM][TN];
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][i][j][r] = 0.f;
    f4 ra0[AP], rb0[BP], ra1[AP], rb1[BP];
    auto load_tiles = [&](int ks, f4 (&ra)[AP], f4 (&rb)[BP]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int idx = tid + 256 * i, r = idx / KQ, kq = idx % KQ;
            ra[i] = *reinterpret_cast<const f4*>(A + (long)(m0 + r) * K + ks * KB + kq * 4);
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            const int idx = tid + 256 * j, br = idx / (BN / 4), bc = idx % (BN / 4);
            rb[j] = *reinterpret_cast<const f4*>(B + (long)(ks * KB + br) * N + n0 + bc * 4);
        }
    };
    auto store_tiles = [&](int buf, const f4 (&ra)[AP], const f4 (&rb)[BP]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int idx = tid + 256 * i, r = idx / KQ, kq = idx % KQ;
            *reinterpret_cast<f4*>(&As[buf][r][kq * 4]) = ra[i];
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            const int idx = tid + 256 * j, br = idx / (BN / 4), bc = idx % (BN / 4);
            *reinterpret_cast<f4*>(&Bs[buf][br][bc * 4]) = rb[j];
        }
    };
    const int a_row = wm * 32 * TM + l31, b_col = wn * 32 * TN + TN * l31;
    auto mma = [&](int buf) __attribute__((always_inline)) {
        float a[2][TM][4], b[2][4][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) unpack(*reinterpret_cast<const f4*>(&As[buf][a_row + 32 * i][4 * half]), a[0][i]);
#pragma unroll
        for (int q = 0; q < 4; ++q) unpack(*reinterpret_cast<const bvec*>(&Bs[buf][4 * half + q][b_col]), b[0][q]);
#pragma unroll
        for (int jg = 0; jg < KB / 8; ++jg) {
            const int cur = jg & 1, nxt = cur ^ 1;
            if (jg + 1 < KB / 8) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    unpack(*reinterpret_cast<const f4*>(&As[buf][a_row + 32 * i][8 * (jg + 1) + 4 * half]), a[nxt][i]);
#pragma unroll
                for (int q = 0; q < 4; ++q) unpack(*reinterpret_cast<const bvec*>(&Bs[buf][8 * (jg + 1) + 4 * half + q][b_col]), b[nxt][q]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[q % NACC][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i][q], b[cur][q][j], acc[q % NACC][i][j], 0, 0, 0);
        }
    };
    const int nks = K / KB;
    // loads and stores are UNCONDITIONAL (the step index is clamped at the tail): with `if (ks + 2 < nks) load` the waitcnt
    // insertion has to assume the branch was skipped and drains vmcnt to 0 before every LDS store -- the two-step lookahead is lost
    load_tiles(0, ra0, rb0);
    store_tiles(0, ra0, rb0);
    load_tiles(min(1, nks - 1), ra1, rb1);
    __syncthreads();
    int ks = 0;
    for (; ks + 1 < nks; ks += 2) {
        load_tiles(min(ks + 2, nks - 1), ra0, rb0);
        mma(0);
        store_tiles(1, ra1, rb1);
        __syncthreads();
        load_tiles(min(ks + 3, nks - 1), ra1, rb1);
        mma(1);
        store_tiles(0, ra0, rb0);
        __syncthreads();
    }
    if (ks < nks) mma(0);
    f32x16 (&accs)[TM][TN] = acc[0];
#pragma unroll
    for (int c = 1; c < NACC; ++c)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) accs[i][j] += acc[c][i][j];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 * TM + 32 * i + 4 * half + (r & 3) + 8 * (r >> 2);
            float* dst = C + (long)row * N + n0 + b_col;
            if (TN == 4) *reinterpret_cast<f4*>(dst) = f4{accs[i][0][r], accs[i][1 % TN][r], accs[i][2 % TN][r], accs[i][3 % TN][r]};
            else if (TN == 2) *reinterpret_cast<f2*>(dst) = f2{accs[i][0][r], accs[i][1 % TN][r]};
            else dst[0] = accs[i][0][r];
        }
}


// v3: three LDS buffers, ONE barrier per step placed before the last fragment group: the step-(s+1) tile is written at the top
// of step s, the barrier sits in the middle of the MFMA stream, and the first fragments of step s+1 are read before step s ends.
template <int WM, int WN, int TM, int TN, int KB, int NSET>
__global__ __launch_bounds__(256) void gemm_v3(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                               int M, int N, int K) {
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN, LDA = KB + 4, LDB = BN + 4;
    constexpr int KQ = KB / 4, AP = BM * KQ / 256, BP = KB * BN / 4 / 256, G = KB / 8;
    static_assert(G % 2 == 0, "fragment register parity");
    typedef typename VecT<TN>::t bvec;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float (*As)[BM][LDA] = reinterpret_cast<float (*)[BM][LDA]>(smem);
    float (*Bs)[KB][LDB] = reinterpret_cast<float (*)[KB][LDB]>(smem + 3 * BM * LDA);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int ntn = N / BN;
    const int bx = blockIdx.x / ntn, by = blockIdx.x % ntn;
    const int m0 = bx * BM, n0 = by * BN;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f4 ra[NSET][AP], rb[NSET][BP];
    auto load_tiles = [&](int ks, f4 (&ra)[AP], f4 (&rb)[BP]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int idx = tid + 256 * i, r = idx / KQ, kq = idx % KQ;
            ra[i] = *reinterpret_cast<const f4*>(A + (long)(m0 + r) * K + ks * KB + kq * 4);
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            const int idx = tid + 256 * j, br = idx / (BN / 4), bc = idx % (BN / 4);
            rb[j] = *reinterpret_cast<const f4*>(B + (long)(ks * KB + br) * N + n0 + bc * 4);
        }
    };
    auto store_tiles = [&](int buf, const f4 (&ra)[AP], const f4 (&rb)[BP]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int idx = tid + 256 * i, r = idx / KQ, kq = idx % KQ;
            *reinterpret_cast<f4*>(&As[buf][r][kq * 4]) = ra[i];
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            const int idx = tid + 256 * j, br = idx / (BN / 4), bc = idx % (BN / 4);
            *reinterpret_cast<f4*>(&Bs[buf][br][bc * 4]) = rb[j];
        }
    };
    const int a_row = wm * 32 * TM + l31, b_col = wn * 32 * TN + TN * l31;
    float a[2][TM][4], b[2][4][TN];
    auto frag = [&](int buf, int g, int set) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TM; ++i) unpack(*reinterpret_cast<const f4*>(&As[buf][a_row + 32 * i][8 * g + 4 * half]), a[set][i]);
#pragma unroll
        for (int q = 0; q < 4; ++q) unpack(*reinterpret_cast<const bvec*>(&Bs[buf][8 * g + 4 * half + q][b_col]), b[set][q]);
    };
    const int nks = K / KB;
    // prologue: tile 0 in LDS buffer 0, register set(s) hold steps 1 (and 2)
    load_tiles(0, ra[0], rb[0]);
    store_tiles(0, ra[0], rb[0]);
    load_tiles(min(1, nks - 1), ra[0], rb[0]);
    if (NSET == 2) load_tiles(min(2, nks - 1), ra[1], rb[1]);
    __syncthreads();
    frag(0, 0, 0);
    int cur = 0;                                   // LDS buffer of this step
    auto step = [&](int s, int set) __attribute__((always_inline)) {
        const int nxt = cur == 2 ? 0 : cur + 1;
        store_tiles(nxt, ra[set], rb[set]);                          // step s+1
        load_tiles(min(s + 1 + NSET, nks - 1), ra[set], rb[set]);    // step s+1+NSET
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (g == G - 1) __syncthreads();
            if (g + 1 < G) frag(cur, g + 1, (g + 1) & 1);
            else frag(nxt, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][i][q], b[g & 1][q][j], acc[i][j], 0, 0, 0);
        }
        cur = nxt;
    };
    int s = 0;
    if (NSET == 2) {
        for (; s + 1 < nks; s += 2) { step(s, 0); step(s + 1, 1); }
        if (s < nks) step(s, 0);
    } else {
        for (; s < nks; ++s) step(s, 0);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 * TM + 32 * i + 4 * half + (r & 3) + 8 * (r >> 2);
            float* dst = C + (long)row * N + n0 + b_col;
            if (TN == 4) *reinterpret_cast<f4*>(dst) = f4{acc[i][0][r], acc[i][1 % TN][r], acc[i][2 % TN][r], acc[i][3 % TN][r]};
            else if (TN == 2) *reinterpret_cast<f2*>(dst) = f2{acc[i][0][r], acc[i][1 % TN][r]};
            else dst[0] = acc[i][0][r];
        }
}

#define LAUNCH3(WM, WN, TM, TN, KB, NSET)                                                                                  \
    {                                                                                                                      \
        constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;                                                                \
        const size_t lds = (3 * BM * (KB + 4) + 3 * KB * (BN + 4)) * sizeof(float);                                        \
        if (M % BM || N % BN || K % KB) return -2;                                                                         \
        hipFuncSetAttribute((const void*)gemm_v3<WM, WN, TM, TN, KB, NSET>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((gemm_v3<WM, WN, TM, TN, KB, NSET>), dim3((M / BM) * (N / BN)), dim3(256), lds, (hipStream_t)stream, A, B, C, M, N, K); \
        return hipGetLastError() == hipSuccess ? 0 : -1;                                                                   \
    }

#define LAUNCH(WM, WN, TM, TN, KB)                                                                                         \
    {                                                                                                                      \
        constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;                                                                \
        const size_t lds = (2 * BM * (KB + 4) + 2 * KB * (BN + 4)) * sizeof(float);                                        \
        if (M % BM || N % BN || K % KB) return -2;                                                                         \
        hipFuncSetAttribute((const void*)gemm_v2<WM, WN, TM, TN, KB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((gemm_v2<WM, WN, TM, TN, KB>), dim3((M / BM) * (N / BN)), dim3(256), lds, (hipStream_t)stream, A, B, C, M, N, K); \
        return hipGetLastError() == hipSuccess ? 0 : -1;                                                                   \
    }

#define LAUNCHN(WM, WN, TM, TN, KB, NA)                                                                                         \
    {                                                                                                                      \
        constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;                                                                \
        const size_t lds = (2 * BM * (KB + 4) + 2 * KB * (BN + 4)) * sizeof(float);                                        \
        if (M % BM || N % BN || K % KB) return -2;                                                                         \
        hipFuncSetAttribute((const void*)gemm_v2<WM, WN, TM, TN, KB, NA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((gemm_v2<WM, WN, TM, TN, KB, NA>), dim3((M / BM) * (N / BN)), dim3(256), lds, (hipStream_t)stream, A, B, C, M, N, K); \
        return hipGetLastError() == hipSuccess ? 0 : -1;                                                                   \
    }


extern "C" int gemm_lab(int cfg, const float* A, const float* B, float* C, int M, int N, int K, void* stream) {
    switch (cfg) {
        case 0: LAUNCH(2, 2, 2, 4, 16)   // 128 x 256
        case 1: LAUNCH(2, 2, 2, 4, 32)
        case 2: LAUNCH(4, 1, 1, 4, 16)   // 128 x 128, wave 32 x 128
        case 3: LAUNCH(2, 2, 2, 2, 16)   // 128 x 128, wave 64 x 64
        case 4: LAUNCH(2, 2, 2, 2, 32)
        case 5: LAUNCH(2, 2, 1, 2, 16)   // 64 x 128
        case 6: LAUNCH(2, 2, 1, 1, 16)   // 64 x 64
        case 7: LAUNCH(2, 2, 1, 1, 32)
        case 8: LAUNCH(4, 1, 1, 2, 32)   // 128 x 64
        case 9: LAUNCH(2, 2, 1, 2, 32)   // 64 x 128, KB 32
        case 10: LAUNCH(1, 4, 4, 1, 16)  // 128 x 128, wave 128 x 32
        case 11: LAUNCH(2, 2, 4, 2, 16)  // 256 x 128
        case 12: LAUNCHN(2, 2, 1, 1, 16, 2)   // 64 x 64, two accumulators
        case 13: LAUNCHN(2, 2, 1, 1, 16, 4)
        case 14: LAUNCHN(2, 2, 1, 1, 32, 2)
        case 15: LAUNCHN(2, 2, 1, 2, 16, 2)   // 64 x 128
        case 20: LAUNCH3(2, 2, 2, 2, 16, 1)   // 128 x 128
        case 21: LAUNCH3(2, 2, 2, 2, 16, 2)
        case 22: LAUNCH3(2, 2, 2, 2, 32, 1)
        case 23: LAUNCH3(2, 2, 1, 1, 16, 2)   // 64 x 64
        case 24: LAUNCH3(2, 2, 1, 1, 32, 1)
        case 25: LAUNCH3(2, 2, 1, 1, 32, 2)
        case 26: LAUNCH3(2, 2, 1, 2, 16, 2)   // 64 x 128
        case 27: LAUNCH3(2, 2, 1, 2, 32, 1)
        case 28: LAUNCH3(2, 2, 4, 2, 16, 1)   // 256 x 128
        case 29: LAUNCH3(2, 2, 2, 4, 16, 1)   // 128 x 256
    }
    return -3;
}
