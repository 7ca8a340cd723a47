// gemm.hip -- dense GEMM  C = act(op(A) op(B) + bias)  for the Keras Dense layers of the path
// (AdaIN MLPs, discriminator heads, latent regressor head, latent discriminator, synthetic
// encoder, RealEncoder heads, LatentGAN).  Arbitrary M/N/K/ld (e.g. latent_dim = 145), so the
// loaders are guarded scalar loads; the multiply is the same 32x32x2 f32 MFMA tile as the
// convolutions (64x64 tile, 4 waves).  Skinny problems with a long K (32768 -> 148 at M = batch)
// are split over K across workgroups and combined with fp32 atomics.
#include "common.h"
#include "mma_tile.h"
#include <stdlib.h>

namespace {

__global__ __launch_bounds__(256) void gemm_kernel(int ta, int tb, int M, int N, int K, const float* __restrict__ A,
                                                   int lda, const float* __restrict__ B, int ldb, float* __restrict__ C,
                                                   int ldc, const float* __restrict__ bias, int act, float slope,
                                                   int k_per_split, int splitk, float* __restrict__ parts = nullptr) {
    constexpr int BM = 64, BN = 64, LDA = BM + 4, LDB = BN + 4;
    __shared__ float As[2][BK][LDA];
    __shared__ float Bs[2][BK][LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int kbeg = blockIdx.z * k_per_split, kend = min(K, kbeg + k_per_split);
    if (kbeg >= kend) return;

    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    float ra[4], rb[4];
    const int q4 = tid & 3, r64 = tid >> 2;      // (4 consecutive k, row) mapping
    const int q16 = tid & 15, r16 = tid >> 4;    // (4 consecutive m/n, k row) mapping

    auto load_tiles = [&](int k0) {
        if (!ta) {
            const int m = m0 + r64;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + q4 * 4 + e;
                ra[e] = (m < M && k < kend) ? A[(long)m * lda + k] : 0.f;
            }
        } else {
            const int k = k0 + r16;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + q16 * 4 + e;
                ra[e] = (m < M && k < kend) ? A[(long)k * lda + m] : 0.f;
            }
        }
        if (!tb) {
            const int k = k0 + r16;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + q16 * 4 + e;
                rb[e] = (n < N && k < kend) ? B[(long)k * ldb + n] : 0.f;
            }
        } else {
            const int n = n0 + r64;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + q4 * 4 + e;
                rb[e] = (n < N && k < kend) ? B[(long)n * ldb + k] : 0.f;
            }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (!ta) As[buf][q4 * 4 + e][r64] = ra[e];
            else As[buf][r16][q16 * 4 + e] = ra[e];
            if (!tb) Bs[buf][r16][q16 * 4 + e] = rb[e];
            else Bs[buf][q4 * 4 + e][r64] = rb[e];
        }
    };

    const int nks = (kend - kbeg + BK - 1) / BK;
    load_tiles(kbeg);
    store_tiles(0);
    __syncthreads();
    for (int ks = 0; ks < nks; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nks) load_tiles(kbeg + (ks + 1) * BK);
        mma_step<1, 1, LDA, LDB>(As[buf], Bs[buf], acc, wm * 32 + l31, wn * 32 + l31, half);
        if (ks + 1 < nks) store_tiles(buf ^ 1);
        __syncthreads();
    }

    const int col = n0 + wn * 32 + l31;
    if (col >= N) return;
    const float bv = (bias && blockIdx.z == 0) ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + 4 * half + (r & 3) + 8 * (r >> 2);
        if (row >= M) continue;
        const float v = acc[0][0][r] + bv;
        if (parts) parts[((long)blockIdx.z * M + row) * N + col] = v;      // deterministic mode: per-split slabs, added in order afterwards
        else if (splitk > 1) unsafeAtomicAdd(&C[(long)row * ldc + col], v);
        else C[(long)row * ldc + col] = cn_apply_act(v, act, slope);
    }
}

// Dense layers with <= 4 outputs and few rows (discriminator / rotation heads: M = batch, N = 1 or 3, K up to 32768): a
// 64x64 MFMA tile would walk the whole K in ONE workgroup (120 us for K = 2048).  One workgroup per (row, K slice)
// instead: 256 lanes stride K, wave shuffles + LDS combine.  bias/act need the full sum, so slices > 1 only without act.
__global__ __launch_bounds__(256) void thin_gemm_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                        const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
                                                        const float* __restrict__ bias, int act, float slope, int kps) {
    const int m = blockIdx.x;
    const int kbeg = blockIdx.y * kps, kend = min(K, kbeg + kps);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* ar = A + (long)m * lda;
    for (int k = kbeg + threadIdx.x; k < kend; k += 256) {
        const float av = ar[k];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < N) acc[j] += av * B[(long)k * ldb + j];
    }
    __shared__ float red[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v = acc[j];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < N) {
        const int j = threadIdx.x;
        const float v = red[0][j] + red[1][j] + red[2][j] + red[3][j];
        if (gridDim.y > 1) unsafeAtomicAdd(&C[(long)m * ldc + j], v + ((bias && blockIdx.y == 0) ? bias[j] : 0.f));
        else C[(long)m * ldc + j] = cn_apply_act(v + (bias ? bias[j] : 0.f), act, slope);
    }
}

// ---------------------------------------------------------------------------------------------
// Small dense layers (round 3).  The path's Dense layers run at M = batch (8 / 16 rows): AdaIN MLPs 145 -> 128 -> 2C, the
// synthetic encoder's 24 per-input layers, the latent discriminator, heads.  ~360 launches per training iteration went
// through the 64 x 64 MFMA tile above, which walks K in 16-deep stages with one global round trip per stage: 10 us for a
// 16 x 145 x 128 product whose arithmetic is nothing.  These kernels issue every load up front instead.
//   row-skinny (M <= 32; C = A op(B)): the whole A block sits in LDS, a thread owns one output column and a quarter of K
//     (M accumulators), the four K quarters are combined through LDS;
//   depth-skinny (K <= 32; C (+)= A^T B, the weight gradient x^T gy of such a layer): a thread owns 4 rows x 1 column.
// ---------------------------------------------------------------------------------------------
template <int MT, bool TB>
__global__ __launch_bounds__(256) void gemm_rows_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                        const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
                                                        const float* __restrict__ bias, int act, float slope) {
    extern __shared__ float sm[];                 // A block [MT][K] (rows >= M are zero), then the partial sums [3][MT][64]
    float* As = sm;
    float* red = sm + MT * K;
    const int tid = threadIdx.x, c = tid & 63, q = tid >> 6;
    for (int i = tid; i < MT * K; i += 256) {
        const int m = i / K, k = i - m * K;
        As[i] = m < M ? A[(long)m * lda + k] : 0.f;
    }
    __syncthreads();
    const int n = blockIdx.x * 64 + c;
    const int kq = (K + 3) / 4, kbeg = q * kq, kend = min(K, kbeg + kq);
    float acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = 0.f;
    if (n < N) {
#pragma unroll 4
        for (int k = kbeg; k < kend; ++k) {
            const float b = TB ? B[(long)n * ldb + k] : B[(long)k * ldb + n];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m] += As[m * K + k] * b;
        }
    }
    if (q > 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m) red[((q - 1) * MT + m) * 64 + c] = acc[m];
    }
    __syncthreads();
    if (q == 0 && n < N) {
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float v = acc[m] + red[(0 * MT + m) * 64 + c] + red[(1 * MT + m) * 64 + c] + red[(2 * MT + m) * 64 + c] + bv;
            if (m < M) C[(long)m * ldc + n] = cn_apply_act(v, act, slope);
        }
    }
}

__global__ __launch_bounds__(256) void gemm_depth_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                         const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
                                                         int accumulate) {
    // C[m][n] (+)= sum_k A[k][m] B[k][n], K <= 32.  Workgroup tile: 16 rows x 64 columns.
    __shared__ float As[32][16];
    __shared__ float Bs[32][64];
    const int tid = threadIdx.x, c = tid & 63, r = tid >> 6;
    const int m0 = blockIdx.x * 16, n0 = blockIdx.y * 64;
    for (int i = tid; i < K * 16; i += 256) {
        const int k = i >> 4, m = i & 15;
        As[k][m] = (m0 + m < M) ? A[(long)k * lda + m0 + m] : 0.f;
    }
    for (int i = tid; i < K * 64; i += 256) {
        const int k = i >> 6, nn = i & 63;
        Bs[k][nn] = (n0 + nn < N) ? B[(long)k * ldb + n0 + nn] : 0.f;
    }
    __syncthreads();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) {
        const float b = Bs[k][c];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += As[k][r * 4 + e] * b;
    }
    const int n = n0 + c;
    if (n >= N) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int m = m0 + r * 4 + e;
        if (m >= M) continue;
        if (accumulate) unsafeAtomicAdd(&C[(long)m * ldc + n], acc[e]);
        else C[(long)m * ldc + n] = acc[e];
    }
}

// ---- many shallow weight-gradient products in ONE launch (round 6): C_j += A_j^T B_j with K_j <= 32 rows -- the dense layers'
// weight gradients of a backward pass (batch <= 32 samples: the AdaIN MLPs, the encoders' and the regressor's dense layers) are
// leaves of the tape; the pass queues them and launches them together at its join.  Per job exactly gemm_depth_kernel's arithmetic.
constexpr int CN_DEPTH_GROUP = 64;
struct DepthJobs {
    const float* a[CN_DEPTH_GROUP];
    const float* b[CN_DEPTH_GROUP];
    float* c[CN_DEPTH_GROUP];
    int m[CN_DEPTH_GROUP], n[CN_DEPTH_GROUP], k[CN_DEPTH_GROUP], lda[CN_DEPTH_GROUP], ldb[CN_DEPTH_GROUP], ldc[CN_DEPTH_GROUP];
    int blk0[CN_DEPTH_GROUP + 1];
    int count;
};

__global__ __launch_bounds__(256) void gemm_depth_grouped_kernel(DepthJobs J) {
    __shared__ float As[32][16];
    __shared__ float Bs[32][64];
    const int bid = blockIdx.x;
    int lo = 0, hi = J.count;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (J.blk0[mid] <= bid) lo = mid;
        else hi = mid;
    }
    const int j = __builtin_amdgcn_readfirstlane(lo), lb = bid - J.blk0[j];
    const int M = J.m[j], N = J.n[j], K = J.k[j], lda = J.lda[j], ldb = J.ldb[j], ldc = J.ldc[j];
    const float* __restrict__ A = J.a[j];
    const float* __restrict__ B = J.b[j];
    float* __restrict__ C = J.c[j];
    const int tx = (M + 15) / 16;
    const int tid = threadIdx.x, c = tid & 63, r = tid >> 6;
    const int m0 = (lb % tx) * 16, n0 = (lb / tx) * 64;
    for (int i = tid; i < K * 16; i += 256) {
        const int k = i >> 4, m = i & 15;
        As[k][m] = (m0 + m < M) ? A[(long)k * lda + m0 + m] : 0.f;
    }
    for (int i = tid; i < K * 64; i += 256) {
        const int k = i >> 6, nn = i & 63;
        Bs[k][nn] = (n0 + nn < N) ? B[(long)k * ldb + n0 + nn] : 0.f;
    }
    __syncthreads();
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) {
        const float b = Bs[k][c];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += As[k][r * 4 + e] * b;
    }
    const int n = n0 + c;
    if (n >= N) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int m = m0 + r * 4 + e;
        if (m >= M) continue;
        unsafeAtomicAdd(&C[(long)m * ldc + n], acc[e]);
    }
}

// ---- many row-skinny dense layers in ONE launch (round 6): C_j = epi(A_j op(B_j) + bias_j) with M_j <= 32 rows -- the six AdaIN
// MLPs of a generator pass layer by layer (hologan_generator.py:119-124), forward and data-gradient side.  Per job exactly
// gemm_rows_kernel's arithmetic (K quarters combined in the same order); epi = activation, or (mask != NULL) the product with
// act'(mask) -- the hidden layer's LeakyReLU derivative taken from its stored output --, or (accumulate) an atomic add into C.
constexpr int CN_ROWS_GROUP = 16;
struct RowsJobs {
    const float* a[CN_ROWS_GROUP];
    const float* b[CN_ROWS_GROUP];
    float* c[CN_ROWS_GROUP];
    const float* bias[CN_ROWS_GROUP];
    const float* mask[CN_ROWS_GROUP];
    int m[CN_ROWS_GROUP], n[CN_ROWS_GROUP], k[CN_ROWS_GROUP], lda[CN_ROWS_GROUP], ldb[CN_ROWS_GROUP], ldc[CN_ROWS_GROUP];
    int tb[CN_ROWS_GROUP], act[CN_ROWS_GROUP], accumulate[CN_ROWS_GROUP];
    float slope[CN_ROWS_GROUP];
    int blk0[CN_ROWS_GROUP + 1];
    int count;
};

template <int MT>
__global__ __launch_bounds__(256) void gemm_rows_grouped_kernel(RowsJobs J) {
    extern __shared__ float sm[];                 // A block [MT][K] (rows >= M are zero), then the partial sums [3][MT][64]
    const int bid = blockIdx.x;
    int lo = 0, hi = J.count;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (J.blk0[mid] <= bid) lo = mid;
        else hi = mid;
    }
    const int j = __builtin_amdgcn_readfirstlane(lo);
    const int M = J.m[j], N = J.n[j], K = J.k[j], lda = J.lda[j], ldb = J.ldb[j], ldc = J.ldc[j], tb = J.tb[j];
    const float* __restrict__ A = J.a[j];
    const float* __restrict__ B = J.b[j];
    float* __restrict__ C = J.c[j];
    float* As = sm;
    float* red = sm + MT * K;
    const int tid = threadIdx.x, c = tid & 63, q = tid >> 6;
    for (int i = tid; i < MT * K; i += 256) {
        const int m = i / K, k = i - m * K;
        As[i] = m < M ? A[(long)m * lda + k] : 0.f;
    }
    __syncthreads();
    const int n = (bid - J.blk0[j]) * 64 + c;
    const int kq = (K + 3) / 4, kbeg = q * kq, kend = min(K, kbeg + kq);
    float acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = 0.f;
    if (n < N) {
        if (tb) {
#pragma unroll 4
            for (int k = kbeg; k < kend; ++k) {
                const float b = B[(long)n * ldb + k];
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] += As[m * K + k] * b;
            }
        } else {
#pragma unroll 4
            for (int k = kbeg; k < kend; ++k) {
                const float b = B[(long)k * ldb + n];
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] += As[m * K + k] * b;
            }
        }
    }
    if (q > 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m) red[((q - 1) * MT + m) * 64 + c] = acc[m];
    }
    __syncthreads();
    if (q == 0 && n < N) {
        const float bv = J.bias[j] ? J.bias[j][n] : 0.f;
        const float* __restrict__ mask = J.mask[j];
        const int act = J.act[j];
        const float slope = J.slope[j];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float v = acc[m] + red[(0 * MT + m) * 64 + c] + red[(1 * MT + m) * 64 + c] + red[(2 * MT + m) * 64 + c] + bv;
            if (m >= M) continue;
            float* dst = C + (long)m * ldc + n;
            if (mask) {
                const float y = mask[(long)m * ldc + n];          // act'(.) from the stored activation output (sign(y) = sign(x))
                const float d = act == CN_ACT_LRELU ? (y > 0.f ? 1.f : slope) : act == CN_ACT_RELU ? (y > 0.f ? 1.f : 0.f)
                                : act == CN_ACT_TANH ? 1.f - y * y : 1.f;
                *dst = v * d;
            } else if (J.accumulate[j]) {
                unsafeAtomicAdd(dst, v);
            } else {
                *dst = cn_apply_act(v, act, slope);
            }
        }
    }
}

__global__ void zero_rows_kernel(float* C, int M, int N, int ldc) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long)M * N) C[(i / N) * ldc + i % N] = 0.f;
}

}  // namespace

static int gemm_launch(int ta, int tb, int m, int n, int k, const float* a, int lda, const float* b, int ldb, float* c,
                       int ldc, const float* bias, int act, float slope, int accumulate, void* stream) {
    CN_CHECK_ARG(m > 0 && n > 0 && k > 0 && a && b && c, "gemm: bad args m=%d n=%d k=%d", m, n, k);
    CN_CHECK_ARG(lda >= (ta ? m : k) && ldb >= (tb ? k : n) && ldc >= n, "gemm: leading dimension too small");
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate && !ta && m <= 32 && n > 4 && (long)(m <= 8 ? 8 : m <= 16 ? 16 : 32) * k <= 8192) {
        const int mt = m <= 8 ? 8 : m <= 16 ? 16 : 32;
        const size_t lds = sizeof(float) * ((size_t)mt * k + 3 * mt * 64);
        dim3 grid(cn_cdiv(n, 64));
#define ROWS(MT_, TB_) hipLaunchKernelGGL((gemm_rows_kernel<MT_, TB_>), grid, dim3(256), lds, s, m, n, k, a, lda, b, ldb, c, ldc, bias, act, slope)
        if (mt == 8) { if (tb) ROWS(8, true); else ROWS(8, false); }
        else if (mt == 16) { if (tb) ROWS(16, true); else ROWS(16, false); }
        else { if (tb) ROWS(32, true); else ROWS(32, false); }
#undef ROWS
        CN_LAUNCH_CHECK();
        return CN_OK;
    }
    if (ta && !tb && k <= 32 && !bias && act == CN_ACT_NONE) {
        dim3 grid(cn_cdiv(m, 16), cn_cdiv(n, 64));
        hipLaunchKernelGGL(gemm_depth_kernel, grid, dim3(256), 0, s, m, n, k, a, lda, b, ldb, c, ldc, accumulate);
        CN_LAUNCH_CHECK();
        return CN_OK;
    }
    if (!accumulate && !ta && !tb && n <= 4 && m <= 256 && k >= 128) {
        int slices = 1;
        if (act == CN_ACT_NONE && k >= 8192 && !cn_det()) slices = k / 4096;      // (deterministic mode: no K slices, no atomics)
        const int kps = (k + slices - 1) / slices;
        slices = (k + kps - 1) / kps;
        if (slices > 1) {
            hipLaunchKernelGGL(zero_rows_kernel, dim3(cn_cdiv((long)m * n, 256)), dim3(256), 0, s, c, m, n, ldc);
            CN_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(thin_gemm_kernel, dim3(m, slices), dim3(256), 0, s, m, n, k, a, lda, b, ldb, c, ldc, bias, act, slope, kps);
        CN_LAUNCH_CHECK();
        return CN_OK;
    }
    const long tiles = (long)cn_cdiv(m, 64) * cn_cdiv(n, 64);
    int splitk = 1;
    if (act == CN_ACT_NONE && tiles < 128 && k >= 1024 && !(cn_det() && (accumulate || ldc != n))) {
        splitk = (int)((256 + tiles - 1) / tiles);
        if (splitk > k / 256) splitk = k / 256;
        if (splitk < 1) splitk = 1;
    }
    int kps = (k + splitk - 1) / splitk;
    kps = (kps + BK - 1) / BK * BK;
    splitk = (k + kps - 1) / kps;
    float* parts = nullptr;
    if (cn_det() && splitk > 1) {
        // deterministic mode: K slices into per-split slabs of the stream's workspace, added in split order by a second launch
        const long cap = (long)(CN_DET_WS_FLOATS / ((size_t)m * n));
        if (splitk > cap) splitk = (int)(cap < 1 ? 1 : cap);
        kps = (k + splitk - 1) / splitk;
        kps = (kps + BK - 1) / BK * BK;
        splitk = (k + kps - 1) / kps;
        if (splitk > 1) {
            parts = cn_det_ws(s, (size_t)splitk * m * n);
            if (!parts) return CN_EINVAL;
        }
    }
    if (splitk > 1 && !accumulate && !parts) {
        hipLaunchKernelGGL(zero_rows_kernel, dim3(cn_cdiv((long)m * n, 256)), dim3(256), 0, s, c, m, n, ldc);
        CN_LAUNCH_CHECK();
    }
    dim3 grid(cn_cdiv(m, 64), cn_cdiv(n, 64), splitk);
    // (the kernel adds with atomics whenever its last argument is > 1: split-K, or accumulation into the caller's C)
    hipLaunchKernelGGL(gemm_kernel, grid, dim3(256), 0, s, ta, tb, m, n, k, a, lda, b, ldb, c, ldc, bias, act, slope, kps,
                       accumulate ? 2 : splitk, parts);
    CN_LAUNCH_CHECK();
    if (parts) return cn_sum_parts(parts, c, splitk, (long)m * n, 0, 1.f, s);
    return CN_OK;
}

extern "C" int cn_gemm(int ta, int tb, int m, int n, int k, const float* a, int lda, const float* b, int ldb, float* c,
                       int ldc, const float* bias, int act, float slope, void* stream) {
    return gemm_launch(ta, tb, m, n, k, a, lda, b, ldb, c, ldc, bias, act, slope, 0, stream);
}

// C += op(A) op(B): the weight gradient of a Dense layer added straight into its slot of the network's gradient arena
extern "C" int cn_gemm_acc(int ta, int tb, int m, int n, int k, const float* a, int lda, const float* b, int ldb, float* c,
                           int ldc, void* stream) {
    return gemm_launch(ta, tb, m, n, k, a, lda, b, ldb, c, ldc, nullptr, CN_ACT_NONE, 0.f, 1, stream);
}

// C_j = epi(A_j op(B_j) + bias_j), m_j <= 32 rows, for every job in one launch per 16 jobs; `jobs` is a HOST array.
extern "C" int cn_gemm_rows_grouped(const CnRowsJob* jobs, int njobs, void* stream) {
    CN_CHECK_ARG(njobs >= 0 && (njobs == 0 || jobs), "cn_gemm_rows_grouped: bad arguments");
    for (int first = 0; first < njobs; first += CN_ROWS_GROUP) {
        RowsJobs J{};
        const int cnt = njobs - first < CN_ROWS_GROUP ? njobs - first : CN_ROWS_GROUP;
        long blocks = 0;
        int mmax = 0, kmax = 0;
        for (int q = 0; q < cnt; ++q) {
            const CnRowsJob& d = jobs[first + q];
            CN_CHECK_ARG(d.a && d.b && d.c && d.m > 0 && d.m <= 32 && d.n > 0 && d.k > 0 && d.lda >= d.k && d.ldb >= (d.tb ? d.k : d.n) && d.ldc >= d.n,
                         "cn_gemm_rows_grouped: job %d: m=%d n=%d k=%d (m <= 32)", first + q, d.m, d.n, d.k);
            CN_CHECK_ARG(!(d.mask && d.accumulate) && !(d.accumulate && (d.bias || d.act != CN_ACT_NONE)),
                         "cn_gemm_rows_grouped: job %d: accumulate excludes mask / bias / activation", first + q);
            J.a[q] = d.a; J.b[q] = d.b; J.c[q] = d.c; J.bias[q] = d.bias; J.mask[q] = d.mask;
            J.m[q] = d.m; J.n[q] = d.n; J.k[q] = d.k; J.lda[q] = d.lda; J.ldb[q] = d.ldb; J.ldc[q] = d.ldc;
            J.tb[q] = d.tb; J.act[q] = d.act; J.accumulate[q] = d.accumulate; J.slope[q] = d.slope;
            J.blk0[q] = (int)blocks;
            blocks += cn_cdiv(d.n, 64);
            mmax = d.m > mmax ? d.m : mmax;
            kmax = d.k > kmax ? d.k : kmax;
        }
        J.blk0[cnt] = (int)blocks;
        J.count = cnt;
        const int mt = mmax <= 8 ? 8 : mmax <= 16 ? 16 : 32;
        CN_CHECK_ARG((long)mt * kmax <= 8192, "cn_gemm_rows_grouped: %d rows x k = %d do not fit the LDS block", mt, kmax);
        const size_t lds = sizeof(float) * ((size_t)mt * kmax + 3 * mt * 64);
        if (mt == 8) hipLaunchKernelGGL((gemm_rows_grouped_kernel<8>), dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, J);
        else if (mt == 16) hipLaunchKernelGGL((gemm_rows_grouped_kernel<16>), dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, J);
        else hipLaunchKernelGGL((gemm_rows_grouped_kernel<32>), dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, J);
        CN_LAUNCH_CHECK();
    }
    return CN_OK;
}

// C_j += A_j^T B_j (A_j: k x m, B_j: k x n, row-major, k <= 32) for every job in one launch per 64 jobs; `jobs` is a HOST array.
extern "C" int cn_gemm_depth_grouped(const CnDepthJob* jobs, int njobs, void* stream) {
    CN_CHECK_ARG(njobs >= 0 && (njobs == 0 || jobs), "cn_gemm_depth_grouped: bad arguments");
    for (int first = 0; first < njobs; first += CN_DEPTH_GROUP) {
        DepthJobs J{};
        const int cnt = njobs - first < CN_DEPTH_GROUP ? njobs - first : CN_DEPTH_GROUP;
        long blocks = 0;
        for (int q = 0; q < cnt; ++q) {
            const CnDepthJob& d = jobs[first + q];
            CN_CHECK_ARG(d.a && d.b && d.c && d.m > 0 && d.n > 0 && d.k > 0 && d.k <= 32 && d.lda >= d.m && d.ldb >= d.n && d.ldc >= d.n,
                         "cn_gemm_depth_grouped: job %d: m=%d n=%d k=%d (k <= 32)", first + q, d.m, d.n, d.k);
            J.a[q] = d.a; J.b[q] = d.b; J.c[q] = d.c;
            J.m[q] = d.m; J.n[q] = d.n; J.k[q] = d.k; J.lda[q] = d.lda; J.ldb[q] = d.ldb; J.ldc[q] = d.ldc;
            J.blk0[q] = (int)blocks;
            blocks += (long)cn_cdiv(d.m, 16) * cn_cdiv(d.n, 64);
        }
        J.blk0[cnt] = (int)blocks;
        J.count = cnt;
        hipLaunchKernelGGL(gemm_depth_grouped_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, J);
        CN_LAUNCH_CHECK();
    }
    return CN_OK;
}
