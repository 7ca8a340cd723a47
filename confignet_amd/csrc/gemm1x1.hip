// gemm1x1.hip -- 1x1 stride-1 convolutions (ResNet-50's bottleneck projections, the generator's projection conv) and their data
// gradients as what they are: a plain row-major product C[M][N] = A[M][K] B, M = all positions, K = cin, N = cout.
//
// The implicit-GEMM kernel pays for generality here: a row decode with integer divisions, the output-row map, a k-major LDS
// image written with scalar stores and read with one ds_read_b32 per MFMA operand.  This kernel keeps both tiles the way they
// arrive:
//   * A rows are K-contiguous in HBM and stay so in LDS ([row][16 + 4 pad]): one ds_write_b128 per 16-byte piece, and -- with the
//     K order of an 8-deep group permuted so that half-wave h owns k = 4h .. 4h+3 -- ONE ds_read_b128 per lane feeds four MFMAs
//     (MFMA step q contracts the pair {q, 4 + q}; any K order is fine as long as A and B agree);
//   * forward: B = filter [K][N], N-contiguous; a lane owns TN ADJACENT output columns (TN * l31 + j), so one 8/16-byte read
//     feeds its TN column tiles and the epilogue stores 8/16 bytes per row;
//   * data gradient (BT): B^T = the ORIGINAL filter [N = cin][K = cout], K-contiguous like A and read like A.
// Three LDS buffers and ONE barrier per step, placed before the last fragment group: the tile of step s+1 is written at the top
// of step s, the barrier sits inside the MFMA stream and the first fragments of step s+1 are read before step s ends.  All loads
// are unconditional (clamped addresses): see igemm_fwd_kernel.  scripts/dev/gemm_lab holds the stand-alone study of this loop
// against the vendor sgemm.
//
// GATHER = true is the same loop for the gathered layers that are not parity-ordered (strided forward convolutions, Conv3D,
// the 3x3 layers Winograd does not take): K walks (tap, 16-channel chunk) tap-major, a row's source offset is recomputed when the
// tap changes, padding taps read a clamped address and are zeroed by a mask on the way into LDS.
#include "common.h"
#include "mma_tile.h"
#include "conv_geom.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int WM, int WN, int TM, int TN, bool BT, bool GATHER>
__global__ __launch_bounds__(256) void gemm1x1_kernel(CnConvGeom g, const float* __restrict__ A, const float* __restrict__ B,
                                                      const float* __restrict__ bias, float* __restrict__ C, int M, int N, int K,
                                                      int act, float slope, int ntm, int ntn, long part_stride, int par,
                                                      const float* __restrict__ res = nullptr) {
    // res (unsplit launches only): a tensor of C's shape added before the activation -- the residual branch of a ResNet block
    // K: channels per tap (the reduction is T * K deep, T = 1 without GATHER)
    // par (GATHER only): parity-ordered rows of a zero-stuffed data gradient / an upsample-folded layer -- tile rows are
    // enumerated class-major (conv_geom.h par_row), a tile inside one class walks its live taps only
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(TN >= 1 && TN <= 3, "column tiles per wave");
    constexpr int KB = 16, G = KB / 8, KQ = KB / 4;
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN, LDA = KB + 4, LDB = BN + 4;
    constexpr int AP = BM * KQ / 256, BP = (BN * KQ + 255) / 256;
    constexpr int BSZ = BT ? BN * LDA : KB * LDB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;                    // [3][BM][LDA]
    float* const Bs = smem + 3 * BM * LDA;     // [3][KB][LDB]  or (BT)  [3][BN][LDA]
    int* const rowmap = reinterpret_cast<int*>(smem + 3 * BM * LDA + 3 * BSZ);   // GATHER: tile row -> output row (or -1)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware order (as igemm_fwd_kernel, mode 2): XCD id % 8 gets a contiguous run of M tiles and, within it, the column tiles
    // of one M tile in consecutive slots -- they read the same A rows, which then come from that XCD's L2
    int bx, by;
    if (GATHER && par) {
        // class-major rows: consecutive M tiles sit in one class (1 / 2 / 2 / 4 live taps); the plain order deals them round-robin
        // over the XCDs, a contiguous run per XCD would give one XCD the 4-tap class
        divmod_pos((int)blockIdx.x, ntm, by, bx);
        if (by >= ntn) return;
    } else {
        const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
        const int q = ntm >> 3, r = ntm & 7;
        const int mine = q + (xcd < r ? 1 : 0);
        int ml;
        divmod_pos(j, ntn, ml, by);
        if (ml >= mine) return;
        bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + ml;
    }
    const int m0 = bx * BM, n0 = by * BN;
    const int T = GATHER ? g.k_d * g.k_h * g.k_w : 1;
    const int cpb = K >> 4;          // KB == 16
    unsigned long long tapmask = T >= 64 ? ~0ull : ((1ull << T) - 1ull);
    if (GATHER && par) {
        int c0, c1;
        par_row(g, m0, M, c0);
        par_row(g, min(m0 + BM, M) - 1, M, c1);
        if (c0 == c1) tapmask = par_tap_mask(g, c0);   // whole tile in one parity class: skip dead taps
    }
    const int nks_all = (GATHER ? __popcll(tapmask) : 1) * cpb;
    const int per_z = (nks_all + gridDim.z - 1) / gridDim.z;
    const int ks_beg = blockIdx.z * per_z, ks_end = min(nks_all, ks_beg + per_z);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    {
        // per-thread source pointers (rows / columns past the end read the last valid one; their results are never stored)
        const float* ap[AP];
        const float* bp[BP];
        int a_lds[AP], b_lds[BP];
        RowInfo ri[AP];
        int aoff[AP];
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int idx = tid + 256 * i, r = idx / KQ, kq = idx % KQ;
            if (GATHER) {
                int mrow = m0 + r, cls;
                if (par) mrow = par_row(g, mrow, M, cls);
                ri[i] = decode_row(g, mrow, M);
                if (kq == 0) rowmap[r] = ri[i].ok ? mrow : -1;
                ap[i] = A + kq * 4;
            } else {
                ap[i] = A + (long)min(m0 + r, M - 1) * K + kq * 4;
            }
            a_lds[i] = r * LDA + kq * 4;
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            const int idx = min(tid + 256 * j, BN * KQ - 1);          // (128 x 96: the second piece exists for half the threads)
            if (BT) {
                const int r = idx / KQ, kq = idx % KQ;
                bp[j] = B + (long)min(n0 + r, N - 1) * K + kq * 4;
                b_lds[j] = r * LDA + kq * 4;
            } else {
                const int br = idx / (BN / 4), bc = idx % (BN / 4);
                bp[j] = B + (long)br * N + min(n0 + bc * 4, N - 4);
                b_lds[j] = br * LDB + bc * 4;
            }
        }
        if (GATHER) __syncthreads();                      // rowmap (an empty K split goes straight to the epilogue)
      if (ks_beg < ks_end) {
        // GATHER: (tap, channel chunk) of the most recent load; load_tiles is called with ks = ks_beg, ks_beg + 1, ... (each call
        // the previous step + 1, or the same step again once the index is clamped at the end)
        int ld_ks = ks_beg, ld_tap = -1, ld_c0 = (ks_beg - (ks_beg / cpb) * cpb) * KB;
        for (int o = ks_beg / cpb; o >= 0; --o) ld_tap += __ffsll((long long)(tapmask >> (ld_tap + 1)));   // the (ks_beg / cpb)-th live tap
        unsigned amask_cur = ~0u;
        long b_tap = 0;                                   // filter offset of the current tap
        auto retap = [&]() __attribute__((always_inline)) {
            int kd, kh, kw;
            tap_decode(g, ld_tap, kd, kh, kw);
            amask_cur = 0;
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                const int off = src_off(g, ri[i], kd, kh, kw);
                aoff[i] = max(off, 0);
                amask_cur |= (off >= 0 ? 1u : 0u) << i;
            }
            b_tap = BT ? (long)(T - 1 - ld_tap) * N * K : (long)ld_tap * K * N;
        };
        if (GATHER) retap();
        f4 ra[2][AP], rb[2][BP];
        unsigned am[2] = {~0u, ~0u};
        auto load_tiles = [&](int ks, f4 (&ra)[AP], f4 (&rb)[BP], unsigned& amk) __attribute__((always_inline)) {
            if (GATHER) {
                if (ks != ld_ks) {
                    ld_ks = ks;
                    ld_c0 += KB;
                    if (ld_c0 == K) {
                        ld_c0 = 0;
                        ld_tap += __ffsll((long long)(tapmask >> (ld_tap + 1)));
                        retap();
                    }
                }
                amk = amask_cur;
#pragma unroll
                for (int i = 0; i < AP; ++i) ra[i] = *reinterpret_cast<const f4*>(ap[i] + aoff[i] + ld_c0);
#pragma unroll
                for (int j = 0; j < BP; ++j) rb[j] = *reinterpret_cast<const f4*>(bp[j] + b_tap + (BT ? (long)ld_c0 : (long)ld_c0 * N));
            } else {
#pragma unroll
                for (int i = 0; i < AP; ++i) ra[i] = *reinterpret_cast<const f4*>(ap[i] + ks * KB);
#pragma unroll
                for (int j = 0; j < BP; ++j) rb[j] = *reinterpret_cast<const f4*>(bp[j] + (BT ? (long)ks * KB : (long)ks * KB * N));
            }
        };
        auto store_tiles = [&](int buf, const f4 (&ra)[AP], const f4 (&rb)[BP], unsigned amk) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                f4 v = ra[i];
                if (GATHER) {
                    const bool live = (amk >> i) & 1u;      // padding taps / rows past the end: zeros
                    v.x = live ? v.x : 0.f; v.y = live ? v.y : 0.f; v.z = live ? v.z : 0.f; v.w = live ? v.w : 0.f;
                }
                *reinterpret_cast<f4*>(As + buf * BM * LDA + a_lds[i]) = v;
            }
#pragma unroll
            for (int j = 0; j < BP; ++j)
                if (BN * KQ % 256 == 0 || tid + 256 * j < BN * KQ) *reinterpret_cast<f4*>(Bs + buf * BSZ + b_lds[j]) = rb[j];
        };
        // fragment addresses: A row (wm, tile i, l31), K quad of this half-wave; B likewise (BT) or [k row][TN adjacent columns]
        const int a_frag = (wm * 32 * TM + l31) * LDA + 4 * half;
        const int b_frag = BT ? (wn * 32 * TN + l31) * LDA + 4 * half : (4 * half) * LDB + wn * 32 * TN + TN * l31;
        float a[2][TM][4], b[2][4][TN];
        auto frag = [&](int buf, int g, int set) __attribute__((always_inline)) {
            const float* as = As + buf * BM * LDA + a_frag + 8 * g;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const f4 v = *reinterpret_cast<const f4*>(as + 32 * i * LDA);
                a[set][i][0] = v.x; a[set][i][1] = v.y; a[set][i][2] = v.z; a[set][i][3] = v.w;
            }
            if (BT) {
                const float* bs = Bs + buf * BSZ + b_frag + 8 * g;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const f4 v = *reinterpret_cast<const f4*>(bs + 32 * j * LDA);
                    b[set][0][j] = v.x; b[set][1][j] = v.y; b[set][2][j] = v.z; b[set][3][j] = v.w;
                }
            } else {
                const float* bs = Bs + buf * BSZ + b_frag + 8 * g * LDB;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (TN == 2) {
                        const f2 v = *reinterpret_cast<const f2*>(bs + q * LDB);
                        b[set][q][0] = v.x; b[set][q][TN - 1] = v.y;
                    } else {
#pragma unroll
                        for (int j = 0; j < TN; ++j) b[set][q][j] = bs[q * LDB + j];
                    }
                }
            }
        };
        const int ks_last = ks_end - 1;
        load_tiles(ks_beg, ra[0], rb[0], am[0]);
        store_tiles(0, ra[0], rb[0], am[0]);
        load_tiles(min(ks_beg + 1, ks_last), ra[0], rb[0], am[0]);
        load_tiles(min(ks_beg + 2, ks_last), ra[1], rb[1], am[1]);
        __syncthreads();
        frag(0, 0, 0);
        int cur = 0;
        auto step = [&](int s, f4 (&ra)[AP], f4 (&rb)[BP], unsigned& amk) __attribute__((always_inline)) {
            const int nxt = cur == 2 ? 0 : cur + 1;
            store_tiles(nxt, ra, rb, amk);                    // step s+1 (loaded two steps ago)
            load_tiles(min(s + 3, ks_last), ra, rb, amk);     // step s+3 into the set just stored
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (g == G - 1) __syncthreads();              // the step-(s+1) tile is complete; everybody is past buffer cur's reads
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
        int s = ks_beg;
        for (; s + 1 < ks_end; s += 2) {
            step(s, ra[0], rb[0], am[0]);
            step(s + 1, ra[1], rb[1], am[1]);
        }
        if (s < ks_end) step(s, ra[0], rb[0], am[0]);
      }
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const bool split = gridDim.z > 1;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lrow = wm * 32 * TM + 32 * i + 4 * half + (r & 3) + 8 * (r >> 2);
            const int row = GATHER ? rowmap[lrow] : m0 + lrow;
            if (row < 0 || row >= M) continue;
            if (!BT && TN == 2) {
                const int col = n0 + wn * 64 + 2 * l31;
                if (col >= N) continue;
                float v0 = acc[i][0][r], v1 = acc[i][TN - 1][r];
                if (bias && blockIdx.z == 0) { v0 += bias[col]; v1 += bias[col + 1]; }
                float* dst = C + (long)row * N + col;
                if (part_stride) *reinterpret_cast<f2*>(dst + (long)blockIdx.z * part_stride) = f2{v0, v1};
                else if (split) { unsafeAtomicAdd(dst, v0); unsafeAtomicAdd(dst + 1, v1); }
                else {
                    if (res) { const f2 rv = *reinterpret_cast<const f2*>(res + (long)row * N + col); v0 += rv.x; v1 += rv.y; }
                    *reinterpret_cast<f2*>(dst) = f2{cn_apply_act(v0, act, slope), cn_apply_act(v1, act, slope)};
                }
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = n0 + wn * 32 * TN + (BT ? 32 * j + l31 : TN * l31 + j);
                    if (col >= N) continue;
                    const float v = acc[i][j][r] + ((bias && blockIdx.z == 0) ? bias[col] : 0.f);
                    float* dst = C + (long)row * N + col;
                    if (part_stride) dst[(long)blockIdx.z * part_stride] = v;
                    else if (split) unsafeAtomicAdd(dst, v);
                    else *dst = cn_apply_act(res ? v + res[(long)row * N + col] : v, act, slope);
                }
            }
        }
}

template <int WM, int WN, int TM, int TN, bool BT, bool GATHER>
int launch(const CnConvGeom& g, const float* A, const float* B, const float* bias, float* C, long M, int N, int K, int act, float slope,
           int splits, long part_stride, int par, hipStream_t s, const float* res) {
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    constexpr size_t lds = sizeof(float) * (3 * (BM * 20 + (BT ? BN * 20 : 16 * (BN + 4))) + BM);
    static bool attr_set = false;
    if (!attr_set) {
        CN_HIP(hipFuncSetAttribute((const void*)gemm1x1_kernel<WM, WN, TM, TN, BT, GATHER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int ntm = cn_cdiv(M, BM), ntn = cn_cdiv(N, BN);
    dim3 grid((unsigned)(par ? ntm * ntn : 8 * cn_cdiv(ntm, 8) * ntn), 1, (unsigned)splits);
    hipLaunchKernelGGL((gemm1x1_kernel<WM, WN, TM, TN, BT, GATHER>), grid, dim3(256), lds, s, g, A, B, bias, C, (int)M, N, K, act, slope, ntm, ntn, part_stride, par, res);
    CN_LAUNCH_CHECK();
    return CN_OK;
}

}  // namespace

// cfg: the implicit-GEMM tile numbering (0 = 128 x 128, 1 = 128 x 64, 2 = 64 x 64, 4 = 128 x 96); the 128 x 32 tile is not provided
// (CN_EUNSUPPORTED: the caller falls back to igemm_fwd_kernel).  bt: B is the original filter [N][K] (data gradient).  Same
// split-K protocol as igemm_fwd_kernel: splits > 1 adds into a zeroed C (or stores slabs at part_stride), bias by split 0, no
// activation.  gp: NULL = the rows of A are the GEMM rows (1x1, stride 1); otherwise the geometry whose gather builds them (vec:
// the caller checks), par = its rows are parity-ordered.
int cn_gemm1x1(const CnConvGeom* gp, int cfg, int bt, const float* A, const float* B, const float* bias, float* C, long M, int N, int K,
               int act, float slope, int splits, long part_stride, int par, hipStream_t s, const float* res) {
    if (res && (splits > 1 || part_stride)) return CN_EUNSUPPORTED;
    if (K % 16 != 0 || N % 4 != 0 || M <= 0 || M > 0x7fffffffL || (par && !gp)) return CN_EUNSUPPORTED;
    static const CnConvGeom none = {};
#define L(WM, WN, TM, TN)                                                                                                          \
    return gp ? (bt ? launch<WM, WN, TM, TN, true, true>(*gp, A, B, bias, C, M, N, K, act, slope, splits, part_stride, par, s, res)      \
                    : launch<WM, WN, TM, TN, false, true>(*gp, A, B, bias, C, M, N, K, act, slope, splits, part_stride, par, s, res))    \
              : (bt ? launch<WM, WN, TM, TN, true, false>(none, A, B, bias, C, M, N, K, act, slope, splits, part_stride, 0, s, res)      \
                    : launch<WM, WN, TM, TN, false, false>(none, A, B, bias, C, M, N, K, act, slope, splits, part_stride, 0, s, res))
    switch (cfg) {
        case 0: L(2, 2, 2, 2);
        case 1: L(2, 2, 2, 1);
        case 2: L(2, 2, 1, 1);
        case 4: L(4, 1, 1, 3);
        default: return CN_EUNSUPPORTED;
    }
#undef L
}
