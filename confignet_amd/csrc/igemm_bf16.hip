// igemm_bf16.hip -- the convolution family in bf16 storage on v_mfma_f32_32x32x16_bf16 (fp32 accumulate): the compute
// path of BASELINE.json configs[2] (bf16 compute, fp32 master weights / statistics / gradients of parameters).
//
// Same implicit GEMM as igemm_conv.hip (M = output positions, K = taps*cin, N = cout; SAME padding, the folded x2
// upsample and the zero-stuffing of strided data gradients are predicates of the gather), but every operand piece is 8
// bf16 = 16 bytes of the REDUCTION axis:
//   activations (N,[D,]H,W,C) bf16: a gathered row piece is 8 consecutive channels = one 16-byte load;
//   filters: bf16 copies with the reduction axis contiguous, made once per weight update (cn_conv_weight_prep_bf16):
//            wf[t][co][ci] for the forward GEMM, wd[t][ci][co] for the data-gradient GEMM (tap flip = index arithmetic);
//   LDS tiles [row][32 k + 8 pad] bf16 (80-byte pitch): ds_write_b128 / ds_read_b128 conflict-free, and a lane's MFMA
//            operand (8 consecutive k of one row) is ONE ds_read_b128.
// A 32-deep stage is two MFMA k-steps.  The bf16 pipe is 16x the fp32 one, so these kernels are bound by L2/HBM and LDS
// traffic, not by the matrix cores: the output tile goes back through LDS and leaves as 16-byte row pieces.
// The filter gradient needs BOTH operands with the position index contiguous; they arrive channel-contiguous, so the
// LDS store transposes (two positions packed per ds_write_b32).
#include "common.h"
#include <stdlib.h>

#include "conv_geom.h"
#include "typed.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KS = 32;          // bf16 elements of the reduction axis per LDS stage
constexpr int LDK = KS + 8;     // LDS row pitch in elements (80 bytes)

typedef short s16x4 __attribute__((ext_vector_type(4)));
union Frag {
    uint4 u;
    bf16x8 v;
    s16x4 h[2];
};

// smallest row pitch (elements) >= w whose byte length is a multiple of 16 and 64 or 192 (mod 256)
constexpr int tr_pitch(int w) {
    int p = (w + 7) / 8 * 8;
    while ((p * 2) % 256 != 64 && (p * 2) % 256 != 192) p += 8;
    return p;
}
__device__ __forceinline__ s16x4 tr_read(const bf16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

__global__ void wprep_bf16_kernel(const float* __restrict__ W, bf16_t* __restrict__ Wf, bf16_t* __restrict__ Wd, int T, int cin,
                                  int cout) {
    const long total = (long)T * cin * cout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        if (Wd) Wd[i] = f32_to_bf16(W[i]);
        if (Wf) {                                    // i indexes wf[t][co][ci]
            const int ci = (int)(i % cin);
            const long r = i / cin;
            const int co = (int)(r % cout), t = (int)(r / cout);
            Wf[i] = f32_to_bf16(W[((long)t * cin + ci) * cout + co]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// forward / data gradient:  Y[m, n] = act( sum_{t,k} X[src(m,t), k] * Wb[t'][n][k] + bias[n] ),  t' = flip ? T-1-t : t
// (g.cin = length of the reduction axis per tap, g.cout = number of output columns)
// ---------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void igemm_bf16_kernel(CnConvGeom g, const bf16_t* __restrict__ X,
                                                         const bf16_t* __restrict__ Wb, const float* __restrict__ bias,
                                                         bf16_t* __restrict__ Y, int act, float slope, int par, int flip) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    constexpr int AP = BM / 64;                    // 16-byte pieces of the A tile per thread per stage (BM rows x 4 pieces)
    constexpr int BP = (BN * 4 + 255) / 256;
    constexpr int LDC = BN + 8;                    // pitch of the output tile staged in LDS
    static_assert(BM * LDC <= 2 * (BM + BN) * LDK, "output tile must fit the operand buffers");
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * (BM + BN) * LDK];
    __shared__ int rowmap[BM];
    bf16_t* As = smem;                             // [2][BM][LDK]
    bf16_t* Bs = smem + 2 * BM * LDK;              // [2][BN][LDK]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, half = lane >> 5, l31 = lane & 31;
    const int M = g.n * g.out_d * g.out_h * g.out_w;
    const int T = g.k_d * g.k_h * g.k_w;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int kq = tid & 3, arow = tid >> 2;

    RowInfo ri[AP];
    unsigned long long tapmask = T >= 64 ? ~0ull : ((1ull << T) - 1ull);
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        int mrow = m0 + arow + 64 * i;
        if (par) {
            int cls;
            mrow = par_row(g, mrow, M, cls);
        }
        ri[i] = decode_row(g, mrow, M);
        if (kq == 0) rowmap[arow + 64 * i] = ri[i].ok ? mrow : -1;
    }
    if (par) {
        int c0, c1;
        par_row(g, m0, M, c0);
        par_row(g, min(m0 + BM, M) - 1, M, c1);
        if (c0 == c1) tapmask = par_tap_mask(g, c0);   // whole tile in one parity class: skip the zero-stuffed taps
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int cpb = (g.cin + KS - 1) / KS;         // stages per tap (the last one may be partly zero: cin = 48)
    const int nks = __popcll(tapmask) * cpb;
    uint4 ra0[AP], rb0[BP], ra1[AP], rb1[BP];      // two register sets: operands of stage s+2 in flight while stage s is multiplied
    int aoff[AP];
    int cur_ord = -1, cur_tap = -1;

    auto load_tiles = [&](int ks, uint4 (&ra)[AP], uint4 (&rb)[BP]) {
        const int ord = ks / cpb;
        const int c0 = (ks - ord * cpb) * KS + kq * 8;
        if (ord != cur_ord) {
            while (cur_ord < ord) {
                cur_tap += __ffsll((long long)(tapmask >> (cur_tap + 1)));
                ++cur_ord;
            }
            int kd, kh, kw;
            tap_decode(g, cur_tap, kd, kh, kw);
#pragma unroll
            for (int i = 0; i < AP; ++i) aoff[i] = src_off(g, ri[i], kd, kh, kw);
        }
        const bool kin = c0 < g.cin;
#pragma unroll
        for (int i = 0; i < AP; ++i)
            ra[i] = (aoff[i] >= 0 && kin) ? *reinterpret_cast<const uint4*>(X + aoff[i] + c0) : make_uint4(0, 0, 0, 0);
        const long wtap = (long)(flip ? T - 1 - cur_tap : cur_tap) * g.cout;
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            const int idx = tid + 256 * j;
            const int brow = idx >> 2, col = n0 + brow;
            rb[j] = (brow < BN && col < g.cout && kin) ? *reinterpret_cast<const uint4*>(Wb + (wtap + col) * g.cin + c0)
                                                       : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tiles = [&](int buf, const uint4 (&ra)[AP], const uint4 (&rb)[BP]) {
#pragma unroll
        for (int i = 0; i < AP; ++i)
            *reinterpret_cast<uint4*>(As + ((size_t)buf * BM + arow + 64 * i) * LDK + kq * 8) = ra[i];
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            const int idx = tid + 256 * j;
            if ((idx >> 2) < BN) *reinterpret_cast<uint4*>(Bs + ((size_t)buf * BN + (idx >> 2)) * LDK + kq * 8) = rb[j];
        }
    };

    const int a_row = wm * 32 * TM + l31, b_row = wn * 32 * TN + l31;
    auto mma_stage = [&](int buf) {
        Frag a[2][TM], b[2][TN];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[s][i].u = *reinterpret_cast<const uint4*>(As + ((size_t)buf * BM + a_row + 32 * i) * LDK + 16 * s + 8 * half);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[s][j].u = *reinterpret_cast<const uint4*>(Bs + ((size_t)buf * BN + b_row + 32 * j) * LDK + 16 * s + 8 * half);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][i].v, b[s][j].v, acc[i][j], 0, 0, 0);
    };
    if (nks > 0) {
        load_tiles(0, ra0, rb0);
        store_tiles(0, ra0, rb0);
    }
    if (nks > 1) load_tiles(1, ra1, rb1);
    __syncthreads();
    int ks = 0;
    for (; ks + 1 < nks; ks += 2) {
        if (ks + 2 < nks) load_tiles(ks + 2, ra0, rb0);
        mma_stage(0);
        store_tiles(1, ra1, rb1);
        __syncthreads();
        if (ks + 3 < nks) load_tiles(ks + 3, ra1, rb1);
        mma_stage(1);
        if (ks + 2 < nks) store_tiles(0, ra0, rb0);
        __syncthreads();
    }
    if (ks < nks) mma_stage(0);
    __syncthreads();

    // epilogue: bias + activation in fp32, bf16 tile through LDS, 16-byte row pieces out.
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    bf16_t* Cs = smem;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int ct = wn * 32 * TN + 32 * j + l31;
        const int col = n0 + ct;
        const float bv = (bias && col < g.cout) ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rbase = wm * 32 * TM + 32 * i + 4 * half;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Cs[(rbase + (r & 3) + 8 * (r >> 2)) * LDC + ct] = f32_to_bf16(cn_apply_act(acc[i][j][r] + bv, act, slope));
        }
    }
    __syncthreads();
    constexpr int NPC = BN / 8;
    for (int idx = tid; idx < BM * NPC; idx += 256) {
        const int row = idx / NPC, pc = idx - row * NPC;
        const int orow = rowmap[row], col = n0 + pc * 8;
        if (orow >= 0 && col < g.cout)
            *reinterpret_cast<uint4*>(Y + (long)orow * g.cout + col) = *reinterpret_cast<const uint4*>(Cs + row * LDC + pc * 8);
    }
}

// ---------------------------------------------------------------------------------------------
// filter gradient:  GW[(t,ci), co] += sum_m X[src(m,t), ci] * GY[m, co]   (fp32 output, split over m, fp32 atomics)
// ---------------------------------------------------------------------------------------------
// Filter-gradient launches: (tap-channel tile, cout tile, row slice) of this workgroup.  tiles_x == 0: the 3-D grid as it is.  Else
// the XCD-aware 1-D order (the fp32 kernels' rule, igemm_conv.hip): workgroup id runs on XCD id % 8, and every tile of ONE row
// slice goes to the same XCD -- the slice of X and GY they all read enters that XCD's L2 once.  false: a padding workgroup.
__device__ __forceinline__ bool wgrad_tile_of_workgroup(int tiles_x, int tiles_y, int nsplits, int& bx, int& by, int& bz) {
    bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
    if (!tiles_x) return true;
    const int TT = tiles_x * tiles_y, id = blockIdx.x;
    const int grp = id / (8 * TT), r = id - grp * 8 * TT;
    bz = grp * 8 + (r & 7);
    if (bz >= nsplits) return false;
    const int t = r >> 3;
    by = t / tiles_x;
    bx = t - by * tiles_x;
    return true;
}

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void igemm_bf16_wgrad_tr_kernel(CnConvGeom g, const bf16_t* __restrict__ X,
                                                               const bf16_t* __restrict__ GY, float* __restrict__ GW,
                                                               int rows_per_split, int tiles_x, int tiles_y, int nsplits) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    constexpr int IPA = BM / 8, IPB = BN / 8;              // 8-channel pieces per position in each tile
    int bx, by, bz;
    if (!wgrad_tile_of_workgroup(tiles_x, tiles_y, nsplits, bx, by, bz)) return;
    constexpr int AT = (IPA * 16 + 255) / 256, BT = (IPB * 16 + 255) / 256;   // (piece, position pair) tasks per thread
    // Row-major [reduction row][channel] images exactly as they sit in memory (16-byte stores, no transposition); the MFMA
    // operands -- 8 consecutive reduction rows of ONE channel per lane -- come out of ds_read_b64_tr_b16: the 16 lanes of a
    // group address a [4 rows][16 channels] block (lane s: row s / 4, channels 4 (s % 4) ..) and lane i receives column i.
    // Row pitch = 64 or 192 (mod 256) bytes: the 4 rows x 64 bytes that 32 lanes read then fall on distinct banks.
    constexpr int PA = tr_pitch(BM), PB = tr_pitch(BN);
    __shared__ __attribute__((aligned(16))) bf16_t As[2][KS][PA];
    __shared__ __attribute__((aligned(16))) bf16_t Bs[2][KS][PB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, half = lane >> 5, l31 = lane & 31;
    const int M = g.n * g.out_d * g.out_h * g.out_w;
    const int T = g.k_d * g.k_h * g.k_w;
    const int Ktot = T * g.cin;
    const int i0 = bx * BM, n0 = by * BN;
    const int mbeg = bz * rows_per_split;
    const int mend = min(M, mbeg + rows_per_split);
    if (mbeg >= mend) return;

    // A tasks: piece ip (8 consecutive (tap, ci) rows: one tap, cin % 8 == 0) x position pair mp; fixed per thread
    int a_ip[AT], a_mp[AT], a_ci[AT], a_kd[AT], a_kh[AT], a_kw[AT];
    bool a_on[AT];
    int p_n[AT][2], p_d[AT][2], p_h[AT][2], p_w[AT][2], p_m[AT][2];
#pragma unroll
    for (int t = 0; t < AT; ++t) {
        const int task = tid + 256 * t;
        a_ip[t] = task % IPA;
        a_mp[t] = task / IPA;
        const int i = i0 + a_ip[t] * 8;
        a_on[t] = a_mp[t] < 16 && i < Ktot;
        const int tap = a_on[t] ? i / g.cin : 0;
        a_ci[t] = a_on[t] ? i - tap * g.cin : 0;
        tap_decode(g, tap, a_kd[t], a_kh[t], a_kw[t]);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int m = mbeg + 2 * a_mp[t] + e;
            p_m[t][e] = m;
            p_w[t][e] = m % g.out_w; m /= g.out_w;
            p_h[t][e] = m % g.out_h; m /= g.out_h;
            p_d[t][e] = m % g.out_d;
            p_n[t][e] = m / g.out_d;
        }
    }
    int b_ip[BT], b_mp[BT];
#pragma unroll
    for (int t = 0; t < BT; ++t) {
        const int task = tid + 256 * t;
        b_ip[t] = task % IPB;
        b_mp[t] = task / IPB;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    uint4 ra[AT][2], rb[BT][2];
    const int nks = (mend - mbeg + KS - 1) / KS;

    auto load_tiles = [&](int ks) {      // called with ks = 0, 1, 2, ... in order (the row coordinates advance by carries)
#pragma unroll
        for (int t = 0; t < AT; ++t) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                RowInfo r;
                r.ok = a_on[t] && p_m[t][e] < mend;
                r.nbase = p_n[t][e] * g.in_d;
                r.vd = p_d[t][e] * g.s_d - g.p_d;
                r.vh = p_h[t][e] * g.s_h - g.p_h;
                r.vw = p_w[t][e] * g.s_w - g.p_w;
                p_m[t][e] += KS;
                p_w[t][e] += KS;
                while (p_w[t][e] >= g.out_w) {
                    p_w[t][e] -= g.out_w;
                    if (++p_h[t][e] == g.out_h) {
                        p_h[t][e] = 0;
                        if (++p_d[t][e] == g.out_d) {
                            p_d[t][e] = 0;
                            ++p_n[t][e];
                        }
                    }
                }
                const int off = src_off(g, r, a_kd[t], a_kh[t], a_kw[t]);
                ra[t][e] = off >= 0 ? *reinterpret_cast<const uint4*>(X + off + a_ci[t]) : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < BT; ++t) {
            const int col = n0 + b_ip[t] * 8;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int m = mbeg + ks * KS + 2 * b_mp[t] + e;
                rb[t][e] = (b_mp[t] < 16 && m < mend && col < g.cout) ? *reinterpret_cast<const uint4*>(GY + (long)m * g.cout + col)
                                                                       : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int t = 0; t < AT; ++t) {
            if (a_mp[t] >= 16) continue;
#pragma unroll
            for (int e = 0; e < 2; ++e) *reinterpret_cast<uint4*>(&As[buf][2 * a_mp[t] + e][a_ip[t] * 8]) = ra[t][e];
        }
#pragma unroll
        for (int t = 0; t < BT; ++t) {
            if (b_mp[t] >= 16) continue;
#pragma unroll
            for (int e = 0; e < 2; ++e) *reinterpret_cast<uint4*>(&Bs[buf][2 * b_mp[t] + e][b_ip[t] * 8]) = rb[t][e];
        }
    };

    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    const int a_col0 = wm * 32 * TM, b_col0 = wn * 32 * TN;
    for (int ks = 0; ks < nks; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nks) load_tiles(ks + 1);
        Frag a[2][TM], b[2][TN];
        {
            // this lane's part of its 16-lane group's block: row (lane & 15) >> 2, channels 16 * ((lane >> 4) & 1) + 4 * (lane & 3) ..
            const int trow = (lane & 15) >> 2, tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int k0 = 16 * s + 8 * half + trow;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    a[s][i].h[0] = tr_read(&As[buf][k0][a_col0 + 32 * i + tcol]);
                    a[s][i].h[1] = tr_read(&As[buf][k0 + 4][a_col0 + 32 * i + tcol]);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    b[s][j].h[0] = tr_read(&Bs[buf][k0][b_col0 + 32 * j + tcol]);
                    b[s][j].h[1] = tr_read(&Bs[buf][k0 + 4][b_col0 + 32 * j + tcol]);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][i].v, b[s][j].v, acc[i][j], 0, 0, 0);
        if (ks + 1 < nks) store_tiles(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * 32 * TN + 32 * j + l31;
        if (col >= g.cout) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rbase = i0 + wm * 32 * TM + 32 * i + 4 * half;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < Ktot) unsafeAtomicAdd(&GW[(long)row * g.cout + col], acc[i][j][r]);
            }
        }
    }
}

template <int WM, int WN, int TM, int TN>
int launch_bf16(const CnConvGeom& g, int par, int flip, const bf16_t* x, const bf16_t* wb, const float* bias, bf16_t* y,
                int act, float slope, hipStream_t s) {
    const long M = (long)g.n * g.out_d * g.out_h * g.out_w;
    dim3 grid(cn_cdiv(M, 32 * WM * TM), cn_cdiv(g.cout, 32 * WN * TN));
    hipLaunchKernelGGL((igemm_bf16_kernel<WM, WN, TM, TN>), grid, dim3(256), 0, s, g, x, wb, bias, y, act, slope, par, flip);
    CN_LAUNCH_CHECK();
    return CN_OK;
}

int conv_bf16(const CnConvGeom& g, int flip, const bf16_t* x, const bf16_t* wb, const float* bias, bf16_t* y, int act,
              float slope, hipStream_t s) {
    if (g.cin % 8 || g.cout % 8) return CN_EUNSUPPORTED;
    const long M = (long)g.n * g.out_d * g.out_h * g.out_w;
    const int par = parity_ordered(g);
    // tile choice: the biggest tile that still gives >= 2 workgroups per CU (256 CUs); these kernels are memory bound,
    // occupancy hides the gather latency
    const long t128 = (long)cn_cdiv(M, 128) * cn_cdiv(g.cout, 128);
    const long t128x64 = (long)cn_cdiv(M, 128) * cn_cdiv(g.cout, 64);
    int cfg;
    if (g.cout <= 32) cfg = 3;
    else if (g.cout > 64 && t128 >= 512) cfg = 0;
    else if (t128x64 >= 512) cfg = 1;
    else cfg = 2;
    if (g.cout % 96 == 0 && g.cout % 128 != 0 && (long)cn_cdiv(M, 128) * (g.cout / 96) >= 256) cfg = 4;
    cn_prof_begin(s, conv_flops(g), conv_bytes(g, 2.0, 2.0, 2.0), CN_FAM_BF16_FWD);
    int e = CN_EUNSUPPORTED;
    // the LDS-DMA main loop (fwd2.hip) where the reduction axis is a whole number of 32-element stages per tap
    if (cfg != 3 && g.cin % 32 == 0 && g.cout >= 48) e = cn_fwd2_bf16(g, cfg, flip, x, wb, bias, y, act, slope, par, s);
    if (e == CN_EUNSUPPORTED)
    switch (cfg) {
        case 3: e = launch_bf16<4, 1, 1, 1>(g, par, flip, x, wb, bias, y, act, slope, s); break;   // 128 x 32
        case 4: e = launch_bf16<4, 1, 1, 3>(g, par, flip, x, wb, bias, y, act, slope, s); break;   // 128 x 96
        case 0: e = launch_bf16<2, 2, 2, 2>(g, par, flip, x, wb, bias, y, act, slope, s); break;   // 128 x 128
        case 1: e = launch_bf16<2, 2, 2, 1>(g, par, flip, x, wb, bias, y, act, slope, s); break;   // 128 x 64
        default: e = launch_bf16<2, 2, 1, 1>(g, par, flip, x, wb, bias, y, act, slope, s); break;  // 64 x 64
    }
    cn_prof_end(s);
    return e;
}

template <int WM, int WN, int TM, int TN>
int launch_bf16_wgrad(const CnConvGeom& g, const bf16_t* x, const bf16_t* gy, float* gw, hipStream_t s) {
    constexpr int BMt = 32 * WM * TM, BNt = 32 * WN * TN;
    const long M = (long)g.n * g.out_d * g.out_h * g.out_w;
    const long Ktot = (long)g.k_d * g.k_h * g.k_w * g.cin;
    const long tiles = (long)cn_cdiv(Ktot, BMt) * cn_cdiv(g.cout, BNt);
    // Row slices (round-5 sweep, profiles/round5_bf16_wgrad_splits.txt).  What the sweep showed: (1) a slice shorter than ~512
    // rows is mostly prologue + the tile's atomic adds; (2) one workgroup more than the CUs hold at once costs a whole extra
    // round -- the kernels hold 3 (128 x 128), 4 (128 x 96) or 5 workgroups per CU -- and the narrow tiles like two rounds;
    // (3) with the XCD order a slice count that is not a multiple of 8 leaves XCDs with one slice more than others.
    constexpr long min_rows = 512;
    long splits;
    {
        const long per_cu = BMt * BNt >= 128 * 128 ? 3 : BMt * BNt >= 128 * 96 ? 4 : 5;
        const long target = 256 * per_cu * (BNt >= 96 ? 1 : 2);
        splits = target / tiles;
        if (splits > M / min_rows) splits = M / min_rows;
        const long fill = std::min<long>(M / 256, (256 + tiles - 1) / tiles);        // ... but at least one workgroup per CU
        if (splits < fill) splits = fill;
        if (splits >= 16) splits &= ~7L;
        if (splits < 1) splits = 1;
    }
    long rows = (M + splits - 1) / splits;
    if (rows < 256) rows = std::min<long>(256, (M + KS - 1) / KS * KS);
    rows = (rows + KS - 1) / KS * KS;
    splits = (M + rows - 1) / rows;
    dim3 grid(cn_cdiv(Ktot, BMt), cn_cdiv(g.cout, BNt), (unsigned)splits);
    int tx = 0, ty = 0;
    if (grid.x * grid.y > 1 && splits >= 16) {
        tx = (int)grid.x; ty = (int)grid.y;
        grid = dim3((unsigned)(cn_cdiv(splits, 8) * 8 * tx * ty), 1, 1);
    }
    hipLaunchKernelGGL((igemm_bf16_wgrad_tr_kernel<WM, WN, TM, TN>), grid, dim3(256), 0, s, g, x, gy, gw, (int)rows, tx, ty, (int)splits);
    CN_LAUNCH_CHECK();
    return CN_OK;
}

}  // namespace

extern "C" int cn_conv_weight_prep_bf16(const float* w, uint16_t* wf, uint16_t* wd, int taps, int cin, int cout, void* stream) {
    CN_CHECK_ARG(w && (wf || wd) && taps > 0 && cin > 0 && cout > 0, "bad weight_prep args");
    const long total = (long)taps * cin * cout;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(wprep_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wf, wd, taps, cin, cout);
    CN_LAUNCH_CHECK();
    return CN_OK;
}

extern "C" int cn_conv_fwd_bf16(const CnConvGeom* gp, const uint16_t* x, const uint16_t* wf, const float* bias, uint16_t* y,
                                int act, float slope, void* stream) {
    if (int e = check_geom(gp)) return e;
    CN_CHECK_ARG(x && wf && y, "NULL tensor");
    CN_CHECK_ARG((((uintptr_t)x | (uintptr_t)wf | (uintptr_t)y) & 15) == 0, "bf16 convolution needs 16-byte aligned tensors");
    return conv_bf16(*gp, 0, x, wf, bias, y, act, slope, (hipStream_t)stream);
}

extern "C" int cn_conv_dgrad_bf16(const CnConvGeom* gp, const uint16_t* gy, const uint16_t* wd, uint16_t* gu, void* stream) {
    if (int e = check_geom(gp)) return e;
    CN_CHECK_ARG(gy && wd && gu, "NULL tensor");
    CN_CHECK_ARG(gp->dl_d == 1 && gp->dl_h == 1 && gp->dl_w == 1, "dgrad of a dilated-input geometry is not defined here");
    CN_CHECK_ARG((((uintptr_t)gy | (uintptr_t)wd | (uintptr_t)gu) & 15) == 0, "bf16 convolution needs 16-byte aligned tensors");
    CnConvGeom d = *gp;
    d.in_d = gp->out_d; d.in_h = gp->out_h; d.in_w = gp->out_w; d.cin = gp->cout;
    d.out_d = gp->in_d << gp->up; d.out_h = gp->in_h << gp->up; d.out_w = gp->in_w << gp->up;
    if (gp->nd == 2) d.out_d = 1;
    d.cout = gp->cin;
    d.s_d = d.s_h = d.s_w = 1;
    d.dl_d = gp->s_d; d.dl_h = gp->s_h; d.dl_w = gp->s_w;
    d.p_d = gp->k_d - 1 - gp->p_d; d.p_h = gp->k_h - 1 - gp->p_h; d.p_w = gp->k_w - 1 - gp->p_w;
    d.up = 0;
    return conv_bf16(d, 1, gy, wd, nullptr, gu, CN_ACT_NONE, 0.f, (hipStream_t)stream);
}

extern "C" int cn_conv_wgrad_bf16(const CnConvGeom* gp, const uint16_t* x, const uint16_t* gy, float* gw, int accumulate,
                                  void* stream) {
    if (int e = check_geom(gp)) return e;
    CN_CHECK_ARG(x && gy && gw, "NULL tensor");
    const CnConvGeom g = *gp;
    if (g.cin % 8 || g.cout % 8) return CN_EUNSUPPORTED;
    CN_CHECK_ARG((((uintptr_t)x | (uintptr_t)gy) & 15) == 0, "bf16 convolution needs 16-byte aligned tensors");
    hipStream_t s = (hipStream_t)stream;
    const long Ktot = (long)g.k_d * g.k_h * g.k_w * g.cin;
    if (!accumulate) {
        if (int ez__ = cn_zero_async(gw, sizeof(float) * Ktot * g.cout, s)) return ez__;
    }
    cn_prof_begin(s, conv_flops(g), conv_bytes(g, 2.0, 2.0, 4.0), CN_FAM_BF16_WGRAD);
    int e;
    if (g.cout <= 32) e = launch_bf16_wgrad<4, 1, 1, 1>(g, x, gy, gw, s);                                     // 128 (tap,ci) x 32 co
    else if (Ktot >= 128 && g.cout % 96 == 0 && g.cout % 128 != 0) e = launch_bf16_wgrad<4, 1, 1, 3>(g, x, gy, gw, s);   // 128 x 96
    else if (Ktot >= 128 && g.cout >= 128) e = launch_bf16_wgrad<2, 2, 2, 2>(g, x, gy, gw, s);                // 128 x 128
    else e = launch_bf16_wgrad<2, 2, 1, 1>(g, x, gy, gw, s);                                                  // 64 x 64
    cn_prof_end(s);
    return e;
}
