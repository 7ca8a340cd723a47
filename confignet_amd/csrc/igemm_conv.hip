// igemm_conv.hip -- N-d convolution family as implicit GEMM on the CDNA4 matrix cores.
//
// fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32 (exact f32, 157 TF peak on MI355X).
// One 256-thread workgroup (4 wave64, 2x2) owns a (64*TM) x (64*TN) output tile; each wave
// owns TM x TN 32x32 MFMA tiles.  K is walked in steps of 16: the gathered activation tile
// and the filter tile are prefetched from HBM/L2 into registers while the previous step is
// multiplied out of LDS (two LDS buffers, one barrier per step).  Both LDS tiles are
// "k-major" ([k][m] / [k][n], row pitch +4 floats) so every MFMA operand read is a
// conflict-free ds_read_b32 of 32 consecutive floats per half-wave.
//
// Layouts: activations channels-last (N,[D,]H,W,C); filters Keras (kd,kh,kw,cin,cout) ==
// row-major [K = taps*cin][cout], i.e. already the B matrix of the GEMM.  The x2 nearest
// upsample that precedes most generator convolutions (hologan_generator.py:139-170) is folded
// into the gather (index >> 1); SAME padding ([TF-2.1] asymmetric, low = total//2) and the
// zero-stuffing of strided data-gradients are predicates of the gather.
#include "common.h"

#include "mma_tile.h"
#include "typed.h"
#include "conv_geom.h"

namespace {

// ---------------------------------------------------------------------------------------------
// forward / data-gradient:  Y[m, co] = act( sum_{t,ci} X[src(m,t), ci] * W[t, ci, co] + bias[co] )
// ---------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN, bool VEC, bool BVEC = true, int KB = BK>
__global__ __launch_bounds__(256) void igemm_fwd_kernel(CnConvGeom g, const float* __restrict__ X,
                                                        const float* __restrict__ W, const float* __restrict__ bias,
                                                        float* __restrict__ Y, int act, float slope, int par,
                                                        int xcd_swizzle, int ntiles_m, int ntiles_n, int bt = 0,
                                                        long part_stride = 0, const float* __restrict__ res = nullptr) {
    // res (unsplit launches only): a tensor of Y's shape added before the activation (residual branch of a ResNet block)
    // bt: W is the ORIGINAL filter [t][n = output channel here][k = reduction channel here] of the convolution whose data
    // gradient this launch computes (tap order reversed, the two channel axes swapped): the B tile is then loaded like the
    // gathered A tile (16-byte pieces along k, transposed on the way into LDS) -- no tap-flipped, channel-transposed copy of
    // every trainable filter per step (cn_conv_weight_tflip: 80 launches per iteration)
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN, LDA = BM + 4, LDB = BN + 4;
    constexpr int KQ = KB / 4;                          // float4 pieces per A row per K step
    constexpr int RPP = 256 / KQ;                       // A rows loaded per pass of the workgroup
    constexpr int AP = BM / RPP, BP = (KB * BN / 4 + 255) / 256;   // float4 loads per thread per K step
    static_assert(!VEC ? KB == BK : true, "scalar gather path uses the 16-deep stage");
    __shared__ float As[2][KB][LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][KB][LDB];
    __shared__ int rowmap[BM];                          // tile row -> output row (or -1)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, half = lane >> 5, l31 = lane & 31;
    const int M = g.n * g.out_d * g.out_h * g.out_w;
    const int T = g.k_d * g.k_h * g.k_w;
    const int Ktot = T * g.cin;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch order); give each XCD a contiguous
    // run of M tiles so that neighbouring tiles (shared input halo rows) hit the same 4 MiB L2.  Bijective for
    // any grid size; placement only affects speed, never results.
    int bx = blockIdx.x, by = blockIdx.y;
    if ((xcd_swizzle & 3) == 2) {
        // 1-D launch over all (M tile, cout tile) pairs: XCD = id % 8 gets a contiguous run of M tiles and, within it, the
        // cout tiles of one M tile in consecutive slots -- they read the same input rows, which then come from that XCD's
        // L2 instead of being fetched once per cout tile
        const int nt = ntiles_n, nmt = ntiles_m;
        const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
        const int q = nmt >> 3, r = nmt & 7;                              // M tiles per XCD: q (+1 for the first r XCDs)
        const int mine = q + (xcd < r ? 1 : 0);
        int ml;
        divmod_pos(j, nt, ml, by);
        if (ml >= mine) return;                                           // padding slot of the rounded-up launch
        bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + ml;
    } else if (xcd_swizzle && !par) {   // parity-ordered rows: classes have 1/2/2/4 live taps, keep them interleaved over XCDs
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bx & 7, idx = bx >> 3;
        bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = bx * BM, n0 = by * BN;

    const int kq = tid % KQ, arow = tid / KQ;
    RowInfo ri[AP];
    unsigned long long tapmask = T >= 64 ? ~0ull : ((1ull << T) - 1ull);
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        int mrow = m0 + arow + RPP * i;
        if (par) {
            int cls;
            mrow = par_row(g, mrow, M, cls);
        }
        ri[i] = decode_row(g, mrow, M);
        if (kq == 0) rowmap[arow + RPP * i] = ri[i].ok ? mrow : -1;
    }
    if (par) {
        int c0, c1;
        par_row(g, m0, M, c0);
        par_row(g, min(m0 + BM, M) - 1, M, c1);
        if (c0 == c1) tapmask = par_tap_mask(g, c0);   // whole tile in one parity class: skip dead taps
    }
    int aoff[AP];

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // two register sets: loads of K step s+2 are issued while step s is multiplied and step s+1 is still in flight, so a
    // gathered operand has two steps of MFMA time to arrive (small tiles give a workgroup only 512 MFMA cycles per step and
    // small problems only 1-2 workgroups per CU: one step did not cover the L2/HBM latency -- MFMA-busy 0.41 on the 64x64 tile)
    float4 ra0[AP], rb0[BP], ra1[AP], rb1[BP];
    const int cpb = VEC ? g.cin / KB : 1;
    const int nks_all = VEC ? __popcll(tapmask) * cpb : (Ktot + BK - 1) / BK;
    // split-K over gridDim.z (small-M problems): this workgroup walks K steps [ks_beg, ks_end)
    const int per_z = (nks_all + gridDim.z - 1) / gridDim.z;
    const int ks_beg = blockIdx.z * per_z, ks_end = min(nks_all, ks_beg + per_z);
    int cur_ord = -1, cur_tap = -1;                     // ordinal among live taps / tap index
    const bool tap_minor = VEC && xcd_swizzle >= 2 && !par && gridDim.z == 1 && T > 1 && T <= 9 && (xcd_swizzle & 4);
    extern __shared__ int offtab[];                     // tap-minor order: gather offset of every (tap, row of this thread)
    if (tap_minor) {
        for (int t = 0; t < T; ++t) {
            int kd, kh, kw;
            tap_decode(g, t, kd, kh, kw);
#pragma unroll
            for (int i = 0; i < AP; ++i) offtab[(t * AP + i) * 256 + tid] = src_off(g, ri[i], kd, kh, kw);
        }
    }

    // K order.  Tap-major (all channel chunks of a tap, then the next tap) recomputes the gather offsets once per tap.
    // Tap-minor (tap_minor != 0: all taps of a channel chunk, then the next chunk) recomputes them every step but keeps
    // the XCD's working set at (tiles in flight) x (rows) x KB channels, so the taps' shifted re-reads of a chunk hit
    // the 4 MiB L2 instead of going back to the fabric (stride-1 layers with many channels).
    auto load_tiles = [&](int ks, float4 (&ra)[AP], float4 (&rb)[BP], unsigned& amask) {
        if (VEC) {
            int c0;
            if (tap_minor) {
                const int chunk = ks / T;
                cur_tap = ks - chunk * T;
                c0 = chunk * KB;
#pragma unroll
                for (int i = 0; i < AP; ++i) aoff[i] = offtab[(cur_tap * AP + i) * 256 + tid];
            } else {
                const int ord = ks / cpb;
                c0 = (ks - ord * cpb) * KB;
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
            }
            const int tap = cur_tap;
            // every load below is UNCONDITIONAL (dead rows / columns read a clamped, valid address and are zeroed on the way
            // into LDS): a load inside a branch makes the compiler's waitcnt insertion assume the branch was skipped, and it
            // then drains vmcnt to 0 before every LDS store -- the two-step lookahead of the register sets would be lost
            amask = 0;
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                ra[i] = *reinterpret_cast<const float4*>(X + max(aoff[i], 0) + c0 + kq * 4);
                amask |= (aoff[i] >= 0 ? 1u : 0u) << i;
            }
            if (bt) {
#pragma unroll
                for (int j = 0; j < BP; ++j) {
                    const int idx = min(tid + 256 * j, KB * BN / 4 - 1);
                    const int nn = min(n0 + idx / KQ, g.cout - 1), kk = (idx % KQ) * 4;
                    rb[j] = *reinterpret_cast<const float4*>(W + ((long)(T - 1 - tap) * g.cout + nn) * g.cin + c0 + kk);
                }
            } else {
#pragma unroll
                for (int j = 0; j < BP; ++j) {
                    const int idx = BVEC ? min(tid + 256 * j, KB * BN / 4 - 1) : tid + 256 * j;
                    const int brow = idx / (BN / 4), col = n0 + (idx % (BN / 4)) * 4;
                    const long kg = (long)tap * g.cin + c0 + brow;
                    if (BVEC) {
                        rb[j] = *reinterpret_cast<const float4*>(W + kg * g.cout + min(col, g.cout - 4));
                    } else {   // thin cout (3-channel image gradients): guarded scalar filter loads
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (col + e < g.cout && idx < KB * BN / 4) ? W[kg * g.cout + col + e] : 0.f;
                        rb[j] = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
            }
        } else {
            amask = ~0u;
#pragma unroll
            for (int i = 0; i < AP; ++i) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = ks * BK + kq * 4 + e;
                    v[e] = 0.f;
                    if (k < Ktot) {
                        const int tap = k / g.cin, ci = k - tap * g.cin;
                        int kd, kh, kw;
                        tap_decode(g, tap, kd, kh, kw);
                        const int off = src_off(g, ri[i], kd, kh, kw);
                        if (off >= 0) v[e] = X[off + ci];
                    }
                }
                ra[i] = make_float4(v[0], v[1], v[2], v[3]);
            }
#pragma unroll
            for (int j = 0; j < BP; ++j) {
                const int idx = tid + 256 * j;
                const int brow = idx / (BN / 4), col = n0 + (idx % (BN / 4)) * 4;
                const long kg = (long)ks * BK + brow;
                rb[j] = (col < g.cout && kg < Ktot && idx < KB * BN / 4)
                            ? *reinterpret_cast<const float4*>(W + kg * g.cout + col)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto store_tiles = [&](int buf, const float4 (&ra)[AP], const float4 (&rb)[BP], unsigned amask) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int r = arow + RPP * i;
            const bool live = (amask >> i) & 1u;     // padding taps / rows past the end: zeros (the load read a clamped address)
            As[buf][kq * 4 + 0][r] = live ? ra[i].x : 0.f;
            As[buf][kq * 4 + 1][r] = live ? ra[i].y : 0.f;
            As[buf][kq * 4 + 2][r] = live ? ra[i].z : 0.f;
            As[buf][kq * 4 + 3][r] = live ? ra[i].w : 0.f;
        }
        if (VEC && bt) {
#pragma unroll
            for (int j = 0; j < BP; ++j) {
                const int idx = tid + 256 * j;
                const int nn = idx / KQ, kk = (idx % KQ) * 4;
                if (idx < KB * BN / 4) {
                    Bs[buf][kk + 0][nn] = rb[j].x;
                    Bs[buf][kk + 1][nn] = rb[j].y;
                    Bs[buf][kk + 2][nn] = rb[j].z;
                    Bs[buf][kk + 3][nn] = rb[j].w;
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            const int idx = tid + 256 * j;
            const int brow = idx / (BN / 4), bcol = (idx % (BN / 4)) * 4;
            if (idx < KB * BN / 4) *reinterpret_cast<float4*>(&Bs[buf][brow][bcol]) = rb[j];
        }
    };

    const int a_col = wm * 32 * TM + l31, b_col = wn * 32 * TN + l31;
    // The steps past the end re-load the last step (clamped index) and store it into the LDS buffer nobody reads again:
    // no branch around a load or a store in the loop (see load_tiles).  Dead filter columns (>= cout) carry whatever the
    // clamped address held: they only reach accumulator columns the epilogue never stores.
    unsigned am0 = 0, am1 = 0;
    if (ks_beg < ks_end) {
        const int ks_last = ks_end - 1;
        load_tiles(ks_beg, ra0, rb0, am0);
        store_tiles(0, ra0, rb0, am0);
        load_tiles(min(ks_beg + 1, ks_last), ra1, rb1, am1);
        __syncthreads();
        int ks = ks_beg;
        for (; ks + 1 < ks_end; ks += 2) {
            // even step: LDS buffer 0 holds step ks, set 1 holds step ks+1 (in flight), step ks+2 goes to set 0
            load_tiles(min(ks + 2, ks_last), ra0, rb0, am0);
            mma_step<TM, TN, LDA, LDB, KB>(As[0], Bs[0], acc, a_col, b_col, half);
            store_tiles(1, ra1, rb1, am1);
            __syncthreads();
            // odd step: buffer 1 holds step ks+1, set 0 holds step ks+2, step ks+3 goes to set 1
            load_tiles(min(ks + 3, ks_last), ra1, rb1, am1);
            mma_step<TM, TN, LDA, LDB, KB>(As[1], Bs[1], acc, a_col, b_col, half);
            store_tiles(0, ra0, rb0, am0);
            __syncthreads();
        }
        if (ks < ks_end) mma_step<TM, TN, LDA, LDB, KB>(As[0], Bs[0], acc, a_col, b_col, half);   // odd number of steps: the last one
    } else {
        // no K step at all (an empty K split; a parity-ordered tile whose class has no live tap: three of the four classes of a 1x1
        // stride-2 data gradient): the epilogue below reads rowmap entries that OTHER waves wrote, and without the loop's barriers
        // nothing ordered those writes before the reads -- a wave that ran ahead stored its (zero) rows through whatever the LDS
        // held before (found with the real halves of the discriminator steps shifted under the generator tail: ResNet-50's
        // 64x64x256 <- 32x32x128 data gradient wrote through stale floats and faulted)
        __syncthreads();
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool split = gridDim.z > 1;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * 32 * TN + 32 * j + l31;
        if (col >= g.cout) continue;
        const float bv = (bias && blockIdx.z == 0) ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rbase = wm * 32 * TM + 32 * i + 4 * half;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rowmap[rbase + (r & 3) + 8 * (r >> 2)];
                if (row < 0) continue;
                const float v = acc[i][j][r] + bv;
                // (deterministic mode: every K split stores its partial tile in its own slab; the slabs are added in order afterwards)
                if (part_stride) Y[(long)blockIdx.z * part_stride + (long)row * g.cout + col] = v;
                else if (split) unsafeAtomicAdd(&Y[(long)row * g.cout + col], v);
                else Y[(long)row * g.cout + col] = cn_apply_act(res ? v + res[(long)row * g.cout + col] : v, act, slope);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// filter gradient:  GW[(t,ci), co] = sum_m X[src(m,t), ci] * GY[m, co]   (split over m, fp32 atomics)
// ---------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN, bool VEC, bool BVEC>
__global__ __launch_bounds__(256) void igemm_wgrad_kernel(CnConvGeom g, const float* __restrict__ X,
                                                          const float* __restrict__ GY, float* __restrict__ GW,
                                                          int rows_per_split, float* __restrict__ parts = nullptr,
                                                          int tiles_x = 0, int tiles_y = 0, int nsplits = 0) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN, LDA = BM + 4, LDB = BN + 4;
    constexpr int AP = BM / 64, BP = (BN + 63) / 64;
    // XCD-aware order (tiles_x != 0: 1-D launch).  Workgroup id runs on XCD id % 8 (observed dispatch order): every (tap, ci) /
    // cout tile of ONE row slice goes to the same XCD, so the slice of X and GY that all of them read is fetched into that
    // XCD's L2 once instead of once per tile on eight different XCDs (PMC: 4 - 5 x the algorithmic bytes at the fabric with the
    // 3-D grid order).  Placement only affects speed.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (tiles_x) {
        const int T = tiles_x * tiles_y, id = blockIdx.x;
        const int grp = id / (8 * T), r = id - grp * 8 * T;
        bz = grp * 8 + (r & 7);
        if (bz >= nsplits) return;
        const int t = r >> 3;
        by = t / tiles_x;
        bx = t - by * tiles_x;
    }
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, half = lane >> 5, l31 = lane & 31;
    const int M = g.n * g.out_d * g.out_h * g.out_w;
    const int T = g.k_d * g.k_h * g.k_w;
    const int Ktot = T * g.cin;
    const int i0 = bx * BM, n0 = by * BN;
    const int mbeg = bz * rows_per_split;
    const int mend = min(M, mbeg + rows_per_split);
    if (mbeg >= mend) return;

    // A loader: float4 along i = (tap, ci); fixed per thread across the whole m loop
    int a_krow[AP], a_col[AP], a_ci[AP][4], a_kd[AP][4], a_kh[AP][4], a_kw[AP][4];
    bool a_ok[AP][4];
#pragma unroll
    for (int ii = 0; ii < AP; ++ii) {
        const int idx = tid + 256 * ii;
        a_krow[ii] = idx / (BM / 4);
        a_col[ii] = (idx % (BM / 4)) * 4;
#pragma unroll
        for (int e = 0; e < (VEC ? 1 : 4); ++e) {
            const int i = i0 + a_col[ii] + e;
            a_ok[ii][e] = i < Ktot;
            const int tap = a_ok[ii][e] ? i / g.cin : 0;
            a_ci[ii][e] = a_ok[ii][e] ? i - tap * g.cin : 0;
            tap_decode(g, tap, a_kd[ii][e], a_kh[ii][e], a_kw[ii][e]);
        }
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[AP], rb[BP];
    const int nks = (mend - mbeg + BK - 1) / BK;

    // output coordinates of this thread's gathered row, advanced by BK rows per K step with carries instead of
    // re-dividing the row index every step (load_tiles is called with ks = 0, 1, 2, ... in order)
    int p_m[AP], p_n[AP], p_d[AP], p_h[AP], p_w[AP];
#pragma unroll
    for (int ii = 0; ii < AP; ++ii) {
        int m = mbeg + a_krow[ii];
        p_m[ii] = m;
        divmod_pos(m, g.out_w, m, p_w[ii]);
        divmod_pos(m, g.out_h, m, p_h[ii]);
        divmod_pos(m, g.out_d, p_n[ii], p_d[ii]);
    }

    auto load_tiles = [&](int ks) {
#pragma unroll
        for (int ii = 0; ii < AP; ++ii) {
            RowInfo r;
            r.ok = p_m[ii] < mend;
            r.nbase = p_n[ii] * g.in_d;
            r.vd = p_d[ii] * g.s_d - g.p_d;
            r.vh = p_h[ii] * g.s_h - g.p_h;
            r.vw = p_w[ii] * g.s_w - g.p_w;
            p_m[ii] += BK;
            p_w[ii] += BK;
            while (p_w[ii] >= g.out_w) {
                p_w[ii] -= g.out_w;
                if (++p_h[ii] == g.out_h) {
                    p_h[ii] = 0;
                    if (++p_d[ii] == g.out_d) {
                        p_d[ii] = 0;
                        ++p_n[ii];
                    }
                }
            }
            if (VEC) {
                const int off = a_ok[ii][0] ? src_off(g, r, a_kd[ii][0], a_kh[ii][0], a_kw[ii][0]) : -1;
                ra[ii] = off >= 0 ? *reinterpret_cast<const float4*>(X + off + a_ci[ii][0])
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int off = a_ok[ii][e] ? src_off(g, r, a_kd[ii][e], a_kh[ii][e], a_kw[ii][e]) : -1;
                    v[e] = off >= 0 ? X[off + a_ci[ii][e]] : 0.f;
                }
                ra[ii] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            const int idx = tid + 256 * j;
            const int krow = idx / (BN / 4), col = n0 + (idx % (BN / 4)) * 4;
            const int m = mbeg + ks * BK + krow;
            const bool ok = m < mend && idx < BK * BN / 4;
            if (BVEC) {
                rb[j] = (ok && col < g.cout) ? *reinterpret_cast<const float4*>(GY + (long)m * g.cout + col)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (ok && col + e < g.cout) ? GY[(long)m * g.cout + col + e] : 0.f;
                rb[j] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int ii = 0; ii < AP; ++ii) *reinterpret_cast<float4*>(&As[buf][a_krow[ii]][a_col[ii]]) = ra[ii];
#pragma unroll
        for (int j = 0; j < BP; ++j) {
            const int idx = tid + 256 * j;
            if (idx < BK * BN / 4) *reinterpret_cast<float4*>(&Bs[buf][idx / (BN / 4)][(idx % (BN / 4)) * 4]) = rb[j];
        }
    };

    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    const int ac = wm * 32 * TM + l31, bc = wn * 32 * TN + l31;
    for (int ks = 0; ks < nks; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nks) load_tiles(ks + 1);
        mma_step<TM, TN, LDA, LDB>(As[buf], Bs[buf], acc, ac, bc, half);
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
                if (row >= Ktot) continue;
                // deterministic mode: this split's partial filter goes to its own slab, the slabs are added in split order
                if (parts) parts[((long)bz * Ktot + row) * g.cout + col] = acc[i][j][r];
                else unsafeAtomicAdd(&GW[(long)row * g.cout + col], acc[i][j][r]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// thin-output direct convolution (cout <= 4): HBM-bound, one thread per output position.
// Used for map_final (32->3, hologan_generator.py:101), the 1x1 3->3 from-RGB conv
// (hologan_discriminator.py:20) and the data-gradient of every 3-channel-input conv.
// ---------------------------------------------------------------------------------------------
template <int CO, bool VEC>
__global__ __launch_bounds__(256) void thin_conv_kernel(CnConvGeom g, const float* __restrict__ X,
                                                        const float* __restrict__ W, const float* __restrict__ bias,
                                                        float* __restrict__ Y, int act, float slope, int par) {
    extern __shared__ __attribute__((aligned(16))) float wsh[];     // [taps*cin][4] filter, broadcast reads
    const int M = g.n * g.out_d * g.out_h * g.out_w;
    const int Ktot = g.k_d * g.k_h * g.k_w * g.cin;
    for (int i = threadIdx.x; i < Ktot; i += 256) {
#pragma unroll
        for (int c = 0; c < 4; ++c) wsh[i * 4 + c] = c < CO ? W[(long)i * CO + c] : 0.f;
    }
    __syncthreads();
    int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    if (par) {
        int cls;
        m = par_row(g, m, M, cls);
    }
    const RowInfo r = decode_row(g, m, M);
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = bias ? bias[c] : 0.f;
    int tap = 0;
    for (int kd = 0; kd < g.k_d; ++kd)
        for (int kh = 0; kh < g.k_h; ++kh)
            for (int kw = 0; kw < g.k_w; ++kw, ++tap) {
                const int off = src_off(g, r, kd, kh, kw);
                if (off < 0) continue;
                const float4* wp = reinterpret_cast<const float4*>(wsh) + tap * g.cin;
                if (VEC) {
                    for (int ci = 0; ci < g.cin; ci += 4) {
                        const float4 xv = *reinterpret_cast<const float4*>(X + off + ci);
                        const float4 w0 = wp[ci], w1 = wp[ci + 1], w2 = wp[ci + 2], w3 = wp[ci + 3];
                        const float wv[4][4] = {{w0.x, w0.y, w0.z, w0.w}, {w1.x, w1.y, w1.z, w1.w},
                                                {w2.x, w2.y, w2.z, w2.w}, {w3.x, w3.y, w3.z, w3.w}};
#pragma unroll
                        for (int c = 0; c < CO; ++c)
                            acc[c] += xv.x * wv[0][c] + xv.y * wv[1][c] + xv.z * wv[2][c] + xv.w * wv[3][c];
                    }
                } else {
                    for (int ci = 0; ci < g.cin; ++ci) {
                        const float xv = X[off + ci];
                        const float4 w0 = wp[ci];
                        const float wv[4] = {w0.x, w0.y, w0.z, w0.w};
#pragma unroll
                        for (int c = 0; c < CO; ++c) acc[c] += xv * wv[c];
                    }
                }
            }
#pragma unroll
    for (int c = 0; c < CO; ++c) Y[(long)m * CO + c] = cn_apply_act(acc[c], act, slope);
}

// Cooperative thin-output convolution: G (= 8 or 16) lanes share one output pixel, lane c4 owning input
// channels [4*c4, 4*c4+4), so every global load instruction is a run of fully used 16-byte pieces
// (G*16 contiguous bytes per pixel), each thread carries PX pixels per filter read (4 broadcast-ish LDS
// reads feed 4*CO*PX FMAs), and the partial sums are combined with G-lane shuffles.  In parity-ordered
// mode (data-gradient of a stride-2 convolution into the 3-channel image) the per-class tap validity and
// coordinate shifts come from a small LDS table instead of per-thread integer divisions.
template <int CO, int G, int PX>
__global__ __launch_bounds__(256) void thin_conv_coop_kernel(CnConvGeom g, const float* __restrict__ X,
                                                             const float* __restrict__ W, const float* __restrict__ bias,
                                                             float* __restrict__ Y, int act, float slope, int par) {
    extern __shared__ __attribute__((aligned(16))) float wsh[];     // [taps*cin][4] filter
    __shared__ int tab[8][32];                                      // par: class x tap -> packed shifts / -1
    constexpr int PPB = 256 / G;                                    // pixel slots per block per step
    const int M = g.n * g.out_d * g.out_h * g.out_w;
    const int T = g.k_d * g.k_h * g.k_w;
    const int Ktot = T * g.cin;
    const int CL = g.cin / 4;
    for (int i = threadIdx.x; i < Ktot; i += 256) {
#pragma unroll
        for (int c = 0; c < 4; ++c) wsh[i * 4 + c] = c < CO ? W[(long)i * CO + c] : 0.f;
    }
    if (par && threadIdx.x < 8 * 32) {
        const int cls = threadIdx.x >> 5, tap = threadIdx.x & 31;
        int e = -1;
        if (cls < g.dl_d * g.dl_h * g.dl_w && tap < T) {
            const int cw = cls % g.dl_w, ch = (cls / g.dl_w) % g.dl_h, cd = cls / (g.dl_w * g.dl_h);
            int kd, kh, kw;
            tap_decode(g, tap, kd, kh, kw);
            const int vd = cd - g.p_d + kd, vh = ch - g.p_h + kh, vw = cw - g.p_w + kw;
            if (vd % g.dl_d == 0 && vh % g.dl_h == 0 && vw % g.dl_w == 0)
                e = ((vd / g.dl_d + 8) << 8) | ((vh / g.dl_h + 8) << 4) | (vw / g.dl_w + 8);   // shifts in [-8, 7]
        }
        tab[cls][tap] = e;
    }
    __syncthreads();
    const int slot = threadIdx.x / G, c4 = threadIdx.x % G;
    const bool lane_on = c4 < CL;
    const int qd_ext = g.out_d / g.dl_d, qh_ext = g.out_h / g.dl_h, qw_ext = g.out_w / g.dl_w;
    const int per = g.n * qd_ext * qh_ext * qw_ext;
    int nb[PX], xd[PX], xh[PX], xw[PX], cls[PX], mrow[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        const int mp = (blockIdx.x * PX + p) * PPB + slot;
        cls[p] = -1;
        mrow[p] = -1;
        nb[p] = xd[p] = xh[p] = xw[p] = 0;
        if (mp >= M) continue;
        if (par) {
            const int c = mp / per;
            int rem = mp - c * per;
            const int cw = c % g.dl_w, chh = (c / g.dl_w) % g.dl_h, cd = c / (g.dl_w * g.dl_h);
            xw[p] = rem % qw_ext; rem /= qw_ext;
            xh[p] = rem % qh_ext; rem /= qh_ext;
            xd[p] = rem % qd_ext;
            const int n = rem / qd_ext;
            nb[p] = n * g.in_d;
            cls[p] = c;
            mrow[p] = ((n * g.out_d + xd[p] * g.dl_d + cd) * g.out_h + xh[p] * g.dl_h + chh) * g.out_w + xw[p] * g.dl_w + cw;
        } else {
            int m = mp;
            int ow, oh, od, nn;
            divmod_pos(m, g.out_w, m, ow);
            divmod_pos(m, g.out_h, m, oh);
            divmod_pos(m, g.out_d, nn, od);
            nb[p] = nn * g.in_d;
            xd[p] = od * g.s_d - g.p_d; xh[p] = oh * g.s_h - g.p_h; xw[p] = ow * g.s_w - g.p_w;
            cls[p] = 0;
            mrow[p] = mp;
        }
    }
    float acc[PX][CO];
#pragma unroll
    for (int p = 0; p < PX; ++p)
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[p][c] = 0.f;
    const int ed = g.in_d << g.up, eh = g.in_h << g.up, ew = g.in_w << g.up;
    int tap = 0;
    for (int kd = 0; kd < g.k_d; ++kd)
        for (int kh = 0; kh < g.k_h; ++kh)
            for (int kw = 0; kw < g.k_w; ++kw, ++tap) {
                float4 xv[PX];
                bool any = false;
#pragma unroll
                for (int p = 0; p < PX; ++p) {
                    xv[p] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (cls[p] < 0 || !lane_on) continue;
                    int qd, qh, qw;
                    if (par) {
                        const int e = tab[cls[p]][tap];
                        if (e < 0) continue;
                        qd = xd[p] + ((e >> 8) & 15) - 8; qh = xh[p] + ((e >> 4) & 15) - 8; qw = xw[p] + (e & 15) - 8;
                        if (qd < 0 || qd >= g.in_d || qh < 0 || qh >= g.in_h || qw < 0 || qw >= g.in_w) continue;
                    } else {
                        qd = xd[p] + kd; qh = xh[p] + kh; qw = xw[p] + kw;
                        if (qd < 0 || qd >= ed || qh < 0 || qh >= eh || qw < 0 || qw >= ew) continue;
                        qd >>= g.up; qh >>= g.up; qw >>= g.up;
                    }
                    const long off = ((((long)nb[p] + qd) * g.in_h + qh) * g.in_w + qw) * g.cin + c4 * 4;
                    xv[p] = *reinterpret_cast<const float4*>(X + off);
                    any = true;
                }
                if (!any) continue;
                const float4* wp = reinterpret_cast<const float4*>(wsh) + (tap * g.cin + c4 * 4);
                const float4 w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3];
                const float wv[4][4] = {{w0.x, w0.y, w0.z, w0.w}, {w1.x, w1.y, w1.z, w1.w},
                                        {w2.x, w2.y, w2.z, w2.w}, {w3.x, w3.y, w3.z, w3.w}};
#pragma unroll
                for (int p = 0; p < PX; ++p)
#pragma unroll
                    for (int c = 0; c < CO; ++c)
                        acc[p][c] += xv[p].x * wv[0][c] + xv[p].y * wv[1][c] + xv[p].z * wv[2][c] + xv[p].w * wv[3][c];
            }
#pragma unroll
    for (int p = 0; p < PX; ++p) {
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            float v = acc[p][c];
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            acc[p][c] = v;
        }
        if (c4 == 0 && mrow[p] >= 0) {
#pragma unroll
            for (int c = 0; c < CO; ++c)
                Y[(long)mrow[p] * CO + c] = cn_apply_act(acc[p][c] + (bias ? bias[c] : 0.f), act, slope);
        }
    }
}

// from-RGB conv (3 -> 3): four pixels = three float4 per tensor per trip
__global__ __launch_bounds__(256) void tiny_wgrad_3x3_kernel(const float* __restrict__ X, const float* __restrict__ GY,
                                                             float* __restrict__ GW, long M, float* __restrict__ parts = nullptr) {
    float acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = 0.f;
    const long quads = M / 4;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < quads; q += (long)gridDim.x * 256) {
        const float4* xp = reinterpret_cast<const float4*>(X + q * 12);
        const float4* gp = reinterpret_cast<const float4*>(GY + q * 12);
        const float4 x0 = xp[0], x1 = xp[1], x2 = xp[2], g0 = gp[0], g1 = gp[1], g2 = gp[2];
        const float xv[12] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w};
        const float gv[12] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w, g2.x, g2.y, g2.z, g2.w};
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i * 3 + j] += xv[p * 3 + i] * gv[p * 3 + j];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (long m = quads * 4; m < M; ++m)
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) acc[i * 3 + j] += X[m * 3 + i] * GY[m * 3 + j];
    __shared__ float sh[4][9];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        float v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) sh[w][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        const float v = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
        if (parts) parts[blockIdx.x * 9 + threadIdx.x] = v;      // deterministic mode: added in block order afterwards
        else unsafeAtomicAdd(&GW[threadIdx.x], v);
    }
}

// filter gradient of a 1x1 convolution between thin tensors (cin, cout <= 4: the from-RGB conv,
// hologan_discriminator.py:20): a plain HBM-bound reduction gw[ci][co] = sum_m x[m][ci] gy[m][co]
__global__ __launch_bounds__(256) void tiny_wgrad_1x1_kernel(const float* __restrict__ X, const float* __restrict__ GY,
                                                             float* __restrict__ GW, long M, int cin, int cout,
                                                             float* __restrict__ parts = nullptr) {
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256) {
        float xv[4] = {0.f, 0.f, 0.f, 0.f}, gv[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < cin; ++i) xv[i] = X[m * cin + i];
        for (int j = 0; j < cout; ++j) gv[j] = GY[m * cout + j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] += xv[i] * gv[j];
    }
    __shared__ float sh[4][16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = acc[i][j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if (lane == 0) sh[w][i * 4 + j] = v;
        }
    __syncthreads();
    if (threadIdx.x < 16) {
        const int i = threadIdx.x >> 2, j = threadIdx.x & 3;
        if (i < cin && j < cout) {
            const float v = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
            if (parts) parts[blockIdx.x * (cin * cout) + i * cout + j] = v;
            else unsafeAtomicAdd(&GW[i * cout + j], v);
        }
    }
}

__global__ void weight_tflip_kernel(const float* __restrict__ W, float* __restrict__ Wt, int T, int cin, int cout) {
    const long total = (long)T * cin * cout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // i indexes Wt[t'][co][ci]
        const int ci = (int)(i % cin);
        const long r = i / cin;
        const int co = (int)(r % cout);
        const int tp = (int)(r / cout);
        Wt[i] = W[((long)(T - 1 - tp) * cin + ci) * cout + co];
    }
}

template <int ND, typename T>
__global__ void sumpool2_kernel(const T* __restrict__ GU, T* __restrict__ GX, int n, int d, int h, int w, int c4) {
    const long total = (long)n * d * h * w * c4;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int cc = (int)(i % c4);
    long t = i / c4;
    const int x = (int)(t % w);
    t /= w;
    const int y = (int)(t % h);
    t /= h;
    const int z = (int)(t % d);
    const int b = (int)(t / d);
    const int H2 = 2 * h, W2 = 2 * w, D2 = ND == 3 ? 2 * d : 1;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int dz = 0; dz < (ND == 3 ? 2 : 1); ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int zz = ND == 3 ? 2 * z + dz : 0;
                const float4 v = ld4<T>(GU + 4 * (((((long)b * D2 + zz) * H2 + 2 * y + dy) * W2 + 2 * x + dx) * c4 + cc));
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
    st4<T>(GX + 4 * i, s);
}

// ---------------------------------------------------------------------------------------------
// Data gradient of a 3x3 stride-2 convolution INTO a 3-channel image (first DiscrBlock of the discriminators and
// of the latent regressor: 29 launches per second-stage iteration).  As a gather -- one output pixel = <= 4 live
// taps x C channels x 3 outputs -- the 32-column MFMA tile spends 10x the useful work on padding.  Transposed,
// every INPUT pixel owns one small dense product
//     P[pixel][tap*3 + co] = sum_c gy[pixel][c] * wt[tap][c][co]        (C x 27, one 32-column MFMA block)
// and each output pixel is the sum of the <= 4 entries of P that land on it (col2im).  One workgroup: (TH+1) x (TW+1)
// input pixels (one halo row/column, on the side the padding fixes) -> P in LDS -> its 2TH x 2TW output pixels.
// No atomics, gy is read once (+ halo), K order is permuted so that a lane's A operand is one float4 load.
template <int NG, typename TI = float>   // C = 8 * NG; TI: storage type of gy (fp32 / bf16)
__global__ __launch_bounds__(256) void s2_image_dgrad_kernel(CnConvGeom g, const TI* __restrict__ GY,
                                                             const float* __restrict__ WT, float* __restrict__ Y) {
    constexpr int TH = 8, TW = 32, RW = TW + 1, R = (TH + 1) * RW, MT = (R + 31) / 32, PS = 28, C = 8 * NG;
    __shared__ float P[MT * 32][PS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    const int tiles_w = (g.in_w + TW - 1) / TW, tiles_h = (g.in_h + TH - 1) / TH;
    int b = blockIdx.x;
    const int tj = b % tiles_w; b /= tiles_w;
    const int ti = b % tiles_h;
    const int n = b / tiles_h;
    const int i0 = ti * TH, j0 = tj * TW;
    // output row y = 2i + p - kh (kh = 0..2): rows [2 i0, 2 i0 + 2 TH) are fed by input rows [i0 + off, i0 + off + TH]
    const int offh = g.p_h == 2 ? -1 : 0, offw = g.p_w == 2 ? -1 : 0;

    // B operand (K x 32 slice of wt, K permuted as below), resident in registers for the whole workgroup
    float breg[NG][4];
#pragma unroll
    for (int jg = 0; jg < NG; ++jg)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 8 * jg + 4 * half + q;
            breg[jg][q] = l31 < 27 ? WT[((l31 / 3) * C + k) * 3 + l31 % 3] : 0.f;
        }

    for (int mt = wave; mt < MT; mt += 4) {
        const int r = mt * 32 + l31;
        const int ri = r / RW, rj = r - ri * RW;
        const int ii = i0 + offh + ri, jj = j0 + offw + rj;
        const bool inb = r < R && ii >= 0 && ii < g.in_h && jj >= 0 && jj < g.in_w;
        // lane (row, half) holds channels 8 jg + 4 half + {0..3}: MFMA step (jg, q) contracts channel pair
        // {8 jg + q, 8 jg + 4 + q} -- any K order is fine as long as A and B agree
        const TI* src = GY + (((long)n * g.in_h + ii) * g.in_w + jj) * C + 4 * half;
        float4 a[NG];
#pragma unroll
        for (int jg = 0; jg < NG; ++jg)
            a[jg] = inb ? ld4<TI>(src + 8 * jg) : make_float4(0.f, 0.f, 0.f, 0.f);
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int jg = 0; jg < NG; ++jg) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[jg].x, breg[jg][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[jg].y, breg[jg][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[jg].z, breg[jg][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[jg].w, breg[jg][3], acc, 0, 0, 0);
        }
        if (l31 < PS) {
#pragma unroll
            for (int q = 0; q < 16; ++q) P[mt * 32 + 4 * half + (q & 3) + 8 * (q >> 2)][l31] = acc[q];
        }
    }
    __syncthreads();

    // col2im: 2TH x 2TW output pixels, 4 per thread
#pragma unroll
    for (int q = 0; q < (2 * TH * 2 * TW) / 256; ++q) {
        const int px = threadIdx.x + 256 * q;
        const int ly = px / (2 * TW), lx = px - ly * (2 * TW);
        const int y = 2 * i0 + ly, x = 2 * j0 + lx;
        if (y >= g.out_h || x >= g.out_w) continue;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int vy = y - g.p_h + kh;
            if (vy < 0 || (vy & 1) || (vy >> 1) >= g.in_h) continue;
            const int ri = (vy >> 1) - (i0 + offh);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int vx = x - g.p_w + kw;
                if (vx < 0 || (vx & 1) || (vx >> 1) >= g.in_w) continue;
                const float* pr = &P[ri * RW + (vx >> 1) - (j0 + offw)][(kh * 3 + kw) * 3];
                s0 += pr[0];
                s1 += pr[1];
                s2 += pr[2];
            }
        }
        float* dst = Y + (((long)n * g.out_h + y) * g.out_w + x) * 3;
        dst[0] = s0;
        dst[1] = s1;
        dst[2] = s2;
    }
}

// Data gradient of a 3x3 stride-1 convolution INTO a 3-channel image (VGG conv1_1 under the perceptual loss, twice per
// generator step).  Same transposition as s2_image_dgrad_kernel: P[pixel][tap*3 + co] = sum_c gy[pixel][c] wt[tap][c][co]
// for the (TH+2) x (TW+2) input pixels around a TH x TW output tile (one 32-column MFMA block per 32 pixels, gy read once
// + halo), then every output pixel sums its 9 entries of P.
template <int NG, typename TI = float>   // C = 8 * NG
__global__ __launch_bounds__(256) void s1_image_dgrad_kernel(CnConvGeom g, const TI* __restrict__ GY,
                                                             const float* __restrict__ WT, float* __restrict__ Y) {
    constexpr int TH = 8, TW = 32, RW = TW + 2, R = (TH + 2) * RW, MT = (R + 31) / 32, PS = 29, C = 8 * NG;
    __shared__ float P[MT * 32][PS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    const int tiles_w = (g.out_w + TW - 1) / TW, tiles_h = (g.out_h + TH - 1) / TH;
    int b = blockIdx.x;
    const int tj = b % tiles_w; b /= tiles_w;
    const int ti = b % tiles_h;
    const int n = b / tiles_h;
    const int i0 = ti * TH - g.p_h, j0 = tj * TW - g.p_w;          // first input row / column of the patch

    float breg[NG][4];
#pragma unroll
    for (int jg = 0; jg < NG; ++jg)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 8 * jg + 4 * half + q;
            breg[jg][q] = l31 < 27 ? WT[((l31 / 3) * C + k) * 3 + l31 % 3] : 0.f;
        }

    for (int mt = wave; mt < MT; mt += 4) {
        const int r = mt * 32 + l31;
        const int ri = r / RW, rj = r - ri * RW;
        const int ii = i0 + ri, jj = j0 + rj;
        const bool inb = r < R && ii >= 0 && ii < g.in_h && jj >= 0 && jj < g.in_w;
        const TI* src = GY + (((long)n * g.in_h + ii) * g.in_w + jj) * C + 4 * half;
        float4 a[NG];
#pragma unroll
        for (int jg = 0; jg < NG; ++jg)
            a[jg] = inb ? ld4<TI>(src + 8 * jg) : make_float4(0.f, 0.f, 0.f, 0.f);
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int jg = 0; jg < NG; ++jg) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[jg].x, breg[jg][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[jg].y, breg[jg][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[jg].z, breg[jg][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[jg].w, breg[jg][3], acc, 0, 0, 0);
        }
        if (l31 < 27) {
#pragma unroll
            for (int q = 0; q < 16; ++q) P[mt * 32 + 4 * half + (q & 3) + 8 * (q >> 2)][l31] = acc[q];
        }
    }
    __syncthreads();

    const int ly = threadIdx.x / TW, lx = threadIdx.x - ly * TW;
    const int y = ti * TH + ly, x = tj * TW + lx;
    if (y >= g.out_h || x >= g.out_w) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const float* pr = &P[(ly + kh) * RW + lx + kw][(kh * 3 + kw) * 3];   // out-of-image pixels hold zeros
            s0 += pr[0];
            s1 += pr[1];
            s2 += pr[2];
        }
    float* dst = Y + (((long)n * g.out_h + y) * g.out_w + x) * 3;
    dst[0] = s0;
    dst[1] = s1;
    dst[2] = s2;
}

// ---------------------------------------------------------------------------------------------
// Epilogue of the image-side forward kernels below: one wave's 32 output pixels (one row segment) x cout channels, accumulators
// in the 32 x 32 MFMA layout (lane = channel, register = pixel), -> bias, activation, store.  yrow: the segment's first pixel;
// pix: pixels of it that exist (>= 32: all).
// staged: the 32 pixels x cout values are one contiguous run of the NHWC output: pass them through LDS, 16 pixels at a time
// ([pixel][channel] = the run's own layout; st: 16 * 32 NB floats of this wave's), and write the run with 16-byte stores -- lane c
// writes bytes 16 c .. of it.  (Straight from the accumulators a lane owns ONE channel of 16 pixels: 4-byte -- in bf16 2-byte --
// stores, 64 of them per row; the bf16 variant took longer than the fp32 one.)  Needs cout % (16 / sizeof(TO)) == 0 and a
// 16-byte aligned tensor; otherwise the element-wise form.
template <int NB, typename TO>
__device__ __forceinline__ void image_row_epilogue(const f32x16 (&acc)[NB], float* st, bool staged, TO* yrow, int pix, int cout,
                                                   const float* __restrict__ bias, int act, float slope) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    if (staged) {
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int col = nb * 32 + l31;
                if (col < cout) {
                    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
                    for (int qq = 0; qq < 8; ++qq)
                        st[(4 * half + (qq & 3) + 8 * (qq >> 2)) * cout + col] = cn_apply_act(acc[nb][8 * ph + qq] + bv, act, slope);
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
            constexpr int CH = 16 / (int)sizeof(TO);                    // channels per 16-byte chunk
            const int cpp = cout / CH, nch = 16 * cpp;
            const int pix_left = pix - 16 * ph;                         // pixels of this half that exist (>= 16: all)
            for (int c = lane; c < nch; c += 64) {
                if (pix_left < 16 && c / cpp >= pix_left) continue;
                const float4 v0 = *reinterpret_cast<const float4*>(st + c * CH);
                if constexpr (sizeof(TO) == 4) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(yrow) + 16 * ph * cout + c * 4) = v0;
                } else {
                    const float4 v1 = *reinterpret_cast<const float4*>(st + c * CH + 4);
                    uint4 o;
                    o.x = (unsigned)f32_to_bf16(v0.x) | ((unsigned)f32_to_bf16(v0.y) << 16);
                    o.y = (unsigned)f32_to_bf16(v0.z) | ((unsigned)f32_to_bf16(v0.w) << 16);
                    o.z = (unsigned)f32_to_bf16(v1.x) | ((unsigned)f32_to_bf16(v1.y) << 16);
                    o.w = (unsigned)f32_to_bf16(v1.z) | ((unsigned)f32_to_bf16(v1.w) << 16);
                    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(yrow) + 16 * ph * cout + c * 8) = o;
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
        }
        return;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = nb * 32 + l31;
        if (col >= cout) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int px = 4 * half + (q & 3) + 8 * (q >> 2);
            if (px < pix) stf<TO>(yrow + (long)px * cout + col, cn_apply_act(acc[nb][q] + bv, act, slope));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// First layers: 3x3 convolution of a 3-channel image (DiscrBlock 0 of both discriminators and the latent regressor,
// VGG conv1_1): K = 27.  The generic kernel gathers those 27 values with per-element integer division (cin = 3 is not
// a float4).  Here a workgroup stages the input patch of an 8 x 32 output tile in LDS once (coalesced rows), the K
// axis is padded to 28 = 14 MFMA steps, rows = output pixels and the whole filter sits in registers.
// (A matching filter-gradient kernel -- rows = the 27 filter rows, K' = pixels -- was tried and dropped: with a
// 27 x cout output every workgroup ends in the same 1296 atomics, and ~75 ns per same-address atomic put it at
// 70-90 us against the generic kernel's 65.)
template <int S, int NB, typename TO = float>   // TO: storage type of the output (fp32 / bf16)
__global__ __launch_bounds__(256) void c3_fwd_kernel(CnConvGeom g, const float* __restrict__ X, const float* __restrict__ W,
                                                     const float* __restrict__ bias, TO* __restrict__ Y, int act, float slope) {
    constexpr int TH = 8, TW = 32, PR = (TH - 1) * S + 3, PC = ((TW - 1) * S + 3) * 3, PCP = PC + 1;
    __shared__ float patch[PR * PCP];
    __shared__ __attribute__((aligned(16))) float stage[4][16 * NB * 32];      // per wave: 16 pixels x cout of the epilogue
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    const int tiles_w = (g.out_w + TW - 1) / TW, tiles_h = (g.out_h + TH - 1) / TH;
    int b = blockIdx.x;
    const int tj = b % tiles_w; b /= tiles_w;
    const int ti = b % tiles_h;
    const int n = b / tiles_h;
    const int oy0 = ti * TH, ox0 = tj * TW, iy0 = oy0 * S - g.p_h, ix0 = ox0 * S - g.p_w;
    // 16-byte stores need whole chunks per pixel and an aligned tensor (else: the element-wise epilogue)
    const bool staged = g.cout % (16 / (int)sizeof(TO)) == 0 && ((uintptr_t)Y & 15) == 0;
    float breg[14][NB];
    int aoff[14];
#pragma unroll
    for (int q = 0; q < 14; ++q) {
        const int k = 2 * q + half;
        const int kh = k / 9, kw = (k - kh * 9) / 3, ci = k - kh * 9 - kw * 3;
        aoff[q] = k < 27 ? kh * PCP + kw * 3 + ci : 0;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int col = nb * 32 + l31;
            breg[q][nb] = (k < 27 && col < g.cout) ? W[k * g.cout + col] : 0.f;
        }
    }
    {
        // patch rows: a thread owns one column of the patch (PC <= 256 / RPP floats per row) and every RPP-th row; ALL its loads
        // are issued before the first LDS store (a load -> wait -> store loop serialises PR x PC / 256 memory round trips per
        // workgroup: that loop, not the MFMA work or the stores, was what the kernel's time consisted of)
        constexpr int PCC = PC <= 128 ? 128 : 256, RPP = 256 / PCC, NR = (PR + RPP - 1) / RPP;
        static_assert(PC <= 256, "patch row wider than the workgroup");
        const int c = threadIdx.x % PCC, rs = threadIdx.x / PCC;
        const int ix = ix0 + c / 3;
        const bool cok = c < PC, xin = cok && ix >= 0 && ix < g.in_w;
        // (branch-free: an out-of-image element loads X[0] and is zeroed afterwards.  As `ok ? X[...] : 0` hipcc put each guarded
        // load in its own exec-mask region and, in the <2, 2, float> instance, waited for it there: seven round trips were left.)
        const long xb = ((long)n * g.in_h * g.in_w + ix0) * 3 + c;
        float pv[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = rs + RPP * i, iy = iy0 + r;
            const bool ok = xin && r < PR && iy >= 0 && iy < g.in_h;
            const float v = X[ok ? xb + (long)iy * g.in_w * 3 : 0L];
            pv[i] = ok ? v : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = rs + RPP * i;
            if (cok && r < PR) patch[r * PCP + c] = pv[i];
        }
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = wave * 2 + rr;
        const int base = r * S * PCP + l31 * S * 3;
        f32x16 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[nb][q] = 0.f;
#pragma unroll
        for (int q = 0; q < 14; ++q) {
            const float a = patch[base + aoff[q]];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, breg[q][nb], acc[nb], 0, 0, 0);
        }
        const int oy = oy0 + r;
        if (oy >= g.out_h) continue;
        image_row_epilogue<NB, TO>(acc, stage[wave], staged, Y + (((long)n * g.out_h + oy) * g.out_w + ox0) * g.cout, g.out_w - ox0, g.cout,
                                   bias, act, slope);
    }
}

// ---------------------------------------------------------------------------------------------
// ResNet-50 conv1 (real_encoder.py:13, keras ResNet50: ZeroPadding2D(3) + 7x7 stride-2 convolution of the 3-channel image): K = 147.
// The generic kernel gathered those with per-element integer division (190 us on 16 images at 256^2 against ~32 us of MFMA work);
// here, as in c3_fwd_kernel, a workgroup stages the 21 x 69-pixel patch of its 8 x 32 output tile once (every load in flight before
// the first LDS store), and walks the filter one kernel row (21 values = 11 MFMA steps, the last half-empty) at a time: the next
// row's filter slice is loaded while this row's MFMAs run, and both of a wave's output rows use it.
template <int NB, typename TO = float, int RPW = 2>      // RPW: output rows per wave (tile = 4 RPW rows x 32 pixels; 1: 96 -> 87 us at 1024 tiles, 39 -> 44 at 512)
__global__ __launch_bounds__(256) void c7s2_fwd_kernel(CnConvGeom g, const float* __restrict__ X, const float* __restrict__ W,
                                                       const float* __restrict__ bias, TO* __restrict__ Y, int act, float slope) {
    constexpr int S = 2, KS = 7, TH = 4 * RPW, TW = 32, PR = (TH - 1) * S + KS, PC = ((TW - 1) * S + KS) * 3, PCP = PC + 1;
    constexpr int KR = KS * 3, NQ = (KR + 1) / 2;
    static_assert(PC <= 256, "patch row wider than the workgroup");
    __shared__ float patch[PR * PCP];
    __shared__ __attribute__((aligned(16))) float stage[4][16 * NB * 32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    const int tiles_w = (g.out_w + TW - 1) / TW, tiles_h = (g.out_h + TH - 1) / TH;
    int b = blockIdx.x;
    const int tj = b % tiles_w; b /= tiles_w;
    const int ti = b % tiles_h;
    const int n = b / tiles_h;
    const int oy0 = ti * TH, ox0 = tj * TW, iy0 = oy0 * S - g.p_h, ix0 = ox0 * S - g.p_w;
    const bool staged = g.cout % (16 / (int)sizeof(TO)) == 0 && ((uintptr_t)Y & 15) == 0;
    float bq[2][NQ][NB];
    auto load_b = [&](int kh, float (*dst)[NB]) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int k = 2 * q + half;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int col = nb * 32 + l31;
                dst[q][nb] = (k < KR && col < g.cout) ? W[(kh * KR + k) * g.cout + col] : 0.f;
            }
        }
    };
    load_b(0, bq[0]);
    {
        const int c = threadIdx.x;
        const int ix = ix0 + c / 3;
        const bool cok = c < PC, xin = cok && ix >= 0 && ix < g.in_w;
        const long xb = ((long)n * g.in_h * g.in_w + ix0) * 3 + c;      // (branch-free loads: see c3_fwd_kernel)
        float pv[PR];
#pragma unroll
        for (int r = 0; r < PR; ++r) {
            const int iy = iy0 + r;
            const bool ok = xin && iy >= 0 && iy < g.in_h;
            const float v = X[ok ? xb + (long)iy * g.in_w * 3 : 0L];
            pv[r] = ok ? v : 0.f;
        }
#pragma unroll
        for (int r = 0; r < PR; ++r)
            if (cok) patch[r * PCP + c] = pv[r];
    }
    __syncthreads();
    f32x16 acc[RPW][NB];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[rr][nb][q] = 0.f;
#pragma unroll
    for (int kh = 0; kh < KS; ++kh) {
        if (kh + 1 < KS) load_b(kh + 1, bq[(kh + 1) & 1]);
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int base = ((wave * RPW + rr) * S + kh) * PCP + l31 * S * 3;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int k = 2 * q + half;
                const float a = patch[base + (k < KR ? k : 0)];           // (k = 21: its filter value is 0, the address stays inside the row)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[rr][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq[kh & 1][q][nb], acc[rr][nb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int oy = oy0 + wave * RPW + rr;
        if (oy >= g.out_h) continue;
        image_row_epilogue<NB, TO>(acc[rr], stage[wave], staged, Y + (((long)n * g.out_h + oy) * g.out_w + ox0) * g.cout, g.out_w - ox0, g.cout,
                                   bias, act, slope);
    }
}

// ---------------------------------------------------------------------------------------------
// map_final of the generator: x2 nearest upsample folded into a 4x4 convolution from C (= 32) channels to the 3-channel
// image, + bias + tanh.  Same transposition as s2_image_dgrad_kernel: every INPUT pixel owns
//     P[pixel][tap*3 + co] = sum_c x[pixel][c] * w[tap][c][co]                 (C x 48: two 32-column MFMA blocks)
// and an output pixel (y, x) adds the 16 entries P[((y+kh-p)>>1, (x+kw-p)>>1)][kh*4+kw] that land on it.  One workgroup:
// (TH+2) x (TW+2) input pixels -> P in LDS -> its 2TH x 2TW output pixels.  The VALU kernel it replaces spent 240 us
// on 16 images at 256^2 (3.2 GFLOP of lane-serial FMAs); here the contraction is 1 GFLOP of MFMA and the pass is
// bounded by reading the input once.
template <int NG>   // C = 8 * NG
__global__ __launch_bounds__(256) void up2k4_rgb_fwd_kernel(CnConvGeom g, const float* __restrict__ X, const float* __restrict__ W,
                                                            const float* __restrict__ bias, float* __restrict__ Y, int act,
                                                            float slope) {
    constexpr int TH = 8, TW = 16, RW = TW + 2, R = (TH + 2) * RW, MT = (R + 31) / 32, PS = 49, C = 8 * NG;
    __shared__ float P[MT * 32 * PS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
    const int tiles_w = (g.in_w + TW - 1) / TW, tiles_h = (g.in_h + TH - 1) / TH;
    int b = blockIdx.x;
    const int tj = b % tiles_w; b /= tiles_w;
    const int ti = b % tiles_h;
    const int n = b / tiles_h;
    const int i0 = ti * TH - 1, j0 = tj * TW - 1;          // first input row / column of the patch (may be -1)
    float breg[NG * 4][2];
#pragma unroll
    for (int q = 0; q < NG * 4; ++q) {
        const int k = 8 * (q >> 2) + 4 * half + (q & 3);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int col = nb * 32 + l31;                 // = tap * 3 + co
            breg[q][nb] = col < 48 ? W[((col / 3) * C + k) * 3 + col % 3] : 0.f;
        }
    }
    for (int mt = wave; mt < MT; mt += 4) {
        const int r = mt * 32 + l31;
        const int ri = r / RW, rj = r - ri * RW;
        const int ii = i0 + ri, jj = j0 + rj;
        const bool inb = r < R && ii >= 0 && ii < g.in_h && jj >= 0 && jj < g.in_w;
        const float* src = X + (((long)n * g.in_h + ii) * g.in_w + jj) * C + 4 * half;
        float4 a[NG];
#pragma unroll
        for (int jg = 0; jg < NG; ++jg)
            a[jg] = inb ? *reinterpret_cast<const float4*>(src + 8 * jg) : make_float4(0.f, 0.f, 0.f, 0.f);
        f32x16 acc[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[nb][q] = 0.f;
#pragma unroll
        for (int jg = 0; jg < NG; ++jg) {
            const float av[4] = {a[jg].x, a[jg].y, a[jg].z, a[jg].w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], breg[jg * 4 + e][nb], acc[nb], 0, 0, 0);
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int col = nb * 32 + l31;
            if (col < 48) {
#pragma unroll
                for (int q = 0; q < 16; ++q) P[(mt * 32 + 4 * half + (q & 3) + 8 * (q >> 2)) * PS + col] = acc[nb][q];
            }
        }
    }
    __syncthreads();
    const float b0 = bias ? bias[0] : 0.f, b1 = bias ? bias[1] : 0.f, b2 = bias ? bias[2] : 0.f;
#pragma unroll
    for (int q = 0; q < (2 * TH * 2 * TW) / 256; ++q) {
        const int px = threadIdx.x + 256 * q;
        const int ly = px / (2 * TW), lx = px - ly * (2 * TW);
        const int y = 2 * (i0 + 1) + ly, x = 2 * (j0 + 1) + lx;
        if (y >= g.out_h || x >= g.out_w) continue;
        float s0 = b0, s1 = b1, s2 = b2;
#pragma unroll
        for (int kh = 0; kh < 4; ++kh) {
            const int uy = y + kh - g.p_h;                 // row of the upsampled image
            if (uy < 0 || uy >= 2 * g.in_h) continue;
            const int ri = (uy >> 1) - i0;
#pragma unroll
            for (int kw = 0; kw < 4; ++kw) {
                const int ux = x + kw - g.p_w;
                if (ux < 0 || ux >= 2 * g.in_w) continue;
                const float* pr = &P[(ri * RW + (ux >> 1) - j0) * PS + (kh * 4 + kw) * 3];
                s0 += pr[0];
                s1 += pr[1];
                s2 += pr[2];
            }
        }
        float* dst = Y + (((long)n * g.out_h + y) * g.out_w + x) * 3;
        dst[0] = cn_apply_act(s0, act, slope);
        dst[1] = cn_apply_act(s1, act, slope);
        dst[2] = cn_apply_act(s2, act, slope);
    }
}

constexpr int g_force_kb16 = 1;      // igemm_fwd_kernel: 16-deep LDS stages (32-deep: measured, no net win)
static int g_tune_cfg = -1;          // cn_conv_tune (sweeps, tests): forced tile / split-K factor / filter-gradient workgroup target
static int g_tune_splits = 0;
static long g_tune_wg_blocks = 0;
constexpr int g_xcd = 1;             // XCD-aware workgroup order
static int g_fwd2_sel = -1;                                                    // cn_conv_loop_select override
constexpr int g_fwd2_min_nks = 1;    // the LDS-DMA loop (fwd2.hip) takes reductions of MORE K steps than this
constexpr int g_fwd2_min_c = 48;     // thinnest layer it takes

template <int WM, int WN, int TM, int TN>
int launch_fwd(const CnConvGeom& g, bool vec, int par, int splits, const float* x, const float* w, const float* bias,
               float* y, int act, float slope, hipStream_t s, int bt = 0, long part_stride = 0, const float* res = nullptr) {
    const long M = (long)g.n * g.out_d * g.out_h * g.out_w;
    dim3 grid(cn_cdiv(M, 32 * WM * TM), cn_cdiv(g.cout, 32 * WN * TN), splits);
    int xcd = g_xcd;
    const int ntm = (int)grid.x, ntn = (int)grid.y;
    constexpr int tap_minor_on = 1;
    // cout tiles of one M tile grouped per XCD (several cout tiles), taps inside channel chunks (wide layers on the big tiles:
    // that is where the tap re-reads miss L2; the offset table costs LDS the small tiles' occupancy cannot spare)
    const bool wide = TM * TN >= 2 && g.cin >= 128;
    if (g_xcd && !par && splits == 1 && ntm >= 64 && (ntn > 1 || wide)) {
        xcd = 2 | ((tap_minor_on && wide) ? 4 : 0);
        const int per_xcd = (ntm + 7) / 8;
        grid = dim3((unsigned)(per_xcd * ntn * 8), 1, 1);
    }
    // 32-deep LDS stages: twice the MFMA work per barrier / per global-load round trip, which is what the
    // smaller tiles need to cover the L2/HBM latency of the gathered operand
    const bool kb32 = vec && g.cin % 32 == 0 && !g_force_kb16;
    const int taps = g.k_d * g.k_h * g.k_w;
    if ((xcd & 4) && (taps < 2 || taps > 9 || !vec)) xcd &= ~4;         // (offset table: taps x AP x 1 KiB of LDS)
    const size_t dyn = (xcd & 4) ? sizeof(int) * taps * (32 * WM * TM / (256 / ((kb32 ? 32 : BK) / 4))) * 256 : 0;   // taps x AP x 256 threads
    if (kb32)
        hipLaunchKernelGGL((igemm_fwd_kernel<WM, WN, TM, TN, true, true, 32>), grid, dim3(256), dyn, s, g, x, w, bias, y, act, slope, par, xcd, ntm, ntn, bt, part_stride, res);
    else if (vec)
        hipLaunchKernelGGL((igemm_fwd_kernel<WM, WN, TM, TN, true>), grid, dim3(256), dyn, s, g, x, w, bias, y, act, slope, par, xcd, ntm, ntn, bt, part_stride, res);
    else
        hipLaunchKernelGGL((igemm_fwd_kernel<WM, WN, TM, TN, false>), grid, dim3(256), 0, s, g, x, w, bias, y, act, slope, par, xcd, ntm, ntn, 0, 0L, res);
    CN_LAUNCH_CHECK();
    return CN_OK;
}

template <int WM, int WN, int TM, int TN>
int launch_wgrad(const CnConvGeom& g, const float* x, const float* gy, float* gw, hipStream_t s) {
    constexpr int BMt = 32 * WM * TM, BNt = 32 * WN * TN;
    const long M = (long)g.n * g.out_d * g.out_h * g.out_w;
    const long Ktot = (long)g.k_d * g.k_h * g.k_w * g.cin;
    const long tiles = (long)cn_cdiv(Ktot, BMt) * cn_cdiv(g.cout, BNt);
    const long wg_blocks = g_tune_wg_blocks > 0 ? g_tune_wg_blocks : 2048;   // sweep 256..4096: flat from 1536 up
    long splits = (wg_blocks + tiles - 1) / tiles;
    float* parts = nullptr;
    if (cn_det()) {
        // deterministic mode: per-split partial filters in the stream's workspace, added in split order by a second launch
        // (every split of every tile writes its whole slab region: no clearing).  As many splits as the workspace holds.
        const long cap = (long)(CN_DET_WS_FLOATS / ((size_t)Ktot * g.cout));
        CN_CHECK_ARG(cap >= 1, "deterministic filter gradient: %ld x %d filter does not fit the workspace", Ktot, g.cout);
        if (splits > cap) splits = cap;
    }
    long rows = (M + splits - 1) / splits;
    if (rows < 256) rows = 256;
    rows = (rows + BK - 1) / BK * BK;
    splits = (M + rows - 1) / rows;
    if (cn_det()) {
        parts = cn_det_ws(s, (size_t)splits * Ktot * g.cout);
        if (!parts) return CN_EINVAL;
    }
    dim3 grid(cn_cdiv(Ktot, BMt), cn_cdiv(g.cout, BNt), (unsigned)splits);
    int tx = 0, ty = 0;
    if (grid.x * grid.y > 1 && splits >= 16) {
        tx = (int)grid.x; ty = (int)grid.y;
        grid = dim3((unsigned)(cn_cdiv(splits, 8) * 8 * tx * ty), 1, 1);
    }
    const bool avec = g.cin % 4 == 0, bvec = g.cout % 4 == 0;
#define WG(A, B) hipLaunchKernelGGL((igemm_wgrad_kernel<WM, WN, TM, TN, A, B>), grid, dim3(256), 0, s, g, x, gy, gw, (int)rows, parts, tx, ty, (int)splits)
    if (avec && bvec) WG(true, true);
    else if (avec) WG(true, false);
    else if (bvec) WG(false, true);
    else WG(false, false);
#undef WG
    CN_LAUNCH_CHECK();
    if (parts) return cn_sum_parts(parts, gw, (int)splits, Ktot * g.cout, 1, 1.f, s);      // gw was cleared (or holds the sum so far)
    return CN_OK;
}

}  // namespace

// bt = 1: w is the original filter of the convolution whose data gradient g describes (see igemm_fwd_kernel); only the
// vectorised implicit-GEMM path takes it -- every other path answers CN_EUNSUPPORTED without launching.
// stats (cn_conv_fwd_stats): the launch must be one that can carry the statistics in its epilogue -- the unsplit LDS-DMA loop with
// tiles inside one sample -- or NOTHING is launched and the answer is CN_EUNSUPPORTED.
static int conv_fwd_impl(const CnConvGeom* gp, const float* x, const float* w, const float* bias, float* y, int act,
                         float slope, void* stream, int bt, const float* res = nullptr, float* stats = nullptr, int stats_mode = 0,
                         float stats_slope = 0.f) {
    if (int e = check_geom(gp)) return e;
    CN_CHECK_ARG(x && w && y, "NULL tensor");
    const CnConvGeom g = *gp;
    hipStream_t s = (hipStream_t)stream;
    const long M = (long)g.n * g.out_d * g.out_h * g.out_w;
    // res: only the unsplit implicit-GEMM launches carry the residual add in their epilogue -- anything else answers
    // CN_EUNSUPPORTED before launching (the caller then adds it with a pass of its own)
    if ((res || stats) && (g.cout <= 4 || g.cin == 3 || g.cout % 4 != 0)) return CN_EUNSUPPORTED;
    if (stats && (bt || cn_det() || g.cin % BK != 0)) return CN_EUNSUPPORTED;
    if (bt && (g.cout <= 4 || g.cin % BK != 0 || g.cout % 4 != 0)) return CN_EUNSUPPORTED;
    if (g.cout <= 4) {
        const bool vec = g.cin % 4 == 0;
        const int par = parity_ordered(g);
        const size_t lds = sizeof(float) * 4 * (size_t)g.k_d * g.k_h * g.k_w * g.cin;
        CN_CHECK_ARG(lds <= 64 * 1024, "thin conv: filter of %zu bytes does not fit the LDS stage", lds);
        const int T = g.k_d * g.k_h * g.k_w, CL = g.cin / 4;
        const bool dl1 = g.dl_d * g.dl_h * g.dl_w == 1;
        if (g.nd == 2 && g.up == 1 && g.k_h == 4 && g.k_w == 4 && g.s_h == 1 && g.s_w == 1 && g.dl_h == 1 && g.dl_w == 1 &&
            g.cout == 3 && g.cin == 32 && g.p_h == 1 && g.p_w == 1 && g.out_h <= 2 * g.in_h && g.out_w <= 2 * g.in_w) {
            dim3 grid((unsigned)(g.n * cn_cdiv(g.in_h, 8) * cn_cdiv(g.in_w, 16)));
            cn_prof_begin(s, conv_flops(g), conv_bytes(g), CN_FAM_THIN);
            hipLaunchKernelGGL((up2k4_rgb_fwd_kernel<4>), grid, dim3(256), 0, s, g, x, w, bias, y, act, slope);
            cn_prof_end(s);
            CN_LAUNCH_CHECK();
            return CN_OK;
        }
        if (g.nd == 2 && g.k_h == 3 && g.k_w == 3 && g.dl_h == 2 && g.dl_w == 2 && g.s_h == 1 && g.s_w == 1 && !g.up &&
            g.cout == 3 && g.cin == 48 && !bias && act == CN_ACT_NONE && g.p_h >= 0 && g.p_h <= 2 && g.p_w >= 0 &&
            g.p_w <= 2 && g.out_h <= 2 * g.in_h && g.out_w <= 2 * g.in_w) {
            dim3 grid((unsigned)(g.n * cn_cdiv(g.in_h, 8) * cn_cdiv(g.in_w, 32)));
            cn_prof_begin(s, conv_flops(g), conv_bytes(g), CN_FAM_S2_IMAGE_DGRAD);
            hipLaunchKernelGGL((s2_image_dgrad_kernel<6>), grid, dim3(256), 0, s, g, x, w, y);
            cn_prof_end(s);
            CN_LAUNCH_CHECK();
            return CN_OK;
        }
        if (g.nd == 2 && g.k_h == 3 && g.k_w == 3 && g.dl_h == 1 && g.dl_w == 1 && g.s_h == 1 && g.s_w == 1 && !g.up &&
            g.cout == 3 && g.cin == 64 && !bias && act == CN_ACT_NONE && g.p_h >= 0 && g.p_h <= 2 && g.p_w >= 0 && g.p_w <= 2) {
            dim3 grid((unsigned)(g.n * cn_cdiv(g.out_h, 8) * cn_cdiv(g.out_w, 32)));
            cn_prof_begin(s, conv_flops(g), conv_bytes(g), CN_FAM_S2_IMAGE_DGRAD);
            hipLaunchKernelGGL((s1_image_dgrad_kernel<8>), grid, dim3(256), 0, s, g, x, w, y);
            cn_prof_end(s);
            CN_LAUNCH_CHECK();
            return CN_OK;
        }
        if (par && g.cin % BK == 0 && act == CN_ACT_NONE) {
            // zero-stuffed data-gradient into a thin image: per pixel only ~taps/4 * cin MACs, the per-pixel
            // bookkeeping of a VALU kernel dominates; the 128x32 MFMA tile with dead-tap skipping is faster
            dim3 grid(cn_cdiv(M, 128), 1, 1);
            cn_prof_begin(s, conv_flops(g), conv_bytes(g), CN_FAM_FWD_128x32);
            hipLaunchKernelGGL((igemm_fwd_kernel<4, 1, 1, 1, true, false>), grid, dim3(256), 0, s, g, x, w, bias, y, act, slope, 1, g_xcd, 0, 0);
            cn_prof_end(s);
            CN_LAUNCH_CHECK();
            return CN_OK;
        }
        if (vec && g.cout == 3 && CL >= 5 && CL <= 16 && T <= 32 && (par || dl1) && g.dl_d * g.dl_h * g.dl_w <= 8) {
            constexpr int PX = 4;
            if (CL <= 8) {
                dim3 grid(cn_cdiv(M, (256 / 8) * PX));
                hipLaunchKernelGGL((thin_conv_coop_kernel<3, 8, PX>), grid, dim3(256), lds, s, g, x, w, bias, y, act, slope, par);
            } else {
                dim3 grid(cn_cdiv(M, (256 / 16) * PX));
                hipLaunchKernelGGL((thin_conv_coop_kernel<3, 16, PX>), grid, dim3(256), lds, s, g, x, w, bias, y, act, slope, par);
            }
            CN_LAUNCH_CHECK();
            return CN_OK;
        }
        dim3 grid(cn_cdiv(M, 256));
#define THIN(CO)                                                                                                      \
    if (vec)                                                                                                          \
        hipLaunchKernelGGL((thin_conv_kernel<CO, true>), grid, dim3(256), lds, s, g, x, w, bias, y, act, slope, par); \
    else                                                                                                              \
        hipLaunchKernelGGL((thin_conv_kernel<CO, false>), grid, dim3(256), lds, s, g, x, w, bias, y, act, slope, par);
        switch (g.cout) {
            case 1: THIN(1); break;
            case 2: THIN(2); break;
            case 3: THIN(3); break;
            default: THIN(4); break;
        }
#undef THIN
        CN_LAUNCH_CHECK();
        return CN_OK;
    }
    if (!bt && g.nd == 2 && g.cin == 3 && g.k_h == 3 && g.k_w == 3 && g.s_h == g.s_w && (g.s_h == 1 || g.s_h == 2) &&
        g.dl_h == 1 && g.dl_w == 1 && !g.up && g.cout > 4 && g.cout <= 64) {
        dim3 grid((unsigned)(g.n * cn_cdiv(g.out_h, 8) * cn_cdiv(g.out_w, 32)));
        cn_prof_begin(s, conv_flops(g), conv_bytes(g), CN_FAM_C3_FWD);
#define C3F(S_, NB_) hipLaunchKernelGGL((c3_fwd_kernel<S_, NB_>), grid, dim3(256), 0, s, g, x, w, bias, y, act, slope)
        if (g.s_h == 1) { if (g.cout <= 32) C3F(1, 1); else C3F(1, 2); }
        else { if (g.cout <= 32) C3F(2, 1); else C3F(2, 2); }
#undef C3F
        cn_prof_end(s);
        CN_LAUNCH_CHECK();
        return CN_OK;
    }
    if (!bt && g.nd == 2 && g.cin == 3 && g.k_h == 7 && g.k_w == 7 && g.s_h == 2 && g.s_w == 2 && g.dl_h == 1 && g.dl_w == 1 &&
        !g.up && g.cout > 4 && g.cout <= 64 && !res) {
        dim3 grid((unsigned)(g.n * cn_cdiv(g.out_h, 8) * cn_cdiv(g.out_w, 32)));
        cn_prof_begin(s, conv_flops(g), conv_bytes(g), CN_FAM_C3_FWD);
        if (g.cout <= 32) hipLaunchKernelGGL((c7s2_fwd_kernel<1>), grid, dim3(256), 0, s, g, x, w, bias, y, act, slope);
        else hipLaunchKernelGGL((c7s2_fwd_kernel<2>), grid, dim3(256), 0, s, g, x, w, bias, y, act, slope);
        cn_prof_end(s);
        CN_LAUNCH_CHECK();
        return CN_OK;
    }
    CN_CHECK_ARG(g.cout % 4 == 0, "cout=%d: implicit-GEMM path needs cout %% 4 == 0", g.cout);
    const bool vec = g.cin % BK == 0;
    const int par = parity_ordered(g) && vec;
    // tile choice: the biggest tile that still gives >= 2 workgroups per CU (256 CUs)
    const long t128 = (long)cn_cdiv(M, 128) * cn_cdiv(g.cout, 128);
    const long t128x64 = (long)cn_cdiv(M, 128) * cn_cdiv(g.cout, 64);
    int cfg;
    long tiles;
    if (g.cout <= 32) { cfg = 3; tiles = (long)cn_cdiv(M, 128) * cn_cdiv(g.cout, 32); }
    else if (g.cout > 64 && t128 >= 512) { cfg = 0; tiles = t128; }
    else if (t128x64 >= 512) { cfg = 1; tiles = t128x64; }
    else { cfg = 2; tiles = (long)cn_cdiv(M, 64) * cn_cdiv(g.cout, 64); }
    // cout = 96 / 192 (discriminator blocks 1-2 and the data gradients of blocks 2-3): a 128 x 96 tile wastes nothing
    // where 128- or 64-wide tiles pad a quarter of their columns
    if (g.cout % 96 == 0 && g.cout % 128 != 0 &&
        (long)cn_cdiv(M, 128) * (g.cout / 96) >= (g.cout == 96 ? 256 : 384)) {
        cfg = 4;
        tiles = (long)cn_cdiv(M, 128) * (g.cout / 96);
    }
    // split-K for small outputs with a long reduction (ResNet stage 4/5, Conv3D at 4^3->8^3)
    int splits = 1;
    // (parity-ordered data gradients: tiles of the 4-tap class carry 4x the K of the 1-tap class, so more, smaller
    // K slices also even out the load -- conv_tune.py dgrad: 123 -> 95 us at M=16384 N=192, 124 -> 104 us at M=4096 N=384)
    const bool par_small = par && cfg == 2;          // 64 x 64 tiles of a parity-ordered data gradient
    const long nks_total = vec ? (long)g.k_d * g.k_h * g.k_w * (g.cin / BK) : 0;
    // a short reduction (<= 8 steps) is all prologue and epilogue: the narrower tile spreads the stores over twice the workgroups
    if (cfg == 0 && vec && !par && nks_total <= 8) { cfg = 1; tiles = t128x64; }
    // one workgroup per CU and a short reduction: the zero pass, the atomics and the separate bias / activation pass of a K
    // split cost more than the idle SIMD slots they would fill (conv_tune.py: M=8192 K=512 N=128 31 -> 25 us unsplit)
    const bool short_full = !par && tiles >= 256 && nks_total < 64;
    if (vec && tiles < (par_small ? 1024 : 512) && !short_full) {
        long nks = nks_total;
        if (par) nks /= (long)g.dl_d * g.dl_h * g.dl_w;
        long want = ((par_small ? 3072 : 1024) + tiles - 1) / tiles;      // aim at ~4 (12) workgroups per CU
        if (want > 16) want = 16;
        const long min_steps = par_small ? 8 : 16;   // average K steps per workgroup
        if (want > nks / min_steps) want = nks / min_steps;
        if (want > 1) splits = (int)want;
    }
    // The LDS-DMA main loop (fwd2.hip) keeps the matrix pipe fed from ONE workgroup per CU (its loads run NS steps ahead of the
    // MFMAs and none of its instructions sits outside an MFMA's shadow), so it does not need the 4 workgroups per CU the
    // register-staged loops are split for -- and every K split it avoids saves the zero pass, a tile of atomics per workgroup and
    // the separate bias / activation pass (13 us of a 60 us launch at M = 4096, K = 2304, N = 256; scripts/dev/fwd2_sweep.py).
    const int fwd2_on = g_fwd2_sel >= 0 ? g_fwd2_sel : 1;
    // (32 output channels: the 128 x 32 tile of the same loop, input channels from 32 up)
    const bool n32 = g.cout == 32 && g.cin >= 32;
    const bool fwd2_takes = fwd2_on && vec && nks_total > g_fwd2_min_nks && ((g.cin >= g_fwd2_min_c && g.cout >= g_fwd2_min_c) || n32) && g.dl_d <= 2 && g.dl_h <= 2 && g.dl_w <= 2 &&
                            (double)g.n * g.in_d * g.in_h * g.in_w * g.cin < 5.3e8 && (double)g.k_d * g.k_h * g.k_w * g.cin * g.cout < 5.3e8;
    if (fwd2_takes) {
        const int T = g.k_d * g.k_h * g.k_w;
        // a parity-ordered 1x1 data gradient (ResNet's strided projections) has ONE live class: only M / (dl_d dl_h dl_w) of
        // its rows do any work, the tiles of the other classes store zeros and leave
        const long Me = (par && T == 1) ? M / ((long)g.dl_d * g.dl_h * g.dl_w) : M;
        long nks = nks_total;
        if (par) nks /= (long)g.dl_d * g.dl_h * g.dl_w;
        splits = 1;
        // parity classes with the same number of live taps (k % dl == 0 on every axis: the upsample-folded layers' class filters)
        // are one balanced launch; the data gradients of the stride-2 3x3 layers mix classes of 1 / 2 / 2 / 4 taps
        const bool par_balanced = par && g.k_d % g.dl_d == 0 && g.k_h % g.dl_h == 0 && g.k_w % g.dl_w == 0;
        if (n32) {
            cfg = 3;
            tiles = cn_cdiv(M, 128);
        } else if (par && T > 1 && !par_balanced) {
            // classes of 1 / 2 / 2 / 4 live taps (a quarter of the rows each): the 64 x 64 tile (128 x 96 for cout = 96 once it fills
            // the chip twice), K slices only for the 64 x 64 tile, where they also even out the load between the classes
            cfg = 2;
            tiles = (long)cn_cdiv(M, 64) * cn_cdiv(g.cout, 64);
            if (g.cout % 96 == 0 && g.cout % 64 != 0 && (long)cn_cdiv(M, 128) * (g.cout / 96) >= 512) {
                cfg = 4;
                tiles = (long)cn_cdiv(M, 128) * (g.cout / 96);
            }
            if (cfg == 2 && tiles < 1024) {
                long want = (1536 + tiles / 2) / tiles;
                if (want > 16) want = 16;
                if (want > nks / 8) want = nks / 8;
                if (want > 1) splits = (int)want;
            }
        } else {
            // Tile and K split together from a cost model of the launch (microseconds): the workgroups of one CU share its matrix
            // pipes, so a launch of W workgroups takes ceil(W / 256) workgroup lifetimes of (K steps) x (MFMA time of a step +
            // what the tile leaves exposed: measured per tile, scripts/dev/fwd2_sweep.py), plus a fixed start / drain, plus --
            // with a K split -- the zero pass, the separate bias / activation pass and one tile of atomics per workgroup.  What
            // the thresholds of the register-staged loops could not see is the quantisation: 384 workgroups on 256 CUs take as
            // long as 512.
            // The 64 x 64 tile wins the tile sweep almost everywhere with this loop (16 accumulator registers and 32 KB of LDS: five
            // workgroups per CU, so their barriers and fills interleave, and 4x finer load balance than a 128 x 128 tile); the one
            // exception is cout = 96, where 64-wide tiles pad a quarter of their columns and the 128 x 96 tile pads nothing.
            struct Cand { int cfg, bm, bn; double step_us; };
            const Cand cands[2] = {{2, 64, 64, 0.265}, {4, 128, 96, 0.68}};
            double best = 0.0;
            bool have = false;
            for (const Cand& c : cands) {
                if (c.cfg == 4 && (g.cout % 96 != 0 || g.cout % 64 == 0 || (long)cn_cdiv(Me, 128) * (g.cout / 96) < 256)) continue;
                const long tl = (long)cn_cdiv(Me, c.bm) * cn_cdiv(g.cout, c.bn);
                const long smax = nks / 8 > 1 ? (nks / 8 > 16 ? 16 : nks / 8) : 1;
                for (long s_ = 1; s_ <= smax; ++s_) {
                    const double waves = (double)cn_cdiv(tl * s_, cn_cu_count());
                    double t = 10.0 + waves * (double)cn_cdiv(nks, s_) * c.step_us * (waves == 1.0 ? 1.06 : 1.0);
                    if (s_ > 1) t += 9.0 + (double)s_ * (double)M * g.cout * 4.0 / 6.0e6;
                    // a split launch cannot carry the residual add / the statistics in its epilogue: the caller then runs a pass of
                    // its own over y (read + write at ~3 TB/s, one more launch) -- priced here so that a fused request splits only
                    // where the split still wins with that pass added
                    if (s_ > 1 && (res || stats)) t += 5.0 + 2.0 * (double)M * g.cout * 4.0 / 3.0e6;
                    if (!have || t < best) { have = true; best = t; cfg = c.cfg; tiles = tl; splits = (int)s_; }
                }
            }
        }
    }
    if (g_tune_cfg >= 0) cfg = g_tune_cfg;                        // tuning overrides (cn_conv_tune; scripts/conv_sweep.py)
    if (g_tune_splits > 0) splits = g_tune_splits;
    if (res && splits > 1) return CN_EUNSUPPORTED;
    int srows = 1, sper = 1;
    if (stats) {
        // rows of one sample (inside one parity class for class-major rows); every tile must lie inside one sample
        const int qd = par ? g.out_d / g.dl_d : g.out_d, qh = par ? g.out_h / g.dl_h : g.out_h, qw = par ? g.out_w / g.dl_w : g.out_w;
        srows = qd * qh * qw;
        sper = par ? g.n * srows : (int)M;
        if (!fwd2_takes || splits > 1 || cfg == 3 || srows % (cfg == 2 ? 64 : 128) != 0) return CN_EUNSUPPORTED;
    }
    float* parts = nullptr;
    if (cn_det() && splits > 1) {
        // deterministic mode: the K splits write partial outputs into the stream's workspace (as many splits as it holds) and a
        // second launch adds them in split order -- no atomics
        const long cap = (long)(CN_DET_WS_FLOATS / ((size_t)M * g.cout));
        if (splits > cap) splits = (int)cap;
        if (splits > 1 && vec) {
            parts = cn_det_ws(s, (size_t)splits * M * g.cout);
            if (!parts) return CN_EINVAL;
        } else {
            splits = 1;
        }
    }
    const int kact = splits > 1 ? CN_ACT_NONE : act;
    if (splits > 1 && !parts) {
        if (int ez__ = cn_zero_async(y, sizeof(float) * M * g.cout, s)) return ez__;
    }
    float* const y_user = y;
    const long part_stride = parts ? (long)M * g.cout : 0;
    if (parts) y = parts;
    cn_prof_begin(s, conv_flops(g), conv_bytes(g), cfg == 0 ? CN_FAM_FWD_128x128 : cfg == 1 ? CN_FAM_FWD_128x64 : cfg == 3 ? CN_FAM_FWD_128x32 : cfg == 4 ? CN_FAM_FWD_128x96 : CN_FAM_FWD_64x64);
    int e = CN_EUNSUPPORTED;
    // the LDS-DMA main loop (fwd2.hip)
    if (fwd2_takes && (cfg != 3 || n32)) {
        const bool plain = !par && g.k_d * g.k_h * g.k_w == 1 && g.s_d == 1 && g.s_h == 1 && g.s_w == 1 && g.dl_d == 1 && g.dl_h == 1 &&
                           g.dl_w == 1 && !g.up && g.p_d == 0 && g.p_h == 0 && g.p_w == 0 && g.out_d == g.in_d && g.out_h == g.in_h &&
                           g.out_w == g.in_w;
        const double xe = (double)g.n * g.in_d * g.in_h * g.in_w * g.cin, we = (double)g.k_d * g.k_h * g.k_w * g.cin * g.cout;
        e = cn_fwd2(plain ? nullptr : &g, cfg, bt, x, w, bias, y, M, g.cout, g.cin, kact, slope, splits, part_stride, plain ? 0 : par, s, res, xe, we,
                    stats, stats_mode, stats_slope, srows, sper);
        if (stats && e != CN_OK) {                   // (cannot happen after the checks above; never fall through to a kernel without them)
            cn_prof_end(s);
            return e == CN_EUNSUPPORTED ? CN_EINVAL : e;
        }
    }
    // everything the LDS-DMA loop does not take (K or cout no multiple of 16 / 4, thin layers, > 2 GiB operands): igemm_fwd_kernel
    if (e == CN_EUNSUPPORTED)
    switch (cfg) {
        case 3: e = launch_fwd<4, 1, 1, 1>(g, vec, par, splits, x, w, bias, y, kact, slope, s, bt, part_stride, res); break;   // 128 x 32
        case 4: e = launch_fwd<4, 1, 1, 3>(g, vec, par, splits, x, w, bias, y, kact, slope, s, bt, part_stride, res); break;   // 128 x 96
        case 0: e = launch_fwd<2, 2, 2, 2>(g, vec, par, splits, x, w, bias, y, kact, slope, s, bt, part_stride, res); break;   // 128 x 128
        case 1: e = launch_fwd<2, 2, 2, 1>(g, vec, par, splits, x, w, bias, y, kact, slope, s, bt, part_stride, res); break;   // 128 x 64
        default: e = launch_fwd<2, 2, 1, 1>(g, vec, par, splits, x, w, bias, y, kact, slope, s, bt, part_stride, res); break;  // 64 x 64
    }
    cn_prof_end(s);
    if (e == CN_OK && parts) e = cn_sum_parts(parts, y_user, splits, (long)M * g.cout, 0, 1.f, s);
    y = y_user;
    if (e == CN_OK && splits > 1 && act != CN_ACT_NONE) e = cn_act_fwd(y, y, (size_t)M * g.cout, act, slope, CN_F32, stream);
    return e;
}

extern "C" int cn_conv_fwd(const CnConvGeom* gp, const float* x, const float* w, const float* bias, float* y, int act,
                           float slope, void* stream) {
    return conv_fwd_impl(gp, x, w, bias, y, act, slope, stream, 0);
}

// y = act(conv(x, w) + bias + res): the residual add of a ResNet block in the convolution's epilogue (real_encoder.py:13 --
// keras ResNet50's `Add` + `Activation("relu")` behind the block's last 1x1 convolution).  Only unsplit implicit-GEMM launches
// carry it; CN_EUNSUPPORTED (nothing launched) otherwise.
extern "C" int cn_conv_fwd_stats(const CnConvGeom* gp, const float* x, const float* w, const float* bias, float* y, int act,
                                 float slope, float* stats, int stats_mode, float stats_slope, void* stream) {
    CN_CHECK_ARG(stats && (stats_mode == 1 || stats_mode == 2), "cn_conv_fwd_stats: stats buffer and mode 1 / 2");
    CN_CHECK_ARG(stats_mode == 1 || act == CN_ACT_NONE, "cn_conv_fwd_stats: mode 2 takes the statistics of the pre-activation output");
    return conv_fwd_impl(gp, x, w, bias, y, act, slope, stream, 0, nullptr, stats, stats_mode, stats_slope);
}

extern "C" int cn_conv_fwd_res(const CnConvGeom* gp, const float* x, const float* w, const float* bias, const float* res, float* y,
                               int act, float slope, void* stream) {
    CN_CHECK_ARG(res, "cn_conv_fwd_res: res is NULL");
    return conv_fwd_impl(gp, x, w, bias, y, act, slope, stream, 0, res);
}

extern "C" int cn_conv_weight_tflip(const float* w, float* wt, int taps, int cin, int cout, void* stream) {
    CN_CHECK_ARG(w && wt && taps > 0 && cin > 0 && cout > 0, "bad tflip args");
    const long total = (long)taps * cin * cout;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(weight_tflip_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wt, taps, cin, cout);
    CN_LAUNCH_CHECK();
    return CN_OK;
}

// First / last layers with mixed storage types (the bf16 path keeps 3-channel images in fp32, everything wider in bf16):
//   * 3x3 convolution of a 3-channel fp32 image written in bf16 (c3_fwd_kernel), and
//   * the data gradient of the stride-2 one INTO the fp32 image from a bf16 output gradient (s2_image_dgrad_kernel, the
//     geometry cn_conv_dgrad_dt builds),
// without a conversion pass over the 48 / 64-channel tensor.  Everything else: CN_EUNSUPPORTED, nothing launched.
extern "C" int cn_conv_fwd_dt(const CnConvGeom* gp, const void* x, int x_dt, const float* w, const float* bias, void* y, int y_dt,
                              int act, float slope, void* stream) {
    if (int e = check_geom(gp)) return e;
    CN_CHECK_ARG(x && w && y, "NULL tensor");
    const CnConvGeom g = *gp;
    hipStream_t s = (hipStream_t)stream;
    if (x_dt == CN_F32 && y_dt == CN_BF16 && g.nd == 2 && g.cin == 3 && g.k_h == 3 && g.k_w == 3 && g.s_h == g.s_w &&
        (g.s_h == 1 || g.s_h == 2) && g.dl_h == 1 && g.dl_w == 1 && !g.up && g.cout > 4 && g.cout <= 64) {
        dim3 grid((unsigned)(g.n * cn_cdiv(g.out_h, 8) * cn_cdiv(g.out_w, 32)));
        cn_prof_begin(s, conv_flops(g), conv_bytes(g, 4.0, 2.0), CN_FAM_C3_FWD);
#define C3F(S_, NB_) hipLaunchKernelGGL((c3_fwd_kernel<S_, NB_, bf16_t>), grid, dim3(256), 0, s, g, (const float*)x, w, bias, (bf16_t*)y, act, slope)
        if (g.s_h == 1) { if (g.cout <= 32) C3F(1, 1); else C3F(1, 2); }
        else { if (g.cout <= 32) C3F(2, 1); else C3F(2, 2); }
#undef C3F
        cn_prof_end(s);
        CN_LAUNCH_CHECK();
        return CN_OK;
    }
    if (x_dt == CN_F32 && y_dt == CN_BF16 && g.nd == 2 && g.cin == 3 && g.k_h == 7 && g.k_w == 7 && g.s_h == 2 && g.s_w == 2 &&
        g.dl_h == 1 && g.dl_w == 1 && !g.up && g.cout > 4 && g.cout <= 64) {
        dim3 grid((unsigned)(g.n * cn_cdiv(g.out_h, 8) * cn_cdiv(g.out_w, 32)));
        cn_prof_begin(s, conv_flops(g), conv_bytes(g, 4.0, 2.0), CN_FAM_C3_FWD);
        if (g.cout <= 32) hipLaunchKernelGGL((c7s2_fwd_kernel<1, bf16_t>), grid, dim3(256), 0, s, g, (const float*)x, w, bias, (bf16_t*)y, act, slope);
        else hipLaunchKernelGGL((c7s2_fwd_kernel<2, bf16_t>), grid, dim3(256), 0, s, g, (const float*)x, w, bias, (bf16_t*)y, act, slope);
        cn_prof_end(s);
        CN_LAUNCH_CHECK();
        return CN_OK;
    }
    if (x_dt == CN_BF16 && y_dt == CN_F32 && g.nd == 2 && g.k_h == 3 && g.k_w == 3 && g.dl_h == 2 && g.dl_w == 2 && g.s_h == 1 &&
        g.s_w == 1 && !g.up && g.cout == 3 && g.cin == 48 && !bias && act == CN_ACT_NONE && g.p_h >= 0 && g.p_h <= 2 && g.p_w >= 0 &&
        g.p_w <= 2 && g.out_h <= 2 * g.in_h && g.out_w <= 2 * g.in_w) {
        dim3 grid((unsigned)(g.n * cn_cdiv(g.in_h, 8) * cn_cdiv(g.in_w, 32)));
        cn_prof_begin(s, conv_flops(g), conv_bytes(g, 2.0, 4.0), CN_FAM_S2_IMAGE_DGRAD);
        hipLaunchKernelGGL((s2_image_dgrad_kernel<6, bf16_t>), grid, dim3(256), 0, s, g, (const bf16_t*)x, w, (float*)y);
        cn_prof_end(s);
        CN_LAUNCH_CHECK();
        return CN_OK;
    }
    return CN_EUNSUPPORTED;
}

extern "C" int cn_conv_dgrad_dt(const CnConvGeom* gp, const void* gy, int gy_dt, const float* w_tflip, void* gu, int gu_dt,
                                void* stream) {
    if (int e = check_geom(gp)) return e;
    if (gp->dl_d != 1 || gp->dl_h != 1 || gp->dl_w != 1) return CN_EUNSUPPORTED;
    CnConvGeom d = *gp;
    d.in_d = gp->out_d; d.in_h = gp->out_h; d.in_w = gp->out_w; d.cin = gp->cout;
    d.out_d = gp->in_d << gp->up; d.out_h = gp->in_h << gp->up; d.out_w = gp->in_w << gp->up;
    if (gp->nd == 2) d.out_d = 1;
    d.cout = gp->cin;
    d.s_d = d.s_h = d.s_w = 1;
    d.dl_d = gp->s_d; d.dl_h = gp->s_h; d.dl_w = gp->s_w;
    d.p_d = gp->k_d - 1 - gp->p_d; d.p_h = gp->k_h - 1 - gp->p_h; d.p_w = gp->k_w - 1 - gp->p_w;
    d.up = 0;
    return cn_conv_fwd_dt(&d, gy, gy_dt, w_tflip, nullptr, gu, gu_dt, CN_ACT_NONE, 0.f, stream);
}

// Data gradient straight from the ORIGINAL filter w [t][cin][cout] (no cn_conv_weight_tflip copy): CN_EUNSUPPORTED (nothing
// launched) where the shape does not reach the vectorised implicit-GEMM kernel -- the caller then uses cn_conv_dgrad.
extern "C" int cn_conv_dgrad_w(const CnConvGeom* gp, const float* gy, const float* w, float* gu, void* stream) {
    if (int e = check_geom(gp)) return e;
    if (gp->dl_d != 1 || gp->dl_h != 1 || gp->dl_w != 1) return CN_EUNSUPPORTED;
    CnConvGeom d = *gp;
    d.in_d = gp->out_d; d.in_h = gp->out_h; d.in_w = gp->out_w; d.cin = gp->cout;
    d.out_d = gp->in_d << gp->up; d.out_h = gp->in_h << gp->up; d.out_w = gp->in_w << gp->up;
    if (gp->nd == 2) d.out_d = 1;
    d.cout = gp->cin;
    d.s_d = d.s_h = d.s_w = 1;
    d.dl_d = gp->s_d; d.dl_h = gp->s_h; d.dl_w = gp->s_w;
    d.p_d = gp->k_d - 1 - gp->p_d; d.p_h = gp->k_h - 1 - gp->p_h; d.p_w = gp->k_w - 1 - gp->p_w;
    d.up = 0;
    return conv_fwd_impl(&d, gy, w, nullptr, gu, CN_ACT_NONE, 0.f, stream, 1);
}

// gu = (data gradient of cn_conv_dgrad_w) + res, res shaped like gu: the SECOND contribution to the gradient of a tensor that
// feeds a convolution AND a skip connection (a ResNet bottleneck's input, real_encoder.py:13: the gradient of keras' `Add`),
// added in the data-gradient launch's epilogue instead of by a separate pass.  Stride-1 layers whose launch is an unsplit
// implicit-GEMM one; CN_EUNSUPPORTED (nothing launched) otherwise -- the caller then adds with a pass of its own.
extern "C" int cn_conv_dgrad_w_res(const CnConvGeom* gp, const float* gy, const float* w, const float* res, float* gu, void* stream) {
    if (int e = check_geom(gp)) return e;
    CN_CHECK_ARG(res, "cn_conv_dgrad_w_res: res is NULL");
    if (gp->dl_d != 1 || gp->dl_h != 1 || gp->dl_w != 1 || gp->up) return CN_EUNSUPPORTED;
    if (gp->s_d != 1 || gp->s_h != 1 || gp->s_w != 1) return CN_EUNSUPPORTED;      // (parity-ordered rows: not with a residual)
    CnConvGeom d = *gp;
    d.in_d = gp->out_d; d.in_h = gp->out_h; d.in_w = gp->out_w; d.cin = gp->cout;
    d.out_d = gp->in_d; d.out_h = gp->in_h; d.out_w = gp->in_w;
    if (gp->nd == 2) d.out_d = 1;
    d.cout = gp->cin;
    d.p_d = gp->k_d - 1 - gp->p_d; d.p_h = gp->k_h - 1 - gp->p_h; d.p_w = gp->k_w - 1 - gp->p_w;
    return conv_fwd_impl(&d, gy, w, nullptr, gu, CN_ACT_NONE, 0.f, stream, 1, res);
}

extern "C" int cn_conv_dgrad(const CnConvGeom* gp, const float* gy, const float* w_tflip, float* gu, void* stream) {
    if (int e = check_geom(gp)) return e;
    CN_CHECK_ARG(gp->dl_d == 1 && gp->dl_h == 1 && gp->dl_w == 1, "dgrad of a dilated-input geometry is not defined here");
    CnConvGeom d = *gp;
    d.in_d = gp->out_d; d.in_h = gp->out_h; d.in_w = gp->out_w; d.cin = gp->cout;
    d.out_d = gp->in_d << gp->up; d.out_h = gp->in_h << gp->up; d.out_w = gp->in_w << gp->up;
    if (gp->nd == 2) d.out_d = 1;
    d.cout = gp->cin;
    d.s_d = d.s_h = d.s_w = 1;
    d.dl_d = gp->s_d; d.dl_h = gp->s_h; d.dl_w = gp->s_w;
    d.p_d = gp->k_d - 1 - gp->p_d; d.p_h = gp->k_h - 1 - gp->p_h; d.p_w = gp->k_w - 1 - gp->p_w;
    d.up = 0;
    return cn_conv_fwd(&d, gy, w_tflip, nullptr, gu, CN_ACT_NONE, 0.f, stream);
}

extern "C" int cn_conv_wgrad(const CnConvGeom* gp, const float* x, const float* gy, float* gw, int accumulate, void* stream) {
    if (int e = check_geom(gp)) return e;
    CN_CHECK_ARG(x && gy && gw, "NULL tensor");
    const CnConvGeom g = *gp;
    hipStream_t s = (hipStream_t)stream;
    const long Ktot = (long)g.k_d * g.k_h * g.k_w * g.cin;
    if (!accumulate) {
        if (int ez__ = cn_zero_async(gw, sizeof(float) * Ktot * g.cout, s)) return ez__;
    }
    if (Ktot <= 4 && g.cout <= 4 && g.k_d * g.k_h * g.k_w == 1 && g.s_h == 1 && g.s_w == 1 && g.s_d == 1 && !g.up &&
        g.p_h == 0 && g.p_w == 0 && g.p_d == 0) {
        const long M = (long)g.n * g.out_d * g.out_h * g.out_w;
        const int blocks = (int)(cn_cdiv(M, 256) > 1024 ? 1024 : cn_cdiv(M, 256));
        float* parts = nullptr;
        if (cn_det()) {                  // deterministic mode: per-workgroup partials, added in workgroup order
            parts = cn_det_ws(s, (size_t)blocks * 16);
            if (!parts) return CN_EINVAL;
        }
        int nb = blocks;
        if (g.cin == 3 && g.cout == 3 && (((uintptr_t)x | (uintptr_t)gy) & 15) == 0) {
            nb = blocks > 512 ? 512 : blocks;
            hipLaunchKernelGGL(tiny_wgrad_3x3_kernel, dim3(nb), dim3(256), 0, s, x, gy, gw, M, parts);
        } else {
            hipLaunchKernelGGL(tiny_wgrad_1x1_kernel, dim3(blocks), dim3(256), 0, s, x, gy, gw, M, g.cin, g.cout, parts);
        }
        CN_LAUNCH_CHECK();
        if (parts) return cn_sum_parts(parts, gw, nb, (long)g.cin * g.cout, 1, 1.f, s);
        return CN_OK;
    }
    cn_prof_begin(s, conv_flops(g), conv_bytes(g), g.cout <= 32 ? CN_FAM_WGRAD_128x32 : (Ktot >= 128 && g.cout % 96 == 0 && g.cout % 128 != 0) ? CN_FAM_WGRAD_128x96 : (Ktot >= 128 && g.cout >= 128) ? CN_FAM_WGRAD_128x128 : CN_FAM_WGRAD_64x64);
    int e;
    if (g.cout <= 32)
        e = launch_wgrad<4, 1, 1, 1>(g, x, gy, gw, s);       // 128 (tap,ci) x 32 co
    else if (Ktot >= 128 && g.cout % 96 == 0 && g.cout % 128 != 0)
        e = launch_wgrad<4, 1, 1, 3>(g, x, gy, gw, s);       // 128 x 96: cout 96 / 192 without column padding
    else if (Ktot >= 128 && g.cout >= 128)
        e = launch_wgrad<2, 2, 2, 2>(g, x, gy, gw, s);       // 128 x 128
    else
        e = launch_wgrad<2, 2, 1, 1>(g, x, gy, gw, s);       // 64 x 64
    cn_prof_end(s);
    return e;
}

// wgrad2.hip
bool cn_wgrad2_ok(const CnConvGeom& g);
size_t cn_wgrad2_workspace_floats(const CnConvGeom& g);
int cn_wgrad2_family(const CnConvGeom& g);
void cn_wgrad2_tune(int cfg, long wg_target);
void cn_wgrad2_stages(int ns);
int cn_wgrad2(const CnConvGeom& g, const float* x, const float* gy, float* gw, int accumulate, float* ws, hipStream_t s, int* parts_out = nullptr);

static bool wgrad2_takes(const CnConvGeom& g) {
    // Every geometry the LDS-DMA kernel can take (round 6: with the slot layout and the XCD-aware slice plan it is at or ahead of
    // the round-3 kernel -- split over rows, fp32 atomics -- on every shape of the iteration, profiles/round6_wgrad_shapes.txt).
    // The round-3 kernel keeps the rest: channel counts that are no multiple of 4, K < 64, > 2 GiB operands.
    const long Ktot = (long)g.k_d * g.k_h * g.k_w * g.cin;
    return cn_wgrad2_ok(g) && Ktot >= 64;
}

// Workspace (bytes) that cn_conv_wgrad_ws needs for this geometry: room for the partial filters of its row splits; 0 = none.
extern "C" size_t cn_conv_wgrad_workspace_bytes(const CnConvGeom* gp) {
    if (!gp || check_geom(gp) != CN_OK || !wgrad2_takes(*gp)) return 0;
    return sizeof(float) * cn_wgrad2_workspace_floats(*gp);
}

// Filter gradient with a CALLER-OWNED workspace (the caller owns all device memory): LDS-DMA main loop, row splits
// through partial slabs in `workspace` + one ordered reduction -- no atomics on the tile, bit-reproducible (wgrad2.hip).
// Geometries the new kernel does not take (channel counts that are no multiple of 4, K < 64, > 2 GiB operands) go to
// cn_conv_wgrad and need no workspace.
static int wgrad_ws_impl(const CnConvGeom* gp, const float* x, const float* gy, float* gw, int accumulate, void* workspace,
                         size_t workspace_bytes, int* parts, void* stream) {
    if (int e = check_geom(gp)) return e;
    CN_CHECK_ARG(x && gy && gw, "NULL tensor");
    if (parts) *parts = 0;
    if (!wgrad2_takes(*gp)) return cn_conv_wgrad(gp, x, gy, gw, accumulate, stream);
    const size_t need = sizeof(float) * cn_wgrad2_workspace_floats(*gp);
    CN_CHECK_ARG(workspace_bytes >= need && (need == 0 || workspace), "cn_conv_wgrad_ws: workspace of %zu bytes, %zu needed", workspace_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    cn_prof_begin(s, conv_flops(*gp), conv_bytes(*gp), cn_wgrad2_family(*gp));
    const int e = cn_wgrad2(*gp, x, gy, gw, accumulate, (float*)workspace, s, parts);
    cn_prof_end(s);
    return e;
}

extern "C" int cn_conv_wgrad_ws(const CnConvGeom* gp, const float* x, const float* gy, float* gw, int accumulate, void* workspace,
                                size_t workspace_bytes, void* stream) {
    return wgrad_ws_impl(gp, x, gy, gw, accumulate, workspace, workspace_bytes, nullptr, stream);
}

// cn_conv_wgrad_ws that leaves the slabs to the caller (include/confignet_hip.h): *parts = 0 -> gw is complete
extern "C" int cn_conv_wgrad_ws_slabs(const CnConvGeom* gp, const float* x, const float* gy, float* gw, int accumulate, void* workspace,
                                      size_t workspace_bytes, int* parts, void* stream) {
    CN_CHECK_ARG(parts, "cn_conv_wgrad_ws_slabs: parts is NULL");
    return wgrad_ws_impl(gp, x, gy, gw, accumulate, workspace, workspace_bytes, parts, stream);
}

extern "C" int cn_sumpool2(const void* gu, void* gx, int nd, int n, int d, int h, int w, int c, int dt, void* stream) {
    CN_CHECK_ARG(gu && gx && (nd == 2 || nd == 3) && c % 4 == 0 && (dt == CN_F32 || dt == CN_BF16), "sumpool2: bad args (c must be a multiple of 4)");
    if (nd == 2) d = 1;
    const long total = (long)n * d * h * w * (c / 4);
    CN_DISPATCH_DT(dt, {
        if (nd == 3)
            hipLaunchKernelGGL((sumpool2_kernel<3, T>), dim3(cn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)gu, (T*)gx, n, d, h, w, c / 4);
        else
            hipLaunchKernelGGL((sumpool2_kernel<2, T>), dim3(cn_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const T*)gu, (T*)gx, n, d, h, w, c / 4);
    });
    CN_LAUNCH_CHECK();
    return CN_OK;
}

// Tuning hook (scripts/conv_sweep.py): force the tile configuration (0 = 128x128, 1 = 128x64, 2 = 64x64, 3 = 128x32, 4 = 128x96;
// -1 = heuristic), the split-K factor of cn_conv_fwd / cn_conv_dgrad (0 = heuristic) and the workgroup target of cn_conv_wgrad
// (0 = default).  Process-wide; not for production use.
extern "C" int cn_conv_loop_select(int loop, int kb, int ns, int np) {
    CN_CHECK_ARG(loop >= -1 && loop <= 1 && (kb == 0 || kb == 16 || kb == 32) && (ns == 0 || ns == 3 || ns == 4) && np >= -1 && np <= 2,
                 "cn_conv_loop_select: bad argument");
    g_fwd2_sel = loop;
    cn_fwd2_tune(kb, ns, np);
    cn_wgrad2_stages(ns);
    return CN_OK;
}

extern "C" int cn_conv_tune(int cfg, int splits, long wg_blocks) {
    g_tune_cfg = cfg;
    g_tune_splits = splits;
    g_tune_wg_blocks = wg_blocks;
    // the same hook steers cn_conv_wgrad_ws: tile 0 / 4 / 2 / 3 -> 128x128 / 128x96 / 64x64 / 128x32, wg_blocks = workgroup target
    cn_wgrad2_tune(cfg == 0 ? 0 : cfg == 4 ? 1 : cfg == 2 ? 2 : cfg == 3 ? 3 : cfg == 5 ? 4 : -1, wg_blocks);
    return CN_OK;
}
