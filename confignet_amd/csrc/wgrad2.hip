// wgrad2.hip -- filter gradient of the convolution family, round 4 (reference op: Conv2D/Conv3DBackpropFilter behind
// tape.gradient in confignet_first_stage.py:472-473,557).
//
//   GW[(t, ci), co] = sum_m X[src(m, t), ci] * GY[m, co]
//
// is a GEMM whose reduction index is the output position m, and BOTH operands are "k-major" in memory as they lie (a row of X
// / GY per m, channels contiguous).  So neither needs a register round trip: every K step's tiles go from HBM/L2 straight into
// LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KB per wave instruction, per-lane source offsets = the gather; an offset
// beyond the buffer descriptor's range reads as zero = padding taps, rows past the slice, columns past the filter), FOUR
// stages deep: the loads of step s+3 are issued while step s is multiplied, so one or two workgroups per CU already keep the
// matrix pipe fed (the round-1..3 kernel staged through registers one step ahead and needed ~8 workgroups per CU to hide it:
// 2048-workgroup launches, i.e. many row splits, each ending in a full tile of fp32 atomics: 3.5x the algorithmic bytes at the
// fabric, MFMA-busy 0.30).  The stages are separate __shared__ objects so that the compiler can tell a ds_read of stage s from
// the DMA destination of stage s+3 (winograd.hip: it drains every outstanding load before a read it cannot prove disjoint).
//
// Row splits write their partial filter to a slab of the caller's workspace (plain stores); one ordered reduction adds the slabs
// and folds the accumulate into the gradient-arena slot.  No atomics on the tile, bit-reproducible by construction: the
// deterministic mode and the default mode are the same code.  A launch with a single split stores (or adds) its tile directly.
#include "common.h"

#include "mma_tile.h"
#include "conv_geom.h"

namespace {

typedef __attribute__((address_space(3))) float lds_float;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    // s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt = imm[3:0] | imm[15:14] << 4, expcnt imm[6:4], lgkmcnt imm[11:8])
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

// WM x WN waves, each TM x TN MFMA tiles: workgroup tile BI = 32 WM TM rows of (tap, ci) by BN = 32 WN TN output channels; KB
// reduction rows per stage.  A lane owns ADJACENT columns of its wave's tiles (column of MFMA lane l, tile t: T * l + t), so the
// TM (TN) operands of one k row are one ds_read_b64 when T == 2.
template <int WM, int WN, int TM, int TN, int KB>
__global__ __launch_bounds__(256) void wgrad2_kernel(CnConvGeom g, const float* __restrict__ X, const float* __restrict__ GY,
                                                     float* __restrict__ out, long slab_stride, int rows_per_split, int tiles_x,
                                                     int tiles_y, int nsplits, int accumulate) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int BI = 32 * WM * TM, BN = 32 * WN * TN;
    constexpr int QA = KB * BI / 256, QB = KB * BN / 256;            // 1 KB wave instructions per stage
    static_assert(QA % 4 == 0, "A pieces divide over the 4 waves");
    constexpr int JA = QA / 4, JB = (QB + 3) / 4, LPW = JA + JB;      // loads per wave per step (dummy pieces keep it uniform)
    constexpr int SA = KB * BI, SB = JB * 4 * 256;                    // floats per stage
    constexpr int NS = 4;                                             // stages
    __shared__ __attribute__((aligned(16))) float SM[NS * (SA + SB)];  // stage s: A at s * SA, B at NS * SA + s * SB

    // XCD-aware 1-D order: workgroup id runs on XCD id % 8; every tile of ONE row slice goes to the same XCD, so the slice of X
    // and GY that all of them read is fetched into that XCD's L2 once.  Placement only affects speed.
    const int ntile = tiles_x * tiles_y, id = blockIdx.x;
    int bz, tt;
    if (nsplits >= 8) {
        const int grp = id / (8 * ntile), rr = id - grp * 8 * ntile;
        bz = grp * 8 + (rr & 7);
        if (bz >= nsplits) return;
        tt = rr >> 3;
    } else {                        // few slices: plain order (grid = nsplits * ntile)
        tt = id / nsplits;
        bz = id - tt * nsplits;
    }
    const int by = tt / tiles_x, bx = tt - by * tiles_x;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, half = lane >> 5, l31 = lane & 31;
    const int M = g.n * g.out_d * g.out_h * g.out_w;
    const int Ktot = g.k_d * g.k_h * g.k_w * g.cin;
    const int i0 = bx * BI, n0 = by * BN;
    const int mbeg = bz * rows_per_split;
    const int mend = min(M, mbeg + rows_per_split);
    const int nks = (mend - mbeg + KB - 1) / KB;

    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(X), 0, (int)((long)g.n * g.in_d * g.in_h * g.in_w * g.cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(GY), 0, (int)((long)M * g.cout * 4), 0x00020000);

    // A pieces of this lane: piece index (wave + 4 j) * 64 + lane -> row = idx / (BI / 4), float4 column idx % (BI / 4); the
    // column (hence tap and channel) is the same for every j, the rows are KB / JA apart
    constexpr int PA = BI / 4;
    const int a_pc = lane % PA;
    const int a_i = i0 + 4 * a_pc;
    const bool a_ok = a_i < Ktot;
    int a_kd, a_kh, a_kw;
    const int a_tap = a_ok ? a_i / g.cin : 0;
    const int a_ci = a_ok ? a_i - a_tap * g.cin : 0;
    tap_decode(g, a_tap, a_kd, a_kh, a_kw);
    // Gather addresses without divisions or branches in the loop: the lane's rows are kept as (n, od, oh, ow) and advanced by
    // the mixed-radix digits of KB with one conditional subtract per digit; a tap's source position is an unsigned range test
    // per axis (dl == 1 in a filter-gradient geometry; the folded x2 upsample is the shift).
    const int bw = a_kw - g.p_w, bh = a_kh - g.p_h, bd = a_kd - g.p_d;
    const unsigned ext_w = (unsigned)(g.in_w << g.up), ext_h = (unsigned)(g.in_h << g.up), ext_d = (unsigned)(g.in_d << g.up);
    int dg_w, dg_t, dg_h, dg_u, dg_d, dg_n;
    divmod_pos(KB, g.out_w, dg_t, dg_w);
    divmod_pos(dg_t, g.out_h, dg_u, dg_h);
    divmod_pos(dg_u, g.out_d, dg_n, dg_d);
    int p_m[JA], p_n[JA], p_d[JA], p_h[JA], p_w[JA];
#pragma unroll
    for (int j = 0; j < JA; ++j) {
        int m = mbeg + ((wave + 4 * j) * 64 + lane) / PA;
        p_m[j] = m;
        divmod_pos(m, g.out_w, m, p_w[j]);
        divmod_pos(m, g.out_h, m, p_h[j]);
        divmod_pos(m, g.out_d, p_n[j], p_d[j]);
    }
    // B pieces: row = idx / (BN / 4), float4 column idx % (BN / 4); pieces past the tile (QB not a multiple of 4) and columns
    // past the filter are dummies (offset out of range: they land as zeros in the stage's padding / unused columns)
    constexpr int PB = BN / 4;
    int b_row[JB], b_col[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const int idx = (wave + 4 * j) * 64 + lane;
        const int col = n0 + 4 * (idx % PB);
        b_row[j] = idx / PB;
        b_col[j] = (wave + 4 * j < QB && col < g.cout) ? col : -1;
    }

    // one stage's loads for K step ks, piece by piece (always issued -- steps past the slice read nothing but keep the vmcnt
    // bookkeeping uniform).  Piece p < JA: the lane's p-th A row; p >= JA: its (p - JA)-th B piece.
    auto issue_piece = [&](int ks, int p) {
        const int st = ks & (NS - 1);
        if (p < JA) {
            const int j = p;
            float* as = SM + st * SA;
            const int vw = p_w[j] * g.s_w + bw, vh = p_h[j] * g.s_h + bh, vd = p_d[j] * g.s_d + bd;
            const bool ok = a_ok & (p_m[j] < mend) & ((unsigned)vw < ext_w) & ((unsigned)vh < ext_h) & ((unsigned)vd < ext_d);
            const int off = (((p_n[j] * g.in_d + (vd >> g.up)) * g.in_h + (vh >> g.up)) * g.in_w + (vw >> g.up)) * g.cin + a_ci;
            const unsigned vo = ok ? (unsigned)off * 4u : 0x80000000u;
            p_m[j] += KB;
            int c;
            p_w[j] += dg_w; c = p_w[j] >= g.out_w; p_w[j] -= c ? g.out_w : 0;
            p_h[j] += dg_h + c; c = p_h[j] >= g.out_h; p_h[j] -= c ? g.out_h : 0;
            p_d[j] += dg_d + c; c = p_d[j] >= g.out_d; p_d[j] -= c ? g.out_d : 0;
            p_n[j] += dg_n + c;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_float*)(as + (wave + 4 * j) * 256), 16, vo, 0, 0, 0);
        } else {
            const int j = p - JA;
            float* bs = SM + NS * SA + st * SB;
            const int m = mbeg + ks * KB + b_row[j];
            // rows past the slice meet zero rows of A (their values only have to be readable: clamp); dummy pieces read as zero
            const unsigned vo = b_col[j] >= 0 ? (unsigned)(min(m, M - 1) * g.cout + b_col[j]) * 4u : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(yres, (lds_float*)(bs + (wave + 4 * j) * 256), 16, vo, 0, 0, 0);
        }
    };
    auto issue = [&](int ks) {
#pragma unroll
        for (int p = 0; p < LPW; ++p) issue_piece(ks, p);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // The operand reads are inline asm: the compiler cannot tell a ds_read of stage s from the LDS-DMA destination of stage s+3
    // and would drain every outstanding load (s_waitcnt vmcnt(0)) before each of them; likewise __syncthreads() -- a release of
    // LDS -- waits for every pending LDS-DMA, so the loop uses the bare s_barrier.  Ordering is done by hand: s_waitcnt vmcnt(2
    // steps) + s_barrier before a stage is read, lgkmcnt before an operand set is used.
    const int ac = wm * 32 * TM + (TM == 2 ? 2 * l31 : l31), bc = wn * 32 * TN + (TN == 2 ? 2 * l31 : l31);
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_float*)SM;
    const unsigned a_lane = lds0 + 4u * (unsigned)(half * BI + ac), b_lane = lds0 + 4u * (unsigned)(NS * SA + half * BN + bc);
    typedef float v2f __attribute__((ext_vector_type(2)));
    constexpr int RD = (TM == 2 ? 1 : TM) + (TN == 2 ? 1 : TN);      // DS instructions per operand set
    static_assert(KB / 2 >= LPW, "one load piece behind each MFMA group");
    auto compute = [&](int st, int ks_next) {
        float a[2][TM], b[2][TN];
        const unsigned ab = a_lane + 4u * (unsigned)(st * SA), bb = b_lane + 4u * (unsigned)(st * SB);
        auto fetch = [&](int kk, int set) {
            const unsigned ap = ab + 4u * (unsigned)(kk * BI), bp = bb + 4u * (unsigned)(kk * BN);
            if (TM == 2) {
                v2f v;
                asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(ap));
                a[set][0] = v.x; a[set][TM - 1] = v.y;
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("ds_read_b32 %0, %1" : "=v"(a[set][i]) : "v"(ap + 128u * i));
            }
            if (TN == 2) {
                v2f v;
                asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(bp));
                b[set][0] = v.x; b[set][TN - 1] = v.y;
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("ds_read_b32 %0, %1" : "=v"(b[set][j]) : "v"(bp + 128u * j));
            }
        };
        // wait until at most RD DS instructions (the next operand set's) are outstanding.  The set's registers are plain INPUTS of the
        // wait and a scheduling barrier follows it, so that the MFMAs that use them stay behind it.  NOT "+v" (read-write) operands:
        // a tied operand that the register allocator does not coalesce becomes a v_mov from the ds_read's destination IN FRONT of
        // the asm, i.e. before the wait -- the copy reads the register before the LDS data has landed.  That was the round-4 form of
        // this loop; right as long as LDS answered within the ~100 cycles between the read and the copy, wrong under LDS contention
        // (another kernel's workgroups on the CU): tile-shaped errors of a few per cent, NaN when the stale register held one.
        // scripts/isa_lds_hazard.py checks the compiled loops for exactly this (tests/test_abi_cpu.py runs it).
        auto ready = [&](int set, bool more) {
            if (TM == 1 && TN == 1) {
                if (more) asm volatile("s_waitcnt lgkmcnt(%2)" : : "v"(a[set][0]), "v"(b[set][0]), "n"(RD));
                else asm volatile("s_waitcnt lgkmcnt(0)" : : "v"(a[set][0]), "v"(b[set][0]));
            } else if (TM == 2 && TN == 2) {
                if (more) asm volatile("s_waitcnt lgkmcnt(%4)" : : "v"(a[set][0]), "v"(a[set][TM - 1]), "v"(b[set][0]), "v"(b[set][TN - 1]), "n"(RD));
                else asm volatile("s_waitcnt lgkmcnt(0)" : : "v"(a[set][0]), "v"(a[set][TM - 1]), "v"(b[set][0]), "v"(b[set][TN - 1]));
            } else {      // TM == 1, TN == 3
                if (more) asm volatile("s_waitcnt lgkmcnt(%4)" : : "v"(a[set][0]), "v"(b[set][0]), "v"(b[set][1 % TN]), "v"(b[set][2 % TN]), "n"(RD));
                else asm volatile("s_waitcnt lgkmcnt(0)" : : "v"(a[set][0]), "v"(b[set][0]), "v"(b[set][1 % TN]), "v"(b[set][2 % TN]));
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        fetch(0, 0);
#pragma unroll
        for (int kk = 0; kk < KB; kk += 2) {
            const int cur = (kk >> 1) & 1;
            if (kk + 2 < KB) fetch(kk + 2, cur ^ 1);
            ready(cur, kk + 2 < KB);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
            // the refill of the stage that step ks-1 used (K step ks+3), one piece in the shadow of each MFMA group: its address
            // arithmetic and the DMA issue run while the matrix pipe works off the group
            if (kk / 2 < LPW) {
                __builtin_amdgcn_sched_barrier(0);
                issue_piece(ks_next, kk / 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // step ks: wait for its loads (issued three steps ago: two younger steps may stay in flight), barrier (everybody's pieces
    // have landed, everybody is done with step ks-1), refill the stage step ks-1 used with step ks+3, multiply
    issue(0);
    issue(1);
    issue(2);
    for (int ks = 0; ks < nks; ++ks) {
        wait_vmcnt<2 * LPW>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        compute(ks & (NS - 1), ks + 3);
        __builtin_amdgcn_sched_barrier(0);
    }
    wait_vmcnt<0>();            // (the dummy tail loads target LDS: they must not outlive the workgroup's allocation)

    // epilogue: MFMA lane l31 / tile j holds output channel n0 + wn*32*TN + (TN == 2 ? 2 l31 + j : 32 j + l31); accumulator r of
    // tile i holds MFMA row rho = 4 half + (r & 3) + 8 (r >> 2), i.e. filter row i0 + wm*32*TM + (TM == 2 ? 2 rho + i : 32 i + rho)
    float* dst = out + (slab_stride ? (long)bz * slab_stride : 0);
    const bool direct_add = !slab_stride && accumulate;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rho = 4 * half + (r & 3) + 8 * (r >> 2);
            const int row = i0 + wm * 32 * TM + (TM == 2 ? 2 * rho + i : 32 * i + rho);
            if (row >= Ktot) continue;
            float* drow = dst + (long)row * g.cout;
            if (TN == 2 && !direct_add) {
                const int col = n0 + wn * 64 + 2 * l31;
                if (col + 1 < g.cout && (g.cout & 1) == 0) {
                    *reinterpret_cast<float2*>(drow + col) = make_float2(acc[i][0][r], acc[i][TN - 1][r]);
                    continue;
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * 32 * TN + (TN == 2 ? 2 * l31 + j : 32 * j + l31);
                if (col >= g.cout) continue;
                if (direct_add) unsafeAtomicAdd(drow + col, acc[i][j][r]);     // (two streams may add into one arena slot)
                else drow[col] = acc[i][j][r];
            }
        }
}

struct Wg2Plan {
    int cfg;          // 0: 128x128, 1: 128x96, 2: 64x64, 3: 128x32
    int bi, bn, kb;
    long tiles_x, tiles_y, splits, rows;
};

long g_wg2_target = 0;     // cn_conv_tune(wg_blocks): workgroup target of the split (0 = heuristic)
int g_wg2_cfg = -1;

Wg2Plan wg2_plan(const CnConvGeom& g) {
    const long M = (long)g.n * g.out_d * g.out_h * g.out_w;
    const long Ktot = (long)g.k_d * g.k_h * g.k_w * g.cin;
    Wg2Plan p;
    int cfg;
    if (g.cout <= 32) cfg = 3;
    else if (Ktot >= 128 && g.cout % 96 == 0 && g.cout % 128 != 0) cfg = 1;
    else if (Ktot >= 128 && g.cout >= 128) cfg = 0;
    else cfg = 2;
    // few rows, many filter elements: the small tile gives enough workgroups without (or with fewer) row splits
    if (cfg == 0 && cn_cdiv(Ktot, 128) * cn_cdiv(g.cout, 128) * cn_cdiv(M, 512) < 128) cfg = 2;
    if (g_wg2_cfg >= 0) cfg = g_wg2_cfg;
    p.cfg = cfg;
    p.bi = cfg == 2 ? 64 : 128;
    p.bn = cfg == 0 ? 128 : cfg == 1 ? 96 : cfg == 2 ? 64 : 32;
    p.kb = cfg == 2 ? 32 : 16;
    p.tiles_x = cn_cdiv(Ktot, p.bi);
    p.tiles_y = cn_cdiv(g.cout, p.bn);
    const long tiles = p.tiles_x * p.tiles_y;
    const long max_splits = cn_cdiv(M, 4 * p.kb);                     // a workgroup is at least 4 K steps long
    long splits;
    if (g_wg2_target > 0) {
        splits = tiles >= g_wg2_target ? 1 : (g_wg2_target + tiles / 2) / tiles;
    } else {
        // cost of `s` row splits in MFMA cycles of one CU: workgroups run two to a CU sharing its matrix pipes, so the launch takes
        // ceil(tiles * s / 256) workgroup lifetimes (MFMA time of the slice + a fixed start / drain) -- plus, for s > 1, the slabs:
        // s partial filters written and read back by the reduction (~3 TB/s through L2 / Infinity Cache) and its launch
        const double mfma_per_row = 32.0 * (p.bi / 32) * (p.bn / 32) / 4.0;       // cycles per reduction row per wave
        const double count = (double)Ktot * g.cout;
        double best = 0.0;
        splits = 1;
        for (long s_ = 1; s_ <= max_splits && s_ <= 256; ++s_) {
            const double rows = (double)cn_cdiv(cn_cdiv(M, s_), p.kb) * p.kb;
            const double life = rows * mfma_per_row + 6000.0;
            double cost = (double)cn_cdiv(tiles * s_, 256) * life;
            if (s_ > 1) cost += 2.0 * s_ * count * 4.0 / 3.0e12 * 2.4e9 / 1.0 + 8000.0;
            if (s_ == 1 || cost < best) { best = cost; splits = s_; }
        }
    }
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    long rows = (M + splits - 1) / splits;
    rows = (rows + p.kb - 1) / p.kb * p.kb;
    p.rows = rows;
    p.splits = (M + rows - 1) / rows;
    return p;
}

}  // namespace

bool cn_wgrad2_ok(const CnConvGeom& g) {
    const double xb = (double)g.n * g.in_d * g.in_h * g.in_w * g.cin * 4.0;
    const double yb = (double)g.n * g.out_d * g.out_h * g.out_w * g.cout * 4.0;
    return g.cin % 4 == 0 && g.cout % 4 == 0 && g.cout > 4 && xb < 2147483647.0 && yb < 2147483647.0 &&
           g.dl_d == 1 && g.dl_h == 1 && g.dl_w == 1;
}

size_t cn_wgrad2_workspace_floats(const CnConvGeom& g) {
    if (!cn_wgrad2_ok(g)) return 0;
    const Wg2Plan p = wg2_plan(g);
    const long Ktot = (long)g.k_d * g.k_h * g.k_w * g.cin;
    return p.splits > 1 ? (size_t)p.splits * Ktot * g.cout : 0;
}

void cn_wgrad2_tune(int cfg, long wg_target) {
    g_wg2_cfg = cfg;
    g_wg2_target = wg_target;
}

int cn_wgrad2_family(const CnConvGeom& g) {
    const Wg2Plan p = wg2_plan(g);
    return p.cfg == 0 ? CN_FAM_WGRAD_128x128 : p.cfg == 1 ? CN_FAM_WGRAD_128x96 : p.cfg == 2 ? CN_FAM_WGRAD_64x64 : CN_FAM_WGRAD_128x32;
}

// gw (+)= filter gradient.  ws: at least cn_wgrad2_workspace_floats(g) floats (may be NULL when that is 0).
int cn_wgrad2(const CnConvGeom& g, const float* x, const float* gy, float* gw, int accumulate, float* ws, hipStream_t s) {
    const Wg2Plan p = wg2_plan(g);
    const long Ktot = (long)g.k_d * g.k_h * g.k_w * g.cin;
    const long count = Ktot * g.cout;
    CN_CHECK_ARG(p.splits == 1 || ws, "filter gradient: %ld row splits need a workspace of %ld floats", p.splits, p.splits * count);
    const long ntile = p.tiles_x * p.tiles_y;
    dim3 grid((unsigned)((p.splits >= 8 ? cn_cdiv(p.splits, 8) * 8 : p.splits) * ntile));
    float* out = p.splits > 1 ? ws : gw;
    const long stride = p.splits > 1 ? count : 0;
#define WG2(WM, WN, TM, TN, KB_) hipLaunchKernelGGL((wgrad2_kernel<WM, WN, TM, TN, KB_>), grid, dim3(256), 0, s, g, x, gy, out, stride, \
                                                     (int)p.rows, (int)p.tiles_x, (int)p.tiles_y, (int)p.splits, accumulate)
    switch (p.cfg) {
        case 0: WG2(2, 2, 2, 2, 16); break;
        case 1: WG2(4, 1, 1, 3, 16); break;
        case 2: WG2(2, 2, 1, 1, 32); break;
        default: WG2(4, 1, 1, 1, 16); break;
    }
#undef WG2
    CN_LAUNCH_CHECK();
    if (p.splits > 1) return cn_sum_parts(ws, gw, (int)p.splits, count, accumulate, 1.f, s);
    return CN_OK;
}
