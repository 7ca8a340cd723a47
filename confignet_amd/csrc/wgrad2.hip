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
#include "slots.h"

namespace {

typedef __attribute__((address_space(3))) float lds_float;
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int w2_task_slot(int q, int nslot, int ntask) { return q * nslot / ntask; }
// LDS-DMA tasks (A piece j: task 4 j + 3; B piece j: task 4 JA + 2 j + 1) that sit in a slot below `limit`
constexpr int w2_dma_before(int ja, int jb, int nslot, int ntask, int limit) {
    int n = 0;
    for (int j = 0; j < ja; ++j) n += w2_task_slot(4 * j + 3, nslot, ntask) < limit ? 1 : 0;
    for (int j = 0; j < jb; ++j) n += w2_task_slot(4 * ja + 2 * j + 1, nslot, ntask) < limit ? 1 : 0;
    return n;
}

// WM x WN waves, each TM x TN MFMA tiles: workgroup tile BI = 32 WM TM rows of (tap, ci) by BN = 32 WN TN output channels; KB
// reduction rows per stage, NS stages.  A lane owns ADJACENT columns of its wave's tiles when T == 2 (column of MFMA lane l, tile
// t: 2 l + t), so the two operands of one k row are one ds_read_b64; otherwise tile t is 32 columns further (ds_read_b32 each).
//
// Round 6: the step is laid out as MFMA slots (slots.h; the layout that took fwd2's main family from 0.49 to 0.61 MFMA-busy).
// The round-4 form of this loop put a whole piece's address arithmetic (~35 VALU with integer multiplies) and its LDS-DMA issue
// (which alone holds the wave's issue ~60 cycles) behind ONE MFMA and started every step with an exposed wait + barrier + LDS
// round trip.  Now, per step of G groups (a group = the MFMAs of GK k pairs, at least 3):
//     groups 0 .. G-2 : MFMA | <= 2 operand reads of the NEXT group (behind the first half of the group's MFMAs) | one task
//     wait: own pieces of step t+1 landed | s_barrier (everybody's landed, everybody done reading stage t)
//     group G-1       : MFMA | operand reads of group 0 of step t+1
// where the tasks are the refill of the stage that step t-1 used (= step t + NS - 1) cut into pieces of <= ~16 VALU: per A piece
// "source offset", "advance w / h", "advance d / n", "issue"; per B piece "offset", "issue" -- spread evenly over the slots in
// front of the barrier by a constant expression.  The gather coordinates are kept in the (strided, tap-shifted) source frame and
// advanced by mixed-radix digits with compare / select only; the three products of the offset are 24-bit multiplies (full rate;
// cn_wgrad2_ok bounds the operands).
template <int WM, int WN, int TM, int TN, int KB, int NS>
__device__ __forceinline__ void wgrad2_body(const CnConvGeom& g, const float* __restrict__ X, const float* __restrict__ GY,
                                            float* __restrict__ out, long slab_stride, int rows_per_split, int tiles_x, int tiles_y,
                                            int nsplits, int accumulate) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int BI = 32 * WM * TM, BN = 32 * WN * TN;
    constexpr int QA = KB * BI / 256, QB = KB * BN / 256;            // 1 KB wave instructions per stage
    static_assert(QA % 4 == 0, "A pieces divide over the 4 waves");
    constexpr int JA = QA / 4, JB = (QB + 3) / 4, LPW = JA + JB;      // loads per wave per step (dummy pieces keep it uniform)
    constexpr int SA = KB * BI, SB = JB * 4 * 256;                    // floats per stage
    constexpr int NMT = TM * TN;                                      // MFMAs per k pair
    constexpr int GK = NMT >= 3 ? 1 : 4 / NMT;                        // k pairs per group
    constexpr int NM = GK * NMT, G = KB / (2 * GK);                   // MFMAs per group, groups per stage
    constexpr int RDA = TM == 2 ? 1 : TM, RDB = TN == 2 ? 1 : TN, RD1 = RDA + RDB, RD = GK * RD1;     // DS instructions per k pair / group
    constexpr int NSL = NM > 1 ? NM / 2 : 1;                          // the reads go behind the first NSL MFMAs of a group
    constexpr int AHEAD = 2, NSET = 4;                                // operands are read AHEAD groups before their MFMAs, NSET register sets
    constexpr int NTASK = 4 * JA + 2 * JB, NSLOT = G * NM;            // refill tasks / MFMA slots of a step
    static_assert(G >= 4 && G % NSET == 0, "operand sets rotate per group and return to set 0 at the step boundary");
    static_assert((NS - 1) * LPW < 64, "vmcnt range");
    // task q sits in slot q * NSLOT / NTASK (even spread, order kept).  The barrier of a step stands in front of group G - AHEAD:
    // LDS-DMA tasks in later slots are issued after the step's wait for "step t+1 has landed" and do not count in it.
    constexpr int VMW = (NS - 3) * LPW + w2_dma_before(JA, JB, NSLOT, NTASK, (G - AHEAD) * NM);     // vmcnt that proves step t+1's pieces
    static_assert(NS >= 3, "stages");
    __shared__ __attribute__((aligned(1024))) float SM[NS * (SA + SB)];  // stage s: A at s * SA, B at NS * SA + s * SB

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

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // (an SGPR: LDS-DMA destinations stay scalar)
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
    constexpr unsigned OOB = 0x80000000u;            // past every descriptor's range: the piece lands as zeros

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
    // A row is kept as its source coordinates v = o * stride + tap - pad per axis (the folded x2 upsample is a shift at use) plus
    // the element offset of its sample; a step advances the row by KB, i.e. by the mixed-radix digits of KB, one conditional
    // subtract per digit.  In bounds <=> an unsigned range test per axis (dl == 1 in a filter-gradient geometry).
    const int up = g.up;
    const int bw = a_kw - g.p_w, bh = a_kh - g.p_h, bd = a_kd - g.p_d;
    const unsigned ext_w = (unsigned)(g.in_w << up), ext_h = (unsigned)(g.in_h << up), ext_d = (unsigned)(g.in_d << up);
    const int Sw = g.cin, Sh = g.in_w * g.cin, Sd = g.in_h * Sh, Sn = g.in_d * Sd;
    int dg_w, dg_t, dg_h, dg_u, dg_d, dg_n;
    divmod_pos(KB, g.out_w, dg_t, dg_w);
    divmod_pos(dg_t, g.out_h, dg_u, dg_h);
    divmod_pos(dg_u, g.out_d, dg_n, dg_d);
    const int st_w = dg_w * g.s_w, st_h = dg_h * g.s_h, st_d = dg_d * g.s_d, st_n = dg_n * Sn;        // a step's advance per axis
    const int wr_w = g.out_w * g.s_w, wr_h = g.out_h * g.s_h, wr_d = g.out_d * g.s_d;                 // a wrap's subtract
    const int lim_w = bw + wr_w, lim_h = bh + wr_h, lim_d = bd + wr_d;                                // wrapped <=> v >= lim
    int p_m[JA], v_w[JA], v_h[JA], v_d[JA], n_off[JA];
#pragma unroll
    for (int j = 0; j < JA; ++j) {
        int m = mbeg + ((wave + 4 * j) * 64 + lane) / PA, pw, ph, pd, pn;
        p_m[j] = m;
        divmod_pos(m, g.out_w, m, pw);
        divmod_pos(m, g.out_h, m, ph);
        divmod_pos(m, g.out_d, pn, pd);
        v_w[j] = pw * g.s_w + bw;
        v_h[j] = ph * g.s_h + bh;
        v_d[j] = pd * g.s_d + bd;
        n_off[j] = pn * Sn + a_ci;
    }
    // B pieces: row = idx / (BN / 4), float4 column idx % (BN / 4); pieces past the tile (QB not a multiple of 4) and columns
    // past the filter are dummies (offset out of range: they land as zeros in the stage's padding / unused columns)
    constexpr int PB = BN / 4;
    int b_m[JB], b_col[JB];
    unsigned b_dead[JB];                             // OOB for a dummy piece, else 0 (ORed into the offset: no branch, no select)
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const int idx = (wave + 4 * j) * 64 + lane;
        const int col = n0 + 4 * (idx % PB);
        const bool live = wave + 4 * j < QB && col < g.cout;
        b_m[j] = mbeg + idx / PB;
        b_col[j] = live ? col : 0;
        b_dead[j] = live ? 0u : OOB;
    }

    // ---- the refill tasks of one step (always run -- steps past the slice read nothing but keep the vmcnt bookkeeping uniform).
    // q < 4 JA: A piece q / 4, part q % 4; then B piece (q - 4 JA) / 2, part (q - 4 JA) % 2.  rs = stage to fill.
    unsigned vo[LPW];
    auto task = [&](auto qc, int rs) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value;
        (void)vo; (void)p_m; (void)v_w; (void)v_h; (void)v_d; (void)n_off; (void)b_m; (void)b_col; (void)b_dead;
        if constexpr (q < 4 * JA) {
            constexpr int j = q / 4, part = q % 4;
            if constexpr (part == 0) {
                const bool ok = a_ok & (p_m[j] < mend) & ((unsigned)v_w[j] < ext_w) & ((unsigned)v_h[j] < ext_h) & ((unsigned)v_d[j] < ext_d);
                const int off = n_off[j] + __mul24(v_d[j] >> up, Sd) + __mul24(v_h[j] >> up, Sh) + __mul24(v_w[j] >> up, Sw);
                vo[j] = ok ? (unsigned)off * 4u : OOB;
            } else if constexpr (part == 1) {
                p_m[j] += KB;
                v_w[j] += st_w;
                const bool cw = v_w[j] >= lim_w;
                v_w[j] -= cw ? wr_w : 0;
                v_h[j] += st_h + (cw ? g.s_h : 0);
            } else if constexpr (part == 2) {
                const bool ch = v_h[j] >= lim_h;
                v_h[j] -= ch ? wr_h : 0;
                v_d[j] += st_d + (ch ? g.s_d : 0);
                const bool cd = v_d[j] >= lim_d;
                v_d[j] -= cd ? wr_d : 0;
                n_off[j] += st_n + (cd ? Sn : 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_float*)(SM + rs * SA + (wave + 4 * j) * 256), 16, vo[j], 0, 0, 0);
            }
        } else {
            constexpr int j = (q - 4 * JA) / 2, part = (q - 4 * JA) % 2;
            if constexpr (part == 0) {
                // rows past the slice meet zero rows of A (their values only have to be readable: clamp); dummy pieces read as zero
                const int m = min(b_m[j], M - 1);
                b_m[j] += KB;
                vo[JA + j] = ((unsigned)(__mul24(m, g.cout) + b_col[j]) * 4u) | b_dead[j];
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(yres, (lds_float*)(SM + NS * SA + rs * SB + (wave + 4 * j) * 256), 16, vo[JA + j], 0, 0, 0);
            }
        }
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
    // LDS -- waits for every pending LDS-DMA, so the loop uses the bare s_barrier.  Ordering is done by hand: s_waitcnt vmcnt +
    // s_barrier before a stage is read, lgkmcnt(0) + a scheduling barrier before an operand set is used.  The destination
    // registers are plain asm OUTPUTS that the MFMAs read directly (no tied operands, no temporaries: a copy the register
    // allocator places between a read and its wait would see the register before the LDS data has landed -- the round-4 form of
    // this loop did that under LDS contention; scripts/isa_lds_hazard.py checks the compiled loops, tests/test_abi_cpu.py runs it).
    const int ac = wm * 32 * TM + (TM == 2 ? 2 * l31 : l31), bc = wn * 32 * TN + (TN == 2 ? 2 * l31 : l31);
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_float*)SM;
    const unsigned a_lane = lds0 + 4u * (unsigned)(half * BI + ac), b_lane = lds0 + 4u * (unsigned)(NS * SA + half * BN + bc);
    v2f a2[NSET][GK], b2[NSET][GK];                  // T == 2: the two adjacent columns of a k pair
    float a1[NSET][GK][TM], b1[NSET][GK][TN];        // else one register per tile
    // the r-th DS instruction of group gq's operand set (k pair r / RD1; A tiles first), base addresses of the stage in ab / bb
    auto read_op = [&](auto rc, auto gc, unsigned ab, unsigned bb) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value, gq = decltype(gc)::value, set = gq % NSET;
        constexpr int k = r / RD1, q = r % RD1, kk = 2 * (GK * gq + k);
        (void)a2; (void)b2; (void)a1; (void)b1;
        if constexpr (q < RDA) {
            if constexpr (TM == 2) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a2[set][k]) : "v"(ab), "n"(4 * kk * BI));
            else asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a1[set][k][TM == 2 ? 0 : q]) : "v"(ab), "n"(4 * kk * BI + 128 * q));
        } else {
            constexpr int qb = q - RDA;
            if constexpr (TN == 2) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(b2[set][k]) : "v"(bb), "n"(4 * kk * BN));
            else asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(b1[set][k][TN == 2 ? 0 : qb]) : "v"(bb), "n"(4 * kk * BN + 128 * qb));
        }
    };
    // the reads of slot m of a group: the m-th of NSL equal shares of the RD instructions
    auto read_slot = [&](auto mc, auto gc, unsigned ab, unsigned bb) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value;
        sl_static_for<RD>([&](auto rc) __attribute__((always_inline)) {
            constexpr int r = decltype(rc)::value;
            if constexpr (m < NSL && r >= (m * RD + NSL - 1) / NSL && r < ((m + 1) * RD + NSL - 1) / NSL) read_op(rc, gc, ab, bb);
        });
    };
    auto mfma_one = [&](auto mc, auto gc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value, set = decltype(gc)::value % NSET;
        constexpr int k = m / NMT, i = (m % NMT) / TN, j = m % TN;
        float av, bv;
        if constexpr (TM == 2) av = a2[set][k][i]; else av = a1[set][k][i];
        if constexpr (TN == 2) bv = b2[set][k][j]; else bv = b1[set][k][j];
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
    };

    if (nks > 0) {
        // prologue: steps 0 .. NS-2 in flight (stage s holds step s)
#pragma unroll 1
        for (int s = 0; s < NS - 1; ++s)
            sl_static_for<NTASK>([&](auto qc) __attribute__((always_inline)) { task(qc, s); });
        sl_wait_vmcnt<(NS - 2) * LPW>();             // step 0: own pieces landed
        __builtin_amdgcn_s_barrier();                //         everybody's
        __builtin_amdgcn_sched_barrier(0);
        sl_static_for<RD>([&](auto rc) __attribute__((always_inline)) { read_op(rc, SlInt<0>{}, a_lane, b_lane); });
        sl_static_for<RD>([&](auto rc) __attribute__((always_inline)) { read_op(rc, SlInt<1>{}, a_lane, b_lane); });
        int st = 0;                                  // stage of step t
        for (int t = 0; t < nks; ++t) {
            const int st_next = st + 1 == NS ? 0 : st + 1, rs = st == 0 ? NS - 1 : st - 1;
            const unsigned a_st = a_lane + 4u * (unsigned)(st * SA), b_st = b_lane + 4u * (unsigned)(st * SB);
            const unsigned a_nx = a_lane + 4u * (unsigned)(st_next * SA), b_nx = b_lane + 4u * (unsigned)(st_next * SB);
            sl_static_for<G>([&](auto gc) __attribute__((always_inline)) {
                constexpr int gq = decltype(gc)::value;
                if constexpr (gq == G - AHEAD) {
                    // every operand of stage st is in registers once the reads retire; step t+1 has landed for this wave;
                    // barrier: for everybody, and everybody is done reading stage st
                    sl_wait_lgkm<0>();
                    sl_wait_vmcnt<VMW>();
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                } else if constexpr (gq < G - AHEAD) {
                    sl_wait_lgkm<(AHEAD - 1) * RD>();    // group gq has landed (LDS reads retire in order; the younger groups may be in flight)
                }
                sl_static_for<NM>([&](auto mc) __attribute__((always_inline)) {
                    constexpr int m = decltype(mc)::value;
                    mfma_one(mc, gc);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (gq + AHEAD < G) read_slot(mc, SlInt<gq + AHEAD>{}, a_st, b_st);
                    else read_slot(mc, SlInt<gq + AHEAD - G>{}, a_nx, b_nx);       // (behind the barrier: step t+1 has landed)
                    sl_static_for<NTASK>([&](auto qc) __attribute__((always_inline)) {
                        constexpr int q = decltype(qc)::value;
                        if constexpr (w2_task_slot(q, NSLOT, NTASK) == gq * NM + m) task(qc, rs);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            st = st_next;
        }
        sl_wait_lgkm<0>();                           // (the read-ahead of the step past the end)
        sl_wait_vmcnt<0>();                          // (the dummy tail loads target LDS: they must not outlive the workgroup's allocation)
    }

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

// (the body is a device function: with generic lambdas directly inside the __global__ template hipcc 7.2 leaves the kernel's
// host-side launch stub undefined)
template <int WM, int WN, int TM, int TN, int KB, int NS>
__global__ __launch_bounds__(256) void wgrad2_kernel(CnConvGeom g, const float* __restrict__ X, const float* __restrict__ GY,
                                                     float* __restrict__ out, long slab_stride, int rows_per_split, int tiles_x,
                                                     int tiles_y, int nsplits, int accumulate) {
    wgrad2_body<WM, WN, TM, TN, KB, NS>(g, X, GY, out, slab_stride, rows_per_split, tiles_x, tiles_y, nsplits, accumulate);
}

struct Wg2Plan {
    int cfg;          // 0: 128x128, 1: 128x96, 2: 64x64, 3: 128x32, 4: 256x64
    int bi, bn, kb;
    long tiles_x, tiles_y, splits, rows;
};

long g_wg2_target = 0;     // cn_conv_tune(wg_blocks): workgroup target of the split (0 = the model below)
int g_wg2_cfg = -1;        // cn_conv_tune(cfg): forced tile
int g_wg2_ns = 0;          // cn_conv_loop_select(ns): forced stage count

// The tiles: rows of (tap, ci) x output channels, reduction rows per stage, and the share of the matrix pipes a CU sustains on the
// tile with one / with two workgroups resident (fitted to the per-shape sweep of round 6, profiles/round6_wgrad_sweep_points.txt).
struct Wg2Tile {
    int cfg, bi, bn, kb;
    double e1, e2;
};
const Wg2Tile WG2_TILES[5] = {{0, 128, 128, 16, 0.72, 0.84}, {1, 128, 96, 16, 0.62, 0.72}, {2, 64, 64, 32, 0.50, 0.56},
                              {3, 128, 32, 32, 0.38, 0.42}, {4, 256, 64, 16, 0.55, 0.63}};

// Estimated duration (us) of the launch + the slab reduction for `s` row slices on tile t.  Workgroups of one slice share an XCD
// (slices dealt round-robin to the 8 XCDs, 32 CUs each, two workgroups resident per CU): the launch lasts as long as the XCD with
// the most slices.  Checked against the 1 096 points of the sweep: its choice is within 5 % of the best measured point in total.
double wg2_cost(const Wg2Tile& t, long M, long Ktot, int cout, long s, long& rows_out, long& splits_out) {
    long rows = (M + s - 1) / s;
    rows = (rows + t.kb - 1) / t.kb * t.kb;
    const long splits = (M + rows - 1) / rows;
    rows_out = rows;
    splits_out = splits;
    const long tiles = (long)cn_cdiv(Ktot, t.bi) * cn_cdiv(cout, t.bn);
    const long per_xcd = splits >= 8 ? (long)cn_cdiv(splits, 8) * tiles : (long)cn_cdiv(tiles * splits, 8);
    const double cyc_row = 32.0 * (t.bi / 32) * (t.bn / 32) / 4.0;           // MFMA cycles per reduction row (4 waves on 4 SIMDs)
    const double life = ((double)rows * cyc_row + 9000.0) / 2400.0;         // us at the full pipe, + prologue / epilogue
    // an XCD (an eighth of the CUs: 32 on an MI355X) holds two workgroups per CU at once; full rounds run two to a CU, the last
    // one alone if it has at most one workgroup per CU
    const long xcd_cus = cn_cu_count() >= 8 ? cn_cu_count() / 8 : 1;
    const long rounds = cn_cdiv(per_xcd, 2 * xcd_cus), rem = per_xcd - 2 * xcd_cus * (rounds - 1);
    double us = (double)(rounds - 1) * 2.0 * life / t.e2 + (rem <= xcd_cus ? life / t.e1 : 2.0 * life / t.e2);
    if (splits > 1) us += 2.0 * (double)splits * (double)Ktot * cout * 4.0 / 2.5e6 + 5.0;      // slabs written + read back, one more launch
    return us;
}

Wg2Plan wg2_plan(const CnConvGeom& g) {
    const long M = (long)g.n * g.out_d * g.out_h * g.out_w;
    const long Ktot = (long)g.k_d * g.k_h * g.k_w * g.cin;
    Wg2Plan p{};
    double best = -1.0;
    for (const Wg2Tile& t : WG2_TILES) {
        if (g_wg2_cfg >= 0 ? t.cfg != g_wg2_cfg
                           : ((t.cfg == 3) != (g.cout <= 32) || (t.cfg == 1 && g.cout % 96 != 0) || (t.cfg == 0 && g.cout < 96) ||
                              (t.cfg == 4 && (g.cout % 64 != 0 || Ktot < 256)) || (t.cfg != 2 && t.cfg != 3 && Ktot < 128)))
            continue;
        const long tiles = (long)cn_cdiv(Ktot, t.bi) * cn_cdiv(g.cout, t.bn);
        const long max_splits = cn_cdiv(M, 4 * t.kb);                 // a workgroup is at least 4 K steps long
        auto consider = [&](long s_) {
            if (s_ < 1) s_ = 1;
            if (s_ > max_splits) s_ = max_splits;
            long rows, splits;
            const double us = wg2_cost(t, M, Ktot, g.cout, s_, rows, splits);
            if (best < 0.0 || us < best) {
                best = us;
                p.cfg = t.cfg; p.bi = t.bi; p.bn = t.bn; p.kb = t.kb;
                p.tiles_x = cn_cdiv(Ktot, t.bi); p.tiles_y = cn_cdiv(g.cout, t.bn);
                p.rows = rows; p.splits = splits;
            }
        };
        if (g_wg2_target > 0) {
            consider(tiles >= g_wg2_target ? 1 : (g_wg2_target + tiles / 2) / tiles);
        } else {
            for (long s_ = 1; s_ < 8 && s_ <= max_splits; ++s_) consider(s_);
            for (long s_ = 8; s_ <= max_splits && s_ <= 1024; s_ += 8) consider(s_);
        }
    }
    if (best < 0.0) {          // (a forced tile that the shape cannot use: the 64 x 64 tile takes everything)
        const int keep = g_wg2_cfg;
        g_wg2_cfg = 2;
        p = wg2_plan(g);
        g_wg2_cfg = keep;
    }
    return p;
}

}  // namespace

bool cn_wgrad2_ok(const CnConvGeom& g) {
    const double xb = (double)g.n * g.in_d * g.in_h * g.in_w * g.cin * 4.0;
    const double yb = (double)g.n * g.out_d * g.out_h * g.out_w * g.cout * 4.0;
    // (24-bit multiplies of the gather: source coordinate x stride of the axis, output row x cout)
    const double sd = (double)g.in_h * g.in_w * g.cin, rows = (double)g.n * g.out_d * g.out_h * g.out_w;
    return g.cin % 4 == 0 && g.cout % 4 == 0 && g.cout > 4 && xb < 2147483647.0 && yb < 2147483647.0 &&
           g.dl_d == 1 && g.dl_h == 1 && g.dl_w == 1 && sd < 8388608.0 && rows < 8388608.0 && g.cout < 8388608;
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

void cn_wgrad2_stages(int ns) { g_wg2_ns = ns; }     // cn_conv_loop_select: 0 = default, 3 / 4 forced

int cn_wgrad2_family(const CnConvGeom& g) {
    const Wg2Plan p = wg2_plan(g);
    return p.cfg == 0 ? CN_FAM_WGRAD_128x128 : p.cfg == 1 ? CN_FAM_WGRAD_128x96 : p.cfg == 2 ? CN_FAM_WGRAD_64x64 : p.cfg == 4 ? CN_FAM_WGRAD_256x64 : CN_FAM_WGRAD_128x32;
}

// gw (+)= filter gradient.  ws: at least cn_wgrad2_workspace_floats(g) floats (may be NULL when that is 0).
// parts_out: NULL = add the slabs here (second launch); else the caller does: *parts_out = slabs written (0: gw is complete).
int cn_wgrad2(const CnConvGeom& g, const float* x, const float* gy, float* gw, int accumulate, float* ws, hipStream_t s, int* parts_out) {
    const Wg2Plan p = wg2_plan(g);
    const long Ktot = (long)g.k_d * g.k_h * g.k_w * g.cin;
    const long count = Ktot * g.cout;
    CN_CHECK_ARG(p.splits == 1 || ws, "filter gradient: %ld row splits need a workspace of %ld floats", p.splits, p.splits * count);
    const long ntile = p.tiles_x * p.tiles_y;
    dim3 grid((unsigned)((p.splits >= 8 ? cn_cdiv(p.splits, 8) * 8 : p.splits) * ntile));
    float* out = p.splits > 1 ? ws : gw;
    const long stride = p.splits > 1 ? count : 0;
#define WG2(WM, WN, TM, TN, KB_, NS_) hipLaunchKernelGGL((wgrad2_kernel<WM, WN, TM, TN, KB_, NS_>), grid, dim3(256), 0, s, g, x, gy, out, stride, \
                                                          (int)p.rows, (int)p.tiles_x, (int)p.tiles_y, (int)p.splits, accumulate)
    const int ns = g_wg2_ns ? g_wg2_ns : 4;
#define WG2N(WM, WN, TM, TN, KB_) do { if (ns == 3) WG2(WM, WN, TM, TN, KB_, 3); else WG2(WM, WN, TM, TN, KB_, 4); } while (0)
    switch (p.cfg) {
        case 0: WG2N(2, 2, 2, 2, 16); break;
        case 1: WG2N(4, 1, 1, 3, 16); break;
        case 2: WG2N(2, 2, 1, 1, 32); break;
        case 4: WG2N(4, 1, 2, 2, 16); break;
        default: WG2(4, 1, 1, 1, 32, 3); break;
    }
#undef WG2N
#undef WG2
    CN_LAUNCH_CHECK();
    if (parts_out) {
        *parts_out = p.splits > 1 ? (int)p.splits : 0;
        return CN_OK;
    }
    if (p.splits > 1) return cn_sum_parts(ws, gw, (int)p.splits, count, accumulate, 1.f, s);
    return CN_OK;
}
