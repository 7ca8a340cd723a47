// fwd2.hip -- forward / data-gradient main loop of the implicit-GEMM family with BOTH operands delivered by LDS-DMA (round 5;
// reference ops: the Conv2D / Conv3D and Conv*BackpropInput behind building_blocks.py:29,65,91 and real_encoder.py:13).
//
//   C[m, n] = sum_{t, c} A[src(m, t), c] * B[(t, c), n]          m = output position, (t, c) = (tap, input channel), n = cout
//
// The round-3 plain-GEMM loop (gemm1x1.hip, deleted in round 6) staged both tiles through registers (global_load -> VGPR -> ds_write_b128 -> barrier -> ds_read) two steps ahead;
// the loads, their address arithmetic, the LDS stores and the operand reads all sit in the instruction stream of the waves that
// issue the MFMAs, and at 2 - 4 waves per SIMD that stream is what the matrix pipe waits for (round 4: the class runs at
// t_mfma + t_rest).  Here a K step's tiles go from L2 straight into LDS (`buffer_load_dwordx4 ... lds`, 1 KB per wave
// instruction, per-lane source offsets = the gather, an offset past the buffer descriptor reads as zero = padding taps, rows
// past the end, columns past the filter), NS stages deep: the loads of step t + NS are issued in the shadow of the MFMAs of step
// t, nothing of a tile ever passes through a VGPR, and the only per-step work left in the wave is NS - independent: a handful
// of address adds, the operand reads and one barrier.
//
// LDS images.  A rows arrive K-contiguous (a row of x per position, channels contiguous) and stay so: [BM][KB] floats, the KB / 4
// 16-byte pieces of a row XOR-swizzled with the row index so that the ds_read_b128 of 16 consecutive rows (one LDS cycle's lane
// group) covers all 64 banks -- an LDS-DMA instruction writes lane L's 16 bytes at byte 16 L of its 1 KB block, so the swizzle
// is applied on the SOURCE side: lane L fetches the piece that belongs at slot L.  One ds_read_b128 per lane feeds four MFMAs
// (K order inside an 8-deep group permuted: half-wave h owns k = 4h .. 4h + 3).  Forward: the filter tile
// [KB][BN] is k-major as it lies in memory, a lane owns TN ADJACENT output columns.  Data gradient from the original filter
// (BT): B^T rows are K-contiguous like A and use A's image.
//
// Schedule of step t (stage t % NS), one barrier per step, placed in the MIDDLE of the step's MFMA stream:
//     read group 1 operands | MFMAs of group 0 | wait: own pieces of step t+1 landed | s_barrier | read group 0 of step t+1 |
//     MFMAs of the last group with the LDS-DMA issue of step t + NS (into the stage just freed) between them
// -- the operand reads of a group are always issued one MFMA group ahead, and the barrier wait overlaps the other waves' MFMAs.
// The operand reads are inline asm with hand-counted lgkmcnt and the barrier is the bare s_barrier, for the two compiler
// reasons stated in wgrad2.hip (a ds_read that may alias an LDS-DMA destination is preceded by vmcnt(0); __syncthreads drains
// the LDS-DMA queue).  scripts/isa_lds_hazard.py replays the compiled loop (tests/test_abi_cpu.py).
#include "common.h"
#include "mma_tile.h"
#include "conv_geom.h"
#include "typed.h"

// Ablation builds (scripts/dev/fwd2_ablate.sh; never defined in the product build): bit 0 = no MFMAs, bit 1 = no LDS-DMA inside the
// loop (the prologue's stages are re-read), bit 2 = no operand reads.  Results are garbage; only the durations mean something.
#ifndef FWD2_ABLATE
#define FWD2_ABLATE 0
#endif

namespace {

typedef __attribute__((address_space(3))) float lds_float;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 f2_bf16x8 __attribute__((ext_vector_type(8)));

template <int N>
__device__ __forceinline__ void f2_wait_vmcnt() {
    // s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt = imm[3:0] | imm[15:14] << 4, expcnt imm[6:4], lgkmcnt imm[11:8])
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

template <int N>
__device__ __forceinline__ void f2_wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N));
    __builtin_amdgcn_sched_barrier(0);
}

template <int N>
struct F2Int {
    static constexpr int value = N;
};

template <int I, int N, class F>
__device__ __forceinline__ void f2_static_for_impl(F&& f) {
    if constexpr (I < N) {
        f(F2Int<I>{});
        f2_static_for_impl<I + 1, N>(f);
    }
}

// f(F2Int<0>) ... f(F2Int<N - 1>): loop indices that are constant expressions inside the body (asm immediates, if constexpr)
template <int N, class F>
__device__ __forceinline__ void f2_static_for(F&& f) {
    f2_static_for_impl<0, N>(f);
}

// (the body is a device function: with generic lambdas directly inside the __global__ template hipcc 7.2 leaves the kernel's
// host-side launch stub undefined)
// BF: bf16 storage on v_mfma_f32_32x32x16_bf16 (the compute path of BASELINE.json configs[2]): A, B and C hold bf16, K counts
// bf16 elements, a stage row is still 64 bytes (32 elements: one 16-byte piece = the 8 consecutive k of one MFMA operand), B is
// always K-contiguous rows ([tap][N][K]: the two prepared filter copies of igemm_bf16.hip; flip = the data gradient walks the
// taps backwards), the output tile goes back through LDS and leaves as 16-byte row pieces.  No K split.
template <int WM, int WN, int TM, int TN, bool BT, bool GATHER, int KB, int NS, int NP, bool BF>
__device__ __forceinline__ void fwd2_body(const CnConvGeom& g, const float* __restrict__ A, const float* __restrict__ B,
                                          const float* __restrict__ bias, float* __restrict__ C, int M, int N, int K,
                                          int act, float slope, int ntm, int ntn, long part_stride, int par,
                                          const float* __restrict__ res, unsigned a_bytes, unsigned b_bytes, int flip,
                                          float* __restrict__ stats, int stats_mode, float stats_slope, int srows, int sper) {
    static_assert(!BF || BT, "bf16: both operands are K-contiguous rows");
    constexpr int ES = BF ? 2 : 4;                   // bytes per element
    constexpr int KE = KB * 4 / ES;                  // elements of the reduction axis per stage
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(KB == 16 || KB == 32, "stage depth");
    constexpr int KP = KB / 4;                       // 16-byte pieces per row and stage
    constexpr int RPI = 64 / KP;                     // rows per 1 KB wave instruction
    constexpr int FSH = KP == 4 ? 2 : 1;             // swizzle: piece ^= (row >> FSH) & (KP - 1)
    constexpr int G = KB / 8;                        // 8-deep MFMA groups per stage
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    // NP = 0: the four MFMA waves also issue the LDS-DMA pieces (behind their last MFMAs).  NP = 1 / 2: that many LOADER waves
    // (waves 4 ..) issue all of them and the MFMA waves never touch the vector memory path: an LDS-DMA instruction holds its
    // wave's issue for 60 - 70 cycles -- longer than the 64-cycle shadow of the MFMA it sits behind -- and the 64 x 64 tile has
    // one per four MFMAs (ablation: 7.7 us of a 56 us launch).
    constexpr int NL = NP > 0 ? NP : 4;              // waves that load
    constexpr int IA = BM / RPI, JA = IA / NL;       // A: wave instructions per stage / per loading wave
    static_assert(IA % NL == 0, "A pieces divide over the loading waves");
    static_assert(NP == 0 || KB == 16, "loader waves: 16-deep stages");
    constexpr int IB = BT ? BN / RPI : KB * BN / 256;
    constexpr int JB = (IB + NL - 1) / NL;
    constexpr int LPW = JA + JB;                     // LDS-DMA instructions per loading wave and step (dummies keep it uniform)
    constexpr int SA = BM * KB, SB = JB * NL * 256;  // floats per stage
    // B reads per operand set: K-contiguous rows one ds_read_b128 per tile; k-major rows one read per k row (8 bytes for two
    // adjacent columns), for a single column two k rows per ds_read2_b32 (rows 64 dwords apart: offsets 0 / 64 and 128 / 192)
    constexpr bool B2 = !BT && TN == 1 && BN == 64;
    constexpr int NB = BT ? TN : B2 ? 2 : 4 * (TN == 2 ? 1 : TN);
    constexpr int RD = TM + NB;                      // DS instructions per operand set
    static_assert((NS - 1) * LPW < 64, "vmcnt range");
    __shared__ __attribute__((aligned(1024))) float SM[NS * (SA + SB)];     // stage s: A at s * SA, B at NS * SA + s * SB
    __shared__ int rowmap[GATHER ? BM : 1];          // GATHER: tile row -> output row (or -1)

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // (an SGPR: LDS-DMA destinations and tile offsets stay scalar)
    const bool loader = NP == 0 || wave >= 4, worker = wave < 4;      // (NP = 0: every wave is both)
    const int lw = NP > 0 ? wave - 4 : wave;                          // index among the loading waves
    const int wm = (wave & 3) / WN, wn = (wave & 3) % WN;
    // workgroup order (parity-ordered launches keep the plain order: their row classes are already dealt out class-major)
    int bx, by;
    if (GATHER && par) {
        divmod_pos((int)blockIdx.x, ntm, by, bx);
        if (by >= ntn) return;
    } else {
        // XCD id % 8 gets a contiguous run of TILES in m-major order (the column tiles of one M tile in consecutive slots: they
        // read the same A rows from that XCD's L2), balanced to within one tile.  (Dealing out whole M tiles would leave XCDs
        // empty when there are fewer than 8 of them -- 128-row tiles at M = 512.)
        const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
        const int nt = ntm * ntn, q = nt >> 3, r = nt & 7;
        if (j >= q + (xcd < r ? 1 : 0)) return;
        const int u = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        divmod_pos(u, ntn, bx, by);
    }
    const int m0 = bx * BM, n0 = by * BN;
    const int T = GATHER ? g.k_d * g.k_h * g.k_w : 1;
    const int cpb = K / KE;
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
    const int nks = max(ks_end - ks_beg, 0);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const __amdgpu_buffer_rsrc_t ares = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (int)a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t bres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, (int)b_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;            // past every descriptor's range: the piece lands as zeros

    // ---- this lane's pieces.  A (and B^T): instruction I = wave + 4 j covers rows I * RPI .. + RPI - 1; lane L writes slot L of
    // the 1 KB block = (row I * RPI + L / KP, piece slot L % KP) and therefore FETCHES piece (L % KP) ^ f(row)
    RowInfo ri[JA];
    int a_base[JA];                                  // element offset of the row's channel 0 at the current tap, or -1
    // f(row) = (row >> FSH) & (KP - 1) with row = I * RPI + L / KP: for 16-deep stages (RPI = 16) I drops out; for 32-deep ones
    // (RPI = 8, NP = 0 only) I = wave + 4 j contributes wave & 1
    const int a_piece = 16 * ((lane % KP) ^ ((((lw & 1) * RPI + lane / KP) >> FSH) & (KP - 1)));     // bytes
#pragma unroll
    for (int j = 0; j < JA; ++j) {
        const int r = (lw + NL * j) * RPI + lane / KP;
        if (!loader) {
            ri[j] = RowInfo{};
            a_base[j] = -1;
        } else if (GATHER) {
            int mrow = m0 + r, cls;
            if (par) mrow = par_row(g, mrow, M, cls);
            ri[j] = decode_row(g, mrow, M);
            if (lane % KP == 0) rowmap[r] = ri[j].ok ? mrow : -1;
            a_base[j] = -1;
        } else {
            a_base[j] = m0 + r < M ? (m0 + r) * K : -1;
        }
    }
    int b_off[JB];                                   // BT: filter row * K (+ piece), else k row * N + column; -1 = dummy piece
    int b_krow[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const int I = lw + NL * j;
        if (BT) {
            const int n = n0 + I * RPI + lane / KP;
            b_off[j] = (I < IB && n < N) ? n * K * ES + a_piece : -1;             // bytes
            b_krow[j] = 0;
        } else {
            const int idx = I * 64 + lane, col = n0 + 4 * (idx % (BN / 4));
            b_krow[j] = idx / (BN / 4);
            b_off[j] = (I < IB && col < N) ? col : -1;
        }
    }
    if (GATHER) __syncthreads();                     // rowmap (before any LDS-DMA is in flight: __syncthreads would drain them)

    // ---- K walk of the loader: (live tap, channel chunk) of the NEXT step to issue.  Everything that changes per step is wave
    // uniform and lives in SGPRs (the channel offset and the filter's tap offset go into the buffer instruction's scalar offset);
    // the per-lane byte offsets change only when the tap does.  Past the last step the loader re-issues the last step (valid
    // addresses, stages nobody reads): the vmcnt bookkeeping stays uniform without a per-piece select.
    int ld_left = nks;                               // steps still to issue
    int ld_tap = -1, ld_c0 = (ks_beg - (ks_beg / cpb) * cpb) * KE;
    if (GATHER)
        for (int o = ks_beg / cpb; o >= 0; --o) ld_tap += __ffsll((long long)(tapmask >> (ld_tap + 1)));   // the (ks_beg / cpb)-th live tap
    else
        ld_tap = 0;
    unsigned a_vo[JA], b_vo[JB];                     // per-lane byte offsets (OOB = the piece lands as zeros)
    int a_so = 0, b_so = 0;                          // scalar byte offsets of the step
    // Source offset of a row at a tap without branches or divisions (the zero-stuffing divisor of this family is 1 or 2: a shift
    // and a mask): v = row origin + tap, dead if negative, odd where the gradient is zero-stuffed, or past the (upsampled) extent.
    const int sh_d = g.dl_d - 1, sh_h = g.dl_h - 1, sh_w = g.dl_w - 1;
    const int ext_d = g.in_d << g.up, ext_h = g.in_h << g.up, ext_w = g.in_w << g.up;
    auto retap = [&]() __attribute__((always_inline)) {
        if (GATHER) {
            int kd, kh, kw;
            tap_decode(g, ld_tap, kd, kh, kw);
#pragma unroll
            for (int j = 0; j < JA; ++j) {
                const int vd = ri[j].vd + kd, vh = ri[j].vh + kh, vw = ri[j].vw + kw;
                const int qd = vd >> sh_d, qh = vh >> sh_h, qw = vw >> sh_w;
                const bool ok = ri[j].ok & ((vd | vh | vw) >= 0) & (((vd & sh_d) | (vh & sh_h) | (vw & sh_w)) == 0) &
                                (qd < ext_d) & (qh < ext_h) & (qw < ext_w);
                const int off = (((ri[j].nbase + (qd >> g.up)) * g.in_h + (qh >> g.up)) * g.in_w + (qw >> g.up)) * g.cin;
                a_vo[j] = ok ? (unsigned)(off * ES + a_piece) : OOB;
            }
        }
    };
    auto set_so = [&]() __attribute__((always_inline)) {
        a_so = ld_c0 * ES;
        b_so = BT ? ((flip ? T - 1 - ld_tap : ld_tap) * N * K + ld_c0) * ES : (ld_tap * K + ld_c0) * N * 4;
    };
#pragma unroll
    for (int j = 0; j < JA; ++j) a_vo[j] = (!GATHER && a_base[j] >= 0) ? (unsigned)(a_base[j] * ES + a_piece) : OOB;
#pragma unroll
    for (int j = 0; j < JB; ++j) b_vo[j] = b_off[j] >= 0 ? (BT ? (unsigned)b_off[j] : (unsigned)(b_krow[j] * N + b_off[j]) * 4u) : OOB;
    retap();
    set_so();
    auto issue_piece = [&](int stage, int p) __attribute__((always_inline)) {
        if (p < JA)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ares, (lds_float*)(SM + stage * SA + (lw + NL * p) * 256), 16, a_vo[p], a_so, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(bres, (lds_float*)(SM + NS * SA + stage * SB + (lw + NL * (p - JA)) * 256), 16,
                                                     b_vo[p - JA], b_so, 0, 0);
    };
    auto advance = [&]() __attribute__((always_inline)) {      // after the last piece of a step
        if (--ld_left > 0) {
            ld_c0 += KE;
            if (ld_c0 == K) {
                ld_c0 = 0;
                if (GATHER) {
                    ld_tap += __ffsll((long long)(tapmask >> (ld_tap + 1)));
                    retap();
                }
            }
            set_so();
        }
    };

    // ---- operand reads (inline asm: see the header).  Byte addresses relative to the stage; group gq reads piece
    // (2 gq + half) ^ f = ((half ^ f) ^ 2 gq): base with gq = 0, then XOR 32 gq on the piece field.
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_float*)SM;
    const unsigned fsw = (unsigned)((l31 >> FSH) & (KP - 1));
    const unsigned a_rd = lds0 + 4u * (unsigned)((wm * 32 * TM + l31) * KB) + 16u * ((unsigned)half ^ fsw);
    const unsigned bt_rd = lds0 + 4u * (unsigned)(NS * SA + (wn * 32 * TN + l31) * KB) + 16u * ((unsigned)half ^ fsw);
    const unsigned bn_rd = lds0 + 4u * (unsigned)(NS * SA + (4 * half) * BN + wn * 32 * TN + TN * l31);
    f4 av[2][TM];
    f4 btv[2][BT ? TN : 1];
    float bnv[2][BT ? 1 : 4][TN];
    unsigned ap_cur = 0, bp_cur = 0;                 // operand base addresses of (stage, group) set by read_base
    auto read_base = [&](int stage, int gq) __attribute__((always_inline)) {
        ap_cur = (a_rd + 4u * (unsigned)(stage * SA)) ^ (32u * (unsigned)gq);
        bp_cur = BT ? (bt_rd + 4u * (unsigned)(stage * SB)) ^ (32u * (unsigned)gq) : bn_rd + 4u * (unsigned)(stage * SB + 8 * gq * BN);
    };
    // the r-th DS instruction of an operand set (r < TM: A tile r; then B), r a compile-time constant
    auto read_op = [&](auto rc, int set) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        (void)av; (void)btv; (void)bnv; (void)ap_cur; (void)bp_cur;       // (named outside the if constexpr: implicit captures)
        if (FWD2_ABLATE & 4) return;
        if constexpr (r < TM) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(av[set][r]) : "v"(ap_cur), "n"(4 * 32 * KB * r));
        } else if constexpr (BT) {
            constexpr int j = r - TM;
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(btv[set][BT ? j : 0]) : "v"(bp_cur), "n"(4 * 32 * KB * j));
        } else if constexpr (B2) {
            constexpr int q = 2 * (r - TM);
            f2 v;
            asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(bp_cur), "n"(BN * q), "n"(BN * (q + 1)));
            bnv[set][BT ? 0 : q][0] = v.x;
            bnv[set][BT ? 0 : q + 1][0] = v.y;
        } else if constexpr (TN == 2) {
            constexpr int q = r - TM;
            f2 v;
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(bp_cur), "n"(4 * BN * q));
            bnv[set][BT ? 0 : q][0] = v.x;
            bnv[set][BT ? 0 : q][TN - 1] = v.y;
        } else {
            constexpr int q = (r - TM) / TN, j = (r - TM) % TN;
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(bnv[set][BT ? 0 : q][j]) : "v"(bp_cur), "n"(4 * BN * q + 4 * j));
        }
    };
    // Note on the TN == 2 form above: the asm writes a temporary pair and two v_mov follow it -- those copies READ the ds_read's
    // destination, so they must sit behind the wait.  They do: the copies are emitted where the values are first used (the MFMA
    // operands), which is behind f2_wait_lgkm; scripts/isa_lds_hazard.py checks exactly this on the compiled loop.
    constexpr int NM = (BF ? 1 : 4) * TM * TN;       // MFMAs per group (fp32: 8 deep = 4 contraction pairs; bf16: one 16-deep MFMA)
    auto mfma_one = [&](auto mc, int set) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value, q = m / (TM * TN), i = (m % (TM * TN)) / TN, j = m % TN;
        if (FWD2_ABLATE & 1) return;
        if constexpr (BF) {
            (void)q;
            union { f4 f; f2_bf16x8 h; } ua, ub;
            ua.f = av[set][i];
            ub.f = btv[set][BT ? j : 0];
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.h, ub.h, acc[i][j], 0, 0, 0);
        } else {
            const float a = av[set][i][q];
            const float b = BT ? btv[set][BT ? j : 0][q] : bnv[set][BT ? 0 : q][j];
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
        }
    };

    if (NP > 0 && !worker) {
        // ---- a loader wave: the same protocol as the MFMA waves' (one barrier per step: step t+1 has landed / stage t is free),
        // nothing but LDS-DMA
        if (nks > 0) {
#pragma unroll 1
            for (int s = 0; s < NS; ++s) {
#pragma unroll
                for (int p = 0; p < LPW; ++p) issue_piece(s, p);
                advance();
            }
            f2_wait_vmcnt<(NS - 1) * LPW>();
            __builtin_amdgcn_s_barrier();
            int st = 0;
#pragma unroll 1
            for (int t = 0; t < nks; ++t) {
                f2_wait_vmcnt<(NS - 2) * LPW>();
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int p = 0; p < LPW; ++p) issue_piece(st, p);
                advance();
                st = st + 1 == NS ? 0 : st + 1;
            }
            f2_wait_vmcnt<0>();
        }
        if (!BF) return;                             // (bf16: the loader waves help to move the output tile out of LDS)
    }
    if (nks > 0 && worker) {
        // prologue: NS steps in flight (stage s holds step s)
        if (NP == 0) {
#pragma unroll 1
            for (int s = 0; s < NS; ++s) {
#pragma unroll
                for (int p = 0; p < LPW; ++p) issue_piece(s, p);
                advance();
            }
            f2_wait_vmcnt<(NS - 1) * LPW>();         // step 0: own pieces landed
        }
        __builtin_amdgcn_s_barrier();                //         everybody's pieces landed
        __builtin_amdgcn_sched_barrier(0);
        read_base(0, 0);
        f2_static_for<RD>([&](auto rc) __attribute__((always_inline)) { read_op(rc, 0); });
        int st = 0;                                  // stage of step t
        // The in-order issue of a wave makes every instruction that is not placed BETWEEN two MFMAs wait for (or delay) the
        // matrix pipe: a v_mfma_f32_32x32x2_f32 holds the pipe 64 cycles, the next MFMA of the wave stalls at issue until then,
        // and whatever sits in between runs in that shadow for free as long as it issues in < 64 cycles.  Round-5 ablation of
        // the first form of this loop (all reads of a group in one block, both LDS-DMA pieces behind one MFMA; 64 x 64 tile,
        // 144 steps, one wave per SIMD): MFMAs alone 27.7 us, LDS-DMA alone 16.3, operand reads alone 10.0, skeleton 12.6, all of
        // it 65.2 -- the SUM.  So every step is laid out as MFMA slots with at most a few instructions behind each.
        for (int t = 0; t < nks; ++t) {
            const int st_next = st + 1 == NS ? 0 : st + 1;
            f2_wait_lgkm<0>();                       // group 0 (read during the previous step's last group)
            // groups 0 .. G-2: multiply group gq, read group gq + 1 behind its first MFMAs
#pragma unroll
            for (int gq = 0; gq + 1 < G; ++gq) {
                read_base(st, gq + 1);
                f2_static_for<NM>([&](auto mc) __attribute__((always_inline)) {
                    constexpr int m = decltype(mc)::value;
                    constexpr int NSL = NM > 1 ? NM / 2 : 1;   // the reads go behind the FIRST half of the MFMAs: an LDS read issued one
                                                         // MFMA (64 cycles) before its wait is not back yet (~100+ cycles)
                    mfma_one(mc, gq & 1);
                    __builtin_amdgcn_sched_barrier(0);
                    f2_static_for<RD>([&](auto rc) __attribute__((always_inline)) {
                        constexpr int r = decltype(rc)::value;
                        if constexpr (m < NSL && r >= (m * RD + NSL - 1) / NSL && r < ((m + 1) * RD + NSL - 1) / NSL) read_op(rc, (gq + 1) & 1);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
                if (gq + 2 < G) f2_wait_lgkm<0>();
            }
            // every operand of stage st is in registers once the reads retire; step t+1 has landed for this wave; barrier: for
            // everybody, and everybody is done reading stage st.  All of it in the shadow of the MFMA issued last.
            f2_wait_lgkm<0>();
            if (NP == 0 && !(FWD2_ABLATE & 2)) f2_wait_vmcnt<(NS - 2) * LPW>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // last group: group 0 of step t+1 behind its first MFMAs, then the refill of stage st (step t + NS)
            read_base(st_next, 0);
            f2_static_for<NM>([&](auto mc) __attribute__((always_inline)) {
                constexpr int m = decltype(mc)::value;
                constexpr int NSL = NM > 1 ? NM / 2 : 1; // reads behind the first half of the MFMAs, LDS-DMA pieces behind the second
                mfma_one(mc, (G - 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                f2_static_for<RD>([&](auto rc) __attribute__((always_inline)) {
                    constexpr int r = decltype(rc)::value;
                    if constexpr (m < NSL && r >= (m * RD + NSL - 1) / NSL && r < ((m + 1) * RD + NSL - 1) / NSL) read_op(rc, G & 1);
                });
                f2_static_for<LPW>([&](auto pc) __attribute__((always_inline)) {
                    constexpr int pp = decltype(pc)::value;
                    constexpr int slot_ = LPW <= NM - NSL ? NM - LPW + pp : NSL + pp * (NM - NSL) / LPW;
                    constexpr int slot = slot_ < NM ? slot_ : NM - 1;
                    if constexpr (slot == m && NP == 0) {
                        if (!(FWD2_ABLATE & 2)) {
                            issue_piece(st, pp);
                            if (pp == LPW - 1) advance();
                        }
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            st = st_next;
        }
        f2_wait_lgkm<0>();                           // (the read-ahead of the step past the end)
        if (NP == 0) f2_wait_vmcnt<0>();             // (the tail's re-issued loads target LDS: they must not outlive the allocation)
    }


    if constexpr (BF) {
        // bf16 epilogue: bias + activation in fp32, the tile as bf16 through LDS (over the operand stages: every wave is past its
        // last operand read and every LDS-DMA has landed once the barrier releases), 16-byte row pieces out
        constexpr int LDC = BN + 8;
        static_assert(BM * LDC * 2 <= NS * (SA + SB) * 4, "output tile fits the operand stages");
        bf16_t* const Cs = reinterpret_cast<bf16_t*>(SM);
        bf16_t* const Y = reinterpret_cast<bf16_t*>(C);
        __syncthreads();
        if (worker) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int ct = wn * 32 * TN + 32 * j + l31, col = n0 + ct;
                const float bv = (bias && col < N) ? bias[col] : 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        Cs[(wm * 32 * TM + 32 * i + 4 * half + (r & 3) + 8 * (r >> 2)) * LDC + ct] =
                            f32_to_bf16(cn_apply_act(acc[i][j][r] + bv, act, slope));
            }
        }
        __syncthreads();
        constexpr int NPC = BN / 8;
        for (int idx = tid; idx < BM * NPC; idx += 256 + 64 * NP) {
            const int row = idx / NPC, pc = idx - row * NPC;
            const int orow = GATHER ? rowmap[row] : (m0 + row < M ? m0 + row : -1), col = n0 + pc * 8;
            if (orow >= 0 && col < N)
                *reinterpret_cast<uint4*>(Y + (long)orow * N + col) = *reinterpret_cast<const uint4*>(Cs + row * LDC + pc * 8);
        }
        return;
    }
    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    // stats (unsplit forward launches whose tiles lie inside one sample; reference building_blocks.py:37-44,97-106): the
    // per-(sample, channel) sums the FOLLOWING normalisation layer needs, taken from the values this epilogue holds anyway instead
    // of a separate pass over the tensor -- mode 1: sum a, sum a^2 of the stored (activated) value (AdaIn after the fused
    // LeakyReLU); mode 2: sum v, sum v^2, sum l, sum l^2 with l = leaky_relu(v, stats_slope) (DiscrBlock: style statistics of the
    // pre-activation tensor + instance-norm statistics of its activation).  stats[k][sample][channel], zeroed by the caller.
    const bool split = gridDim.z > 1;
    float sacc[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) sacc[j][k] = 0.f;
    auto stat = [&](int j, float v, float a) __attribute__((always_inline)) {
        if (stats_mode == 1) {
            sacc[j][0] += a;
            sacc[j][1] += a * a;
        } else {
            const float l = v > 0.f ? v : v * stats_slope;
            sacc[j][0] += v;
            sacc[j][1] += v * v;
            sacc[j][2] += l;
            sacc[j][3] += l * l;
        }
    };
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
                    const float a0 = cn_apply_act(v0, act, slope), a1 = cn_apply_act(v1, act, slope);
                    *reinterpret_cast<f2*>(dst) = f2{a0, a1};
                    if (stats) { stat(0, v0, a0); stat(TN - 1, v1, a1); }
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
                    else {
                        const float vr = res ? v + res[(long)row * N + col] : v, a = cn_apply_act(vr, act, slope);
                        *dst = a;
                        if (stats) stat(j, vr, a);
                    }
                }
            }
        }
    if (stats && !split && !part_stride) {
        // the two half-waves hold the two row halves of the same columns; then one atomic per (column, sum) and wave
        const int sample = (m0 - (m0 / sper) * sper) / srows;        // (parity-ordered rows: class-major, samples inside a class)
        const int nsamp = sper / srows;                              // (NOT g.n: the plain 1x1 product is launched without a geometry)
        const int nk = stats_mode == 1 ? 2 : 4;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * 32 * TN + (BT ? 32 * j + l31 : TN * l31 + j);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k >= nk) break;
                const float t = sacc[j][k] + __shfl_xor(sacc[j][k], 32);
                if (half == 0 && col < N) unsafeAtomicAdd(stats + ((long)k * nsamp + sample) * N + col, t);
            }
        }
    }
}

template <int WM, int WN, int TM, int TN, bool BT, bool GATHER, int KB, int NS, int NP, bool BF>
__global__ __launch_bounds__(256 + 64 * NP) void fwd2_kernel(CnConvGeom g, const float* __restrict__ A, const float* __restrict__ B,
                                                   const float* __restrict__ bias, float* __restrict__ C, int M, int N, int K,
                                                   int act, float slope, int ntm, int ntn, long part_stride, int par,
                                                   const float* __restrict__ res, unsigned a_bytes, unsigned b_bytes, int flip,
                                                   float* __restrict__ stats, int stats_mode, float stats_slope, int srows, int sper) {
    fwd2_body<WM, WN, TM, TN, BT, GATHER, KB, NS, NP, BF>(g, A, B, bias, C, M, N, K, act, slope, ntm, ntn, part_stride, par, res, a_bytes,
                                                          b_bytes, flip, stats, stats_mode, stats_slope, srows, sper);
}

template <int WM, int WN, int TM, int TN, bool BT, bool GATHER, int KB, int NS, int NP, bool BF = false>
int launch2(const CnConvGeom& g, const float* A, const float* B, const float* bias, float* C, long M, int N, int K, int act, float slope,
            int splits, long part_stride, int par, hipStream_t s, const float* res, unsigned a_bytes, unsigned b_bytes, int flip = 1,
            float* stats = nullptr, int stats_mode = 0, float stats_slope = 0.f, int srows = 1, int sper = 1) {
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    const int ntm = cn_cdiv(M, BM), ntn = cn_cdiv(N, BN);
    dim3 grid((unsigned)(par ? ntm * ntn : 8 * cn_cdiv((long)ntm * ntn, 8)), 1, (unsigned)splits);
    hipLaunchKernelGGL((fwd2_kernel<WM, WN, TM, TN, BT, GATHER, KB, NS, NP, BF>), grid, dim3(256 + 64 * NP), 0, s, g, A, B, bias, C, (int)M, N, K, act,
                       slope, ntm, ntn, part_stride, par, res, a_bytes, b_bytes, flip, stats, stats_mode, stats_slope, srows, sper);
    CN_LAUNCH_CHECK();
    return CN_OK;
}

int g_fwd2_kb = 0;      // cn_conv_loop_select (tests / sweeps): 0 = per-tile default
int g_fwd2_ns = 0;
int g_fwd2_np = -1;     // loader waves: -1 = per-tile default

}  // namespace

void cn_fwd2_tune(int kb, int ns, int np) {
    g_fwd2_kb = kb;
    g_fwd2_ns = ns;
    g_fwd2_np = np;
}

// The contract stated in common.h: tile cfg 0 / 1 / 2 / 4, bt = B is the original filter [N][K] (data gradient), split-K
// protocol of igemm_fwd_kernel, gp = NULL for the plain 1x1 stride-1 product, else the geometry whose gather builds the rows (par:
// parity-ordered).  x_elems / w_elems: sizes of the two tensors (the buffer descriptors' ranges).
int cn_fwd2(const CnConvGeom* gp, int cfg, int bt, const float* A, const float* B, const float* bias, float* C, long M, int N, int K,
            int act, float slope, int splits, long part_stride, int par, hipStream_t s, const float* res, double x_elems, double w_elems,
            float* stats, int stats_mode, float stats_slope, int srows, int sper) {
    if ((res || stats) && (splits > 1 || part_stride)) return CN_EUNSUPPORTED;
    if (stats && bt) return CN_EUNSUPPORTED;
    if (K % 16 != 0 || N % 4 != 0 || M <= 0 || M > 0x7fffffffL || (par && !gp)) return CN_EUNSUPPORTED;
    if (gp && (gp->dl_d > 2 || gp->dl_h > 2 || gp->dl_w > 2)) return CN_EUNSUPPORTED;     // (the gather's shift-and-mask form)
    if (x_elems * 4.0 >= 2147483647.0 || w_elems * 4.0 >= 2147483647.0) return CN_EUNSUPPORTED;      // 32-bit byte offsets
    const unsigned ab = (unsigned)(x_elems * 4.0), bb = (unsigned)(w_elems * 4.0);
    static const CnConvGeom none = {};
    // Loop variant.  Defaults: 16-deep stages, FOUR stages, no loader waves.  How they were arrived at (all three are decided by
    // alternating bench.py runs on one box, scripts/dev/pipe_ab.sh -- isolated timings of one shape say something else for each):
    //  * stages: in isolation three stages (24 - 48 KB: more workgroups per CU) win for the big tiles and for >= 2048 workgroups
    //    (64 x 64: 388 vs 418 us at M = 1 310 720) and four for small launches (84 vs 111 us at M = 8 192 with two K slices); in the
    //    pipelined iteration, where other lines' kernels share the CUs, that rule measured 396.0 against 398.5 images/s for four everywhere;
    //  * loader waves (NP = 2) for 64 x 64 launches of <= 768 workgroups: 52 -> 48.5 us on the ablation shape, 266 -> 303 us at 8 192
    //    workgroups; in the pipelined iteration 392.4 -> 394.8 images/s fp32 and 711 -> 719 bf16 WITHOUT them;
    //  * 32-deep stages (half the steps and barriers; 64 x 64 tile only): +-0 per shape, +0.6 % on the iteration while the small
    //    launches ran with loader waves, 396.7 against 397.0 once they did not; 32 for every launch loses 1.5 %.  (bf16: cn_fwd2_bf16.)
    // g_fwd2_kb / _ns / _np: 0 / 0 / -1 = these defaults, else forced by cn_conv_loop_select.
    int kb = g_fwd2_kb ? g_fwd2_kb : 16;
    int ns = g_fwd2_ns ? g_fwd2_ns : 4;
    if (kb == 32 && (K % 32 != 0 || cfg != 2)) kb = 16;            // 32-deep stages: the 64 x 64 tile only
    if (kb == 32) ns = 3;
    int np = g_fwd2_np >= 0 ? g_fwd2_np : 0;
    if (cfg != 2 || kb == 32) np = 0;
#define L3(WM, WN, TM, TN, KB_, NS_, NP_)                                                                                                          \
    return gp ? (bt ? launch2<WM, WN, TM, TN, true, true, KB_, NS_, NP_>(*gp, A, B, bias, C, M, N, K, act, slope, splits, part_stride, par, s, res, ab, bb, 1, stats, stats_mode, stats_slope, srows, sper)   \
                    : launch2<WM, WN, TM, TN, false, true, KB_, NS_, NP_>(*gp, A, B, bias, C, M, N, K, act, slope, splits, part_stride, par, s, res, ab, bb, 1, stats, stats_mode, stats_slope, srows, sper)) \
              : (bt ? launch2<WM, WN, TM, TN, true, false, KB_, NS_, NP_>(none, A, B, bias, C, M, N, K, act, slope, splits, part_stride, 0, s, res, ab, bb, 1, stats, stats_mode, stats_slope, srows, sper)   \
                    : launch2<WM, WN, TM, TN, false, false, KB_, NS_, NP_>(none, A, B, bias, C, M, N, K, act, slope, splits, part_stride, 0, s, res, ab, bb, 1, stats, stats_mode, stats_slope, srows, sper))
#define L2(WM, WN, TM, TN)                 \
    if (ns == 3) { L3(WM, WN, TM, TN, 16, 3, 0); } \
    else { L3(WM, WN, TM, TN, 16, 4, 0); }
    switch (cfg) {
        case 0: L2(2, 2, 2, 2);
        case 1: L2(2, 2, 2, 1);
        case 2:
            if (kb == 32) { L3(2, 2, 1, 1, 32, 3, 0); }
            if (np == 1) { if (ns == 3) { L3(2, 2, 1, 1, 16, 3, 1); } else { L3(2, 2, 1, 1, 16, 4, 1); } }
            if (np == 2) { if (ns == 3) { L3(2, 2, 1, 1, 16, 3, 2); } else { L3(2, 2, 1, 1, 16, 4, 2); } }
            L2(2, 2, 1, 1);
        case 3: L2(4, 1, 1, 1);
        case 4: L2(4, 1, 1, 3);
        default: return CN_EUNSUPPORTED;
    }
#undef L2
#undef L3
}

// The bf16 family on the same loop (igemm_bf16.hip's contract for the forward / data-gradient GEMM: x and y bf16, wb = the prepared
// filter copy [tap][N][K] with K = g.cin contiguous, flip = walk the taps backwards).  tile cfg as above; np as cn_fwd2's loader
// waves.  CN_EUNSUPPORTED (nothing launched) unless K % 32 == 0 and N % 8 == 0.
int cn_fwd2_bf16(const CnConvGeom& g, int cfg, int flip, const void* x, const void* wb, const float* bias, void* y, int act, float slope,
                 int par, hipStream_t s) {
    const long M = (long)g.n * g.out_d * g.out_h * g.out_w;
    const int N = g.cout, K = g.cin;
    if (K % 32 != 0 || N % 8 != 0 || M <= 0 || M > 0x7fffffffL) return CN_EUNSUPPORTED;
    if (g.dl_d > 2 || g.dl_h > 2 || g.dl_w > 2) return CN_EUNSUPPORTED;
    const double xe = (double)g.n * g.in_d * g.in_h * g.in_w * g.cin, we = (double)g.k_d * g.k_h * g.k_w * g.cin * g.cout;
    if (xe * 2.0 >= 2147483647.0 || we * 2.0 >= 2147483647.0) return CN_EUNSUPPORTED;
    const unsigned ab = (unsigned)(xe * 2.0), bb = (unsigned)(we * 2.0);
    const float* A = reinterpret_cast<const float*>(x);
    const float* B = reinterpret_cast<const float*>(wb);
    float* C = reinterpret_cast<float*>(y);
    const long wgs = (long)cn_cdiv(M, cfg == 2 ? 64 : 128) * cn_cdiv(N, cfg == 0 ? 128 : cfg == 4 ? 96 : 64);
    // loader waves: none by default (in isolation they pay where a CU holds at most ~three workgroups -- 34 -> 28 us at 256
    // workgroups, 105 -> 139 us at 4 096 --; in the pipelined iteration 711 -> 719 images/s without them: see cn_fwd2)
    const int np = g_fwd2_np >= 0 ? g_fwd2_np : 0;
    const int ns = g_fwd2_ns ? g_fwd2_ns : ((cfg == 0 || cfg == 4 || wgs >= 2048) ? 3 : 4);
#define LB(WM, WN, TM, TN, NS_, NP_) \
    return launch2<WM, WN, TM, TN, true, true, 16, NS_, NP_, true>(g, A, B, bias, C, M, N, K, act, slope, 1, 0, par, s, nullptr, ab, bb, flip)
#define LB2(WM, WN, TM, TN, NP_)   \
    if (ns == 3) { LB(WM, WN, TM, TN, 3, NP_); } \
    else { LB(WM, WN, TM, TN, 4, NP_); }
    // 32-deep stages (64 bf16 elements): half the steps, hence half the barriers, of a loop whose step holds only two 32-cycle MFMAs
    // per wave -- for launches of at most ~four workgroups per CU on the 64- and 128 x 64 tiles (M = 4 096, K = 2 304, N = 256:
    // 26 -> 22 us; 4 096 x 256 x 1 024: 25 -> 19; 8 192 x 4 608 x 512: 65 -> 55).  Larger launches lose (48 / 72 KB of LDS per
    // workgroup: 524 288 x 576 x 64 84 -> 114 us), so does the 128 x 128 tile (96 KB) and a single-step reduction
    // (scripts/dev/bf16_kb_ab.sh).
    constexpr int kb_env = 32;
    constexpr long kb_wgs = 1024;
#define LB32(WM, WN, TM, TN) \
    return launch2<WM, WN, TM, TN, true, true, 32, 3, 0, true>(g, A, B, bias, C, M, N, K, act, slope, 1, 0, par, s, nullptr, ab, bb, flip)
    if (kb_env == 32 && K % 64 == 0 && wgs <= kb_wgs && (long)K * g.k_d * g.k_h * g.k_w >= 128) {
        switch (cfg) {
            case 1: LB32(2, 2, 2, 1);
            case 2: LB32(2, 2, 1, 1);
            default: break;
        }
    }
#undef LB32
    switch (cfg) {
        case 0: if (np) { LB2(2, 2, 2, 2, 2); } LB2(2, 2, 2, 2, 0);
        case 1: if (np) { LB2(2, 2, 2, 1, 2); } LB2(2, 2, 2, 1, 0);
        case 2: if (np) { LB2(2, 2, 1, 1, 2); } LB2(2, 2, 1, 1, 0);
        case 4: if (np) { LB2(4, 1, 1, 3, 2); } LB2(4, 1, 1, 3, 0);
        default: return CN_EUNSUPPORTED;
    }
#undef LB2
#undef LB
}
