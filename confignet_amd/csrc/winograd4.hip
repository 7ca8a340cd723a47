// winograd4.hip -- 3x3 stride-1 SAME convolutions as Winograd F(4x4, 3x3) on the fp32 matrix cores (round 4).
//
// F(4x4, 3x3) computes a 4x4 output tile from a 6x6 input patch with 36 multiplies per (ci, co) instead of 144:
//     Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A            (Lavin & Gray, interpolation points 0, +-1, +-2, inf)
// i.e. 36 independent GEMMs  M_p[tile, co] = sum_ci V_p[tile, ci] U_p[ci, co]  -- 2.25 MFMA products per output pixel and
// (ci, co) where F(2x2, 3x3) (winograd.hip) issues 4 and the direct implicit GEMM 9.  Used where the layer is large enough to
// fill the chip with its bigger blocks (VGG-19 conv1_2 .. conv3_4 of the perceptual loss, perceptual_loss.py:19-41: 6.6 of
// the 8.3 ms of F(2x2) time per iteration); everything else stays on F(2x2).  Exact in real arithmetic; in fp32 the larger
// transform coefficients (up to 8 / 5 / 1/24) cost about one more digit than F(2x2) -- held to the same 2e-4 bar against the
// float64 oracle (tests/test_ops_gpu.py).
//
// One workgroup = 12 waves = a block of 4 x 8 tiles (16 x 32 output pixels) x 64 output channels.  Wave w owns row i = w % 6 of
// the 6 x 6 transform domain for the channel half cb = w / 6: six 32 x 32 accumulators (96 registers; three waves per SIMD), so
// the row pass of the output transform is per-lane arithmetic and only its column pass goes through LDS.
//
// Measured (round 4, scripts/dev/wino_bench.py, profiles/round4_winograd_f4x4.txt): 1.25 - 1.35 x faster than F(2x2) on the VGG-19
// layers it takes, at 0.33 - 0.47 of the fp32 MFMA peak on the work it issues.  Ablation on conv3_x (141 us): multiply alone 71 us,
// transform alone 23 us, loads alone 52 us, skeleton (launch, prologue, 64 barriers, epilogue) 41 us -- the step's parts run back
// to back more than side by side.  Tried and dropped: register-blocked multiply with 8-byte filter reads + partial-sum epilogue
// (accumulator spills: 1.4 - 2.3 x slower), one fused instruction stream of MFMAs and transform with hand-counted LDS waits
// (spills again under the 168-register cap of 3 waves per SIMD: 4.9 x slower), K walked from a rotated start per block (no change).
// K (= cin) is walked 4 channels at a time:
//   * the filter slice U[p][4][64] of all 36 positions and the raw (16+2) x (32+2)-pixel input block (8 channels = two steps
//     at a time: one 32-byte sector per pixel) arrive by LDS-DMA (`buffer_load ... lds`, per-lane source offsets; an offset
//     beyond the descriptor's range reads as zero = the image border costs nothing), one step ahead of their use;
//   * every wave transforms its row i of the next step's patches (B^T d B; lane = (tile, channel), the wave's row is uniform,
//     so the row pass is one fused multiply-add chain with wave-uniform coefficients) into the k-major V planes;
//   * every wave multiplies its six positions of the current step: 12 MFMAs, operands straight from LDS.
// The DMA destinations (U / raw stages) are only ever read through inline-asm ds_reads inside the loop: the compiler cannot
// prove a ds_read of one stage disjoint from the DMA destination of the other and would drain every outstanding load first
// (winograd.hip found the same; there the two input buffers are separate objects).
#include "common.h"

#include "mma_tile.h"

namespace {

typedef __attribute__((address_space(3))) float lds_float4;
typedef float v2f __attribute__((ext_vector_type(2)));

struct Wino4Geom {
    int n, h, w, cin, cout, bh, bw;     // bh x bw blocks of 16 x 32 output pixels per image
};

// U[p][k][n] = (G g G^T)[p], G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1], of
// g = w[., ., ci, co] (forward: k = ci, n = co) or of the flipped, channel-swapped filter (data gradient: k = co, n = ci).
__global__ void wino4_filter_kernel(const float* __restrict__ W, float* __restrict__ U, int cin, int cout, int dgrad) {
    const long total = (long)cin * cout;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int co = (int)(e % cout), ci = (int)(e / cout);
        float g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
                g[a][b] = dgrad ? W[((long)((2 - a) * 3 + (2 - b)) * cin + ci) * cout + co] : W[((long)(a * 3 + b) * cin + ci) * cout + co];
        float t[6][3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const float g0 = g[0][b], g1 = g[1][b], g2 = g[2][b];
            t[0][b] = 0.25f * g0;
            t[1][b] = (-1.f / 6.f) * (g0 + g1 + g2);
            t[2][b] = (-1.f / 6.f) * (g0 - g1 + g2);
            t[3][b] = (1.f / 24.f) * g0 + (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
            t[4][b] = (1.f / 24.f) * g0 - (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
            t[5][b] = g2;
        }
        const long k = dgrad ? co : ci, nn = dgrad ? ci : co;
        const long kdim = dgrad ? cout : cin, ndim = dgrad ? cin : cout;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const float t0 = t[a][0], t1 = t[a][1], t2 = t[a][2];
            float u[6];
            u[0] = 0.25f * t0;
            u[1] = (-1.f / 6.f) * (t0 + t1 + t2);
            u[2] = (-1.f / 6.f) * (t0 - t1 + t2);
            u[3] = (1.f / 24.f) * t0 + (1.f / 12.f) * t1 + (1.f / 6.f) * t2;
            u[4] = (1.f / 24.f) * t0 - (1.f / 12.f) * t1 + (1.f / 6.f) * t2;
            u[5] = t2;
#pragma unroll
            for (int b = 0; b < 6; ++b) U[((long)(a * 6 + b) * kdim + k) * ndim + nn] = u[b];
        }
    }
}

#ifndef W4_ABLATE
#define W4_ABLATE 0      // dev (scripts/dev/wino4_ablate.sh): 1 no MFMA, 2 no input transform, 4 no loads in the loop, 8 no output transform / stores, 16 one K step
#endif
constexpr int W4_KC = 4;                 // channels per K step
constexpr int W4_T = 32;                 // tiles per workgroup (4 rows x 8 columns of 4 x 4 output pixels)
constexpr int W4_C = 64;                 // output channels per workgroup
constexpr int W4_VP = 40;                // tile pitch of a V row: 40 = 8 mod 32, the transform's (tile, channel) stores hit 32 banks
constexpr int W4_U = 36 * W4_KC * W4_C;  // floats of one filter stage (36 wave-instruction pieces of 1 KB)
constexpr int W4_RAW_PIECES = 21;        // raw stage: 18 rows x 36 slot columns x 8 channels = 20.25 KB, rounded up
constexpr int W4_RAW = W4_RAW_PIECES * 256;
constexpr int W4_BUF = 2 * W4_U + 2 * W4_RAW;     // floats: 116.7 KB (the epilogue's exchange buffer needs 96 KB of it)


__global__ __launch_bounds__(768) void wino4_fwd_kernel(Wino4Geom g, const float* __restrict__ X, const float* __restrict__ U,
                                                        const float* __restrict__ bias, float* __restrict__ Y, int act, float slope) {
    __shared__ __attribute__((aligned(16))) float Vs[2][36][W4_KC][W4_VP];      // V planes, k-major: [stage][position][k][tile]
    extern __shared__ __attribute__((aligned(16))) float BUF[];                 // U0 | U1 | RAW0 | RAW1 (DMA destinations)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wi = wave % 6, cb = wave / 6;         // position row / channel half of this wave's accumulators

    // workgroup -> (block of tiles, 64 output channels); consecutive workgroups (same channel block, neighbouring pixels) share an XCD
    const int nblk = g.n * g.bh * g.bw, total = nblk * (g.cout / W4_C);
    int wg = blockIdx.x;
    if (total % 8 == 0) wg = (wg & 7) * (total >> 3) + (wg >> 3);
    const int blk = wg % nblk, co0 = (wg / nblk) * W4_C;
    int img, brem, by, bx;
    divmod_pos(blk, g.bh * g.bw, img, brem);
    divmod_pos(brem, g.bw, by, bx);

    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, g.n * g.h * g.w * g.cin * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ures = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(U), 0, 36 * g.cin * g.cout * 4, 0x00020000);

    // filter pieces of this wave: p = wave + 12 j; lane -> k row (lane >> 4), float4 column (lane & 15); step advance in an SGPR
    unsigned uoff[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int p = wave + 12 * j;
        uoff[j] = (unsigned)((p * g.cin + (lane >> 4)) * g.cout + co0 + (lane & 15) * 4) * 4u;
    }
    // raw pieces: q = wave + 12 j (< 21); lane -> slot (q * 64 + lane) >> 1, channel half (lane & 1).  Slot of block pixel
    // (r, c), r < 18, c < 34:  (r * 4 + (c & 3)) * 9 + (c >> 2) -- the pixels one tile column apart are neighbours, so the
    // transform's 8-byte reads of a wave spread over the banks.
    unsigned xoff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = wave + 12 * j, sl = (q * 64 + lane) >> 1;
        const int r = sl / 36, rem = sl - r * 36, cm = rem / 9, c = (rem - cm * 9) * 4 + cm;
        const int yy = by * 16 - 1 + r, xx = bx * 32 - 1 + c;
        const bool in = q < W4_RAW_PIECES && r < 18 && c < 34 && yy >= 0 && yy < g.h && xx >= 0 && xx < g.w;
        // (in pixel rows with bit 2 set the two 4-channel halves of a slot are SWAPPED: the transform's reads of two tile rows,
        // 4 pixel rows apart, then fall into different banks -- a 32-lane group of 4 channels x 4 tile columns x 2 tile rows
        // covers all 32 banks)
        xoff[j] = in ? (unsigned)(((img * g.h + yy) * g.w + xx) * g.cin + (((lane & 1) ^ (r >> 2)) & 1) * 4) * 4u : 0x80000000u;
    }
    auto issue_u = [&](int s) {            // filter slice of K step s -> U stage s & 1
        float* dst = BUF + (s & 1) * W4_U;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ures, (lds_float4*)(dst + (wave + 12 * j) * 256), 16, uoff[j], s * W4_KC * g.cout * 4, 0, 0);
    };
    auto issue_raw = [&](int a) {          // input block, channels 8 a .. 8 a + 7 -> raw stage a & 1
        float* dst = BUF + 2 * W4_U + (a & 1) * W4_RAW;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (wave + 12 * j < W4_RAW_PIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (lds_float4*)(dst + (wave + 12 * j) * 256), 16, xoff[j], a * 8 * 4, 0, 0);
    };

    const unsigned buf0 = (unsigned)(unsigned long long)(lds_float4*)BUF;

    // ---- input transform of K step s: V = B^T d B for row i = wave % 6 of the transform domain ---------------------
    // B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]; row i reads patch rows
    // r_q with coefficients c_q (wave-uniform): i = 0: (0, 2, 4 | 4, -5, 1), i = 5: (1, 3, 5 | 4, -5, 1), else rows 1..4.
    // lane = (tile, channel): tile (wave / 6) * 16 + lane / 4, channel lane % 4 of the step; all 12 waves share the work.
    // lane = (channel tk, tile column txl + 4 txh, tile row 2 (wave / 6) + tyl); the tile's index in the V planes (= MFMA row) is
    // tt = txl + 4 tyl + 8 txh + 16 (wave / 6): a 32-lane group's stores (8 tiles x 4 channel rows of pitch 40) hit 32 banks.
    const int tk = lane & 3, txl = (lane >> 2) & 3, tyl = (lane >> 4) & 1, txh = lane >> 5;
    const int ttx = txl + 4 * txh, tty = 2 * (wave / 6) + tyl, tt = txl + 4 * tyl + 8 * txh + 16 * (wave / 6);
    int trow[4];
    float tco[4];
    {
        const int i = wi;
        trow[0] = i == 0 ? 0 : 1;
        trow[1] = i == 0 ? 2 : (i == 5 ? 3 : 2);
        trow[2] = i == 0 ? 4 : (i == 5 ? 5 : 3);
        trow[3] = 4;
        tco[0] = i == 0 ? 4.f : i == 1 ? -4.f : i == 2 ? 4.f : i == 3 ? -2.f : i == 4 ? 2.f : 4.f;
        tco[1] = i == 0 ? -5.f : i == 1 ? -4.f : i == 2 ? -4.f : i == 3 ? -1.f : i == 4 ? -1.f : -5.f;
        tco[2] = i == 0 ? 1.f : i == 1 ? 1.f : i == 2 ? -1.f : i == 3 ? 2.f : i == 4 ? -2.f : 1.f;
        tco[3] = (i == 0 || i == 5) ? 0.f : 1.f;
    }
    // byte address of patch element (row q of this wave's row list, column j) of the lane's tile and channel, less the stage /
    // sub-step term: a per-lane base per row (4 registers) + an immediate for the column
    unsigned rbase[4], rsw[4];                       // rsw: the row's half-swap bit << 4
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 4 * tty + trow[q];
        rbase[q] = buf0 + 4u * (unsigned)(2 * W4_U) + 32u * (unsigned)(r * 36 + ttx) + 4u * (unsigned)tk;
        rsw[q] = (unsigned)((r >> 2) & 1) << 4;
    }
#define W4_RD(q, j, dst) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(ra[q]), "i"((((j) & 3) * 9 + ((j) >> 2)) * 32))
    auto transform = [&](int s) {
        const unsigned so = 4u * (unsigned)(((s >> 1) & 1) * W4_RAW), sub = (unsigned)(s & 1) << 4;
        unsigned ra[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) ra[q] = rbase[q] + so + (sub ^ rsw[q]);
        // Round 6: the 24 patch elements of the lane's row are requested ahead of their use, a column at a time, and a column pair is
        // reduced as soon as ITS reads have retired (LDS reads retire in order) while the younger columns are still in flight --
        // one exposed LDS round trip per step instead of three back to back (round-5 form: read 8, wait, reduce, read 8, ...).
        // Never more than 12 reads outstanding (the counter holds 15).  Measured: -1 % (179.5 against 180.8 us on conv1_2, 138.8
        // against 140.2 on conv3_x): with three waves per SIMD another wave already fills a wave's LDS wait; what bounds the
        // step is operand delivery from L2 (profiles/round5_wino4_ablation.txt), not latency inside a wave.
        float d[6][4], t[6];
#define W4_COL(c) W4_RD(0, c, d[c][0]); W4_RD(1, c, d[c][1]); W4_RD(2, c, d[c][2]); W4_RD(3, c, d[c][3]);
#define W4_WAIT(pending) asm volatile("s_waitcnt lgkmcnt(" #pending ")"); __builtin_amdgcn_sched_barrier(0);
#define W4_RED(j0, j1)                                                                                                         \
        t[j0] = tco[0] * d[j0][0] + tco[1] * d[j0][1] + tco[2] * d[j0][2] + tco[3] * d[j0][3];                                 \
        t[j1] = tco[0] * d[j1][0] + tco[1] * d[j1][1] + tco[2] * d[j1][2] + tco[3] * d[j1][3];                                 \
        __builtin_amdgcn_sched_barrier(0);
        W4_COL(0) W4_COL(1) W4_COL(2)
        W4_WAIT(4)                 // columns 0, 1 are back (column 2 may be in flight)
        W4_COL(3) W4_COL(4)
        __builtin_amdgcn_sched_barrier(0);
        W4_RED(0, 1)
        W4_WAIT(4)                 // columns 2, 3 (column 4 may be in flight)
        W4_COL(5)
        __builtin_amdgcn_sched_barrier(0);
        W4_RED(2, 3)
        W4_WAIT(0)
        W4_RED(4, 5)
#undef W4_WAIT
#undef W4_RED
#undef W4_COL
        float* vdst = &Vs[s & 1][wi * 6][tk][tt];
        constexpr int VPL = W4_KC * W4_VP;          // floats between two positions
        vdst[0 * VPL] = 4.f * t[0] - 5.f * t[2] + t[4];
        vdst[1 * VPL] = -4.f * (t[1] + t[2]) + t[3] + t[4];
        vdst[2 * VPL] = 4.f * (t[1] - t[2]) - t[3] + t[4];
        vdst[3 * VPL] = -2.f * t[1] - t[2] + 2.f * t[3] + t[4];
        vdst[4 * VPL] = 2.f * t[1] - t[2] - 2.f * t[3] + t[4];
        vdst[5 * VPL] = 4.f * t[1] - 5.f * t[3] + t[5];
    };

    f32x16 acc[6];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // ---- multiply of K step s: the wave's six positions x two k pairs = 12 MFMAs.  Round 6: every operand read is inline asm with an
    // immediate offset, the 24 reads run as a rolling window in front of the MFMAs (pair j of k pair 0 is waited for with 12 younger
    // reads in flight; the reads of k pair 1 are issued one pair behind each MFMA of k pair 0) -- the round-5 form read 12, waited for
    // all of them, multiplied 6, and did that twice per step.  Never more than 14 reads outstanding (the counter holds 15).
#define W4_RDU(j, kk, dst) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(ub), "i"((((j) * W4_KC + 2 * (kk)) * W4_C) * 4))
#define W4_RDA(j, kk, dst) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(vb), "i"((((j) * W4_KC + 2 * (kk)) * W4_VP) * 4))
    const unsigned ubase = buf0 + 4u * (unsigned)((wi * 6 * W4_KC + half) * W4_C + cb * 32 + l31);
    const unsigned vbase = (unsigned)(unsigned long long)(lds_float4*)&Vs[0][wi * 6][half][l31];
    auto multiply = [&](int s) {
        const unsigned ub = ubase + 4u * (unsigned)((s & 1) * W4_U);
        const unsigned vb = vbase + 4u * (unsigned)((s & 1) * 36 * W4_KC * W4_VP);
        float a[2][6], b[2][6];
#define W4_PAIR(j, kk) W4_RDA(j, kk, a[kk][j]); W4_RDU(j, kk, b[kk][j]);
#define W4_MFMA(j, kk)                                                                                                         \
        if (!(W4_ABLATE & 1)) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][j], b[kk][j], acc[j], 0, 0, 0);              \
        else acc[j][0] += a[kk][j] * b[kk][j];                                                                                 \
        __builtin_amdgcn_sched_barrier(0);
#define W4_WAIT(pending) asm volatile("s_waitcnt lgkmcnt(" #pending ")"); __builtin_amdgcn_sched_barrier(0);
        W4_PAIR(0, 0) W4_PAIR(1, 0) W4_PAIR(2, 0) W4_PAIR(3, 0) W4_PAIR(4, 0) W4_PAIR(5, 0) W4_PAIR(0, 1)
        W4_WAIT(12) W4_MFMA(0, 0) W4_PAIR(1, 1)
        W4_WAIT(12) W4_MFMA(1, 0) W4_PAIR(2, 1)
        W4_WAIT(12) W4_MFMA(2, 0) W4_PAIR(3, 1)
        W4_WAIT(12) W4_MFMA(3, 0) W4_PAIR(4, 1)
        W4_WAIT(12) W4_MFMA(4, 0) W4_PAIR(5, 1)
        W4_WAIT(12) W4_MFMA(5, 0)
        W4_WAIT(10) W4_MFMA(0, 1)
        W4_WAIT(8) W4_MFMA(1, 1)
        W4_WAIT(6) W4_MFMA(2, 1)
        W4_WAIT(4) W4_MFMA(3, 1)
        W4_WAIT(2) W4_MFMA(4, 1)
        W4_WAIT(0) W4_MFMA(5, 1)
#undef W4_WAIT
#undef W4_MFMA
#undef W4_PAIR
    };

    const int nks = (W4_ABLATE & 16) ? 1 : g.cin / W4_KC, nraw = g.cin / 8;
    issue_raw(0);
    issue_u(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
    __syncthreads();
    if (nraw > 1) issue_raw(1);
    if (nks > 1) issue_u(1);
    transform(0);
    // two of a SIMD's three waves (w, w + 4, w + 8 share one) multiply first and transform afterwards, the third the other way
    // round: right after the barrier one wave's transform runs in the shadow of the other waves' MFMAs
    const bool mul_first = ((wave >> 2) & 1) == 0;
    for (int s = 0; s < nks; ++s) {
        __builtin_amdgcn_s_waitcnt(0x0F70);      // the loads issued a step ago have landed ...
        __syncthreads();                         // ... everybody's; everybody is past step s-1 and V(s) is written
        if (s >= 1 && !(W4_ABLATE & 4)) {
            if (s + 1 < nks) issue_u(s + 1);
            if ((s & 1) && (s + 3) / 2 < nraw) issue_raw((s + 3) / 2);
        }
#pragma nounroll
        for (int ph = 0; ph < 2; ++ph) {             // (a loop so that each body exists once: two inlined copies spilled accumulators)
            if ((ph == 0) == mul_first) multiply(s);
            else if (s + 1 < nks && !(W4_ABLATE & 2)) transform(s + 1);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();

    // ---- output transform Y = A^T M A, A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1].  The row pass (over
    // the wave's six positions j) is per-lane arithmetic; its results R[i][y][tile][channel] of one channel half go through LDS
    // (96 KB over the now idle stages), the column pass (over i) is thread-parallel over (tile, channel, y).
    float* E = BUF;
    if (W4_ABLATE & 8) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) t += acc[j][r];
        if (t == 12345.f) Y[tid] = t;
        return;
    }
    for (int rb = 0; rb < 2; ++rb) {
        if (cb == rb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int tl = 4 * half + (e & 3) + 8 * (e >> 2);
                const float a0 = acc[0][e], a1 = acc[1][e], a2 = acc[2][e], a3 = acc[3][e], a4 = acc[4][e], a5 = acc[5][e];
                const float s12 = a1 + a2, d12 = a1 - a2, s34 = a3 + a4, d34 = a3 - a4;
                float* dst = E + ((wi * 4) * 32 + tl) * 32 + l31;
                dst[0 * 1024] = a0 + s12 + s34;
                dst[1 * 1024] = d12 + 2.f * d34;
                dst[2 * 1024] = s12 + 4.f * s34;
                dst[3 * 1024] = d12 + 8.f * d34 + a5;
            }
        }
        __syncthreads();
        // column pass: a thread owns 4 consecutive channels of one (tile, y): 16-byte LDS reads and 16-byte stores
        for (int idx = tid; idx < 1024; idx += 768) {
            const int c4 = idx & 7, tl = (idx >> 3) & 31, y = idx >> 8;
            float4 m[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) m[i] = *reinterpret_cast<const float4*>(E + ((i * 4 + y) * 32 + tl) * 32 + 4 * c4);
            const int co = co0 + rb * 32 + 4 * c4;
            const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
            const int otx = (tl & 3) + 4 * ((tl >> 3) & 1), oty = ((tl >> 2) & 1) + 2 * (tl >> 4);     // (the V planes' tile order)
            const int row0 = by * 16 + 4 * oty, col = bx * 32 + 4 * otx + y;
#define W4_O(f)                                                                                                          \
            const float s12##f = m[1].f + m[2].f, d12##f = m[1].f - m[2].f, s34##f = m[3].f + m[4].f, d34##f = m[3].f - m[4].f; \
            const float o0##f = m[0].f + s12##f + s34##f + bv.f, o1##f = d12##f + 2.f * d34##f + bv.f,                     \
                        o2##f = s12##f + 4.f * s34##f + bv.f, o3##f = d12##f + 8.f * d34##f + m[5].f + bv.f;
            W4_O(x) W4_O(y) W4_O(z) W4_O(w)
#undef W4_O
            float* dst = Y + ((long)(img * g.h + row0) * g.w + col) * g.cout + co;
            const long rs = (long)g.w * g.cout;
            *reinterpret_cast<float4*>(dst) = make_float4(cn_apply_act(o0x, act, slope), cn_apply_act(o0y, act, slope), cn_apply_act(o0z, act, slope), cn_apply_act(o0w, act, slope));
            *reinterpret_cast<float4*>(dst + rs) = make_float4(cn_apply_act(o1x, act, slope), cn_apply_act(o1y, act, slope), cn_apply_act(o1z, act, slope), cn_apply_act(o1w, act, slope));
            *reinterpret_cast<float4*>(dst + 2 * rs) = make_float4(cn_apply_act(o2x, act, slope), cn_apply_act(o2y, act, slope), cn_apply_act(o2z, act, slope), cn_apply_act(o2w, act, slope));
            *reinterpret_cast<float4*>(dst + 3 * rs) = make_float4(cn_apply_act(o3x, act, slope), cn_apply_act(o3y, act, slope), cn_apply_act(o3z, act, slope), cn_apply_act(o3w, act, slope));
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int cn_conv_wino4_filter(const float* w, float* u, int cin, int cout, int dgrad, void* stream) {
    CN_CHECK_ARG(w && u && cin > 0 && cout > 0, "wino4_filter: bad args");
    const long total = (long)cin * cout;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(wino4_filter_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, u, cin, cout, dgrad);
    CN_LAUNCH_CHECK();
    return CN_OK;
}

// x (n, h, w, cin) -> y (n, h, w, cout): 3x3, stride 1, SAME; u from cn_conv_wino4_filter ([36][cin][cout]).
// Returns CN_EUNSUPPORTED (nothing launched) unless h % 16 == 0, w % 32 == 0, cin % 16 == 0 and cout % 64 == 0.
extern "C" int cn_conv_fwd_wino4(int n, int h, int w, int cin, int cout, const float* x, const float* u, const float* bias,
                                 float* y, int act, float slope, void* stream) {
    CN_CHECK_ARG(x && u && y && n > 0 && h > 0 && w > 0, "conv_fwd_wino4: bad args");
    if (h % 16 || w % 32 || cin % 16 || cout % W4_C) return CN_EUNSUPPORTED;
    CN_CHECK_ARG((double)n * h * w * (cin > cout ? cin : cout) * 4.0 < 2147483647.0 && 144.0 * cin * cout < 2147483647.0,
                 "tensor exceeds 2^31 bytes (buffer descriptors, 32-bit offsets)");
    Wino4Geom g{n, h, w, cin, cout, h / 16, w / 32};
    const long nblk = (long)n * g.bh * g.bw;
    constexpr size_t lds = sizeof(float) * W4_BUF;           // + 36 KB static (V planes)
    static bool attr_set = false;
    if (!attr_set) {
        CN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wino4_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipStream_t s = (hipStream_t)stream;
    // MFMA work that contributes to the result: 36 products per 4x4 output tile and (ci, co) pair
    cn_prof_begin(s, 2.0 * 36.0 * (double)nblk * W4_T * cin * cout, 4.0 * ((double)n * h * w * (cin + cout) + 36.0 * cin * cout), CN_FAM_WINO);
    hipLaunchKernelGGL(wino4_fwd_kernel, dim3((unsigned)(nblk * (cout / W4_C))), dim3(768), lds, s, g, x, u, bias, y, act, slope);
    cn_prof_end(s);
    CN_LAUNCH_CHECK();
    return CN_OK;
}
