// winograd.hip -- 3x3 stride-1 SAME convolutions (forward and data gradient) as Winograd F(2x2, 3x3) on the fp32 matrix cores.
//
// More than a third of the iteration's convolution time is 2-D 3x3 stride-1 layers with wide channels (the VGG-19
// perceptual stack: 4 forward + 2 data-gradient passes per generator step; the 3x3 convolutions of ResNet-50).  F(2x2,3x3)
// computes a 2x2 output tile from a 4x4 input patch with 16 multiplies per (ci, co) instead of 36:
//     Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A
// i.e. 16 independent GEMMs  M_p[tile, co] = sum_ci V_p[tile, ci] U_p[ci, co]  (p = position in the 4x4 transform domain)
// -- 4/9 of the MFMA work of the direct implicit GEMM, exact in real arithmetic (fp32: a few more roundings, ~1e-6 relative).
//
// One workgroup = 64 tiles x 64 output channels; each of its 4 waves owns a 32 x 32 block of ALL 16 positions
// (16 accumulators of v_mfma_f32_32x32x2_f32 = the whole AGPR file), so the output transform A^T M A is per-lane
// arithmetic on registers.  K (= cin) is walked in steps of 8: every thread gathers the 4x4 patch of one (tile, channel
// pair) (8-byte loads, 4 lanes = one 32-byte sector), transforms it in registers and writes the 16 V planes to LDS
// (k-major, like igemm_conv.hip); the filter U[p][ci][co] is transformed once per weight update (cn_conv_wino_filter).
// A step is 64 MFMAs per wave (4096 cycles): the next step's global loads are issued before them and have that long to land.
#include "common.h"

#include "mma_tile.h"

namespace {

constexpr int WKB = 8;      // input channels per step
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct WinoGeom {
    int n, h, w, cin, cout, th, tw, bh, bw;     // th = ceil(h/2), tw = ceil(w/2) tiles per image, in bh x bw blocks of 8 x 8
};

// U[p][ci][co] = (G g G^T)[p] of g = w[.,.,ci,co] (forward) or of the flipped, channel-swapped filter (data gradient:
// the "input channels" of that convolution are the forward cout): U[p][k][n] with k = co, n = ci.
__global__ void wino_filter_kernel(const float* __restrict__ W, float* __restrict__ U, int cin, int cout, int dgrad) {
    const long total = (long)cin * cout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % cout), ci = (int)(i / cout);
        float g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b)
                g[a][b] = dgrad ? W[((long)((2 - a) * 3 + (2 - b)) * cin + ci) * cout + co] : W[((long)(a * 3 + b) * cin + ci) * cout + co];
        float t[4][3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            t[0][b] = g[0][b];
            t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
            t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
            t[3][b] = g[2][b];
        }
        const long k = dgrad ? co : ci, nn = dgrad ? ci : co;
        const long kdim = dgrad ? cout : cin, ndim = dgrad ? cin : cout;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]), u3 = t[a][2];
            U[((long)(a * 4 + 0) * kdim + k) * ndim + nn] = u0;
            U[((long)(a * 4 + 1) * kdim + k) * ndim + nn] = u1;
            U[((long)(a * 4 + 2) * kdim + k) * ndim + nn] = u2;
            U[((long)(a * 4 + 3) * kdim + k) * ndim + nn] = u3;
        }
    }
}

__global__ __launch_bounds__(512, 1) void wino_fwd_kernel(WinoGeom g, const float* __restrict__ X, const float* __restrict__ U,
                                                          const float* __restrict__ bias, float* __restrict__ Y, int act, float slope) {
    constexpr int BT = 64, BC = 64, LDV = BT + 4, LDU = BC;
    constexpr int RAW_SLOTS = 324, RAW_FLOATS = 12 * 64 * 4;    // 18 x 18 pixels x 8 channels, rounded up to 12 wave loads
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float (*Vs)[16][WKB][LDV] = reinterpret_cast<float (*)[16][WKB][LDV]>(smem);                                   // [2]
    float (*Us)[16][WKB][LDU] = reinterpret_cast<float (*)[16][WKB][LDU]>(smem + 2 * 16 * WKB * LDV);              // [2]
    // The two input-block buffers are separate objects: the compiler drains every outstanding LDS-DMA before a ds_read it
    // cannot prove disjoint from the DMA's destination, and two regions of one array indexed by thread-dependent
    // offsets it cannot.
    __shared__ __attribute__((aligned(16))) float Raw0[RAW_FLOATS], Raw1[RAW_FLOATS];
    int* tilebase = reinterpret_cast<int*>(smem + 2 * 16 * WKB * (LDV + LDU));                                      // [BT] pixel index of output (2th, 2tw), or -1
    int* tilehw = tilebase + BT;                                                                                    // [BT] (2th << 16) | 2tw
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ph = wave >> 2, wt = (wave >> 1) & 1, wc = wave & 1, half = lane >> 5, l31 = lane & 31;

    // Workgroup -> (8 x 8 block of tiles, 64 output channels).  Logically consecutive workgroups (same channel block = same
    // filter tile, neighbouring pixels) are sent to the same XCD so that they share its L2.
    const int nblk = g.n * g.bh * g.bw, total = nblk * (g.cout / BC);
    int wg = blockIdx.x;
    if (total % 8 == 0) wg = (wg & 7) * (total >> 3) + (wg >> 3);
    const int blk = wg % nblk, c0 = (wg / nblk) * BC;
    int img, brem, bty, btx;
    divmod_pos(blk, g.bh * g.bw, img, brem);
    divmod_pos(brem, g.bw, bty, btx);

    // this thread's transform task: tile (tid >> 3) = (ty, tx) of the block, channel (tid & 7) of every step
    const int gt = tid >> 3, ch = tid & 7, ty = gt >> 3, tx = gt & 7;
    if (ch == 0) {
        const int th = bty * 8 + ty, tw = btx * 8 + tx;
        const bool valid = th < g.th && tw < g.tw;
        tilebase[gt] = valid ? (img * g.h + 2 * th) * g.w + 2 * tw : -1;
        tilehw[gt] = ((2 * th) << 16) | (2 * tw);
    }
    // Both operands are fetched through buffer descriptors: one instruction per 1 KB piece (per-lane byte offset in a VGPR,
    // the step's advance in an SGPR), and an offset beyond the descriptor's range reads as zero -- the zero padding of the
    // image border costs nothing.
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, g.n * g.h * g.w * g.cin * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ures = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(U), 0, 16 * g.cin * g.cout * 4, 0x00020000);
    // Filter tile: one piece = 4 k rows x 64 channels of one position (lane = (k & 3) * 16 + float4 column); 16 positions x
    // 2 k halves = 32 pieces per step, 4 per wave, through registers (a load early in the step, a ds_write_b128 late).
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    unsigned uoff[4];
    f32x4 ru[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int piece = wave * 4 + j, p = piece >> 1, k0 = (piece & 1) * 4;
        uoff[j] = (unsigned)((p * g.cin + k0 + (lane >> 4)) * g.cout + c0 + (lane & 15) * 4) * 4u;
    }
    auto load_filter_piece = [&](int j, int ks) { ru[j] = __builtin_amdgcn_raw_buffer_load_b128(ures, uoff[j], ks * WKB * g.cout * 4, 0); };
    auto store_filter_piece = [&](int j, int buf) {
        const int piece = wave * 4 + j, p = piece >> 1, k0 = (piece & 1) * 4;
        *reinterpret_cast<f32x4*>(&Us[buf][p][k0 + (lane >> 4)][(lane & 15) * 4]) = ru[j];
    };
    // Input block: the 18 x 18 pixels under the 8 x 8 tiles (each pixel once, not once per tile that covers it) x 8 channels,
    // by LDS-DMA (no registers).  Slot of pixel (r, c): ((2r + (c & 1)) * 9 + (c >> 1)) -- even and odd columns apart, so that
    // the tiles of a block row read consecutive 32-byte slots (conflict-free).  A piece is 32 slots; lane -> slot (lane >> 1),
    // channel half (lane & 1); pieces 0..7 by the 8 waves, 8..11 by waves 0..3.
    unsigned xoff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = (j * 8 + wave) * 64 + lane, slot = q >> 1;
        const int r = slot / 18, rem = slot - r * 18, par = rem >= 9, c = 2 * (rem - 9 * par) + par;
        const int yy = bty * 16 - 1 + r, xx = btx * 16 - 1 + c;
        const bool in = slot < RAW_SLOTS && yy >= 0 && yy < g.h && xx >= 0 && xx < g.w;
        xoff[j] = in ? (unsigned)(((img * g.h + yy) * g.w + xx) * g.cin + (q & 1) * 4) * 4u : 0x80000000u;
    }
    auto load_input_piece = [&](int j, int ks, int rbuf) {   // rbuf = ks & 1, as a literal
        if (j == 1 && wave >= 4) return;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (__attribute__((address_space(3))) float*)&(rbuf ? Raw1 : Raw0)[(j * 8 + wave) * 256], 16,
                                                 xoff[j], ks * WKB * 4, 0, 0);
    };
    // V = B^T d B of the thread's channel, cut into 64 items (16 patch reads, 16 results of the column pass, 16 results each
    // followed by its write to a V plane) so that the step below can put two behind each MFMA
    float dd[16], tt[16], vv = 0.f;
    const int rawt = ((4 * ty) * 9 + tx) * 8 + ch;
    auto transform_item = [&](int rbuf, int vbuf, int it) {
        if (it < 16) {
            const int i = it >> 2, j = it & 3;
            dd[it] = (rbuf ? Raw1 : Raw0)[rawt + (i * 18 + (j & 1) * 9 + (j >> 1)) * 8];
        } else if (it < 32) {
            const int j = (it - 16) >> 2, s_ = it & 3;
            tt[s_ * 4 + j] = s_ == 0 ? dd[0 * 4 + j] - dd[2 * 4 + j] : s_ == 1 ? dd[1 * 4 + j] + dd[2 * 4 + j]
                           : s_ == 2 ? dd[2 * 4 + j] - dd[1 * 4 + j] : dd[1 * 4 + j] - dd[3 * 4 + j];
        } else {
            const int k = (it - 32) >> 1, i = k >> 2, s_ = k & 3;
            if ((it & 1) == 0)
                vv = s_ == 0 ? tt[i * 4 + 0] - tt[i * 4 + 2] : s_ == 1 ? tt[i * 4 + 1] + tt[i * 4 + 2]
                   : s_ == 2 ? tt[i * 4 + 2] - tt[i * 4 + 1] : tt[i * 4 + 1] - tt[i * 4 + 3];
            else
                Vs[vbuf][k][ch][gt] = vv;
        }
    };

    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    const int nks = g.cin / WKB;                 // even (cin % 16 == 0)
    const int trow = wt * 32 + l31, ccol = wc * 32 + l31;
    // One step: the wave's 32 MFMAs on (V, U)(ks) in LDS buffer `buf` (positions 8 ph .. 8 ph + 7, 4 k pairs), and a few small
    // items behind each of them:
    //   * the filter tile of step ks+1 (4 loads early, 4 LDS writes late) and the input block of step ks+2 (LDS-DMA),
    //   * the input block of step ks+1 (in LDS since the previous step) transformed into the other V buffer,
    //   * the operands of the next group of 4 MFMAs read from LDS.
    // Two waves share a SIMD (the two position halves): what one wave issues besides MFMAs runs under the other's MFMAs.
    auto step = [&](int ks, int buf) {
        const int ksf = ks + 1 < nks ? ks + 1 : nks - 1, ksx = ks + 2 < nks ? ks + 2 : nks - 2 + buf;   // (the tail re-fetches, unused)
        __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): the next input block is in LDS
        __syncthreads();
        float a[2][4], b[2][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[0][e] = Vs[buf][8 * ph + e][half][trow];
            b[0][e] = Us[buf][8 * ph + e][half][ccol];
        }
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            // MFMA m: k pair (m >> 3), position 8 ph + (m & 7); its operands were read one group (4 MFMAs) ago
            const int grp = m >> 2, e = m & 3, cur = grp & 1, q = m & 7;
            __builtin_amdgcn_sched_barrier(0);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][e], b[cur][e], acc[q], 0, 0, 0);
            if (grp + 1 < 8) {
                const int kk = ((grp + 1) >> 1) * 2, pn = 8 * ph + 4 * ((grp + 1) & 1) + e;
                a[cur ^ 1][e] = Vs[buf][pn][kk + half][trow];
                b[cur ^ 1][e] = Us[buf][pn][kk + half][ccol];
            }
            transform_item(buf ^ 1, buf ^ 1, 2 * m);
            transform_item(buf ^ 1, buf ^ 1, 2 * m + 1);
            if (m < 4) load_filter_piece(m, ksf);
            else if (m < 6) load_input_piece(m - 4, ksx, buf);
            else if (m >= 28) store_filter_piece(m - 28, buf ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

#pragma unroll
    for (int j = 0; j < 4; ++j) load_filter_piece(j, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) load_input_piece(j, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) load_input_piece(j, 1, 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) store_filter_piece(j, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 64; ++it) transform_item(0, 0, it);
    for (int ks = 0; ks < nks; ks += 2) {
        step(ks, 0);
        step(ks + 1, 1);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // nothing may still be writing LDS when the workgroup retires
    __syncthreads();

    // Output transform Y = A^T M A per (tile, co), A^T = [1 1 1 0; 0 1 -1 -1].  A wave holds rows a = 2 ph, 2 ph + 1 of M (the 8
    // positions of one element in the same lane / register index): it forms its part of both output rows, keeps the part
    // of output row ph, hands the other to its partner wave through LDS (over the operand buffers, no longer read), adds
    // what it receives and stores output row ph.
    // C/D layout: col = lane&31 -> co, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -> tile
    float* xchg = smem + ((wt * 2 + wc) * 2) * 16 * 2 * 64;      // [sender ph][r][j][lane]
    float keep[16][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        __builtin_amdgcn_sched_barrier(0);      // one element's 8 accumulator reads at a time
        float tm[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float m0 = acc[j][r], m1 = acc[4 + j][r];
            tm[0][j] = ph ? m0 : m0 + m1;
            tm[1][j] = ph ? -m0 - m1 : m1;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float y0 = tm[i][0] + tm[i][1] + tm[i][2], y1 = tm[i][1] - tm[i][2] - tm[i][3];
            if (i == ph) {
                keep[r][0] = y0;
                keep[r][1] = y1;
            } else {
                xchg[((ph * 16 + r) * 2 + 0) * 64 + lane] = y0;
                xchg[((ph * 16 + r) * 2 + 1) * 64 + lane] = y1;
            }
        }
    }
    __syncthreads();
    const int co = c0 + wc * 32 + l31;
    const float bv = bias ? bias[co] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int tl = wt * 32 + 4 * half + (r & 3) + 8 * (r >> 2);
        const int base = tilebase[tl];
        const int hw = tilehw[tl], oy = hw >> 16, ox = hw & 0xffff;
        const float y0 = keep[r][0] + xchg[(((ph ^ 1) * 16 + r) * 2 + 0) * 64 + lane];
        const float y1 = keep[r][1] + xchg[(((ph ^ 1) * 16 + r) * 2 + 1) * 64 + lane];
        if (base < 0 || oy + ph >= g.h) continue;
        float* dst = Y + ((long)base + ph * g.w) * g.cout + co;
        dst[0] = cn_apply_act(y0 + bv, act, slope);
        if (ox + 1 < g.w) dst[g.cout] = cn_apply_act(y1 + bv, act, slope);
    }
}

}  // namespace

extern "C" int cn_conv_wino_filter(const float* w, float* u, int cin, int cout, int dgrad, void* stream) {
    CN_CHECK_ARG(w && u && cin > 0 && cout > 0, "wino_filter: bad args");
    const long total = (long)cin * cout;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(wino_filter_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, u, cin, cout, dgrad);
    CN_LAUNCH_CHECK();
    return CN_OK;
}

// x (n, h, w, cin) -> y (n, h, w, cout): 3x3, stride 1, SAME; u from cn_conv_wino_filter ([16][cin][cout]).
// Returns CN_EUNSUPPORTED (nothing launched) unless cin % 16 == 0 and cout % 64 == 0.
extern "C" int cn_conv_fwd_wino(int n, int h, int w, int cin, int cout, const float* x, const float* u, const float* bias,
                                float* y, int act, float slope, void* stream) {
    CN_CHECK_ARG(x && u && y && n > 0 && h > 0 && w > 0, "conv_fwd_wino: bad args");
    if (cin % (2 * WKB) || cout % 64) return CN_EUNSUPPORTED;
    CN_CHECK_ARG((double)n * h * w * (cin > cout ? cin : cout) * 4.0 < 2147483647.0 && 64.0 * cin * cout < 2147483647.0, "tensor exceeds 2^31 bytes (buffer descriptors, 32-bit offsets)");
    WinoGeom g{n, h, w, cin, cout, (h + 1) / 2, (w + 1) / 2, (h + 15) / 16, (w + 15) / 16};
    const long ntiles = (long)n * g.th * g.tw, nblk = (long)n * g.bh * g.bw;
    constexpr size_t lds = sizeof(float) * (2 * 16 * WKB * (68 + 64)) + sizeof(int) * 128;     // + 24 KB static (input blocks)
    static bool attr_set = false;
    if (!attr_set) {
        CN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wino_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipStream_t s = (hipStream_t)stream;
    // MFMA work that contributes to the result: 16 products per 2x2 output tile and (ci, co) pair
    cn_prof_begin(s, 2.0 * 16.0 * (double)ntiles * cin * cout,
                  4.0 * ((double)n * h * w * (cin + cout) + 16.0 * cin * cout), CN_FAM_WINO);
    hipLaunchKernelGGL(wino_fwd_kernel, dim3((unsigned)(nblk * (cout / 64))), dim3(512), lds, s, g, x, u, bias, y, act, slope);
    cn_prof_end(s);
    CN_LAUNCH_CHECK();
    return CN_OK;
}
