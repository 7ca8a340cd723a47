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

__global__ __launch_bounds__(256, 1) void wino_fwd_kernel(WinoGeom g, const float* __restrict__ X, const float* __restrict__ U,
                                                          const float* __restrict__ bias, float* __restrict__ Y, int act, float slope) {
    constexpr int BT = 64, BC = 64, LDV = BT + 4, LDU = BC;    // Us / Raw unpadded: filled by LDS-DMA loads (wave-contiguous)
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
    const int wt = wave >> 1, wc = wave & 1, half = lane >> 5, l31 = lane & 31;

    // Workgroup -> (8 x 8 block of tiles, 64 output channels).  Logically consecutive workgroups (same channel block = same
    // filter tile, neighbouring pixels) are sent to the same XCD so that they share its L2.
    const int nblk = g.n * g.bh * g.bw, total = nblk * (g.cout / BC);
    int wg = blockIdx.x;
    if (total % 8 == 0) wg = (wg & 7) * (total >> 3) + (wg >> 3);
    const int blk = wg % nblk, c0 = (wg / nblk) * BC;
    const int img = blk / (g.bh * g.bw), brem = blk - img * (g.bh * g.bw), bty = brem / g.bw, btx = brem - bty * g.bw;

    // this thread's transform task: tile (tid >> 2) = (ty, tx) of the block, channel pair (tid & 3) of every step
    const int gt = tid >> 2, cp = tid & 3, ty = gt >> 3, tx = gt & 7;
    if (cp == 0) {
        const int th = bty * 8 + ty, tw = btx * 8 + tx;
        const bool valid = th < g.th && tw < g.tw;
        tilebase[gt] = valid ? (img * g.h + 2 * th) * g.w + 2 * tw : -1;
        tilehw[gt] = ((2 * th) << 16) | (2 * tw);
    }
    // Both operands arrive by LDS-DMA through buffer descriptors: one instruction per 1 KB piece (per-lane byte offset in a
    // VGPR, the step's advance in an SGPR), no staging registers (16 accumulators leave none to spare), and an offset
    // beyond the descriptor's range reads as zero -- the zero padding of the image border costs nothing.
    const __amdgpu_buffer_rsrc_t xres = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, g.n * g.h * g.w * g.cin * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ures = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(U), 0, 16 * g.cin * g.cout * 4, 0x00020000);
    // Filter tile: one piece = 4 k rows x 64 channels of one position (lane = (k & 3) * 16 + float4 column); 16 positions x
    // 2 k halves = 32 pieces per step, 8 per wave.
    unsigned uoff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int piece = wave * 8 + j, p = piece >> 1, k0 = (piece & 1) * 4;
        uoff[j] = (unsigned)((p * g.cin + k0 + (lane >> 4)) * g.cout + c0 + (lane & 15) * 4) * 4u;
    }
    // Input block: the 18 x 18 pixels under the 8 x 8 tiles (each pixel once, not once per tile that covers it) x 8 channels.
    // Slot of pixel (r, c): ((2r + (c & 1)) * 9 + (c >> 1)) -- even and odd columns apart, so that the 8 tiles of a block row
    // read 8 consecutive 32-byte slots (conflict-free ds_read_b64).  A piece is 32 slots; lane -> slot (lane >> 1), channel
    // half (lane & 1); 3 pieces per wave.
    unsigned xoff[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int q = (wave * 3 + j) * 64 + lane, slot = q >> 1;
        const int r = slot / 18, rem = slot - r * 18, par = rem >= 9, c = 2 * (rem - 9 * par) + par;
        const int yy = bty * 16 - 1 + r, xx = btx * 16 - 1 + c;
        const bool in = slot < RAW_SLOTS && yy >= 0 && yy < g.h && xx >= 0 && xx < g.w;
        xoff[j] = in ? (unsigned)(((img * g.h + yy) * g.w + xx) * g.cin + (q & 1) * 4) * 4u : 0x80000000u;
    }
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 ru[8];
    auto load_filter_piece = [&](int j, int ks) {
        ru[j] = __builtin_amdgcn_raw_buffer_load_b128(ures, uoff[j], ks * WKB * g.cout * 4, 0);
    };
    auto store_filter_piece = [&](int j, int buf) {
        const int piece = wave * 8 + j, p = piece >> 1, k0 = (piece & 1) * 4;
        *reinterpret_cast<f32x4*>(&Us[buf][p][k0 + (lane >> 4)][(lane & 15) * 4]) = ru[j];
    };
    auto load_input_piece = [&](int j, int ks, int rbuf) {   // rbuf = ks & 1, as a literal
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xres, (__attribute__((address_space(3))) float*)&(rbuf ? Raw1 : Raw0)[(wave * 3 + j) * 256], 16,
                                                 xoff[j], ks * WKB * 4, 0, 0);
    };
    // V = B^T d B of the thread's two channels, cut into 64 pieces so that the step below can put one behind each MFMA:
    // 16 patch reads (both channels in one ds_read_b64), 16 results of the column pass, 32 results written to their V planes
    float2 dd[16], tt[16];
    const int rawt = ((4 * ty) * 9 + tx) * 8 + cp * 2;
    auto transform_piece = [&](int rbuf, int vbuf, int m) {
        if (m < 16) {
            const int i = m >> 2, j = m & 3;
            dd[m] = *reinterpret_cast<const float2*>(&(rbuf ? Raw1 : Raw0)[rawt + (i * 18 + (j & 1) * 9 + (j >> 1)) * 8]);
        } else if (m < 32) {
            const int j = (m - 16) >> 2, s_ = m & 3;
            auto d = [&](int i) { return dd[i * 4 + j]; };
            float2 v;
            if (s_ == 0) v = make_float2(d(0).x - d(2).x, d(0).y - d(2).y);
            else if (s_ == 1) v = make_float2(d(1).x + d(2).x, d(1).y + d(2).y);
            else if (s_ == 2) v = make_float2(d(2).x - d(1).x, d(2).y - d(1).y);
            else v = make_float2(d(1).x - d(3).x, d(1).y - d(3).y);
            tt[s_ * 4 + j] = v;
        } else {
            const int idx = m - 32, e = idx & 1, i = idx >> 3, s_ = (idx >> 1) & 3;
            auto t = [&](int c) { return e ? tt[i * 4 + c].y : tt[i * 4 + c].x; };
            const float v = s_ == 0 ? t(0) - t(2) : s_ == 1 ? t(1) + t(2) : s_ == 2 ? t(2) - t(1) : t(1) - t(3);
            Vs[vbuf][i * 4 + s_][2 * cp + e][gt] = v;
        }
    };

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    const int nks = g.cin / WKB;                 // even (cin % 16 == 0)
    const int trow = wt * 32 + l31, ccol = wc * 32 + l31;
    // One step: the matrix cores work on (V, U)(ks) in LDS buffer `buf` while, one small piece behind each of the 64 MFMAs
    // (a wave has its SIMD to itself: whatever is not issued in an MFMA's 64-cycle shadow stalls the matrix pipe),
    //   * the filter tile of step ks+1 and the input block of step ks+2 are requested (8 + 3 instructions),
    //   * the input block of step ks+1 (in LDS since the previous step) is transformed into the other V buffer,
    //   * the operands of the next group of 4 MFMAs are read from LDS.
    // The loads have most of a step (> 3000 cycles) to land before the barrier that opens the next step.
    auto step = [&](int ks, int buf) {
        const int ksf = ks + 1 < nks ? ks + 1 : nks - 1, ksx = ks + 2 < nks ? ks + 2 : nks - 2 + buf;   // (the tail re-fetches, unused)
        __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): this step's filter tile and the next input block are in LDS
        __syncthreads();
        float a[2][4], b[2][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[0][e] = Vs[buf][e][half][trow];
            b[0][e] = Us[buf][e][half][ccol];
        }
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            // MFMA m: k pair (m >> 4), position 4 * ((m >> 2) & 3) + (m & 3); its operands were read one group (4 MFMAs) ago
            const int grp = m >> 2, e = m & 3, cur = grp & 1, p = 4 * (grp & 3) + e;
            __builtin_amdgcn_sched_barrier(0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][e], b[cur][e], acc[p], 0, 0, 0);
            if (grp + 1 < 16) {
                const int kk = ((grp + 1) >> 2) * 2, pn = 4 * ((grp + 1) & 3) + e;
                a[cur ^ 1][e] = Vs[buf][pn][kk + half][trow];
                b[cur ^ 1][e] = Us[buf][pn][kk + half][ccol];
            }
            transform_piece(buf ^ 1, buf ^ 1, m);
            if (m < 8) load_filter_piece(m, ksf);
            else if (m >= 56) store_filter_piece(m - 56, buf ^ 1);
            else if (m < 11) load_input_piece(m - 8, ksx, buf);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

#pragma unroll
    for (int j = 0; j < 8; ++j) load_filter_piece(j, 0);
#pragma unroll
    for (int j = 0; j < 3; ++j) load_input_piece(j, 0, 0);
#pragma unroll
    for (int j = 0; j < 3; ++j) load_input_piece(j, 1, 1);
#pragma unroll
    for (int j = 0; j < 8; ++j) store_filter_piece(j, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 64; ++m) transform_piece(0, 0, m);
    for (int ks = 0; ks < nks; ks += 2) {
        step(ks, 0);
        step(ks + 1, 1);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // nothing may still be writing LDS when the workgroup retires
    __syncthreads();

    // output transform Y = A^T M A per (tile, co): the 16 positions of one element sit in the same lane / register index.
    // C/D layout: col = lane&31 -> co, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -> tile
    const int co = c0 + wc * 32 + l31;
    const float bv = bias ? bias[co] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        __builtin_amdgcn_sched_barrier(0);      // one element's 16 accumulator reads at a time (hoisting all 256 exhausts the VGPRs)
        const int tl = wt * 32 + 4 * half + (r & 3) + 8 * (r >> 2);
        const int base = tilebase[tl];
        if (base < 0) continue;
        const int hw = tilehw[tl], oy = hw >> 16, ox = hw & 0xffff;
        float tm[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tm[0][j] = acc[0 * 4 + j][r] + acc[1 * 4 + j][r] + acc[2 * 4 + j][r];
            tm[1][j] = acc[1 * 4 + j][r] - acc[2 * 4 + j][r] - acc[3 * 4 + j][r];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            if (oy + a >= g.h) continue;
            const float y0 = tm[a][0] + tm[a][1] + tm[a][2], y1 = tm[a][1] - tm[a][2] - tm[a][3];
            float* dst = Y + ((long)base + a * g.w) * g.cout + co;
            dst[0] = cn_apply_act(y0 + bv, act, slope);
            if (ox + 1 < g.w) dst[g.cout] = cn_apply_act(y1 + bv, act, slope);
        }
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
    cn_prof_begin(s, 2.0 * 16.0 * (double)ntiles * cin * cout);
    hipLaunchKernelGGL(wino_fwd_kernel, dim3((unsigned)(nblk * (cout / 64))), dim3(256), lds, s, g, x, u, bias, y, act, slope);
    cn_prof_end(s);
    CN_LAUNCH_CHECK();
    return CN_OK;
}
