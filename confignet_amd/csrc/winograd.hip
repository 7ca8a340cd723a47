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

struct WinoGeom {
    int n, h, w, cin, cout, th, tw;     // th = ceil(h/2), tw = ceil(w/2) tiles per image
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
    constexpr int BT = 64, BC = 64, LDV = BT + 4, LDU = BC;    // Us unpadded: filled by global_load_lds (wave-contiguous)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float (*Vs)[16][WKB][LDV] = reinterpret_cast<float (*)[16][WKB][LDV]>(smem);                                   // [2]
    float (*Us)[16][WKB][LDU] = reinterpret_cast<float (*)[16][WKB][LDU]>(smem + 2 * 16 * WKB * LDV);              // [2]
    int* tilebase = reinterpret_cast<int*>(smem + 2 * 16 * WKB * (LDV + LDU));                                      // [BT] pixel index of output (2th, 2tw), or -1
    int* tilehw = tilebase + BT;                                                                                    // [BT] (2th << 16) | 2tw
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wt = wave >> 1, wc = wave & 1, half = lane >> 5, l31 = lane & 31;
    const int tiles_img = g.th * g.tw, ntiles = g.n * tiles_img;
    const int t0 = blockIdx.x * BT, c0 = blockIdx.y * BC;

    // this thread's gather task: tile (tid >> 2), channel pair (tid & 3) of every step
    const int gt = tid >> 2, cp = tid & 3;
    int gbase = -1, gy0 = 0, gx0 = 0;
    {
        const int t = t0 + gt;
        if (t < ntiles) {
            const int nimg = t / tiles_img, r = t - nimg * tiles_img;
            const int th = r / g.tw, tw = r - th * g.tw;
            gy0 = 2 * th - 1;
            gx0 = 2 * tw - 1;
            gbase = nimg * g.h * g.w;
            if (cp == 0) {
                tilebase[gt] = gbase + 2 * th * g.w + 2 * tw;
                tilehw[gt] = ((2 * th) << 16) | (2 * tw);
            }
        } else if (cp == 0) {
            tilebase[gt] = -1;
            tilehw[gt] = 0;
        }
    }
    // offsets of the 16 patch pixels; pixels outside the image load element 0 and are zeroed through pmask (straight-line
    // loads: 16 divergent branches per step otherwise)
    int poff[16];
    unsigned pmask = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int yy = gy0 + i, xx = gx0 + j;
            const bool in = gbase >= 0 && yy >= 0 && yy < g.h && xx >= 0 && xx < g.w;
            poff[i * 4 + j] = in ? (gbase + yy * g.w + xx) * g.cin + 2 * cp : 0;
            pmask |= in ? 1u << (i * 4 + j) : 0u;
        }

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    float2 rd[16];
    // The filter tile goes global -> LDS directly (no staging registers: 16 accumulators leave none to spare).  One wave
    // instruction moves 4 k rows x 64 channels of one position (lane = (k & 3) * 16 + float4 column) into 1 KB of LDS.
    auto load_step = [&](int ks, int buf) {
        const int ci0 = ks * WKB;
#pragma unroll
        for (int j = 0; j < 8; ++j) {                        // 16 positions x 2 k halves = 32 wave loads, 8 per wave
            const int piece = wave * 8 + j, p = piece >> 1, k0 = (piece & 1) * 4;
            const float* src = U + ((long)p * g.cin + ci0 + k0 + (lane >> 4)) * g.cout + c0 + (lane & 15) * 4;
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) float*)&Us[buf][p][k0][0], 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q)
            rd[q] = *reinterpret_cast<const float2*>(X + poff[q] + ci0);
    };
    auto store_step = [&](int buf) {
        // V = B^T d B for the two channels of this thread
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float d[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) d[q] = (pmask >> q) & 1 ? (e ? rd[q].y : rd[q].x) : 0.f;
            float t[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t[0 * 4 + j] = d[0 * 4 + j] - d[2 * 4 + j];
                t[1 * 4 + j] = d[1 * 4 + j] + d[2 * 4 + j];
                t[2 * 4 + j] = d[2 * 4 + j] - d[1 * 4 + j];
                t[3 * 4 + j] = d[1 * 4 + j] - d[3 * 4 + j];
            }
            const int k = 2 * cp + e;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Vs[buf][i * 4 + 0][k][gt] = t[i * 4 + 0] - t[i * 4 + 2];
                Vs[buf][i * 4 + 1][k][gt] = t[i * 4 + 1] + t[i * 4 + 2];
                Vs[buf][i * 4 + 2][k][gt] = t[i * 4 + 2] - t[i * 4 + 1];
                Vs[buf][i * 4 + 3][k][gt] = t[i * 4 + 1] - t[i * 4 + 3];
            }
        }
    };

    const int nks = g.cin / WKB;
    load_step(0, 0);
    store_step(0);
    __syncthreads();
    const int trow = wt * 32 + l31, ccol = wc * 32 + l31;
    for (int ks = 0; ks < nks; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nks) load_step(ks + 1, buf ^ 1);
        {
            // 16 groups of 4 MFMAs (k pair kk, positions 4q..4q+3); the operands of group i+1 are fetched from LDS before the
            // MFMAs of group i are issued (two small register sets) -- a bounded software pipeline instead of 128 hoisted reads
            float a[2][4], b[2][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[0][e] = Vs[buf][e][half][trow];
                b[0][e] = Us[buf][e][half][ccol];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int cur = i & 1, nxt = cur ^ 1;
                if (i + 1 < 16) {
                    const int kk = ((i + 1) >> 2) * 2, q = (i + 1) & 3;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a[nxt][e] = Vs[buf][4 * q + e][kk + half][trow];
                        b[nxt][e] = Us[buf][4 * q + e][kk + half][ccol];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                const int q = i & 3;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[4 * q + e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][e], b[cur][e], acc[4 * q + e], 0, 0, 0);
            }
        }
        if (ks + 1 < nks) store_step(buf ^ 1);
        __syncthreads();
    }

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
// Returns CN_EUNSUPPORTED (nothing launched) unless cin % 8 == 0 and cout % 64 == 0.
extern "C" int cn_conv_fwd_wino(int n, int h, int w, int cin, int cout, const float* x, const float* u, const float* bias,
                                float* y, int act, float slope, void* stream) {
    CN_CHECK_ARG(x && u && y && n > 0 && h > 0 && w > 0, "conv_fwd_wino: bad args");
    if (cin % WKB || cout % 64 || cin < WKB) return CN_EUNSUPPORTED;
    CN_CHECK_ARG((double)n * h * w * (cin > cout ? cin : cout) < 2147483647.0, "tensor exceeds 2^31 elements (32-bit offsets)");
    WinoGeom g{n, h, w, cin, cout, (h + 1) / 2, (w + 1) / 2};
    const long ntiles = (long)n * g.th * g.tw;
    constexpr size_t lds = sizeof(float) * (2 * 16 * WKB * (68 + 64)) + sizeof(int) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        CN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wino_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipStream_t s = (hipStream_t)stream;
    // MFMA work actually issued: 16 products per 2x2 output tile and (ci, co) pair
    cn_prof_begin(s, 2.0 * 16.0 * (double)ntiles * cin * cout);
    hipLaunchKernelGGL(wino_fwd_kernel, dim3(cn_cdiv(ntiles, 64), cout / 64), dim3(256), lds, s, g, x, u, bias, y, act, slope);
    cn_prof_end(s);
    CN_LAUNCH_CHECK();
    return CN_OK;
}
