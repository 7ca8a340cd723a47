// thin_wgrad.hip -- filter gradient of a 2-D stride-1 convolution with <= 4 output channels (map_final 32 -> 3 behind the folded
// x2 upsample, hologan_generator.py:101): gw[kh][kw][ci][co] = sum_{n,oy,ox} xu[n, oy+kh-p, ox+kw-p, ci] * gy[n, oy, ox, co].
// The implicit-GEMM filter gradient runs this shape at 6 TFLOP/s (a 3-column B tile on a 32-column MFMA block, 16-row steps whose
// gather is recomputed per tap: 257 us for 8 x 256 x 256 pixels); the work is 0.8 G multiply-adds over 23 MB of operands.  Here a
// workgroup stages the input patch of an 8 x 32 output tile and the gy tile in LDS, a thread owns one (tap, 4-channel group) with
// 4 x cout accumulators and walks the tile's pixels (gy is an LDS broadcast, x a 16-byte LDS read), the workgroups' partial
// filters go to scratch and are added in order by a second launch -- no atomics, deterministic.
#include "common.h"
#include "conv_geom.h"

namespace {

constexpr int TH = 8, TW = 32;            // output pixels of a tile
constexpr int NWG = 512;                  // workgroups = partial filters

template <int CO>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(CnConvGeom g, const float* __restrict__ X, const float* __restrict__ GY,
                                                         float* __restrict__ part, int tiles_y, int tiles_x, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int up = g.up, cin = g.cin, CG = cin / 4, T = g.k_h * g.k_w;
    const int PH0 = ((TH + g.k_h - 1) >> up) + 2, PW0 = ((TW + g.k_w - 1) >> up) + 2;     // stored-grid patch extents (upper bounds)
    // (PH below: partitions of the tile's 8 rows: 1, 2, 4 or 8)
    float* patch = sm;                                     // [PH0][PW0][cin]
    float* gyt = sm + PH0 * PW0 * cin;                     // [TH*TW][4]
    const int combos = T * CG;
    int PH = 256 / combos;                                 // row partitions of a tile
    PH = PH >= 8 ? 8 : PH >= 4 ? 4 : PH >= 2 ? 2 : 1;
    const int tid = threadIdx.x;
    const int role = tid % combos, ppart = tid / combos;
    const bool live = tid < combos * PH;
    const int cg = role % CG, tap = role / CG, kh = tap / g.k_w, kw = tap - kh * g.k_w;
    float acc[4][CO];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[e][c] = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
        const int oy0 = ty * TH, ox0 = tx * TW;
        // first virtual (upsampled) source coordinate the tile touches and its stored-grid pixel (floor shift: -1 >> 1 == -1):
        // the patch starts there and out-of-image entries are staged as zeros, so the inner loop needs no bounds test
        const int vy0 = oy0 - g.p_h, vx0 = ox0 - g.p_w;
        const int sy0 = vy0 >> up, sx0 = vx0 >> up;
        __syncthreads();                                   // previous tile fully consumed
        for (int i = tid; i < PH0 * PW0 * CG; i += 256) {
            const int c4 = i % CG, px = (i / CG) % PW0, py = i / (CG * PW0);
            const int sy = sy0 + py, sx = sx0 + px;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (sy >= 0 && sx >= 0 && sy < g.in_h && sx < g.in_w)
                v = *reinterpret_cast<const float4*>(X + (((long)n * g.in_h + sy) * g.in_w + sx) * cin + c4 * 4);
            *reinterpret_cast<float4*>(patch + (py * PW0 + px) * cin + c4 * 4) = v;
        }
        for (int i = tid; i < TH * TW; i += 256) {
            const int py = i / TW, px = i - py * TW;
            const int oy = oy0 + py, ox = ox0 + px;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (oy < g.out_h && ox < g.out_w) {
                const float* p = GY + (((long)n * g.out_h + oy) * g.out_w + ox) * g.cout;
#pragma unroll
                for (int c = 0; c < CO; ++c) v[c] = p[c];
            }
            *reinterpret_cast<float4*>(gyt + i * 4) = make_float4(v[0], v[1], v[2], v[3]);
        }
        __syncthreads();
        if (live) {
            for (int py = ppart; py < TH; py += PH) {      // the pixel partitions split the tile's rows
                const float* prow = patch + (((vy0 + py + kh) >> up) - sy0) * PW0 * cin + cg * 4;
                const float* grow = gyt + py * TW * 4;
                const int vx = vx0 + kw;
#pragma unroll 8
                for (int px = 0; px < TW; ++px) {
                    const float4 xv = *reinterpret_cast<const float4*>(prow + (((vx + px) >> up) - sx0) * cin);
                    const float4 gv = *reinterpret_cast<const float4*>(grow + px * 4);
                    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int c = 0; c < CO; ++c) acc[e][c] += xs[e] * gs[c];
                }
            }
        }
    }
    // combine the pixel partitions (through the patch area), one partial filter per workgroup: part[wg][tap][ci][co]
    __syncthreads();
    float* red = sm;                                       // [PH][combos][4*CO]
    if (live) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < CO; ++c) red[((ppart * combos) + role) * (4 * CO) + e * CO + c] = acc[e][c];
    }
    __syncthreads();
    for (int i = tid; i < combos * 4 * CO; i += 256) {
        float t = 0.f;
        for (int p = 0; p < PH; ++p) t += red[(p * combos) * (4 * CO) + i];
        const int r = i / (4 * CO), e = (i / CO) % 4, c = i % CO;
        const int cg2 = r % CG, tap2 = r / CG;
        part[(long)blockIdx.x * (T * cin * g.cout) + ((long)tap2 * cin + cg2 * 4 + e) * g.cout + c] = t;
    }
}

}  // namespace

extern "C" int cn_conv_wgrad_thin_partials(void) { return NWG; }

// Filter gradient for cout <= 4 (see the header of this file).  scratch: cn_conv_wgrad_thin_partials() * taps * cin * cout floats.
// accumulate: add to gw.  CN_EUNSUPPORTED (nothing launched) for every other geometry.
extern "C" int cn_conv_wgrad_thin(const CnConvGeom* gp, const float* x, const float* gy, float* scratch, float* gw, int accumulate,
                                  void* stream) {
    if (int e = check_geom(gp)) return e;
    const CnConvGeom g = *gp;
    const int T = g.k_h * g.k_w;
    if (g.nd != 2 || g.cout > 4 || g.cin % 4 || g.cin > 64 || g.s_h != 1 || g.s_w != 1 || g.dl_h != 1 || g.dl_w != 1 || T > 16 ||
        T * (g.cin / 4) > 256 || g.cin < 8)
        return CN_EUNSUPPORTED;
    CN_CHECK_ARG(x && gy && scratch && gw, "conv_wgrad_thin: NULL");
    hipStream_t s = (hipStream_t)stream;
    const int tiles_y = cn_cdiv(g.out_h, TH), tiles_x = cn_cdiv(g.out_w, TW);
    const int ntiles = g.n * tiles_y * tiles_x;
    const int nwg = ntiles < NWG ? ntiles : NWG;
    const int PH0 = ((TH + g.k_h - 1) >> g.up) + 2, PW0 = ((TW + g.k_w - 1) >> g.up) + 2;
    size_t lds = sizeof(float) * ((size_t)PH0 * PW0 * g.cin + TH * TW * 4);
    const size_t red = sizeof(float) * 256 * 4 * g.cout;
    if (lds < red) lds = red;
    if (lds > 64 * 1024) return CN_EUNSUPPORTED;
    const long count = (long)T * g.cin * g.cout;
#define TW_LAUNCH(CO_) hipLaunchKernelGGL((thin_wgrad_kernel<CO_>), dim3(nwg), dim3(256), lds, s, g, x, gy, scratch, tiles_y, tiles_x, ntiles)
    switch (g.cout) {
        case 1: TW_LAUNCH(1); break;
        case 2: TW_LAUNCH(2); break;
        case 3: TW_LAUNCH(3); break;
        default: TW_LAUNCH(4); break;
    }
#undef TW_LAUNCH
    CN_LAUNCH_CHECK();
    return cn_sum_parts(scratch, gw, nwg, count, accumulate, 1.f, s);
}
