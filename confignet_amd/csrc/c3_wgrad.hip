// c3_wgrad.hip -- filter gradient of the 3x3 convolutions of a 3-channel image (K = 27: from-RGB block of the discriminators,
// VGG conv1_1; stride 1 or 2, TF SAME padding, cout <= 64).
//
//   GW[(kh, kw, ci), co] = sum over output pixels p of X[src(p, kh, kw), ci] * GY[p, co]
//
// 0.7 GFLOP against 63 MB of reads (gy once + the image): an HBM-bound shape that the generic split-over-rows kernel runs at
// 10 TFLOP/s (63 us fp32 / 80 us with bf16 activations, 17 launches per iteration) because every 16-row step gathers 27
// scalars per row.  Here a WAVE owns tiles of 64 consecutive output pixels of one output row: it stages the three input
// rows under the tile (coalesced dword loads, zero outside the image) and the gy tile (16-byte loads) in its private part of
// LDS and feeds v_mfma_f32_32x32x2_f32 from there: rows i = (kh, kw, ci) padded 27 -> 32, columns co in two 32-wide blocks,
// reduction over the tile's pixels.  No atomics (the 27 x cout output would make every workgroup end in the same 1296 of
// them): the four waves of a workgroup are added through LDS, every workgroup writes one partial filter, and a second small
// kernel adds the partials.
#include "common.h"
#include "typed.h"

#include "mma_tile.h"

namespace {

constexpr int C3W_TILE = 64;           // output pixels per tile

template <int S, typename T>
__global__ __launch_bounds__(256) void c3_wgrad_kernel(CnConvGeom g, const float* __restrict__ X, const T* __restrict__ GY,
                                                       float* __restrict__ partial, int tiles_x, int ntiles) {
    constexpr int PW = C3W_TILE * S + 2;                 // input columns under a tile
    constexpr int XF = 3 * PW * 3;                       // floats of the staged input rows
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int cout = g.cout, gpitch = (cout + 3) & ~3;
    const int per_wave = XF + C3W_TILE * gpitch;
    float* xs = smem + wave * per_wave;                  // [3][PW][3]
    float* gs = xs + XF;                                 // [64][gpitch]
    // operand A: row i = (kh, kw, ci) -> offset inside xs of pixel 0's tap; rows 27..31 read a zero kept at xs[XF - 1]... no:
    // they are masked after the read
    const int kh = l31 / 9, kwci = l31 - kh * 9;
    const int a_off = kh * PW * 3 + kwci;
    const bool a_on = l31 < 27;

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // Tile-invariant per-lane slots of the two staging copies: element e = lane + 64 k of an input row (3 PW floats) and
    // piece e of the gy tile (pixel e / q4, channels 4 (e % q4) ..).
    constexpr int XI = (PW * 3 + 63) / 64, GI = 16;
    const int q4 = gpitch >> 2;
    int g_p[GI], g_c[GI];
#pragma unroll
    for (int k = 0; k < GI; ++k) {
        const int e = lane + 64 * k;
        g_p[k] = e / q4;
        g_c[k] = (e - g_p[k] * q4) * 4;
    }
    float xr[3][XI];
    float4 gr[GI];
    // global -> registers (the next tile's loads fly while the current tile is multiplied)
    auto load_tile = [&](int t) {
        const int tx = t % tiles_x, row = t / tiles_x, oy = row % g.out_h, n = row / g.out_h;
        const int ox0 = tx * C3W_TILE, npx = min(C3W_TILE, g.out_w - ox0);
        const int ix0 = ox0 * S - g.p_w;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = oy * S - g.p_h + r;
            const bool rok = iy >= 0 && iy < g.in_h;
            const float* src = X + ((long)(n * g.in_h + (rok ? iy : 0)) * g.in_w + ix0) * 3;
#pragma unroll
            for (int k = 0; k < XI; ++k) {
                const int e = lane + 64 * k, ix = ix0 + e / 3;
                xr[r][k] = (rok && e < PW * 3 && ix >= 0 && ix < g.in_w) ? src[e] : 0.f;
            }
        }
        const T* gsrc = GY + ((long)(n * g.out_h + oy) * g.out_w + ox0) * cout;
#pragma unroll
        for (int k = 0; k < GI; ++k)
            gr[k] = g_p[k] < npx ? ld4<T>(gsrc + (long)g_p[k] * cout + g_c[k]) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < XI; ++k)
                if (lane + 64 * k < PW * 3) xs[r * PW * 3 + lane + 64 * k] = xr[r][k];
#pragma unroll
        for (int k = 0; k < GI; ++k)
            if (g_p[k] < C3W_TILE) *reinterpret_cast<float4*>(&gs[g_p[k] * gpitch + g_c[k]]) = gr[k];
    };

    const int nwaves = gridDim.x * 4;
    int t = blockIdx.x * 4 + wave;
    if (t < ntiles) load_tile(t);
    while (t < ntiles) {
        store_tile();                                    // (in-order LDS: behind the previous tile's operand reads)
        const int tn = t + nwaves;
        if (tn < ntiles) load_tile(tn);
        // ---- 32 pixel pairs x 2 column blocks; the operands of pair pp + 1 are read from LDS before the MFMAs of pair pp
        // are issued (two register sets), as in mma_tile.h ----
        {
            const bool c0 = l31 < cout, c1 = 32 + l31 < cout;
            const float* ap = xs + a_off + half * 3 * S;
            const float* bp = gs + half * gpitch + l31;
            float a[2], b0[2], b1[2];
            a[0] = ap[0];
            b0[0] = c0 ? bp[0] : 0.f;
            b1[0] = c1 ? bp[32] : 0.f;
#pragma unroll
            for (int pp = 0; pp < C3W_TILE / 2; ++pp) {
                const int cur = pp & 1, nxt = cur ^ 1;
                if (pp + 1 < C3W_TILE / 2) {
                    a[nxt] = ap[(pp + 1) * 2 * 3 * S];
                    b0[nxt] = c0 ? bp[(pp + 1) * 2 * gpitch] : 0.f;
                    b1[nxt] = c1 ? bp[(pp + 1) * 2 * gpitch + 32] : 0.f;
                }
                __builtin_amdgcn_sched_barrier(0);
                const float av = a_on ? a[cur] : 0.f;
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0[cur], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1[cur], acc[1], 0, 0, 0);
            }
        }
        t = tn;
    }
    // ---- add the four waves through LDS, write the workgroup's partial filter [27][cout] ----
    __syncthreads();
    float* red = smem;                                   // [4][2][16][64]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * 2 + j) * 16 + r) * 64 + lane] = acc[j][r];
    __syncthreads();
    // C/D layout: col = lane & 31 -> co (block j), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) -> i
    for (int e = tid; e < 2 * 16 * 64; e += 256) {
        const int ln = e & 63, r = (e >> 6) & 15, j = e >> 10;
        const float v = red[((0 * 2 + j) * 16 + r) * 64 + ln] + red[((1 * 2 + j) * 16 + r) * 64 + ln] +
                        red[((2 * 2 + j) * 16 + r) * 64 + ln] + red[((3 * 2 + j) * 16 + r) * 64 + ln];
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), co = 32 * j + (ln & 31);
        if (i < 27 && co < cout) partial[((long)blockIdx.x * 27 + i) * cout + co] = v;
    }
}

// gw[i] (+)= sum over workgroups of partial[wg][i].  A workgroup adds 16 outputs: 16 lanes x 16 groups of partials, every thread's
// loads independent (a serial loop over the partials is a chain of HBM latencies: 512 of them took 90 us).
__global__ __launch_bounds__(256) void c3_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ gw, int nparts,
                                                              int count, int accumulate) {
    __shared__ float red[16][17];
    const int o = threadIdx.x & 15, grp = threadIdx.x >> 4, i = blockIdx.x * 16 + o;
    float s = 0.f;
    if (i < count) {
#pragma unroll 16
        for (int p = grp; p < nparts; p += 16) s += partial[(long)p * count + i];
    }
    red[grp][o] = s;
    __syncthreads();
    if (grp == 0 && i < count) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][o];
        if (accumulate) unsafeAtomicAdd(&gw[i], t);      // (a slot of a gradient arena may be added to from two streams at once)
        else gw[i] = t;
    }
}

}  // namespace

// Number of partial filters (27 x cout floats each) the caller provides as scratch.
extern "C" int cn_conv_wgrad_c3_partials(void) { return 512; }

// Filter gradient of a 3x3 convolution of a 3-channel fp32 image (g->cin == 3, stride 1 or 2, no dilation / upsample,
// cout <= 64 and a multiple of 4); gy in fp32 or bf16 (gy_dt).  scratch: cn_conv_wgrad_c3_partials() * 27 * cout floats.  accumulate: add to gw.
// Returns CN_EUNSUPPORTED (nothing launched) for other geometries.
extern "C" int cn_conv_wgrad_c3(const CnConvGeom* gp, const float* x, const void* gy, int gy_dt, float* scratch, float* gw,
                                int accumulate, void* stream) {
    CN_CHECK_ARG(gp && x && gy && scratch && gw && (gy_dt == CN_F32 || gy_dt == CN_BF16), "conv_wgrad_c3: bad args");
    const CnConvGeom& g = *gp;
    if (!(g.nd == 2 && g.cin == 3 && g.k_h == 3 && g.k_w == 3 && g.k_d == 1 && g.s_h == g.s_w && (g.s_h == 1 || g.s_h == 2) &&
          g.dl_h == 1 && g.dl_w == 1 && g.up == 0 && g.cout <= 64 && g.cout >= 4 && g.cout % 4 == 0))
        return CN_EUNSUPPORTED;
    CN_CHECK_ARG((double)g.n * g.out_h * g.out_w * g.cout < 2147483647.0, "tensor exceeds 2^31 elements");
    const int tiles_x = cn_cdiv(g.out_w, C3W_TILE), ntiles = g.n * g.out_h * tiles_x;
    const int nparts = cn_conv_wgrad_c3_partials();
    const int pw = C3W_TILE * g.s_h + 2, gpitch = (g.cout + 3) & ~3;
    size_t lds = sizeof(float) * 4 * (size_t)(3 * pw * 3 + C3W_TILE * gpitch);
    if (lds < sizeof(float) * 4 * 2 * 16 * 64) lds = sizeof(float) * 4 * 2 * 16 * 64;
    hipStream_t s = (hipStream_t)stream;
    cn_prof_begin(s, 2.0 * 27.0 * g.cout * (double)g.n * g.out_h * g.out_w,
                  4.0 * (double)g.n * g.in_h * g.in_w * 3 + (gy_dt == CN_BF16 ? 2.0 : 4.0) * g.n * g.out_h * g.out_w * g.cout + 4.0 * 27 * g.cout,
                  CN_FAM_C3_WGRAD);
#define C3W(S_, T_)                                                                                                              \
    do {                                                                                                                         \
        static bool attr_set = false;                                                                                            \
        if (!attr_set) {                                                                                                         \
            CN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(c3_wgrad_kernel<S_, T_>),                                   \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));                                  \
            attr_set = true;                                                                                                     \
        }                                                                                                                        \
        hipLaunchKernelGGL((c3_wgrad_kernel<S_, T_>), dim3(nparts), dim3(256), lds, s, g, x, (const T_*)gy, scratch, tiles_x,    \
                           ntiles);                                                                                              \
    } while (0)
    if (gy_dt == CN_F32) {
        if (g.s_h == 1) C3W(1, float); else C3W(2, float);
    } else {
        if (g.s_h == 1) C3W(1, bf16_t); else C3W(2, bf16_t);
    }
#undef C3W
    cn_prof_end(s);
    CN_LAUNCH_CHECK();
    const int count = 27 * g.cout;
    hipLaunchKernelGGL(c3_wgrad_reduce_kernel, dim3(cn_cdiv(count, 16)), dim3(256), 0, s, scratch, gw, nparts, count, accumulate);
    CN_LAUNCH_CHECK();
    return CN_OK;
}
