// upfold.hip -- filters of a convolution that follows a x2 nearest upsample, collapsed per output-parity class.
//
// keras UpSampling2D/3D + ConvND(k, SAME) (hologan_generator.py:139-170) reads, for output o = 2 i + p along one axis, the
// upsampled rows u = o - P + kk (kk < k, P = low padding), i.e. the STORED rows floor(u / 2): only 2 (k = 3) or 2-3 (k = 4)
// distinct ones, so the k taps of a parity class pre-sum to k2-ish taps on the stored grid -- an exact identity:
//     y[2 s + tau] += Wc(tau) x[s],   tau = o - 2 s in [tau_min, tau_min + k2),   Wc(tau) = sum of the w[kk] that map to tau.
// That is the data-gradient structure of a stride-2 convolution "conv2" (kernel k2, stride 2, low padding -tau_min) from the
// output grid to the stored grid, so the three GEMMs of the layer run on existing kernels with 8/27 (3-D, k = 3) or
// 6.25/16 (2-D, k = 4) of the multiply-adds and without the upsampled gradient tensor:
//     forward        = parity-ordered zero-stuffed convolution (dl = 2) with  wf[a'][ci][co] = Wc(tau_min + k2-1-a')
//     data gradient  = conv2 forward (stored extent, no sum-pool pass)   with  wd[a][co][ci]  = Wc(tau_min + a)^T
//     filter gradient= conv2 filter gradient gw2[a][co][ci], scattered back:  gw[kk][ci][co] = sum_{a ~ kk} gw2[a][co][ci]
// This file holds the two small maps between w (k taps) and the class filters (k2 taps per axis).
#include "common.h"

namespace {

struct UpfoldTab {
    int nd, k[3], k2[3];
    int a_of[3][4][2];      // per axis, original tap kk, parity p: the conv2 tap a = tau - tau_min it contributes to
};

__device__ __forceinline__ bool contributes(const UpfoldTab& t, int axis, int kk, int a) {
    return t.a_of[axis][kk][0] == a || t.a_of[axis][kk][1] == a;
}

// one thread per (a_d, a_h, a_w, ci, co) of the class filters
__global__ void upfold_weights_kernel(UpfoldTab t, const float* __restrict__ W, float* __restrict__ Wf, float* __restrict__ Wd,
                                      int cin, int cout) {
    const int T2 = t.k2[0] * t.k2[1] * t.k2[2];
    const long total = (long)T2 * cin * cout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int co = (int)(i % cout);
        long r = i / cout;
        const int ci = (int)(r % cin);
        const int a = (int)(r / cin);                       // linear conv2 tap (a_d, a_h, a_w)
        const int aw = a % t.k2[2], ah = (a / t.k2[2]) % t.k2[1], ad = a / (t.k2[2] * t.k2[1]);
        float s = 0.f;
        for (int kd = 0; kd < t.k[0]; ++kd) {
            if (!contributes(t, 0, kd, ad)) continue;
            for (int kh = 0; kh < t.k[1]; ++kh) {
                if (!contributes(t, 1, kh, ah)) continue;
                for (int kw = 0; kw < t.k[2]; ++kw) {
                    if (!contributes(t, 2, kw, aw)) continue;
                    s += W[((long)((kd * t.k[1] + kh) * t.k[2] + kw) * cin + ci) * cout + co];
                }
            }
        }
        if (Wd) Wd[((long)a * cout + co) * cin + ci] = s;
        if (Wf) {
            const int af = ((t.k2[0] - 1 - ad) * t.k2[1] + (t.k2[1] - 1 - ah)) * t.k2[2] + (t.k2[2] - 1 - aw);
            Wf[((long)af * cin + ci) * cout + co] = s;
        }
    }
}

// gw[kk][ci][co] (+)= sum over the conv2 taps a that tap kk contributes to of gw2[a][co][ci]
__global__ void upfold_wgrad_kernel(UpfoldTab t, const float* __restrict__ GW2, float* __restrict__ GW, int cin, int cout,
                                    int accumulate) {
    const int T = t.k[0] * t.k[1] * t.k[2];
    const long total = (long)T * cin * cout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % cin);                      // ci fastest: coalesced reads of gw2[a][co][:]
        long r = i / cin;
        const int co = (int)(r % cout);
        const int kk = (int)(r / cout);
        const int kw = kk % t.k[2], kh = (kk / t.k[2]) % t.k[1], kd = kk / (t.k[2] * t.k[1]);
        float s = 0.f;
        for (int pd = 0; pd < (t.nd == 3 ? 2 : 1); ++pd)
            for (int ph = 0; ph < 2; ++ph)
                for (int pw = 0; pw < 2; ++pw) {
                    const int ad = t.nd == 3 ? t.a_of[0][kd][pd] : 0, ah = t.a_of[1][kh][ph], aw = t.a_of[2][kw][pw];
                    s += GW2[((long)((ad * t.k2[1] + ah) * t.k2[2] + aw) * cout + co) * cin + ci];
                }
        float* dst = GW + ((long)kk * cin + ci) * cout + co;
        // accumulate = 1: atomic (the two branches of a forked step may add into one gradient-arena slot from two streams at
        // once); accumulate = 2: plain read-modify-write (the caller guarantees a single writer: 4x faster on the 3.5 M-entry filters)
        if (accumulate == 1) unsafeAtomicAdd(dst, s);
        else if (accumulate) *dst += s;
        else *dst = s;
    }
}

int make_tab(UpfoldTab& t, int nd, const int* k, const int* pad_lo) {
    t.nd = nd;
    for (int ax = 0; ax < 3; ++ax) {
        const int kk_n = k[ax], P = pad_lo[ax];
        t.k[ax] = kk_n;
        if (kk_n == 1 && ax < 3 - nd) {                     // unused depth axis of a 2-D layer
            t.k2[ax] = 1;
            for (int kk = 0; kk < 4; ++kk) t.a_of[ax][kk][0] = t.a_of[ax][kk][1] = 0;
            continue;
        }
        CN_CHECK_ARG(kk_n >= 2 && kk_n <= 4, "upfold: kernel extent %d not in 2..4", kk_n);
        int tau[4][2], tmin = 1 << 30, tmax = -(1 << 30);
        for (int kk = 0; kk < kk_n; ++kk)
            for (int p = 0; p < 2; ++p) {
                const int u = p - P + kk;                   // upsampled row read by output o = p (i = 0)
                const int s = u >= 0 ? u / 2 : -((-u + 1) / 2);   // floor(u / 2)
                tau[kk][p] = p - 2 * s;
                if (tau[kk][p] < tmin) tmin = tau[kk][p];
                if (tau[kk][p] > tmax) tmax = tau[kk][p];
            }
        t.k2[ax] = tmax - tmin + 1;
        for (int kk = 0; kk < 4; ++kk)
            for (int p = 0; p < 2; ++p) t.a_of[ax][kk][p] = kk < kk_n ? tau[kk][p] - tmin : -1;
    }
    return CN_OK;
}

}  // namespace

extern "C" int cn_upfold_weights(const float* w, float* wf, float* wd, int nd, const int* k3, const int* pad_lo3, int cin,
                                 int cout, int* k2_out3, int* pad2_out3, void* stream) {
    CN_CHECK_ARG(w && (wf || wd) && (nd == 2 || nd == 3) && k3 && pad_lo3 && cin > 0 && cout > 0, "upfold_weights: bad args");
    UpfoldTab t;
    if (int e = make_tab(t, nd, k3, pad_lo3)) return e;
    for (int ax = 0; ax < 3; ++ax) {
        if (k2_out3) k2_out3[ax] = t.k2[ax];
        // low padding of conv2 = -tau_min = k2 - 1 - (largest tau) ... recomputed from the table: tap a = 0 is tau_min
        if (pad2_out3) {
            int tmin = 1 << 30;
            if (t.k[ax] == 1 && ax < 3 - nd) tmin = 0;
            else
                for (int kk = 0; kk < t.k[ax]; ++kk)
                    for (int p = 0; p < 2; ++p) {
                        const int u = p - pad_lo3[ax] + kk;
                        const int s = u >= 0 ? u / 2 : -((-u + 1) / 2);
                        if (p - 2 * s < tmin) tmin = p - 2 * s;
                    }
            pad2_out3[ax] = -tmin;
        }
    }
    const long total = (long)t.k2[0] * t.k2[1] * t.k2[2] * cin * cout;
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(upfold_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, w, wf, wd, cin, cout);
    CN_LAUNCH_CHECK();
    return CN_OK;
}

extern "C" int cn_upfold_wgrad(const float* gw2, float* gw, int nd, const int* k3, const int* pad_lo3, int cin, int cout,
                               int accumulate, void* stream) {
    CN_CHECK_ARG(gw2 && gw && (nd == 2 || nd == 3) && k3 && pad_lo3 && cin > 0 && cout > 0, "upfold_wgrad: bad args");
    UpfoldTab t;
    if (int e = make_tab(t, nd, k3, pad_lo3)) return e;
    const long total = (long)t.k[0] * t.k[1] * t.k[2] * cin * cout;
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(upfold_wgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, gw2, gw, cin, cout, accumulate);
    CN_LAUNCH_CHECK();
    return CN_OK;
}
