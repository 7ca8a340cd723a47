// norm_coef.hip -- the O(N*C) coefficient algebra of AdaIN / instance norm / style statistics as single
// tiny kernels (one thread per (n,c), or per c for the parameter gradients that reduce over n).
#include "common.h"
#include "typed.h"

namespace {

__global__ void norm_coef_fwd_kernel(int mode, const float* __restrict__ s1, const float* __restrict__ s2,
                                     const float* __restrict__ p1, const float* __restrict__ p2, float* __restrict__ A,
                                     float* __restrict__ B, float* __restrict__ sm, float* __restrict__ sr, int N, int C,
                                     float invS, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    const float mu = s1[i] * invS;
    const float var = fmaxf(s2[i] * invS - mu * mu, 0.f);
    sm[i] = mu;
    if (mode == 0) {
        const float r = rsqrtf(var + eps);
        const float a = r * (p1[n * 2 * C + c] + 1.f);
        sr[i] = r;
        A[i] = a;
        B[i] = p1[n * 2 * C + C + c] - mu * a;
    } else if (mode == 1) {
        const float q = 1.f / (sqrtf(var) + eps);
        const float a = p1[c] * q;
        sr[i] = q;
        A[i] = a;
        B[i] = p2[c] - mu * a;
    } else {
        const float sd = sqrtf(var + eps);
        sr[i] = sd;
        A[n * 2 * C + c] = mu;
        A[n * 2 * C + C + c] = sd;
    }
}

__global__ void norm_coef_bwd_kernel(int mode, const float* __restrict__ t1, const float* __restrict__ t2,
                                     const float* __restrict__ sm, const float* __restrict__ sr,
                                     const float* __restrict__ p1, float* __restrict__ c1, float* __restrict__ c2,
                                     float* __restrict__ c0, float* __restrict__ gp1, int N, int C, float invS, float eps,
                                     int coef_blocks, float* __restrict__ ggamma, float* __restrict__ gbeta) {
    if ((int)blockIdx.x >= coef_blocks) {
        // instance norm (mode 1): the workgroups behind the coefficient ones reduce the parameter gradients over n, one channel per
        // thread in sample order (round 6: was a second launch, inorm_param_grad_kernel -- same arithmetic, same bits)
        const int c = ((int)blockIdx.x - coef_blocks) * blockDim.x + threadIdx.x;
        if (c >= C) return;
        float gg = 0.f, gb = 0.f;
        for (int n = 0; n < N; ++n) {
            const int i = n * C + c;
            gg += sr[i] * (t2[i] - sm[i] * t1[i]);
            gb += t1[i];
        }
        ggamma[c] = gg;
        gbeta[c] = gb;
        return;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    const float mu = sm[i];
    if (mode == 0) {
        // y = xhat*(s+1)+b ; gx = r(s+1) gy - r(s+1) mean(gy) - r xhat (s+1) gs/S
        const float r = sr[i], sp1 = p1[n * 2 * C + c] + 1.f;
        const float gs = r * (t2[i] - mu * t1[i]);
        const float k1 = r * sp1;
        const float k2 = -r * r * sp1 * gs * invS;
        c1[i] = k1;
        c2[i] = k2;
        c0[i] = -k1 * t1[i] * invS - k2 * mu;
        gp1[n * 2 * C + c] = gs;
        gp1[n * 2 * C + C + c] = t1[i];
    } else if (mode == 1) {
        // y = (a-mu) q gamma + beta, q = 1/(sigma+eps): ga = q g^ - q mean(g^) - q^2 G/(S sigma) (a-mu)
        const float q = sr[i], gam = p1[c];
        const float sigma = fmaxf(1.f / q - eps, 1e-20f);
        const float G = gam * (t2[i] - mu * t1[i]);
        const float k1 = q * gam;
        const float k2 = -q * q * G * invS / sigma;
        c1[i] = k1;
        c2[i] = k2;
        c0[i] = -k1 * t1[i] * invS - k2 * mu;
    } else {
        // style = [mu | sd]: gx = gmu/S + gsd (x-mu)/(S sd)
        const float sd = sr[i];
        const float gmu = t1[n * 2 * C + c], gsd = t1[n * 2 * C + C + c];
        const float k2 = gsd * invS / sd;
        c2[i] = k2;
        c0[i] = gmu * invS - k2 * mu;
    }
}

}  // namespace

extern "C" int cn_norm_coef_fwd(int mode, const float* s1, const float* s2, const float* p1, const float* p2, float* A,
                                float* B, float* save_mean, float* save_r, int n, int c, int S, float eps, void* stream) {
    CN_CHECK_ARG(mode >= 0 && mode <= 2 && s1 && s2 && A && save_mean && save_r && n > 0 && c > 0 && S > 0, "norm_coef_fwd: bad args");
    CN_CHECK_ARG(mode == 2 || (p1 && B), "norm_coef_fwd: missing parameter tensor");
    CN_CHECK_ARG(mode != 1 || p2, "norm_coef_fwd: instance norm needs beta");
    hipLaunchKernelGGL(norm_coef_fwd_kernel, dim3(cn_cdiv((long)n * c, 256)), dim3(256), 0, (hipStream_t)stream, mode, s1, s2, p1,
                       p2, A, B, save_mean, save_r, n, c, 1.f / (float)S, eps);
    CN_LAUNCH_CHECK();
    return CN_OK;
}

extern "C" int cn_norm_coef_bwd(int mode, const float* t1, const float* t2, const float* save_mean, const float* save_r,
                                const float* p1, float* c1, float* c2, float* c0, float* gp1, float* gp2, int n, int c, int S,
                                float eps, void* stream) {
    CN_CHECK_ARG(mode >= 0 && mode <= 2 && t1 && save_mean && save_r && c2 && c0 && n > 0 && c > 0 && S > 0, "norm_coef_bwd: bad args");
    CN_CHECK_ARG(mode == 2 || (t2 && c1 && p1 && gp1), "norm_coef_bwd: missing tensor");
    CN_CHECK_ARG(mode != 1 || gp2, "norm_coef_bwd: instance norm needs dbeta");
    hipStream_t s = (hipStream_t)stream;
    const int coef_blocks = cn_cdiv((long)n * c, 256), param_blocks = mode == 1 ? cn_cdiv(c, 256) : 0;
    hipLaunchKernelGGL(norm_coef_bwd_kernel, dim3(coef_blocks + param_blocks), dim3(256), 0, s, mode, t1, t2, save_mean, save_r,
                       p1, c1, c2, c0, mode == 0 ? gp1 : nullptr, n, c, 1.f / (float)S, eps, coef_blocks, mode == 1 ? gp1 : nullptr,
                       mode == 1 ? gp2 : nullptr);
    CN_LAUNCH_CHECK();
    return CN_OK;
}

// =============================================================================================
// R1 penalty without a second-order tape.  d/dtheta |g|^2 with g = d out/d x equals
// 2 * d/dtheta JVP_x(out)(v) at v = g held constant, so the penalty's gradient is an ordinary first-order
// backward pass through a TANGENT forward pass of the discriminator.  These kernels are the O(N*C)
// coefficient algebra of the tangent ("dual") DiscrBlock tail (style statistics + LeakyReLU + instance norm):
//   a = lrelu(x), ta = lrelu'(x) tx, tmu = mean(ta), P = mean((a-mu) ta), tsigma = P/sigma, tq = -q^2 tsigma
//   ty = gamma (q (ta - tmu) + tq (a - mu))            = C1*ta + C2*a + C0
//   tstyle = [mean(tx) | mean((x-m) tx)/sd]
// and of its backward (cotangents h of ty, u of tstyle):
//   g_tx = lrelu'(x) (K1*h + K2*a + K0) + D2*x + D0
//   g_x  = lrelu'(x) (kh*h + kt*ta + ka*a + kc) + et*tx + ex*x + e0       (the second-order terms)
// =============================================================================================
namespace {

__global__ void dual_coef_fwd_kernel(const float* __restrict__ T1, const float* __restrict__ T2,
                                     const float* __restrict__ U1, const float* __restrict__ U2,
                                     const float* __restrict__ mean, const float* __restrict__ qv,
                                     const float* __restrict__ sm, const float* __restrict__ ssd,
                                     const float* __restrict__ gamma, float* __restrict__ C1, float* __restrict__ C2,
                                     float* __restrict__ C0, float* __restrict__ tstyle, int N, int C, float invS, float eps,
                                     int n_style, int period) {
    // rows [0, n_style): style branch (U*, tstyle indexed by the row); rows [n_style, N): instance-norm branch (T*, C* indexed
    // by row - n_style); the primal statistics hold `period` samples and row r of either branch reads sample r % period
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    if (n >= n_style) {
        if (!C1) return;
        const int r = n - n_style;
        const int j = r * C + c, p = (r % period) * C + c;
        const float mu = mean[p], q = qv[p], g = gamma[c];
        const float sigma = fmaxf(1.f / q - eps, 1e-20f);
        const float tmu = T1[j] * invS;
        const float P = T2[j] * invS - mu * tmu;
        const float tq = -q * q * P / sigma;
        C1[j] = g * q;
        C2[j] = g * tq;
        C0[j] = -g * q * tmu - g * tq * mu;
    } else if (tstyle) {
        const int p = (n % period) * C + c;
        const float m = sm[p], sd = ssd[p];
        const float tm = U1[i] * invS;
        tstyle[n * 2 * C + c] = tm;
        tstyle[n * 2 * C + C + c] = (U2[i] * invS - m * tm) / sd;
    }
}

struct DualBwdOut {
    float *K1, *K2, *K0, *D2, *D0, *kh, *kt, *ka, *kc, *et, *ex, *e0, *ggamma, *ggamma_rows;
};

__global__ void dual_coef_bwd_kernel(const float* __restrict__ H1, const float* __restrict__ H2p,
                                     const float* __restrict__ E, const float* __restrict__ u,
                                     const float* __restrict__ T1, const float* __restrict__ T2,
                                     const float* __restrict__ U1, const float* __restrict__ U2,
                                     const float* __restrict__ mean, const float* __restrict__ qv,
                                     const float* __restrict__ sm, const float* __restrict__ ssd,
                                     const float* __restrict__ gamma, DualBwdOut o, int N, int C, float invS, float eps,
                                     int n_style, int period) {
    // row layout as in dual_coef_fwd_kernel
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    if (n >= n_style) {
        if (!H1) return;
        const int r = n - n_style;
        i = r * C + c;
        const int p = (r % period) * C + c;
        const float mu = mean[p], q = qv[p], g = gamma[c];
        const float sigma = fmaxf(1.f / q - eps, 1e-20f);
        const float tmu = T1[i] * invS;
        const float P = T2[i] * invS - mu * tmu;
        const float tq = -q * q * P / sigma;
        const float h1 = H1[i], h2 = H2p[i] - mu * h1, e = E[i];
        // adjoint of the tangent map (same form as the first-order instance-norm backward)
        const float k2 = -g * q * q * h2 * invS / sigma;
        o.K1[i] = g * q;
        o.K2[i] = k2;
        o.K0[i] = -g * q * h1 * invS - k2 * mu;
        // second-order terms w.r.t. the primal activation a
        const float ka = g * (-q * q * (e - tmu * h1) + h2 * (2.f * q * q * q * P / sigma + q * q * P / (sigma * sigma))) * invS / sigma;
        const float k0 = (g * h2 * q * q * tmu / sigma - g * tq * h1) * invS;
        o.kh[i] = g * tq;
        o.kt[i] = -g * h2 * q * q * invS / sigma;
        o.ka[i] = ka;
        o.kc[i] = k0 - ka * mu;
        o.ggamma_rows[i] = q * e - q * tmu * h1 + tq * h2;      // summed over the rows in row order afterwards (no atomics)
    } else if (u) {
        const int p = (n % period) * C + c;
        const float m = sm[p], sd = ssd[p];
        const float um = u[n * 2 * C + c], usd = u[n * 2 * C + C + c];
        const float tm = U1[i] * invS;
        const float Q = U2[i] * invS - m * tm;
        const float d2 = usd * invS / sd;
        o.D2[i] = d2;
        o.D0[i] = um * invS - d2 * m;
        const float ex = -usd * Q * invS / (sd * sd * sd);
        o.et[i] = usd * invS / sd;
        o.ex[i] = ex;
        o.e0[i] = -usd * tm * invS / sd - ex * m;
    }
}

// g_x = lrelu'(x) (kh*h + kt*ta + ka*lrelu(x) + kc) + et*tx + ex*x + e0 ; any of the two groups may be absent
template <typename T>
__global__ void dual_gx_kernel(const T* __restrict__ h, const T* __restrict__ ta, const T* __restrict__ tx,
                               const T* __restrict__ x, const float* __restrict__ kh, const float* __restrict__ kt,
                               const float* __restrict__ ka, const float* __restrict__ kc, const float* __restrict__ et,
                               const float* __restrict__ ex, const float* __restrict__ e0, T* __restrict__ out,
                               long total4, int S, int C, float slope, int nrep, const float* __restrict__ K1 = nullptr,
                               const float* __restrict__ K2 = nullptr, const float* __restrict__ K0 = nullptr,
                               const float* __restrict__ D2 = nullptr, const float* __restrict__ D0 = nullptr,
                               T* __restrict__ out_tx = nullptr, int ta_is_tx = 0) {
    // ta_is_tx: `ta` points at the tangent INPUT rows of the heads (tx behind its first N samples); ta = lrelu'(x) tx is formed here
    // out_tx (round 6, cn_dual_tail_gx_tx): the gradient w.r.t. the stacked tangent input from the SAME pass over h and x --
    // rows [0, N): D2 x + D0 (the head that left through the style statistics), rows [(1 + j) N, (2 + j) N): lrelu'(x) (K1 h_j + K2
    // lrelu(x) + K0) -- what two cn_nc_lin2 launches (one more read of h) computed before
    // total4 = N*S*C/4 elements of x / out; h, ta and their coefficients hold nrep*N samples (the heads of a batched tangent
    // pass that share this primal activation): their contributions are summed here, in head order
    const int C4 = C / 4;
    const long NC = (total4 / S / C4) * (long)C;         // N * C: coefficient rows per head
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % C4);
        const int n = (int)((i / C4) / S);
        const long ci = (long)n * C + (long)cg * 4;
        const float4 xv = ld4<T>(x + 4 * i);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        float r[4] = {0.f, 0.f, 0.f, 0.f};
        if (kh) {
            for (int j = 0; j < nrep; ++j) {
                const float4 hv = ld4<T>(h + 4 * (i + j * total4));
                const float4 tv = ld4<T>(ta + 4 * (i + j * total4));
                const float hs[4] = {hv.x, hv.y, hv.z, hv.w};
                float ts[4] = {tv.x, tv.y, tv.z, tv.w};
                if (ta_is_tx) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) ts[e] *= xs[e] > 0.f ? 1.f : slope;
                }
                const long cj = ci + j * NC;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float mk = xs[e] > 0.f ? 1.f : slope;
                    r[e] += mk * (kh[cj + e] * hs[e] + kt[cj + e] * ts[e] + ka[cj + e] * xs[e] * mk + kc[cj + e]);
                }
                if (out_tx) {
                    float w[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // (the operation order of nc_lin2_rows_kernel with flags 2 | 4: bias, + a1 x1, + a2 lrelu(x2), times lrelu'(x2))
                        float v = K0[cj + e];
                        v += K1[cj + e] * hs[e];
                        v += K2[cj + e] * (xs[e] > 0.f ? xs[e] : xs[e] * slope);
                        w[e] = v * (xs[e] > 0.f ? 1.f : slope);
                    }
                    st4<T>(out_tx + 4 * (i + (long)(j + 1) * total4), make_float4(w[0], w[1], w[2], w[3]));
                }
            }
        }
        if (et) {
            const float4 tv = ld4<T>(tx + 4 * i);
            const float ts[4] = {tv.x, tv.y, tv.z, tv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] += et[ci + e] * ts[e] + ex[ci + e] * xs[e] + e0[ci + e];
        }
        if (out_tx) {
            float w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = D0[ci + e] + D2[ci + e] * xs[e];
            st4<T>(out_tx + 4 * i, make_float4(w[0], w[1], w[2], w[3]));
        }
        st4<T>(out + 4 * i, make_float4(r[0], r[1], r[2], r[3]));
    }
}

}  // namespace

extern "C" int cn_dual_tail_coef_fwd(const float* T1, const float* T2, const float* U1, const float* U2, const float* mean,
                                     const float* q, const float* sm, const float* ssd, const float* gamma, float* C1,
                                     float* C2, float* C0, float* tstyle, int n, int c, int S, float eps, int n_style, int period,
                                     void* stream) {
    CN_CHECK_ARG(n > 0 && c > 0 && S > 0 && (C1 || tstyle) && n_style >= 0 && n_style <= n && period > 0, "dual_tail_coef_fwd: bad args");
    CN_CHECK_ARG((n_style == 0 || tstyle) && (n_style == n || C1), "dual_tail_coef_fwd: a row range without its output tensors");
    CN_CHECK_ARG(!C1 || (T1 && T2 && mean && q && gamma && C2 && C0), "dual_tail_coef_fwd: missing instance-norm tensors");
    CN_CHECK_ARG(!tstyle || (U1 && U2 && sm && ssd), "dual_tail_coef_fwd: missing style tensors");
    hipLaunchKernelGGL(dual_coef_fwd_kernel, dim3(cn_cdiv((long)n * c, 256)), dim3(256), 0, (hipStream_t)stream, T1, T2, U1, U2,
                       mean, q, sm, ssd, gamma, C1, C2, C0, tstyle, n, c, 1.f / (float)S, eps, n_style, period);
    CN_LAUNCH_CHECK();
    return CN_OK;
}

extern "C" int cn_dual_tail_coef_bwd(const float* H1, const float* H2p, const float* E, const float* u, const float* T1,
                                     const float* T2, const float* U1, const float* U2, const float* mean, const float* q,
                                     const float* sm, const float* ssd, const float* gamma, float* const* out13, int n, int c,
                                     int S, float eps, int n_style, int period, void* stream) {
    // out13[13] (a 14th entry): scratch of (n - n_style) * c floats for the per-row dgamma terms
    CN_CHECK_ARG(n > 0 && c > 0 && S > 0 && out13 && (H1 || u) && n_style >= 0 && n_style <= n && period > 0, "dual_tail_coef_bwd: bad args");
    CN_CHECK_ARG((n_style == 0 || u) && (n_style == n || H1), "dual_tail_coef_bwd: a row range without its input tensors");
    DualBwdOut o;
    o.K1 = out13[0]; o.K2 = out13[1]; o.K0 = out13[2]; o.D2 = out13[3]; o.D0 = out13[4];
    o.kh = out13[5]; o.kt = out13[6]; o.ka = out13[7]; o.kc = out13[8];
    o.et = out13[9]; o.ex = out13[10]; o.e0 = out13[11]; o.ggamma = out13[12]; o.ggamma_rows = out13[13];
    CN_CHECK_ARG(!H1 || (H2p && E && T1 && T2 && mean && q && gamma && o.K1 && o.K2 && o.K0 && o.kh && o.kt && o.ka && o.kc && o.ggamma && o.ggamma_rows),
                 "dual_tail_coef_bwd: missing instance-norm tensors");
    CN_CHECK_ARG(!u || (U1 && U2 && sm && ssd && o.D2 && o.D0 && o.et && o.ex && o.e0), "dual_tail_coef_bwd: missing style tensors");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(dual_coef_bwd_kernel, dim3(cn_cdiv((long)n * c, 256)), dim3(256), 0, s, H1, H2p, E, u, T1, T2, U1, U2, mean,
                       q, sm, ssd, gamma, o, n, c, 1.f / (float)S, eps, n_style, period);
    CN_LAUNCH_CHECK();
    if (H1) return cn_sum_parts(o.ggamma_rows, o.ggamma, n - n_style, c, 0, 1.f, s);
    return CN_OK;
}

extern "C" int cn_dual_tail_gx(const void* h, const void* ta, const void* tx, const void* x, const float* kh,
                               const float* kt, const float* ka, const float* kc, const float* et, const float* ex,
                               const float* e0, void* out, int n, int s, int c, float slope, int nrep, int dt, void* stream) {
    CN_CHECK_ARG(x && out && n > 0 && s > 0 && c > 0 && c % 4 == 0 && (kh || et) && nrep >= 1 && (dt == CN_F32 || dt == CN_BF16), "dual_tail_gx: bad args");
    CN_CHECK_ARG(!kh || (h && ta && kt && ka && kc), "dual_tail_gx: missing instance-norm tensors");
    CN_CHECK_ARG(!et || (tx && ex && e0), "dual_tail_gx: missing style tensors");
    const long total4 = (long)n * s * (c / 4);
    long blocks = (total4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    CN_DISPATCH_DT(dt, hipLaunchKernelGGL((dual_gx_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const T*)h,
                                          (const T*)ta, (const T*)tx, (const T*)x, kh, kt, ka, kc, et, ex, e0, (T*)out, total4, s, c, slope, nrep));
    CN_LAUNCH_CHECK();
    return CN_OK;
}

// cn_dual_tail_gx + the gradient w.r.t. the stacked tangent input in the same pass (round 6): out_tx holds (1 + nrep) N samples --
// the style head's rows D2 x + D0 first, then per head lrelu'(x) (K1 h + K2 lrelu(x) + K0) -- instead of two cn_nc_lin2 launches
// that read h and x again (losses.py:75-82, the R1 penalty's tangent pass through building_blocks.py:100-106).  ta_is_tx: `ta` holds the
// heads' tangent INPUT rows and ta = lrelu'(x) tx is formed in the pass (the forward pass then never stores ta).
extern "C" int cn_dual_tail_gx_tx(const void* h, const void* ta, const void* tx, const void* x, const float* kh, const float* kt,
                                  const float* ka, const float* kc, const float* et, const float* ex, const float* e0,
                                  const float* K1, const float* K2, const float* K0, const float* D2, const float* D0, void* out,
                                  void* out_tx, int n, int s, int c, float slope, int nrep, int ta_is_tx, int dt, void* stream) {
    CN_CHECK_ARG(h && ta && tx && x && out && out_tx && kh && kt && ka && kc && et && ex && e0 && K1 && K2 && K0 && D2 && D0,
                 "dual_tail_gx_tx: NULL tensor");
    CN_CHECK_ARG(n > 0 && s > 0 && c > 0 && c % 4 == 0 && nrep >= 1 && (dt == CN_F32 || dt == CN_BF16), "dual_tail_gx_tx: bad args");
    const long total4 = (long)n * s * (c / 4);
    long blocks = (total4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    CN_DISPATCH_DT(dt, hipLaunchKernelGGL((dual_gx_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const T*)h,
                                          (const T*)ta, (const T*)tx, (const T*)x, kh, kt, ka, kc, et, ex, e0, (T*)out, total4, s, c, slope, nrep,
                                          K1, K2, K0, D2, D0, (T*)out_tx, ta_is_tx));
    CN_LAUNCH_CHECK();
    return CN_OK;
}
