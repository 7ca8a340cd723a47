// norm_coef.hip -- the O(N*C) coefficient algebra of AdaIN / instance norm / style statistics as single
// tiny kernels (one thread per (n,c), or per c for the parameter gradients that reduce over n).
#include "common.h"

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
                                     float* __restrict__ c0, float* __restrict__ gp1, int N, int C, float invS, float eps) {
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

// instance-norm parameter gradients: reduce over n
__global__ void inorm_param_grad_kernel(const float* __restrict__ t1, const float* __restrict__ t2,
                                        const float* __restrict__ sm, const float* __restrict__ sr,
                                        float* __restrict__ ggamma, float* __restrict__ gbeta, int N, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float gg = 0.f, gb = 0.f;
    for (int n = 0; n < N; ++n) {
        const int i = n * C + c;
        gg += sr[i] * (t2[i] - sm[i] * t1[i]);
        gb += t1[i];
    }
    ggamma[c] = gg;
    gbeta[c] = gb;
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
    hipLaunchKernelGGL(norm_coef_bwd_kernel, dim3(cn_cdiv((long)n * c, 256)), dim3(256), 0, s, mode, t1, t2, save_mean, save_r,
                       p1, c1, c2, c0, mode == 0 ? gp1 : nullptr, n, c, 1.f / (float)S, eps);
    CN_LAUNCH_CHECK();
    if (mode == 1) {
        hipLaunchKernelGGL(inorm_param_grad_kernel, dim3(cn_cdiv(c, 256)), dim3(256), 0, s, t1, t2, save_mean, save_r, gp1, gp2, n, c);
        CN_LAUNCH_CHECK();
    }
    return CN_OK;
}
