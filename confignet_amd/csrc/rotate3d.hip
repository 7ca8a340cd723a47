// rotate3d.hip -- rigid 3-D resample of the (N,G,G,G,C) feature volume about its centre
// (transform_3d_grid_tf, confignet_utils.py:63-120): q = R (p - ctr) + ctr, clamp to [0,G-1],
// trilinear (x, then y, then z).  Coordinates are computed in-kernel from the 3x3 matrix; no
// coordinate tensors and no materialised gathers.  HBM-bound: 8 taps/voxel mostly from L2.
#include "common.h"

namespace {

struct Taps {
    int x0, x1, y0, y1, z0, z1;
    float dx, dy, dz;
    bool px, py, pz;   // clip_by_value passes the gradient (raw coordinate inside [0, G-1])
};

__device__ __forceinline__ Taps make_taps(const float* __restrict__ R, int p, int G) {
    const float ctr = 0.5f * (float)(G - 1);
    const int k = p % G, j = (p / G) % G, i = p / (G * G);
    const float px = (float)i - ctr, py = (float)j - ctr, pz = (float)k - ctr;
    float q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) q[a] = R[a * 3 + 0] * px + R[a * 3 + 1] * py + R[a * 3 + 2] * pz + ctr;
    Taps t;
    const float hi = (float)(G - 1);
    t.px = q[0] >= 0.f && q[0] <= hi;
    t.py = q[1] >= 0.f && q[1] <= hi;
    t.pz = q[2] >= 0.f && q[2] <= hi;
#pragma unroll
    for (int a = 0; a < 3; ++a) q[a] = fminf(fmaxf(q[a], 0.f), hi);
    const float fx = floorf(q[0]), fy = floorf(q[1]), fz = floorf(q[2]);
    t.x0 = (int)fx; t.y0 = (int)fy; t.z0 = (int)fz;
    t.x1 = min(t.x0 + 1, G - 1); t.y1 = min(t.y0 + 1, G - 1); t.z1 = min(t.z0 + 1, G - 1);
    t.dx = q[0] - fx; t.dy = q[1] - fy; t.dz = q[2] - fz;
    return t;
}

__device__ __forceinline__ float4 lerp4(float4 a, float4 b, float t) {
    const float s = 1.f - t;
    return make_float4(a.x * s + b.x * t, a.y * s + b.y * t, a.z * s + b.z * t, a.w * s + b.w * t);
}
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// grid (ceil(P*C4/256), N)
__global__ __launch_bounds__(256) void rotate3d_fwd_kernel(const float* __restrict__ grid, const float* __restrict__ rot,
                                                           float* __restrict__ out, int G, int C4) {
    const int n = blockIdx.y;
    const int P = G * G * G;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P * C4) return;
    const int c = i % C4, p = i / C4;
    const Taps t = make_taps(rot + n * 9, p, G);
    const float4* g4 = reinterpret_cast<const float4*>(grid) + (long)n * P * C4;
    auto at = [&](int x, int y, int z) { return g4[((long)(x * G + y) * G + z) * C4 + c]; };
    const float4 c00 = lerp4(at(t.x0, t.y0, t.z0), at(t.x1, t.y0, t.z0), t.dx);
    const float4 c01 = lerp4(at(t.x0, t.y0, t.z1), at(t.x1, t.y0, t.z1), t.dx);
    const float4 c10 = lerp4(at(t.x0, t.y1, t.z0), at(t.x1, t.y1, t.z0), t.dx);
    const float4 c11 = lerp4(at(t.x0, t.y1, t.z1), at(t.x1, t.y1, t.z1), t.dx);
    const float4 c0 = lerp4(c00, c10, t.dy);
    const float4 c1 = lerp4(c01, c11, t.dy);
    reinterpret_cast<float4*>(out)[((long)n * P + p) * C4 + c] = lerp4(c0, c1, t.dz);
}

__device__ __forceinline__ void atomic_add4(float* p, float4 v, float w) {
    unsafeAtomicAdd(p + 0, v.x * w);
    unsafeAtomicAdd(p + 1, v.y * w);
    unsafeAtomicAdd(p + 2, v.z * w);
    unsafeAtomicAdd(p + 3, v.w * w);
}

// grid (ceil(P/256), N): one thread per output voxel, loop over channel groups
__global__ __launch_bounds__(256) void rotate3d_bwd_kernel(const float* __restrict__ grid, const float* __restrict__ rot,
                                                           const float* __restrict__ gout, float* __restrict__ ggrid,
                                                           float* __restrict__ grot, int G, int C4) {
    const int n = blockIdx.y;
    const int P = G * G * G;
    const int p = blockIdx.x * 256 + threadIdx.x;
    float gq[3] = {0.f, 0.f, 0.f};
    float pc[3] = {0.f, 0.f, 0.f};
    if (p < P) {
        const Taps t = make_taps(rot + n * 9, p, G);
        const float ctr = 0.5f * (float)(G - 1);
        pc[0] = (float)(p / (G * G)) - ctr; pc[1] = (float)((p / G) % G) - ctr; pc[2] = (float)(p % G) - ctr;
        const float4* g4 = reinterpret_cast<const float4*>(grid) + (long)n * P * C4;
        float* gg = ggrid + (long)n * P * C4 * 4;
        const float4* go4 = reinterpret_cast<const float4*>(gout) + ((long)n * P + p) * C4;
        const float wx0 = 1.f - t.dx, wy0 = 1.f - t.dy, wz0 = 1.f - t.dz;
        const long o000 = ((long)(t.x0 * G + t.y0) * G + t.z0) * C4, o100 = ((long)(t.x1 * G + t.y0) * G + t.z0) * C4;
        const long o001 = ((long)(t.x0 * G + t.y0) * G + t.z1) * C4, o101 = ((long)(t.x1 * G + t.y0) * G + t.z1) * C4;
        const long o010 = ((long)(t.x0 * G + t.y1) * G + t.z0) * C4, o110 = ((long)(t.x1 * G + t.y1) * G + t.z0) * C4;
        const long o011 = ((long)(t.x0 * G + t.y1) * G + t.z1) * C4, o111 = ((long)(t.x1 * G + t.y1) * G + t.z1) * C4;
        for (int c = 0; c < C4; ++c) {
            const float4 go = go4[c];
            atomic_add4(gg + (o000 + c) * 4, go, wx0 * wy0 * wz0);
            atomic_add4(gg + (o100 + c) * 4, go, t.dx * wy0 * wz0);
            atomic_add4(gg + (o001 + c) * 4, go, wx0 * wy0 * t.dz);
            atomic_add4(gg + (o101 + c) * 4, go, t.dx * wy0 * t.dz);
            atomic_add4(gg + (o010 + c) * 4, go, wx0 * t.dy * wz0);
            atomic_add4(gg + (o110 + c) * 4, go, t.dx * t.dy * wz0);
            atomic_add4(gg + (o011 + c) * 4, go, wx0 * t.dy * t.dz);
            atomic_add4(gg + (o111 + c) * 4, go, t.dx * t.dy * t.dz);
            if (grot) {
                const float4 c000 = g4[o000 + c], c100 = g4[o100 + c], c001 = g4[o001 + c], c101 = g4[o101 + c];
                const float4 c010 = g4[o010 + c], c110 = g4[o110 + c], c011 = g4[o011 + c], c111 = g4[o111 + c];
                const float4 c00 = lerp4(c000, c100, t.dx), c01 = lerp4(c001, c101, t.dx);
                const float4 c10 = lerp4(c010, c110, t.dx), c11 = lerp4(c011, c111, t.dx);
                const float4 c0 = lerp4(c00, c10, t.dy), c1 = lerp4(c01, c11, t.dy);
                gq[2] += dot4(go, sub4(c1, c0));
                gq[1] += dot4(go, lerp4(sub4(c10, c00), sub4(c11, c01), t.dz));
                const float4 ex0 = lerp4(sub4(c100, c000), sub4(c110, c010), t.dy);
                const float4 ex1 = lerp4(sub4(c101, c001), sub4(c111, c011), t.dy);
                gq[0] += dot4(go, lerp4(ex0, ex1, t.dz));
            }
        }
        if (!t.px) gq[0] = 0.f;
        if (!t.py) gq[1] = 0.f;
        if (!t.pz) gq[2] = 0.f;
    }
    if (!grot) return;
    // block reduction of the 9 outer-product entries gq[a] * pc[b]
    __shared__ float sh[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            float v = gq[a] * pc[b];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            __syncthreads();
            if (lane == 0) sh[w] = v;
            __syncthreads();
            if (threadIdx.x == 0) unsafeAtomicAdd(&grot[n * 9 + a * 3 + b], sh[0] + sh[1] + sh[2] + sh[3]);
        }
}

// Backward through LDS: one workgroup owns (sample n, CC channels) and keeps the whole G^3 x CC gradient
// slab (<= 128 KiB of the CU's 160 KiB LDS) on chip; the 8-tap scatter of every output voxel is an LDS
// atomic add, the slab is written to HBM once with plain stores (no global atomics, no clearing pass).
constexpr int SLAB_FLOATS = 32768;
__global__ __launch_bounds__(256) void rotate3d_bwd_lds_kernel(const float* __restrict__ grid, const float* __restrict__ rot,
                                                               const float* __restrict__ gout, float* __restrict__ ggrid,
                                                               float* __restrict__ grot, int G, int C, int CC) {
    __shared__ float slab[SLAB_FLOATS];
    __shared__ float red[4];
    const int n = blockIdx.y, c0 = blockIdx.x * CC;
    const int P = G * G * G;
    const int cc = min(CC, C - c0);
    for (int i = threadIdx.x; i < P * CC; i += 256) slab[i] = 0.f;
    __syncthreads();
    const float* R = rot + n * 9;
    const float ctr = 0.5f * (float)(G - 1);
    float g9[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) g9[i] = 0.f;
    const float* gsrc = grid + (long)n * P * C;
    for (int p = threadIdx.x; p < P; p += 256) {
        const Taps t = make_taps(R, p, G);
        const float wx0 = 1.f - t.dx, wy0 = 1.f - t.dy, wz0 = 1.f - t.dz;
        const int o000 = (t.x0 * G + t.y0) * G + t.z0, o100 = (t.x1 * G + t.y0) * G + t.z0;
        const int o001 = (t.x0 * G + t.y0) * G + t.z1, o101 = (t.x1 * G + t.y0) * G + t.z1;
        const int o010 = (t.x0 * G + t.y1) * G + t.z0, o110 = (t.x1 * G + t.y1) * G + t.z0;
        const int o011 = (t.x0 * G + t.y1) * G + t.z1, o111 = (t.x1 * G + t.y1) * G + t.z1;
        const float* go = gout + ((long)n * P + p) * C + c0;
        float gq0 = 0.f, gq1 = 0.f, gq2 = 0.f;
        for (int c = 0; c < cc; ++c) {
            const float gv = go[c];
            atomicAdd(&slab[o000 * CC + c], gv * wx0 * wy0 * wz0);
            atomicAdd(&slab[o100 * CC + c], gv * t.dx * wy0 * wz0);
            atomicAdd(&slab[o001 * CC + c], gv * wx0 * wy0 * t.dz);
            atomicAdd(&slab[o101 * CC + c], gv * t.dx * wy0 * t.dz);
            atomicAdd(&slab[o010 * CC + c], gv * wx0 * t.dy * wz0);
            atomicAdd(&slab[o110 * CC + c], gv * t.dx * t.dy * wz0);
            atomicAdd(&slab[o011 * CC + c], gv * wx0 * t.dy * t.dz);
            atomicAdd(&slab[o111 * CC + c], gv * t.dx * t.dy * t.dz);
            if (grot) {
                const float c000 = gsrc[(long)o000 * C + c0 + c], c100 = gsrc[(long)o100 * C + c0 + c];
                const float c001 = gsrc[(long)o001 * C + c0 + c], c101 = gsrc[(long)o101 * C + c0 + c];
                const float c010 = gsrc[(long)o010 * C + c0 + c], c110 = gsrc[(long)o110 * C + c0 + c];
                const float c011 = gsrc[(long)o011 * C + c0 + c], c111 = gsrc[(long)o111 * C + c0 + c];
                const float c00 = c000 * wx0 + c100 * t.dx, c01 = c001 * wx0 + c101 * t.dx;
                const float c10 = c010 * wx0 + c110 * t.dx, c11 = c011 * wx0 + c111 * t.dx;
                const float cc0 = c00 * wy0 + c10 * t.dy, cc1 = c01 * wy0 + c11 * t.dy;
                gq2 += gv * (cc1 - cc0);
                gq1 += gv * ((c10 - c00) * wz0 + (c11 - c01) * t.dz);
                const float ex0 = (c100 - c000) * wy0 + (c110 - c010) * t.dy;
                const float ex1 = (c101 - c001) * wy0 + (c111 - c011) * t.dy;
                gq0 += gv * (ex0 * wz0 + ex1 * t.dz);
            }
        }
        if (grot) {
            if (!t.px) gq0 = 0.f;
            if (!t.py) gq1 = 0.f;
            if (!t.pz) gq2 = 0.f;
            const float pc0 = (float)(p / (G * G)) - ctr, pc1 = (float)((p / G) % G) - ctr, pc2 = (float)(p % G) - ctr;
            g9[0] += gq0 * pc0; g9[1] += gq0 * pc1; g9[2] += gq0 * pc2;
            g9[3] += gq1 * pc0; g9[4] += gq1 * pc1; g9[5] += gq1 * pc2;
            g9[6] += gq2 * pc0; g9[7] += gq2 * pc1; g9[8] += gq2 * pc2;
        }
    }
    __syncthreads();
    float* dst = ggrid + (long)n * P * C + c0;
    for (int i = threadIdx.x; i < P * cc; i += 256) {
        const int p = i / cc, c = i - p * cc;
        dst[(long)p * C + c] = slab[p * CC + c];
    }
    if (grot) {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            float v = g9[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            __syncthreads();
            if (lane == 0) red[w] = v;
            __syncthreads();
            if (threadIdx.x == 0) unsafeAtomicAdd(&grot[n * 9 + i], red[0] + red[1] + red[2] + red[3]);
        }
    }
}

// Deterministic backward (cn_set_deterministic): ONE wave owns (sample n, 8 channels) and walks the output voxels in index
// order; lane = (tap, channel) -- the eight taps of a voxel go to eight different addresses of the wave's LDS slab, except
// where a clamped coordinate makes two taps coincide, and there the coinciding tap has weight exactly 0 (q clamped onto the
// border => fractional part 0), so the order of those two adds cannot change the sum.  Contributions of different voxels to
// one address are added in voxel order.  The rotation-matrix gradient goes to per-workgroup partials (added in order after).
__global__ __launch_bounds__(64) void rotate3d_bwd_det_kernel(const float* __restrict__ grid, const float* __restrict__ rot,
                                                              const float* __restrict__ gout, float* __restrict__ ggrid,
                                                              float* __restrict__ grot_parts, int G, int C) {
    __shared__ float slab[SLAB_FLOATS];                 // [voxel][8 channels]
    const int n = blockIdx.y, c0 = blockIdx.x * 8;
    const int P = G * G * G;
    const int lane = threadIdx.x, tap = lane >> 3, ch = lane & 7;
    const bool live = c0 + ch < C;
    for (int i = lane; i < P * 8; i += 64) slab[i] = 0.f;
    __syncthreads();
    const float* R = rot + n * 9;
    const float ctr = 0.5f * (float)(G - 1);
    const float* gsrc = grid + (long)n * P * C;
    float g9[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) g9[i] = 0.f;
    constexpr int U = 4;                                // voxels per trip: their loads are in flight together
    for (int p0 = 0; p0 < P; p0 += U) {
        Taps t[U];
        int o[U];
        float w[U], gv[U], src[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = min(p0 + u, P - 1);
            t[u] = make_taps(R, p, G);
            const int xs = (tap & 1) ? t[u].x1 : t[u].x0, ys = (tap & 2) ? t[u].y1 : t[u].y0, zs = (tap & 4) ? t[u].z1 : t[u].z0;
            w[u] = ((tap & 1) ? t[u].dx : 1.f - t[u].dx) * ((tap & 2) ? t[u].dy : 1.f - t[u].dy) * ((tap & 4) ? t[u].dz : 1.f - t[u].dz);
            o[u] = (xs * G + ys) * G + zs;
            gv[u] = (live && p0 + u < P) ? gout[((long)n * P + p) * C + c0 + ch] : 0.f;
            src[u] = (live && grot_parts) ? gsrc[(long)o[u] * C + c0 + ch] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = p0 + u;
            if (p >= P) break;
            atomicAdd(&slab[o[u] * 8 + ch], gv[u] * w[u]);      // (LDS; distinct addresses within the wave but for zero-weight twins)
            if (grot_parts) {
                // d out / d q through the eight source values of this lane's channel: every lane needs all eight -> shuffles
                float c8[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) c8[k] = __shfl(src[u], k * 8 + ch, 64);
                if (tap == 0) {
                    const Taps& tt = t[u];
                    const float wx0 = 1.f - tt.dx, wy0 = 1.f - tt.dy, wz0 = 1.f - tt.dz;
                    // tap bit 0 = x, bit 1 = y, bit 2 = z
                    const float c000 = c8[0], c100 = c8[1], c010 = c8[2], c110 = c8[3], c001 = c8[4], c101 = c8[5], c011 = c8[6], c111 = c8[7];
                    const float c00 = c000 * wx0 + c100 * tt.dx, c01 = c001 * wx0 + c101 * tt.dx;
                    const float c10 = c010 * wx0 + c110 * tt.dx, c11 = c011 * wx0 + c111 * tt.dx;
                    const float cc0 = c00 * wy0 + c10 * tt.dy, cc1 = c01 * wy0 + c11 * tt.dy;
                    float gq2 = gv[u] * (cc1 - cc0);
                    float gq1 = gv[u] * ((c10 - c00) * wz0 + (c11 - c01) * tt.dz);
                    const float ex0 = (c100 - c000) * wy0 + (c110 - c010) * tt.dy;
                    const float ex1 = (c101 - c001) * wy0 + (c111 - c011) * tt.dy;
                    float gq0 = gv[u] * (ex0 * wz0 + ex1 * tt.dz);
                    if (!tt.px) gq0 = 0.f;
                    if (!tt.py) gq1 = 0.f;
                    if (!tt.pz) gq2 = 0.f;
                    const float pc0 = (float)(p / (G * G)) - ctr, pc1 = (float)((p / G) % G) - ctr, pc2 = (float)(p % G) - ctr;
                    g9[0] += gq0 * pc0; g9[1] += gq0 * pc1; g9[2] += gq0 * pc2;
                    g9[3] += gq1 * pc0; g9[4] += gq1 * pc1; g9[5] += gq1 * pc2;
                    g9[6] += gq2 * pc0; g9[7] += gq2 * pc1; g9[8] += gq2 * pc2;
                }
            }
        }
    }
    __syncthreads();
    float* dst = ggrid + (long)n * P * C + c0;
    for (int i = lane; i < P * 8; i += 64) {
        const int p = i >> 3, c = i & 7;
        if (c0 + c < C) dst[(long)p * C + c] = slab[i];
    }
    if (grot_parts) {
        // lanes 0..7 (tap 0) hold one channel each: add them in channel order
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            float v = 0.f;
            for (int k = 0; k < 8; ++k) v += __shfl(g9[i], k, 64);
            if (lane == 0) grot_parts[((long)blockIdx.x * gridDim.y + n) * 9 + i] = v;     // [channel block][n][9]
        }
    }
}

}  // namespace

namespace {
// euler_angles_to_matrix (confignet_utils.py:122-145) and its transposed Jacobian: one thread per sample.  (As 9 products of
// sines and cosines in torch this was ~110 launches per generator pass forward and ~150 backward, on 8-element vectors.)
__global__ void euler_matrix_kernel(const float* __restrict__ a, float* __restrict__ R, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s0, c0, s1, c1, s2, c2;
    sincosf(a[3 * i + 0], &s0, &c0);
    sincosf(a[3 * i + 1], &s1, &c1);
    sincosf(a[3 * i + 2], &s2, &c2);
    float* r = R + 9 * i;
    r[0] = c2 * c1;                 r[1] = -s2;      r[2] = c2 * s1;
    r[3] = s0 * s1 + c0 * c1 * s2;  r[4] = c0 * c2;  r[5] = c0 * s2 * s1 - c1 * s0;
    r[6] = c1 * s0 * s2 - c0 * s1;  r[7] = c2 * s0;  r[8] = c0 * c1 + s0 * s1 * s2;
}

__global__ void euler_matrix_bwd_kernel(const float* __restrict__ a, const float* __restrict__ gR, float* __restrict__ ga, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s0, c0, s1, c1, s2, c2;
    sincosf(a[3 * i + 0], &s0, &c0);
    sincosf(a[3 * i + 1], &s1, &c1);
    sincosf(a[3 * i + 2], &s2, &c2);
    const float* g = gR + 9 * i;
    ga[3 * i + 0] = g[3] * (c0 * s1 - s0 * c1 * s2) + g[4] * (-s0 * c2) + g[5] * (-s0 * s2 * s1 - c1 * c0) +
                    g[6] * (c1 * c0 * s2 + s0 * s1) + g[7] * (c2 * c0) + g[8] * (-s0 * c1 + c0 * s1 * s2);
    ga[3 * i + 1] = g[0] * (-c2 * s1) + g[2] * (c2 * c1) + g[3] * (s0 * c1 - c0 * s1 * s2) + g[5] * (c0 * s2 * c1 + s1 * s0) +
                    g[6] * (-s1 * s0 * s2 - c0 * c1) + g[8] * (-c0 * s1 + s0 * c1 * s2);
    ga[3 * i + 2] = g[0] * (-s2 * c1) + g[1] * (-c2) + g[2] * (-s2 * s1) + g[3] * (c0 * c1 * c2) + g[4] * (-c0 * s2) +
                    g[5] * (c0 * c2 * s1) + g[6] * (c1 * s0 * c2) + g[7] * (-s2 * s0) + g[8] * (s0 * s1 * c2);
}
}  // namespace

extern "C" int cn_euler_matrix(const float* angles, float* rot, int n, void* stream) {
    CN_CHECK_ARG(angles && rot && n > 0, "euler_matrix: bad args");
    hipLaunchKernelGGL(euler_matrix_kernel, dim3(cn_cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, angles, rot, n);
    CN_LAUNCH_CHECK();
    return CN_OK;
}

extern "C" int cn_euler_matrix_bwd(const float* angles, const float* grot, float* gangles, int n, void* stream) {
    CN_CHECK_ARG(angles && grot && gangles && n > 0, "euler_matrix_bwd: bad args");
    hipLaunchKernelGGL(euler_matrix_bwd_kernel, dim3(cn_cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, angles, grot, gangles, n);
    CN_LAUNCH_CHECK();
    return CN_OK;
}

extern "C" int cn_rotate3d_fwd(const float* grid, const float* rot, float* out, int n, int g, int c, void* stream) {
    CN_CHECK_ARG(grid && rot && out && n > 0 && g > 1 && c > 0 && c % 4 == 0, "rotate3d: bad args (c %% 4 == 0 required)");
    const long work = (long)g * g * g * (c / 4);
    hipLaunchKernelGGL(rotate3d_fwd_kernel, dim3(cn_cdiv(work, 256), n), dim3(256), 0, (hipStream_t)stream, grid, rot, out, g, c / 4);
    CN_LAUNCH_CHECK();
    return CN_OK;
}

extern "C" int cn_rotate3d_bwd(const float* grid, const float* rot, const float* gout, float* ggrid, float* grot, int n,
                               int g, int c, void* stream) {
    CN_CHECK_ARG(grid && rot && gout && ggrid && n > 0 && g > 1 && c > 0 && c % 4 == 0, "rotate3d_bwd: bad args");
    hipStream_t s = (hipStream_t)stream;
    const long P = (long)g * g * g;
    if (cn_det() && P * 8 <= SLAB_FLOATS) {
        const int cblocks = cn_cdiv(c, 8);
        float* parts = nullptr;
        if (grot) {
            parts = cn_det_ws(s, (size_t)cblocks * n * 9);
            if (!parts) return CN_EINVAL;
        }
        hipLaunchKernelGGL(rotate3d_bwd_det_kernel, dim3(cblocks, n), dim3(64), 0, s, grid, rot, gout, ggrid, parts, g, c);
        CN_LAUNCH_CHECK();
        if (grot) return cn_sum_parts(parts, grot, cblocks, (long)n * 9, 0, 1.f, s);
        return CN_OK;
    }
    if (grot) {
        if (int ez__ = cn_zero_async(grot, sizeof(float) * n * 9, s)) return ez__;
    }
    if (P * 4 <= SLAB_FLOATS) {
        int CC = (int)(SLAB_FLOATS / P);
        if (CC > c) CC = c;
        CC = CC / 4 * 4;
        // fewer channels per workgroup while that still adds workgroups on idle CUs (n = 8, C = 128: 128 -> 256 workgroups)
        while (CC > 4 && (long)cn_cdiv(c, CC) * n < 256) CC -= 4;
        hipLaunchKernelGGL(rotate3d_bwd_lds_kernel, dim3(cn_cdiv(c, CC), n), dim3(256), 0, s, grid, rot, gout, ggrid, grot, g, c, CC);
    } else {
        if (int ez__ = cn_zero_async(ggrid, sizeof(float) * n * P * c, s)) return ez__;
        hipLaunchKernelGGL(rotate3d_bwd_kernel, dim3(cn_cdiv(P, 256), n), dim3(256), 0, s, grid, rot, gout, ggrid, grot, g, c / 4);
    }
    CN_LAUNCH_CHECK();
    return CN_OK;
}
