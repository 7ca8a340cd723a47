// elementwise.hip -- HBM-bound kernels of the path: per-(sample,channel) statistics and affine
// maps (LayerNorm/AdaIN, instance norm, style statistics, BN inference, bias gradients, global
// average pool), activations, pooling, loss reductions, image pre/post-processing, Adam+EMA.
// All are single-pass, float4-vectorised over the channel (fastest) axis where C % 4 == 0.
#include "common.h"
#include "typed.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ float lrelu(float v, float s) { return v > 0.f ? v : v * s; }
__device__ __forceinline__ float4 lrelu4(float4 v, float s) {
    return make_float4(lrelu(v.x, s), lrelu(v.y, s), lrelu(v.z, s), lrelu(v.w, s));
}

// derivative of an activation evaluated from its OUTPUT o
__device__ __forceinline__ float act_deriv(float o, int act, float slope) {
    if (act == CN_ACT_LRELU) return o > 0.f ? 1.f : slope;
    if (act == CN_ACT_RELU) return o > 0.f ? 1.f : 0.f;
    if (act == CN_ACT_TANH) return 1.f - o * o;
    return 1.f;
}

// ---- nc_reduce: (N,S,C) -> (N,C) sums.  grid (cblk, sblk, n); block (TX c-groups, TY rows) ----
template <int V, typename T>   // V = 4: 4-wide channel groups, V = 1: scalar channels; T: storage type of x1 / x2
__global__ __launch_bounds__(256) void nc_reduce_kernel(const T* __restrict__ x1, const T* __restrict__ x2,
                                                        float* __restrict__ s1, float* __restrict__ s2, int S, int C,
                                                        int rows_per_block, int flags, float slope, int period2,
                                                        T* __restrict__ dact_out = nullptr, int dact = 0,
                                                        float* __restrict__ parts = nullptr, const T* __restrict__ x3 = nullptr,
                                                        const float* __restrict__ coef = nullptr, T* __restrict__ scaled_out = nullptr,
                                                        int dact_on = 0) {
    // dact_on: fused activation backward -- a = x1 * act'(x2) (x2 = the activation's OUTPUT); a is written to dact_out (if given),
    // coef[c] * a to scaled_out (if given: the input gradient of an inference-mode BatchNorm), s1 = sum a, s2 = sum a * (x3 if
    // given, else f2(x2)): the whole backward of conv -> BN(inference) -> ReLU in one pass (cn_bn_act_bwd)
    const int CG = C / V;                          // channel groups
    const int tx = threadIdx.x, ty = threadIdx.y, TX = blockDim.x, TY = blockDim.y;
    const int cg = blockIdx.x * TX + tx;
    const int n = blockIdx.z;
    const int sbeg = blockIdx.y * rows_per_block, send = min(S, sbeg + rows_per_block);
    float a1[V], a2[V];
#pragma unroll
    for (int e = 0; e < V; ++e) a1[e] = a2[e] = 0.f;
    if (cg < CG) {
        const long base = (long)n * S * C + (long)cg * V;
        const long base2 = (long)(period2 ? n % period2 : n) * S * C + (long)cg * V;     // x2 may hold fewer samples (tiled)
        int s = sbeg + ty;
        if (V == 4) {
            // four rows in flight per thread: with ~2 workgroups per CU (the same-address atomics of the epilogue bound the workgroup
            // count) one dependent 16-byte load per array kept 16 KB per CU on the wire -- 3.4 TB/s on the largest tensors, 5.6 now
            for (; s + 3 * TY < send; s += 4 * TY) {
                float4 va[4], vb[4], vc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) va[u] = ld4<T>(x1 + base + (long)(s + u * TY) * C);
                if (x2) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) vb[u] = ld4<T>(x2 + base2 + (long)(s + u * TY) * C);
                }
                if (dact_on && x3) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) vc[u] = ld4<T>(x3 + base + (long)(s + u * TY) * C);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float4 a = va[u], b = vb[u];
                    if (flags & 1) a = lrelu4(a, slope);
                    if (x2) { if (flags & 2) b = lrelu4(b, slope); }
                    if (dact_on) {
                        a.x *= act_deriv(b.x, dact, slope); a.y *= act_deriv(b.y, dact, slope);
                        a.z *= act_deriv(b.z, dact, slope); a.w *= act_deriv(b.w, dact, slope);
                        if (dact_out) st4<T>(dact_out + base + (long)(s + u * TY) * C, a);
                        if (scaled_out) {
                            const float4 k4 = *reinterpret_cast<const float4*>(coef + (long)cg * V);
                            st4<T>(scaled_out + base + (long)(s + u * TY) * C, make_float4(a.x * k4.x, a.y * k4.y, a.z * k4.z, a.w * k4.w));
                        }
                        if (x3) b = vc[u];
                    }
                    if (!x2) b = a;
                    a1[0] += a.x; a1[1 % V] += a.y; a1[2 % V] += a.z; a1[3 % V] += a.w;
                    a2[0] += a.x * b.x; a2[1 % V] += a.y * b.y; a2[2 % V] += a.z * b.z; a2[3 % V] += a.w * b.w;
                }
            }
        }
        for (; s < send; s += TY) {
            float a[V], b[V];
            if (V == 4) {
                float4 va = ld4<T>(x1 + base + (long)s * C);
                if (flags & 1) va = lrelu4(va, slope);
                a[0] = va.x; a[1 % V] = va.y; a[2 % V] = va.z; a[3 % V] = va.w;
                if (x2) {
                    float4 vb = ld4<T>(x2 + base2 + (long)s * C);
                    if (flags & 2) vb = lrelu4(vb, slope);
                    b[0] = vb.x; b[1 % V] = vb.y; b[2 % V] = vb.z; b[3 % V] = vb.w;
                }
                if (dact_on) {
#pragma unroll
                    for (int e = 0; e < V; ++e) a[e] *= act_deriv(b[e], dact, slope);
                    if (dact_out) st4<T>(dact_out + base + (long)s * C, make_float4(a[0], a[1 % V], a[2 % V], a[3 % V]));
                    if (scaled_out) {
                        const float4 k4 = *reinterpret_cast<const float4*>(coef + (long)cg * V);
                        st4<T>(scaled_out + base + (long)s * C, make_float4(a[0] * k4.x, a[1 % V] * k4.y, a[2 % V] * k4.z, a[3 % V] * k4.w));
                    }
                    if (x3) {
                        const float4 vc = ld4<T>(x3 + base + (long)s * C);
                        b[0] = vc.x; b[1 % V] = vc.y; b[2 % V] = vc.z; b[3 % V] = vc.w;
                    }
                }
            } else {
                a[0] = ldf<T>(x1 + base + (long)s * C);
                if (flags & 1) a[0] = lrelu(a[0], slope);
                if (x2) {
                    b[0] = ldf<T>(x2 + base2 + (long)s * C);
                    if (flags & 2) b[0] = lrelu(b[0], slope);
                }
                if (dact_on) {
                    a[0] *= act_deriv(b[0], dact, slope);
                    if (dact_out) stf<T>(dact_out + base + (long)s * C, a[0]);
                    if (scaled_out) stf<T>(scaled_out + base + (long)s * C, a[0] * coef[cg]);
                    if (x3) b[0] = ldf<T>(x3 + base + (long)s * C);
                }
            }
#pragma unroll
            for (int e = 0; e < V; ++e) {
                a1[e] += a[e];
                a2[e] += a[e] * (x2 ? b[e] : a[e]);
            }
        }
    }
    __shared__ float red[2][256 * V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        red[0][(ty * TX + tx) * V + e] = a1[e];
        red[1][(ty * TX + tx) * V + e] = a2[e];
    }
    __syncthreads();
    if (ty == 0 && cg < CG) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float t1 = 0.f, t2 = 0.f;
            for (int y = 0; y < TY; ++y) {
                t1 += red[0][(y * TX + tx) * V + e];
                t2 += red[1][(y * TX + tx) * V + e];
            }
            const long o = (long)n * C + (long)cg * V + e;
            if (parts) {                           // deterministic mode: per-row-block partials, added in block order afterwards
                const long nc = (long)gridDim.z * C;
                if (s1) parts[(long)blockIdx.y * nc + o] = t1;
                if (s2) parts[((long)gridDim.y + blockIdx.y) * nc + o] = t2;
            } else if (gridDim.y == 1) {           // the only workgroup of this (n, channel block): plain stores
                if (s1) s1[o] = t1;
                if (s2) s2[o] = t2;
            } else {
                if (s1) unsafeAtomicAdd(&s1[o], t1);
                if (s2) unsafeAtomicAdd(&s2[o], t2);
            }
        }
    }
}

// ---- nc_reduce4: the four statistics of a DiscrBlock's tail in ONE pass over its pre-activation tensor ----
// out[0] = sum x, out[1] = sum x^2 (the layer style, confignet_utils.py:147-159), out[2] = sum l, out[3] = sum l^2 with
// l = leaky_relu(x) (the instance normalisation behind the activation, building_blocks.py:100-106); out is (4, N, C), zeroed by
// the caller unless gridDim.y == 1.  Same grid as nc_reduce_kernel<4>.
template <typename T>
__global__ __launch_bounds__(256) void nc_reduce4_kernel(const T* __restrict__ x, float* __restrict__ out, int S, int C,
                                                         int rows_per_block, float slope, float* __restrict__ parts) {
    const int CG = C / 4;
    const int tx = threadIdx.x, ty = threadIdx.y, TX = blockDim.x, TY = blockDim.y;
    const int cg = blockIdx.x * TX + tx;
    const int n = blockIdx.z;
    const int sbeg = blockIdx.y * rows_per_block, send = min(S, sbeg + rows_per_block);
    float acc[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[q][e] = 0.f;
    if (cg < CG) {
        const long base = (long)n * S * C + (long)cg * 4;
        int s = sbeg + ty;
        for (; s + 3 * TY < send; s += 4 * TY) {       // four rows in flight (see nc_reduce_kernel)
            float4 vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) vv[u] = ld4<T>(x + base + (long)(s + u * TY) * C);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 l = lrelu4(vv[u], slope);
                const float a[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w}, b[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0][e] += a[e];
                    acc[1][e] += a[e] * a[e];
                    acc[2][e] += b[e];
                    acc[3][e] += b[e] * b[e];
                }
            }
        }
        for (; s < send; s += TY) {
            const float4 v = ld4<T>(x + base + (long)s * C);
            const float4 l = lrelu4(v, slope);
            const float a[4] = {v.x, v.y, v.z, v.w}, b[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0][e] += a[e];
                acc[1][e] += a[e] * a[e];
                acc[2][e] += b[e];
                acc[3][e] += b[e] * b[e];
            }
        }
    }
    __shared__ float red[4][256 * 4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[q][(ty * TX + tx) * 4 + e] = acc[q][e];
    __syncthreads();
    if (ty == 0 && cg < CG) {
        const long nc = (long)gridDim.z * C;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = 0.f;
                for (int y = 0; y < TY; ++y) t += red[q][(y * TX + tx) * 4 + e];
                const long o = (long)n * C + (long)cg * 4 + e;
                if (parts) parts[((long)q * gridDim.y + blockIdx.y) * nc + o] = t;     // deterministic mode: ordered second pass
                else if (gridDim.y == 1) out[q * nc + o] = t;
                else unsafeAtomicAdd(&out[q * nc + o], t);
            }
    }
}

// ---- nc_reduce_hxt: the three reductions of the R1 tangent tail's backward pass in ONE pass over its cotangent (round 6) ----
// out[0] = sum h, out[1] = sum h lrelu(x), out[2] = sum h ta per (n, c); h and ta hold gridDim.z samples, x `period` samples
// (read through the sample period); out is (3, N, C), zeroed by the caller unless gridDim.y == 1.  Grid as nc_reduce_kernel<4>.
template <typename T>
__global__ __launch_bounds__(256) void nc_reduce_hxt_kernel(const T* __restrict__ h, const T* __restrict__ x, const T* __restrict__ ta,
                                                            float* __restrict__ out, int S, int C, int rows_per_block, float slope,
                                                            int period, float* __restrict__ parts, int ta_is_tx) {
    // ta_is_tx: the third operand is the tangent INPUT tx; ta = lrelu'(x) tx is formed here (it is never stored)
    const int CG = C / 4;
    const int tx = threadIdx.x, ty = threadIdx.y, TX = blockDim.x, TY = blockDim.y;
    const int cg = blockIdx.x * TX + tx;
    const int n = blockIdx.z;
    const int sbeg = blockIdx.y * rows_per_block, send = min(S, sbeg + rows_per_block);
    float acc[3][4];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[q][e] = 0.f;
    if (cg < CG) {
        const long base = (long)n * S * C + (long)cg * 4;
        const long base2 = (long)(n % period) * S * C + (long)cg * 4;
        int s = sbeg + ty;
        for (; s + 3 * TY < send; s += 4 * TY) {       // four rows in flight (see nc_reduce_kernel)
            float4 vh[4], vx[4], vt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) vh[u] = ld4<T>(h + base + (long)(s + u * TY) * C);
#pragma unroll
            for (int u = 0; u < 4; ++u) vx[u] = ld4<T>(x + base2 + (long)(s + u * TY) * C);
#pragma unroll
            for (int u = 0; u < 4; ++u) vt[u] = ld4<T>(ta + base + (long)(s + u * TY) * C);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 l = lrelu4(vx[u], slope);
                const float a[4] = {vh[u].x, vh[u].y, vh[u].z, vh[u].w}, b[4] = {l.x, l.y, l.z, l.w},
                            xr[4] = {vx[u].x, vx[u].y, vx[u].z, vx[u].w};
                float t[4] = {vt[u].x, vt[u].y, vt[u].z, vt[u].w};
                if (ta_is_tx) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] *= xr[e] > 0.f ? 1.f : slope;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0][e] += a[e];
                    acc[1][e] += a[e] * b[e];
                    acc[2][e] += a[e] * t[e];
                }
            }
        }
        for (; s < send; s += TY) {
            const float4 vh = ld4<T>(h + base + (long)s * C), vxr = ld4<T>(x + base2 + (long)s * C), l = lrelu4(vxr, slope),
                         vt = ld4<T>(ta + base + (long)s * C);
            const float a[4] = {vh.x, vh.y, vh.z, vh.w}, b[4] = {l.x, l.y, l.z, l.w}, xr[4] = {vxr.x, vxr.y, vxr.z, vxr.w};
            float t[4] = {vt.x, vt.y, vt.z, vt.w};
            if (ta_is_tx) {
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] *= xr[e] > 0.f ? 1.f : slope;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0][e] += a[e];
                acc[1][e] += a[e] * b[e];
                acc[2][e] += a[e] * t[e];
            }
        }
    }
    __shared__ float red[3][256 * 4];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[q][(ty * TX + tx) * 4 + e] = acc[q][e];
    __syncthreads();
    if (ty == 0 && cg < CG) {
        const long nc = (long)gridDim.z * C;
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = 0.f;
                for (int y = 0; y < TY; ++y) t += red[q][(y * TX + tx) * 4 + e];
                const long o = (long)n * C + (long)cg * 4 + e;
                if (parts) parts[((long)q * gridDim.y + blockIdx.y) * nc + o] = t;     // deterministic mode: ordered second pass
                else if (gridDim.y == 1) out[q * nc + o] = t;
                else unsafeAtomicAdd(&out[q * nc + o], t);
            }
    }
}

// ---- nc_lin2: y = A1*f1(x1) + A2*f2(x2) + B ----
template <int V, typename T>
__global__ __launch_bounds__(256) void nc_lin2_kernel(const T* __restrict__ x1, const float* __restrict__ a1,
                                                      const T* __restrict__ x2, const float* __restrict__ a2,
                                                      const float* __restrict__ bb, const float* __restrict__ a3,
                                                      const float* __restrict__ b3, T* __restrict__ y, long total_g,
                                                      int S, int C, int cstride, int flags, float slope, int period2) {
    const int CG = C / V;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total_g; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % CG);
        const long row = i / CG;
        const int n = (int)(row / S);
        const long ci = (long)n * cstride + (long)cg * V;
        const long i2 = period2 ? (((long)(n % period2) * S + (row - (long)n * S)) * CG + cg) : i;   // x2 tiled over samples
        float r[V];
#pragma unroll
        for (int e = 0; e < V; ++e) r[e] = bb ? bb[ci + e] : 0.f;
        float raw2[V];
        if (x2) {
#pragma unroll
            for (int e = 0; e < V; ++e) raw2[e] = ldf<T>(x2 + i2 * V + e);
        }
        if (x1) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float v = ldf<T>(x1 + i * V + e);
                if (flags & 1) v = lrelu(v, slope);
                if ((flags & 16) && x2) v *= raw2[e] > 0.f ? 1.f : slope;      // x1 lrelu'(x2): a tangent through the activation, not stored
                r[e] += (a1 ? a1[ci + e] : 1.f) * v;
            }
        }
        if (x2) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float v = raw2[e];
                if (flags & 2) v = lrelu(v, slope);
                r[e] += (a2 ? a2[ci + e] : 1.f) * v;
            }
            if (flags & 4) {
#pragma unroll
                for (int e = 0; e < V; ++e) r[e] *= raw2[e] > 0.f ? 1.f : slope;
            }
            if (a3) {
#pragma unroll
                for (int e = 0; e < V; ++e) r[e] += a3[ci + e] * raw2[e] + (b3 ? b3[ci + e] : 0.f);
            }
        }
        if (flags & 8) {
#pragma unroll
            for (int e = 0; e < V; ++e) r[e] = fmaxf(r[e], 0.f);
        }
        if (V == 4) st4<T>(y + i * 4, make_float4(r[0], r[1 % V], r[2 % V], r[3 % V]));
        else stf<T>(y + i, r[0]);
    }
}

// nc_lin2 for tensors big enough to care: grid (gx, n) with gx*256 a multiple of the channel-group count, so a
// thread keeps ONE channel group for its whole grid-stride walk through a sample: no per-element div/mod (the
// 64-bit ones of the generic kernel cost more than the 48 bytes they index) and the (n, c) coefficients live in
// registers.  G = float4 groups per sample.
template <int V, typename T>
__global__ __launch_bounds__(256) void nc_lin2_rows_kernel(const T* __restrict__ x1, const float* __restrict__ a1,
                                                           const T* __restrict__ x2, const float* __restrict__ a2,
                                                           const float* __restrict__ bb, const float* __restrict__ a3,
                                                           const float* __restrict__ b3, T* __restrict__ y, int G,
                                                           int CG, int cstride, int flags, float slope, int period2) {
    const int t0 = blockIdx.x * 256 + threadIdx.x;
    const int adv = gridDim.x * 256;
    const int cg = t0 % CG;
    const long ci = (long)blockIdx.y * cstride + (long)cg * V;
    float k1[V], k2[V], kb[V], k3[V], kb3[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        k1[e] = a1 ? a1[ci + e] : 1.f;
        k2[e] = a2 ? a2[ci + e] : 1.f;
        kb[e] = bb ? bb[ci + e] : 0.f;
        k3[e] = a3 ? a3[ci + e] : 0.f;
        kb3[e] = (a3 && b3) ? b3[ci + e] : 0.f;
    }
    const long base = (long)blockIdx.y * G;
    const long base2 = (long)(period2 ? blockIdx.y % period2 : blockIdx.y) * G;      // x2 tiled over samples
    const bool f1 = flags & 1, f2 = flags & 2, fm = flags & 4, fr = flags & 8, fd = flags & 16;
    for (int j = t0; j < G; j += adv) {
        const long i = (base + j) * V;
        const long i2 = (base2 + j) * V;
        float v1[V], v2[V], r[V];
        if (V == 4) {
            if (x1) {
                const float4 t = ld4<T>(x1 + i);
                v1[0] = t.x; v1[1 % V] = t.y; v1[2 % V] = t.z; v1[3 % V] = t.w;
            }
            if (x2) {
                const float4 t = ld4<T>(x2 + i2);
                v2[0] = t.x; v2[1 % V] = t.y; v2[2 % V] = t.z; v2[3 % V] = t.w;
            }
        } else {
            if (x1) v1[0] = ldf<T>(x1 + i);
            if (x2) v2[0] = ldf<T>(x2 + i2);
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            r[e] = kb[e];
            if (x1) {
                float u = f1 ? lrelu(v1[e], slope) : v1[e];
                if (fd && x2) u *= v2[e] > 0.f ? 1.f : slope;                   // x1 lrelu'(x2)
                r[e] += k1[e] * u;
            }
            if (x2) {
                r[e] += k2[e] * (f2 ? lrelu(v2[e], slope) : v2[e]);
                if (fm) r[e] *= v2[e] > 0.f ? 1.f : slope;
                if (a3) r[e] += k3[e] * v2[e] + kb3[e];
            }
            if (fr) r[e] = fmaxf(r[e], 0.f);
        }
        if (V == 4) st4<T>(y + i, make_float4(r[0], r[1 % V], r[2 % V], r[3 % V]));
        else stf<T>(y + i, r[0]);
    }
}

// ---- norm_apply: nc_lin2_rows_kernel<4> with the coefficient algebra of cn_norm_coef_fwd / _bwd INLINE (round 6) ----
// Every thread of the apply pass owns one 4-channel group of one sample and used to LOAD its (n, c) coefficients, which a
// one-thread-per-(n, c) launch had just computed from the statistics; it now computes them itself (a dozen flops) and the
// coefficient launch is gone: AdaIn / instance norm apply and backward are reduce + this, two launches instead of three.
// dir 0 (forward):  y = k1 f1(x1) + kb,  (k1, kb) from (s1, s2, p1, p2) as norm_coef_fwd_kernel modes 0 / 1; the threads of
//   workgroup column 0 also store mean / r for the backward pass.
// dir 1 (backward): gx = (k1 gy + k2 f2(x) + kb) [lrelu'(x)] [+ a3 x + b3],  (k1, k2, kb) from (t1, t2, mean, r, p1) as
//   norm_coef_bwd_kernel; workgroup column 0 stores d[s|b] (mode 0); workgroup (0, 0) reduces d gamma / d beta over the samples
//   in sample order (mode 1) -- the same arithmetic and order as the separate kernels.
struct NormApplyArgs {
    int mode, dir, N, C;
    float invS, eps;
    const float* s1; const float* s2;      // dir 0: sum x, sum x^2 (of f1(x1));  dir 1: t1 = sum gy, t2 = sum gy f2(x)
    const float* p1; const float* p2;      // mode 0: [s | b] (N, 2C);  mode 1: gamma, beta (C)
    float* sm; float* sr;                  // dir 0: written;  dir 1: read
    float* gp1; float* gp2;                // dir 1: mode 0 d[s | b] (N, 2C);  mode 1: d gamma, d beta (C)
};

template <typename T>
__global__ __launch_bounds__(256) void norm_apply_rows_kernel(NormApplyArgs A, const T* __restrict__ x1, const T* __restrict__ x2,
                                                              const float* __restrict__ a3, const float* __restrict__ b3,
                                                              T* __restrict__ y, int G, int CG, int flags, float slope, int period2) {
    constexpr int V = 4;
    const int t0 = blockIdx.x * 256 + threadIdx.x;
    const int adv = gridDim.x * 256;
    const int cg = t0 % CG;
    const int n = blockIdx.y, C = A.C;
    const long ci = (long)n * C + (long)cg * V;
    float k1[V], k2[V], kb[V], k3[V], kb3[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const int c = cg * V + e;
        const long i = ci + e;
        k2[e] = 0.f;
        if (A.dir == 0) {
            const float mu = A.s1[i] * A.invS;
            const float var = fmaxf(A.s2[i] * A.invS - mu * mu, 0.f);
            float rr, a;
            if (A.mode == 0) {
                rr = rsqrtf(var + A.eps);
                a = rr * (A.p1[(long)n * 2 * C + c] + 1.f);
                kb[e] = A.p1[(long)n * 2 * C + C + c] - mu * a;
            } else {
                rr = 1.f / (sqrtf(var) + A.eps);
                a = A.p1[c] * rr;
                kb[e] = A.p2[c] - mu * a;
            }
            k1[e] = a;
            if (t0 < CG) { A.sm[i] = mu; A.sr[i] = rr; }
        } else {
            const float mu = A.sm[i], t1 = A.s1[i], t2 = A.s2[i];
            if (A.mode == 0) {
                const float r = A.sr[i], sp1 = A.p1[(long)n * 2 * C + c] + 1.f;
                const float gs = r * (t2 - mu * t1);
                const float q1 = r * sp1;
                const float q2 = -r * r * sp1 * gs * A.invS;
                k1[e] = q1; k2[e] = q2;
                kb[e] = -q1 * t1 * A.invS - q2 * mu;
                if (t0 < CG) { A.gp1[(long)n * 2 * C + c] = gs; A.gp1[(long)n * 2 * C + C + c] = t1; }
            } else {
                const float q = A.sr[i], gam = A.p1[c];
                const float sigma = fmaxf(1.f / q - A.eps, 1e-20f);
                const float Gm = gam * (t2 - mu * t1);
                const float q1 = q * gam;
                const float q2 = -q * q * Gm * A.invS / sigma;
                k1[e] = q1; k2[e] = q2;
                kb[e] = -q1 * t1 * A.invS - q2 * mu;
            }
        }
        k3[e] = a3 ? a3[i] : 0.f;
        kb3[e] = (a3 && b3) ? b3[i] : 0.f;
    }
    if (A.dir == 1 && A.mode == 1 && blockIdx.x == 0 && blockIdx.y == 0 && A.gp1) {
        // d gamma / d beta: one thread per channel, samples in order (norm_coef_bwd_kernel's parameter workgroups)
        for (int c = threadIdx.x; c < C; c += 256) {
            float gg = 0.f, gb = 0.f;
            for (int m = 0; m < A.N; ++m) {
                const long i = (long)m * C + c;
                gg += A.sr[i] * (A.s2[i] - A.sm[i] * A.s1[i]);
                gb += A.s1[i];
            }
            A.gp1[c] = gg;
            A.gp2[c] = gb;
        }
    }
    const long base = (long)blockIdx.y * G;
    const long base2 = (long)(period2 ? blockIdx.y % period2 : blockIdx.y) * G;
    const bool f1 = flags & 1, f2 = flags & 2, fm = flags & 4;
    for (int j = t0; j < G; j += adv) {
        const long i = (base + j) * V;
        const long i2 = (base2 + j) * V;
        float v1[V], v2[V], r[V];
        {
            const float4 t = ld4<T>(x1 + i);
            v1[0] = t.x; v1[1] = t.y; v1[2] = t.z; v1[3] = t.w;
        }
        if (x2) {
            const float4 t = ld4<T>(x2 + i2);
            v2[0] = t.x; v2[1] = t.y; v2[2] = t.z; v2[3] = t.w;
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            r[e] = kb[e];
            r[e] += k1[e] * (f1 ? lrelu(v1[e], slope) : v1[e]);
            if (x2) {
                r[e] += k2[e] * (f2 ? lrelu(v2[e], slope) : v2[e]);
                if (fm) r[e] *= v2[e] > 0.f ? 1.f : slope;
                if (a3) r[e] += k3[e] * v2[e] + kb3[e];
            }
        }
        st4<T>(y + i, make_float4(r[0], r[1], r[2], r[3]));
    }
}

// ---- streaming maps: one float4 per lane per trip when the pointers are 16-byte aligned (they are for every tensor
// the host side allocates), scalar tail / fallback otherwise.  VEC is decided by the launcher. ----

template <bool VEC, typename T>
__global__ void act_fwd_kernel(const T* x, T* y, size_t n, int act, float slope) {   // may run in place
    const size_t stride = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (VEC) {
        for (size_t i = t; i < n / 4; i += stride) {
            float4 v = ld4<T>(x + 4 * i);
            v.x = cn_apply_act(v.x, act, slope); v.y = cn_apply_act(v.y, act, slope);
            v.z = cn_apply_act(v.z, act, slope); v.w = cn_apply_act(v.w, act, slope);
            st4<T>(y + 4 * i, v);
        }
        for (size_t i = (n / 4) * 4 + t; i < n; i += stride) stf<T>(y + i, cn_apply_act(ldf<T>(x + i), act, slope));
    } else {
        for (size_t i = t; i < n; i += stride) stf<T>(y + i, cn_apply_act(ldf<T>(x + i), act, slope));
    }
}

template <bool VEC, typename T>
__global__ void act_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ y, T* __restrict__ gx,
                               size_t n, int act, float slope) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (VEC) {
        for (size_t i = t; i < n / 4; i += stride) {
            const float4 o = ld4<T>(y + 4 * i), g = ld4<T>(gy + 4 * i);
            st4<T>(gx + 4 * i, make_float4(g.x * act_deriv(o.x, act, slope), g.y * act_deriv(o.y, act, slope),
                                           g.z * act_deriv(o.z, act, slope), g.w * act_deriv(o.w, act, slope)));
        }
        for (size_t i = (n / 4) * 4 + t; i < n; i += stride) stf<T>(gx + i, ldf<T>(gy + i) * act_deriv(ldf<T>(y + i), act, slope));
    } else {
        for (size_t i = t; i < n; i += stride) stf<T>(gx + i, ldf<T>(gy + i) * act_deriv(ldf<T>(y + i), act, slope));
    }
}

template <bool VEC, typename T>
__global__ void axpby_kernel(const T* __restrict__ x, const T* __restrict__ y, T* __restrict__ o, size_t n,
                             float a, float b) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (VEC) {
        for (size_t i = t; i < n / 4; i += stride) {
            const float4 u = ld4<T>(x + 4 * i);
            float4 r = make_float4(a * u.x, a * u.y, a * u.z, a * u.w);
            if (y) {
                const float4 w = ld4<T>(y + 4 * i);
                r.x += b * w.x; r.y += b * w.y; r.z += b * w.z; r.w += b * w.w;
            }
            st4<T>(o + 4 * i, r);
        }
        for (size_t i = (n / 4) * 4 + t; i < n; i += stride) stf<T>(o + i, a * ldf<T>(x + i) + (y ? b * ldf<T>(y + i) : 0.f));
    } else {
        for (size_t i = t; i < n; i += stride) stf<T>(o + i, a * ldf<T>(x + i) + (y ? b * ldf<T>(y + i) : 0.f));
    }
}

template <bool VEC, typename T>
__global__ void mul_kernel(const T* __restrict__ x, const T* __restrict__ y, T* __restrict__ o, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (VEC) {
        for (size_t i = t; i < n / 4; i += stride) {
            const float4 u = ld4<T>(x + 4 * i), w = ld4<T>(y + 4 * i);
            st4<T>(o + 4 * i, make_float4(u.x * w.x, u.y * w.y, u.z * w.z, u.w * w.w));
        }
        for (size_t i = (n / 4) * 4 + t; i < n; i += stride) stf<T>(o + i, ldf<T>(x + i) * ldf<T>(y + i));
    } else {
        for (size_t i = t; i < n; i += stride) stf<T>(o + i, ldf<T>(x + i) * ldf<T>(y + i));
    }
}

// dst[i] = (TD) src[i]: storage-type conversion (fp32 <-> bf16, round to nearest even)
template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ src, TD* __restrict__ dst, size_t n, int vec) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        for (size_t i = t; i < n / 4; i += stride) st4<TD>(dst + 4 * i, ld4<TS>(src + 4 * i));
        for (size_t i = (n / 4) * 4 + t; i < n; i += stride) stf<TD>(dst + i, ldf<TS>(src + i));
    } else {
        for (size_t i = t; i < n; i += stride) stf<TD>(dst + i, ldf<TS>(src + i));
    }
}

__device__ __forceinline__ float block_sum(float v) {
    __shared__ float sh[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

template <typename T>
__global__ __launch_bounds__(256) void sqdiff_sum_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                         float* __restrict__ out, size_t n, float scale, int vec,
                                                         float* __restrict__ parts) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        for (size_t i = t0; i < n / 4; i += stride) {
            const float4 u = ld4<T>(a + 4 * i), w = ld4<T>(b + 4 * i);
            const float d0 = u.x - w.x, d1 = u.y - w.y, d2 = u.z - w.z, d3 = u.w - w.w;
            acc += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        }
        for (size_t i = (n / 4) * 4 + t0; i < n; i += stride) {
            const float d = ldf<T>(a + i) - ldf<T>(b + i);
            acc += d * d;
        }
    } else {
        for (size_t i = t0; i < n; i += stride) {
            const float d = ldf<T>(a + i) - ldf<T>(b + i);
            acc += d * d;
        }
    }
    const float t = block_sum(acc);
    if (threadIdx.x == 0) {
        if (parts) parts[blockIdx.x] = t;           // deterministic mode: the partials are added in block order afterwards
        else unsafeAtomicAdd(out, t * scale);
    }
}

// grid (blocks_per_row, n)
__global__ __launch_bounds__(256) void row_sumsq_kernel(const float* __restrict__ x, float* __restrict__ out, size_t row,
                                                        int vec, float* __restrict__ parts) {
    const float* p = x + (size_t)blockIdx.y * row;
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {           // row % 4 == 0 and x 16-byte aligned
        for (size_t i = t0; i < row / 4; i += stride) {
            const float4 u = reinterpret_cast<const float4*>(p)[i];
            acc += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w;
        }
    } else {
        for (size_t i = t0; i < row; i += stride) acc += p[i] * p[i];
    }
    const float t = block_sum(acc);
    if (threadIdx.x == 0) {
        if (parts) parts[(size_t)blockIdx.x * gridDim.y + blockIdx.y] = t;      // [block][row]: summed over blocks in order afterwards
        else unsafeAtomicAdd(&out[blockIdx.y], t);
    }
}

template <typename T>
__global__ void row_scale_kernel(const T* __restrict__ x, const float* __restrict__ s, T* __restrict__ o,
                                 size_t row, float k, int vec, const T* __restrict__ x2 = nullptr) {
    // o = (x - x2) * s[row] * k (x2 optional): with x2 the backward of a squared-difference loss term in one pass
    const float f = s[blockIdx.y] * k;
    const size_t base = (size_t)blockIdx.y * row;
    const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        for (size_t i = t0; i < row / 4; i += stride) {
            float4 u = ld4<T>(x + base + 4 * i);
            if (x2) {
                const float4 w = ld4<T>(x2 + base + 4 * i);
                u.x -= w.x; u.y -= w.y; u.z -= w.z; u.w -= w.w;
            }
            st4<T>(o + base + 4 * i, make_float4(u.x * f, u.y * f, u.z * f, u.w * f));
        }
    } else {
        for (size_t i = t0; i < row; i += stride) stf<T>(o + base + i, (ldf<T>(x + base + i) - (x2 ? ldf<T>(x2 + base + i) : 0.f)) * f);
    }
}

// The backward of a tapped, activated layer of a feature loss in ONE pass (round 6): o = (g + (y - target) s[row] k) act'(y) --
// the squared-difference term's gradient, the gradient arriving from the next layer (g, optional) and the layer's own
// activation derivative -- instead of cn_row_scale_diff + an add + cn_act_bwd (perceptual_loss.py:74-82: the four VGG taps).
template <typename T>
__global__ void tap_bwd_kernel(const T* __restrict__ y, const T* __restrict__ tgt, const T* __restrict__ g, const float* __restrict__ s,
                               T* __restrict__ o, size_t row, float k, int vec, int act, float slope) {
    const float f = s[blockIdx.y] * k;
    const size_t base = (size_t)blockIdx.y * row;
    const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        for (size_t i = t0; i < row / 4; i += stride) {
            const float4 u = ld4<T>(y + base + 4 * i), w = ld4<T>(tgt + base + 4 * i);
            float4 r = make_float4((u.x - w.x) * f, (u.y - w.y) * f, (u.z - w.z) * f, (u.w - w.w) * f);
            if (g) {
                const float4 q = ld4<T>(g + base + 4 * i);
                r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
            }
            st4<T>(o + base + 4 * i, make_float4(r.x * act_deriv(u.x, act, slope), r.y * act_deriv(u.y, act, slope),
                                                 r.z * act_deriv(u.z, act, slope), r.w * act_deriv(u.w, act, slope)));
        }
    } else {
        for (size_t i = t0; i < row; i += stride) {
            const float u = ldf<T>(y + base + i);
            const float r = (u - ldf<T>(tgt + base + i)) * f + (g ? ldf<T>(g + base + i) : 0.f);
            stf<T>(o + base + i, r * act_deriv(u, act, slope));
        }
    }
}

__global__ void masked_diff_kernel(const float* __restrict__ a, const float* __restrict__ b, const uint8_t* __restrict__ m,
                                   float* __restrict__ o, size_t pixels, int c) {
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < pixels; p += (size_t)gridDim.x * blockDim.x) {
        const float mk = (float)m[p];
        for (int e = 0; e < c; ++e) o[p * c + e] = (a[p * c + e] - (b ? b[p * c + e] : 0.f)) * mk;
    }
}

template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w, int c,
                                   int oh, int ow, int k, int s, int pad) {
    const long total = (long)n * oh * ow * c;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % c);
        long t = i / c;
        const int ox = (int)(t % ow);
        t /= ow;
        const int oy = (int)(t % oh);
        const int b = (int)(t / oh);
        float best = -INFINITY;
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx) {
                const int iy = oy * s - pad + dy, ix = ox * s - pad + dx;
                const bool in = iy >= 0 && iy < h && ix >= 0 && ix < w;
                const float v = in ? ldf<T>(x + (((long)b * h + iy) * w + ix) * c + ch) : 0.f;
                if (in || pad > 0) best = fmaxf(best, v);
            }
        stf<T>(y + i, best);
    }
}

// AveragePooling2D((3, 3), strides 1, padding "same") [TF-2.1]: the mean over the window cells INSIDE the image (tf.nn.avg_pool
// does not count padding), float4 over channels.  InceptionV3's pool branches (metrics path).
template <int V, typename T>
__global__ void avgpool3_same_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w, int c) {
    const int CG = c / V;
    const long total = (long)n * h * w * CG;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % CG);
        long t = i / CG;
        const int ox = (int)(t % w);
        t /= w;
        const int oy = (int)(t % h);
        const int b = (int)(t / h);
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        int cnt = 0;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int iy = oy + dy, ix = ox + dx;
                if (iy < 0 || iy >= h || ix < 0 || ix >= w) continue;
                const T* p = x + (((long)b * h + iy) * w + ix) * c + (long)cg * V;
                if (V == 4) {
                    const float4 v = ld4<T>(p);
                    acc[0] += v.x; acc[1 % V] += v.y; acc[2 % V] += v.z; acc[3 % V] += v.w;
                } else {
                    acc[0] += ldf<T>(p);
                }
                ++cnt;
            }
        const float r = 1.f / (float)cnt;
        T* q = y + (((long)b * h + oy) * w + ox) * c + (long)cg * V;
        if (V == 4) st4<T>(q, make_float4(acc[0] * r, acc[1 % V] * r, acc[2 % V] * r, acc[3 % V] * r));
        else stf<T>(q, acc[0] * r);
    }
}

// Gradient goes to the first maximum of each window in row-major window order (a zero-padding cell that wins takes it
// nowhere).  Written as a GATHER over input elements -- every element re-evaluates the (at most ceil(k/s)^2) windows
// that contain it -- so there are no atomics: deterministic, and gx may be stored in bf16.
template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ gy, T* __restrict__ gx,
                                   int n, int h, int w, int c, int oh, int ow, int k, int s, int pad) {
    const long total = (long)n * h * w * c;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % c);
        long t = i / c;
        const int ix0 = (int)(t % w);
        t /= w;
        const int iy0 = (int)(t % h);
        const int b = (int)(t / h);
        float acc = 0.f;
        // windows (oy, ox) with oy*s - pad <= iy0 < oy*s - pad + k
        const int oy_lo = max(0, (iy0 + pad - k + s) / s), oy_hi = min(oh - 1, (iy0 + pad) / s);
        const int ox_lo = max(0, (ix0 + pad - k + s) / s), ox_hi = min(ow - 1, (ix0 + pad) / s);
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                float best = -INFINITY;
                int ay = -1, ax = -1;
                bool arg_in = false;
                for (int dy = 0; dy < k; ++dy)
                    for (int dx = 0; dx < k; ++dx) {
                        const int iy = oy * s - pad + dy, ix = ox * s - pad + dx;
                        const bool in = iy >= 0 && iy < h && ix >= 0 && ix < w;
                        if (!in && pad == 0) continue;
                        const float v = in ? ldf<T>(x + (((long)b * h + iy) * w + ix) * c + ch) : 0.f;
                        if (v > best) {
                            best = v;
                            ay = iy; ax = ix; arg_in = in;
                        }
                    }
                if (arg_in && ay == iy0 && ax == ix0) acc += ldf<T>(gy + (((long)b * oh + oy) * ow + ox) * c + ch);
            }
        stf<T>(gx + i, acc);
    }
}

// Generic windows (ResNet-50's 3x3 / stride 2 / pad 1 pool) on 4-channel groups with 32-bit index arithmetic: same gather and tie
// rule as maxpool_bwd_kernel, 16-byte accesses (199 -> 69 us on 8 x 128 x 128 x 64).
template <typename T>
__global__ void maxpool_bwd4_kernel(const T* __restrict__ x, const T* __restrict__ gy, T* __restrict__ gx,
                                    int n, int h, int w, int c4, int oh, int ow, int k, int s, int pad) {
    const int total = n * h * w * c4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int cg = i % c4;
        int t = i / c4;
        const int ix0 = t % w;
        t /= w;
        const int iy0 = t % h;
        const int b = t / h;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const int oy_lo = max(0, (iy0 + pad - k + s) / s), oy_hi = min(oh - 1, (iy0 + pad) / s);
        const int ox_lo = max(0, (ix0 + pad - k + s) / s), ox_hi = min(ow - 1, (ix0 + pad) / s);
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                int arg[4] = {-1, -1, -1, -1};             // window position (dy * k + dx) of the first maximum; -2 = a padding cell
                for (int dy = 0; dy < k; ++dy)
                    for (int dx = 0; dx < k; ++dx) {
                        const int iy = oy * s - pad + dy, ix = ox * s - pad + dx;
                        const bool in = iy >= 0 && iy < h && ix >= 0 && ix < w;
                        if (!in && pad == 0) continue;
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (in) v = ld4<T>(x + ((long)(b * h + iy) * w + ix) * c4 * 4 + cg * 4);
                        const float vs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (vs[e] > best[e]) { best[e] = vs[e]; arg[e] = in ? dy * k + dx : -2; }
                    }
                const int mine = (iy0 - (oy * s - pad)) * k + (ix0 - (ox * s - pad));
                const float4 g = ld4<T>(gy + ((long)(b * oh + oy) * ow + ox) * c4 * 4 + cg * 4);
                const float gs[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (arg[e] == mine) acc[e] += gs[e];
            }
        st4<T>(gx + (long)i * 4, make_float4(acc[0], acc[1], acc[2], acc[3]));
    }
}

// The VGG pools (2x2, stride 2, no padding, even extents, C % 4 == 0): one thread owns one window of one 4-channel group --
// four 16-byte reads of x, one of gy, four 16-byte writes of gx, no index arithmetic per element (the generic kernel above
// re-derives the windows of every element with 64-bit div/mod: 286 us on the 8 x 256 x 256 x 64 tensor, 302 MB of traffic).
// Same tie rule: the first maximum in (dy, dx) scan order takes the gradient.
template <typename T>
__global__ void maxpool2_bwd_kernel(const T* __restrict__ x, const T* __restrict__ gy, T* __restrict__ gx, long windows4,
                                    int oh, int ow, int c4, int relu = 0, const T* __restrict__ tgt = nullptr,
                                    const float* __restrict__ sc = nullptr, int sc_rows = 1, float kk = 0.f) {
    // relu / tgt (round 6, cn_maxpool2_bwd_act): x is the ReLU OUTPUT that was pooled -- the routed gradient (plus, at a tapped
    // layer, the feature loss' own term (x - tgt) sc[sample] kk) is multiplied by relu'(x) here, so the full-size gradient between
    // the pool's and the activation's backward is never written
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < windows4; i += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(i % c4);
        long t = i / c4;
        const int ox = (int)(t % ow);
        t /= ow;
        const int oy = (int)(t % oh);
        const long b = t / oh;
        const long row = (long)ow * 2 * c4 * 4;                       // floats per input row
        const T* px = x + ((b * oh * 2 + oy * 2) * (long)ow * 2 + ox * 2) * c4 * 4 + cg * 4;
        T* pg = gx + (px - x);
        const float4 v00 = ld4<T>(px), v01 = ld4<T>(px + c4 * 4), v10 = ld4<T>(px + row), v11 = ld4<T>(px + row + c4 * 4);
        const float4 g = ld4<T>(gy + i * 4);
        const float a[4][4] = {{v00.x, v01.x, v10.x, v11.x}, {v00.y, v01.y, v10.y, v11.y}, {v00.z, v01.z, v10.z, v11.z}, {v00.w, v01.w, v10.w, v11.w}};
        const float gg[4] = {g.x, g.y, g.z, g.w};
        float o[4][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int arg = 0;
            float best = a[e][0];
#pragma unroll
            for (int j = 1; j < 4; ++j)
                if (a[e][j] > best) { best = a[e][j]; arg = j; }
#pragma unroll
            for (int j = 0; j < 4; ++j) o[e][j] = (j == arg && best > -INFINITY) ? gg[e] : 0.f;
        }
        if (tgt) {
            const T* pt = tgt + (px - x);
            const float4 t00 = ld4<T>(pt), t01 = ld4<T>(pt + c4 * 4), t10 = ld4<T>(pt + row), t11 = ld4<T>(pt + row + c4 * 4);
            const float tt[4][4] = {{t00.x, t01.x, t10.x, t11.x}, {t00.y, t01.y, t10.y, t11.y}, {t00.z, t01.z, t10.z, t11.z}, {t00.w, t01.w, t10.w, t11.w}};
            const float f = sc[sc_rows > 1 ? b : 0] * kk;
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j) o[e][j] = (a[e][j] - tt[e][j]) * f + o[e][j];
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j) o[e][j] = a[e][j] > 0.f ? o[e][j] : 0.f;
        }
        st4<T>(pg, make_float4(o[0][0], o[1][0], o[2][0], o[3][0]));
        st4<T>(pg + c4 * 4, make_float4(o[0][1], o[1][1], o[2][1], o[3][1]));
        st4<T>(pg + row, make_float4(o[0][2], o[1][2], o[2][2], o[3][2]));
        st4<T>(pg + row + c4 * 4, make_float4(o[0][3], o[1][3], o[2][3], o[3][3]));
    }
}

__global__ void chan_affine3_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t pixels, int p0, int p1,
                                        int p2, float scale, float o0, float o1, float o2) {
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < pixels; p += (size_t)gridDim.x * blockDim.x) {
        const float v0 = x[p * 3 + 0], v1 = x[p * 3 + 1], v2 = x[p * 3 + 2];
        const float v[3] = {v0, v1, v2};
        y[p * 3 + 0] = scale * v[p0] + o0;
        y[p * 3 + 1] = scale * v[p1] + o1;
        y[p * 3 + 2] = scale * v[p2] + o2;
    }
}

__global__ void chan_affine3_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, size_t pixels, int p0,
                                        int p1, int p2, float scale) {
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < pixels; p += (size_t)gridDim.x * blockDim.x) {
        float g[3];
        g[p0] = scale * gy[p * 3 + 0];
        g[p1] = scale * gy[p * 3 + 1];
        g[p2] = scale * gy[p * 3 + 2];
        gx[p * 3 + 0] = g[0];
        gx[p * 3 + 1] = g[1];
        gx[p * 3 + 2] = g[2];
    }
}

__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void gan_loss_fwd_kernel(const float* __restrict__ s, float* __restrict__ out, int n,
                                                           float label) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += label * softplus_f(-s[i]) + (1.f - label) * softplus_f(s[i]);
    const float t = block_sum(acc);
    if (threadIdx.x == 0) out[0] = t / (float)n;
}

__global__ void gan_loss_bwd_kernel(const float* __restrict__ s, const float* __restrict__ gout, float* __restrict__ gs,
                                    int n, float label) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) gs[i] = gout[0] * (-label * sigmoid_f(-s[i]) + (1.f - label) * sigmoid_f(s[i])) / (float)n;
}

__global__ void adam_kernel(float* __restrict__ th, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, float* __restrict__ ema, size_t n, const float* __restrict__ lr_ptr,
                            float b1, float b2, float eps, float alpha) {
    const float lr_t = lr_ptr[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        const float t = th[i] - lr_t * mi / (sqrtf(vi) + eps);
        m[i] = mi;
        v[i] = vi;
        th[i] = t;
        if (ema) ema[i] = alpha * ema[i] + (1.f - alpha) * t;
    }
}

__global__ void ema_kernel(float* __restrict__ ema, const float* __restrict__ th, size_t n, float alpha) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        ema[i] = alpha * ema[i] + (1.f - alpha) * th[i];
}

__global__ void gather_u8_kernel(const uint8_t* __restrict__ pool, const int64_t* __restrict__ idx,
                                 const uint8_t* __restrict__ flip, float* __restrict__ out, int n, int h, int w, int c) {
    const long per = (long)h * w * c;
    const long total = (long)n * per;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        const long r = i % per;
        const int ch = (int)(r % c);
        const int x = (int)((r / c) % w);
        const int y = (int)(r / ((long)c * w));
        const int sx = (flip && flip[b]) ? w - 1 - x : x;
        const uint8_t v = pool[(size_t)idx[b] * per + ((long)y * w + sx) * c + ch];
        out[i] = __fsub_rn(__fdiv_rn((float)v, 127.5f), 1.0f);    // = NumPy's float32 (v / 127.5) - 1, bit for bit
    }
}

__global__ void to_uint8_kernel(const float* __restrict__ x, uint8_t* __restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = fminf(fmaxf(x[i], -1.f), 1.f);
        // two separately rounded fp32 operations like NumPy's (clip(x) + 1) * 127.5 (no fma contraction: the byte must be
        // bit-exact), then astype(uint8)'s truncation toward zero
        o[i] = (uint8_t)__fmul_rn(__fadd_rn(v, 1.f), 127.5f);
    }
}

inline int ew_blocks(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b > 8192 ? 8192 : (b ? b : 1));
}

}  // namespace

static int nc_reduce_launch(const void* x1, const void* x2, float* s1, float* s2, int n, int s, int c, int flags,
                            float slope, int dt, void* stream, void* dact_out, int dact, const void* x3 = nullptr,
                            const float* coef = nullptr, void* scaled_out = nullptr, int dact_on = -1) {
    if (dact_on < 0) dact_on = dact_out != nullptr;
    CN_CHECK_ARG(x1 && (s1 || s2) && n > 0 && s > 0 && c > 0 && (dt == CN_F32 || dt == CN_BF16), "nc_reduce: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (flags & 16) {
        // outputs were cleared by the caller (per-step zero pool: one clearing launch per step, not per call)
    } else if (s1 && s2 == s1 + (size_t)n * c) {
        if (int ez__ = cn_zero_async(s1, sizeof(float) * 2 * (size_t)n * c, st)) return ez__;
    } else {
        if (s1) {
            if (int ez__ = cn_zero_async(s1, sizeof(float) * (size_t)n * c, st)) return ez__;
        }
        if (s2) {
            if (int ez__ = cn_zero_async(s2, sizeof(float) * (size_t)n * c, st)) return ez__;
        }
    }
    const int V = (c % 4 == 0) ? 4 : 1;
    const int CG = c / V;
    int TX = 1;
    while (TX < CG && TX < 64) TX <<= 1;
    const int TY = 256 / TX;
    const int cblk = cn_cdiv(CG, TX);
    // ~512 workgroups in total, at least 4*TY rows each
    constexpr long red_blocks = 512;   // sweep: fewer, longer workgroups = shorter same-address atomic tails
    long want = red_blocks / ((long)cblk * n);
    if (cn_det()) {                 // deterministic mode: as many row blocks as the per-stream workspace holds partials for
        const long cap = (long)(CN_DET_WS_FLOATS / (2 * (size_t)n * c));
        if (want > cap) want = cap;
    }
    if (want < 1) want = 1;
    if (want > 256) want = 256;     // same-address atomics serialise (~100 ns each): 2048 per address cost 200 us
    long rpb = (s + want - 1) / want;
    if (rpb < 4 * TY) rpb = 4 * TY;
    const int sblk = cn_cdiv(s, rpb);
    dim3 grid(cblk, sblk, n), block(TX, TY);
    const int period2 = flags >> 8;                 // bits 8..: x2 holds `period2` samples, used for sample n as n % period2
    CN_CHECK_ARG(period2 == 0 || (x2 && n % period2 == 0), "nc_reduce: bad x2 period %d for n = %d", period2, n);
    float* parts = nullptr;
    if (cn_det() && sblk > 1) {
        parts = cn_det_ws(st, 2 * (size_t)sblk * n * c);
        if (!parts) return CN_EINVAL;
    }
    CN_DISPATCH_DT(dt, {
        const T* p1 = (const T*)x1; const T* p2 = (const T*)x2;
        if (V == 4) hipLaunchKernelGGL((nc_reduce_kernel<4, T>), grid, block, 0, st, p1, p2, s1, s2, s, c, (int)rpb, flags & 255, slope, period2, (T*)dact_out, dact, parts, (const T*)x3, coef, (T*)scaled_out, dact_on);
        else hipLaunchKernelGGL((nc_reduce_kernel<1, T>), grid, block, 0, st, p1, p2, s1, s2, s, c, (int)rpb, flags & 255, slope, period2, (T*)dact_out, dact, parts, (const T*)x3, coef, (T*)scaled_out, dact_on);
    });
    CN_LAUNCH_CHECK();
    if (parts) {
        if (s1) { if (int e = cn_sum_parts(parts, s1, sblk, (long)n * c, 0, 1.f, st)) return e; }
        if (s2) { if (int e = cn_sum_parts(parts + (size_t)sblk * n * c, s2, sblk, (long)n * c, 0, 1.f, st)) return e; }
    }
    return CN_OK;
}

// out (4, n, c) = sum x, sum x^2, sum lrelu(x), sum lrelu(x)^2 over s: the style statistics and the instance-norm statistics of a
// DiscrBlock's pre-activation tensor in one pass (c % 4 == 0).  flags bit4: `out` is already zero.
// (sum h, sum h lrelu(x), sum h ta) per (n, c) in one pass: out (3, n, c); x holds `period` samples (n % period == 0); flags & 16:
// out is already zero; flags & 32: `ta` is the tangent input tx and ta = lrelu'(x) tx is formed in the pass.  The backward reductions of the DiscrBlock tail's tangent (losses.py:75-82 through building_blocks.py:100-106).
extern "C" int cn_nc_reduce_hxt(const void* h, const void* x, const void* ta, float* out, int n, int s, int c, float slope, int period,
                                int flags, int dt, void* stream) {
    CN_CHECK_ARG(h && x && ta && out && n > 0 && s > 0 && c > 0 && c % 4 == 0 && period > 0 && n % period == 0 && (dt == CN_F32 || dt == CN_BF16),
                 "nc_reduce_hxt: bad args");
    hipStream_t st = (hipStream_t)stream;
    const int CG = c / 4;
    int TX = 1;
    while (TX < CG && TX < 64) TX <<= 1;
    const int TY = 256 / TX;
    const int cblk = cn_cdiv(CG, TX);
    long want = 512 / ((long)cblk * n);
    if (cn_det()) {
        const long cap = (long)(CN_DET_WS_FLOATS / (3 * (size_t)n * c));
        if (want > cap) want = cap;
    }
    if (want < 1) want = 1;
    if (want > 256) want = 256;
    long rpb = (s + want - 1) / want;
    if (rpb < 4 * TY) rpb = 4 * TY;
    const int sblk = cn_cdiv(s, rpb);
    float* parts = nullptr;
    if (cn_det() && sblk > 1) {
        parts = cn_det_ws(st, 3 * (size_t)sblk * n * c);
        if (!parts) return CN_EINVAL;
    } else if (sblk > 1 && !(flags & 16)) {
        if (int ez__ = cn_zero_async(out, sizeof(float) * 3 * (size_t)n * c, st)) return ez__;
    }
    dim3 grid(cblk, sblk, n), block(TX, TY);
    CN_DISPATCH_DT(dt, hipLaunchKernelGGL((nc_reduce_hxt_kernel<T>), grid, block, 0, st, (const T*)h, (const T*)x, (const T*)ta, out, s, c, (int)rpb,
                                          slope, period, parts, (flags & 32) ? 1 : 0));
    CN_LAUNCH_CHECK();
    if (parts) {
        for (int q = 0; q < 3; ++q)
            if (int e = cn_sum_parts(parts + (size_t)q * sblk * n * c, out + (size_t)q * n * c, sblk, (long)n * c, 0, 1.f, st)) return e;
    }
    return CN_OK;
}

extern "C" int cn_nc_reduce4(const void* x, float* out, int n, int s, int c, float slope, int flags, int dt, void* stream) {
    CN_CHECK_ARG(x && out && n > 0 && s > 0 && c > 0 && c % 4 == 0 && (dt == CN_F32 || dt == CN_BF16), "nc_reduce4: bad args");
    hipStream_t st = (hipStream_t)stream;
    const int CG = c / 4;
    int TX = 1;
    while (TX < CG && TX < 64) TX <<= 1;
    const int TY = 256 / TX;
    const int cblk = cn_cdiv(CG, TX);
    long want = 512 / ((long)cblk * n);
    if (cn_det()) {
        const long cap = (long)(CN_DET_WS_FLOATS / (4 * (size_t)n * c));
        if (want > cap) want = cap;
    }
    if (want < 1) want = 1;
    if (want > 256) want = 256;
    long rpb = (s + want - 1) / want;
    if (rpb < 4 * TY) rpb = 4 * TY;
    const int sblk = cn_cdiv(s, rpb);
    float* parts = nullptr;
    if (cn_det() && sblk > 1) {
        parts = cn_det_ws(st, 4 * (size_t)sblk * n * c);
        if (!parts) return CN_EINVAL;
    } else if (sblk > 1 && !(flags & 16)) {
        if (int ez__ = cn_zero_async(out, sizeof(float) * 4 * (size_t)n * c, st)) return ez__;
    }
    dim3 grid(cblk, sblk, n), block(TX, TY);
    CN_DISPATCH_DT(dt, hipLaunchKernelGGL((nc_reduce4_kernel<T>), grid, block, 0, st, (const T*)x, out, s, c, (int)rpb, slope, parts));
    CN_LAUNCH_CHECK();
    if (parts) {
        for (int q = 0; q < 4; ++q)
            if (int e = cn_sum_parts(parts + (size_t)q * sblk * n * c, out + (size_t)q * n * c, sblk, (long)n * c, 0, 1.f, st)) return e;
    }
    return CN_OK;
}

extern "C" int cn_nc_reduce(const void* x1, const void* x2, float* s1, float* s2, int n, int s, int c, int flags,
                            float slope, int dt, void* stream) {
    return nc_reduce_launch(x1, x2, s1, s2, n, s, c, flags, slope, dt, stream, nullptr, 0);
}

// Activation backward fused with the bias gradient: gx = gy * act'(y) (y = the activation's output), gb[n][c] = sum_s gx --
// one pass over gy and y instead of an act_bwd pass plus a reduction pass over its result.  flags: bit 4 (16) as cn_nc_reduce.
extern "C" int cn_act_bwd_bias(const void* gy, const void* y, void* gx, float* gb, int n, int s, int c, int act, float slope,
                               int flags, int dt, void* stream) {
    CN_CHECK_ARG(gy && y && gx && gb, "act_bwd_bias: NULL");
    return nc_reduce_launch(gy, y, gb, nullptr, n, s, c, flags & 16, slope, dt, stream, gx, act);
}

namespace {
// dst[c] (+)= sum_r src[r][c]: the partial rows of a per-channel reduction, added in row order (no atomics)
__global__ void sum_rows_into_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float t = 0.f;
    for (int r = 0; r < rows; ++r) t += src[(long)r * cols + c];
    if (accumulate) unsafeAtomicAdd(&dst[c], t);        // (two streams of a forked step may add into one slot at once)
    else dst[c] = t;
}
}  // namespace

extern "C" int cn_sum_rows_into(const float* src, float* dst, int rows, int cols, int accumulate, void* stream) {
    CN_CHECK_ARG(src && dst && rows > 0 && cols > 0, "sum_rows_into: bad args");
    hipLaunchKernelGGL(sum_rows_into_kernel, dim3(cn_cdiv(cols, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, rows, cols, accumulate);
    CN_LAUNCH_CHECK();
    return CN_OK;
}

// Backward of conv -> BatchNorm(inference: y0 = a[c] x + shift[c] (+ residual)) -> activation in ONE pass (keras ResNet50 blocks
// of the real encoder, real_encoder.py:13-20): g = gy * act'(y) (y = the activation's output; act NONE: g = gy), gx = a[c] * g
// written to gx, g itself to g_out if the residual branch needs it (else NULL), sum_g[r][c] = sum g (-> d shift),
// sum_gx[r][c] = sum g * x (-> d a); n = partial rows the per-channel sums are spread over, s rows each.  flags: bit4 as cn_nc_reduce.
extern "C" int cn_bn_act_bwd(const void* gy, const void* y, const void* x, const float* a, void* g_out, void* gx, float* sum_g,
                             float* sum_gx, int n, int s, int c, int act, int flags, int dt, void* stream) {
    CN_CHECK_ARG(gy && y && x && a && gx && sum_g && sum_gx, "bn_act_bwd: NULL");
    return nc_reduce_launch(gy, y, sum_g, sum_gx, n, s, c, flags & 16, 0.f, dt, stream, g_out, act, x, a, gx, 1);
}

extern "C" int cn_nc_reduce_dact(const void* x1, const void* x2, float* s1, float* s2, void* dact_out, int n, int s, int c,
                                 int flags, float slope, int act, int dt, void* stream) {
    CN_CHECK_ARG(x1 && x2 && s1, "nc_reduce_dact: NULL");          // (dact_out may be NULL: the sums only)
    return nc_reduce_launch(x1, x2, s1, s2, n, s, c, flags, slope, dt, stream, dact_out, act, nullptr, nullptr, nullptr, 1);
}

extern "C" int cn_nc_lin2(const void* x1, const float* a1, const void* x2, const float* a2, const float* bb,
                          const float* a3, const float* b3, void* y, int n, int s, int c, int cstride, int flags,
                          float slope, int dt, void* stream) {
    CN_CHECK_ARG(y && n > 0 && s > 0 && c > 0 && (cstride == 0 || cstride == c) && (dt == CN_F32 || dt == CN_BF16), "nc_lin2: bad args");
    CN_CHECK_ARG(x1 || x2 || bb, "nc_lin2: nothing to compute");
    const int V = (c % 4 == 0) ? 4 : 1;
    const long total = (long)n * s * (c / V);
    hipStream_t st = (hipStream_t)stream;
    const int period2 = flags >> 8;                 // bits 8..: x2 holds `period2` samples, used for sample n as n % period2
    CN_CHECK_ARG(period2 == 0 || (x2 && cstride && n % period2 == 0), "nc_lin2: bad x2 period %d for n = %d", period2, n);
    flags &= 255;
    {
        // big tensors: per-sample grid with a fixed channel group per thread (nc_lin2_rows_kernel)
        const int CG = c / V;
        const int ny = cstride ? n : 1;
        const long G = total / ny;                 // groups per grid row (per_channel: the whole tensor)
        int gcd = CG, r256 = 256;
        while (r256) { const int t = gcd % r256; gcd = r256; r256 = t; }
        const int q = CG / gcd;                    // gx must be a multiple of q
        if (G >= 16384 && G < 2147483647L - 256 * 8192L && q <= 32) {
            long gx = (G + 256 * 4 - 1) / (256 * 4);           // ~4 groups per thread
            if (gx * ny > 8192) gx = 8192 / ny;
            if (gx < 1) gx = 1;
            gx = (gx + q - 1) / q * q;
            CN_DISPATCH_DT(dt, {
                if (V == 4) hipLaunchKernelGGL((nc_lin2_rows_kernel<4, T>), dim3((unsigned)gx, ny), dim3(256), 0, st, (const T*)x1, a1, (const T*)x2, a2, bb, a3, b3, (T*)y, (int)G, CG, cstride, flags, slope, period2);
                else hipLaunchKernelGGL((nc_lin2_rows_kernel<1, T>), dim3((unsigned)gx, ny), dim3(256), 0, st, (const T*)x1, a1, (const T*)x2, a2, bb, a3, b3, (T*)y, (int)G, CG, cstride, flags, slope, period2);
            });
            CN_LAUNCH_CHECK();
            return CN_OK;
        }
    }
    CN_DISPATCH_DT(dt, {
        if (V == 4) hipLaunchKernelGGL((nc_lin2_kernel<4, T>), dim3(ew_blocks(total)), dim3(256), 0, st, (const T*)x1, a1, (const T*)x2, a2, bb, a3, b3, (T*)y, total, s, c, cstride, flags, slope, period2);
        else hipLaunchKernelGGL((nc_lin2_kernel<1, T>), dim3(ew_blocks(total)), dim3(256), 0, st, (const T*)x1, a1, (const T*)x2, a2, bb, a3, b3, (T*)y, total, s, c, cstride, flags, slope, period2);
    });
    CN_LAUNCH_CHECK();
    return CN_OK;
}

// AdaIn (mode 0) / instance norm (mode 1) apply (dir 0) or backward map (dir 1) with the coefficient algebra inline: see
// norm_apply_rows_kernel.  dir 0: y = k1 f1(x1) + kb from (sa = sum, sb = sum of squares, p1, p2), writes save_mean / save_r.
// dir 1: y = gx from x1 = gy, x2 = x, (sa, sb) = (sum gy, sum gy f2(x)), reads save_mean / save_r, writes gp1 (/ gp2); a3 / b3: an
// additional a3 x + b3 (the style statistics' gradient).  flags as cn_nc_lin2 (bits 0, 1, 2; bits 8..: the period of x2).
// CN_EUNSUPPORTED (nothing launched) unless c % 4 == 0 and the tensor is big enough for the per-sample grid.
extern "C" int cn_norm_apply(int mode, int dir, const void* x1, const void* x2, const float* sa, const float* sb, const float* p1,
                             const float* p2, float* save_mean, float* save_r, float* gp1, float* gp2, const float* a3,
                             const float* b3, void* y, int n, int s, int c, float eps, int flags, float slope, int dt, void* stream) {
    CN_CHECK_ARG((mode == 0 || mode == 1) && (dir == 0 || dir == 1) && x1 && sa && sb && p1 && save_mean && save_r && y && n > 0 && s > 0 && c > 0 &&
                 (dt == CN_F32 || dt == CN_BF16), "norm_apply: bad args");
    CN_CHECK_ARG(mode == 0 || dir == 1 || p2, "norm_apply: instance norm needs beta");
    CN_CHECK_ARG(dir == 0 || (x2 && gp1 && (mode == 0 || gp2)), "norm_apply: backward needs x and the parameter gradients' tensors");
    if (c % 4 != 0) return CN_EUNSUPPORTED;
    const int period2 = flags >> 8;
    CN_CHECK_ARG(period2 == 0 || (x2 && n % period2 == 0), "norm_apply: bad x2 period %d for n = %d", period2, n);
    flags &= 255;
    const int CG = c / 4;
    const long G = (long)s * CG;
    int gcd = CG, r256 = 256;
    while (r256) { const int t = gcd % r256; gcd = r256; r256 = t; }
    const int q = CG / gcd;
    if (!(G >= 16384 && G < 2147483647L - 256 * 8192L && q <= 32)) return CN_EUNSUPPORTED;
    long gx = (G + 256 * 4 - 1) / (256 * 4);
    if (gx * n > 8192) gx = 8192 / n;
    if (gx < 1) gx = 1;
    gx = (gx + q - 1) / q * q;
    if (gx * 256 < CG) return CN_EUNSUPPORTED;          // (workgroup column 0 must cover every channel group: it writes the side outputs)
    NormApplyArgs A;
    A.mode = mode; A.dir = dir; A.N = n; A.C = c; A.invS = 1.f / (float)s; A.eps = eps;
    A.s1 = sa; A.s2 = sb; A.p1 = p1; A.p2 = p2; A.sm = save_mean; A.sr = save_r; A.gp1 = gp1; A.gp2 = gp2;
    CN_DISPATCH_DT(dt, hipLaunchKernelGGL((norm_apply_rows_kernel<T>), dim3((unsigned)gx, n), dim3(256), 0, (hipStream_t)stream, A, (const T*)x1,
                                          (const T*)x2, a3, b3, (T*)y, (int)G, CG, flags, slope, period2));
    CN_LAUNCH_CHECK();
    return CN_OK;
}

#define EW_LAUNCH(kernel, n, ...)                                                                              \
    hipLaunchKernelGGL(kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);             \
    CN_LAUNCH_CHECK();                                                                                         \
    return CN_OK;

static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
// 4-wide access of a tensor stored as `dt` needs 16-byte (fp32) / 8-byte (bf16) alignment (NULL counts as aligned)
static inline bool alv(const void* p, int dt) { return ((uintptr_t)p & (dt == CN_BF16 ? 7 : 15)) == 0; }
// 4-wide form of a streaming map when every pointer is aligned (quarter the grid: four elements per lane per trip)
#define EW_LAUNCH_VT(kernel, n, vec, dt, ...)                                                                                    \
    CN_CHECK_ARG((dt) == CN_F32 || (dt) == CN_BF16, "bad dtype code %d", (int)(dt));                                                \
    CN_DISPATCH_DT(dt, {                                                                                                          \
        if (vec) hipLaunchKernelGGL((kernel<true, T>), dim3(ew_blocks(((n) + 3) / 4)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((kernel<false, T>), dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);             \
    });                                                                                                                           \
    CN_LAUNCH_CHECK();                                                                                                            \
    return CN_OK;

extern "C" int cn_act_fwd(const void* x, void* y, size_t numel, int act, float slope, int dt, void* stream) {
    CN_CHECK_ARG(x && y, "act_fwd: NULL");
    if (!numel) return CN_OK;
    EW_LAUNCH_VT(act_fwd_kernel, numel, alv(x, dt) && alv(y, dt), dt, (const T*)x, (T*)y, numel, act, slope)
}
extern "C" int cn_act_bwd(const void* gy, const void* y, void* gx, size_t numel, int act, float slope, int dt, void* stream) {
    CN_CHECK_ARG(gy && y && gx, "act_bwd: NULL");
    if (!numel) return CN_OK;
    EW_LAUNCH_VT(act_bwd_kernel, numel, alv(gy, dt) && alv(y, dt) && alv(gx, dt), dt, (const T*)gy, (const T*)y, (T*)gx, numel, act, slope)
}
extern "C" int cn_axpby(const void* x, const void* y, void* out, size_t numel, float a, float b, int dt, void* stream) {
    CN_CHECK_ARG(x && out, "axpby: NULL");
    if (!numel) return CN_OK;
    EW_LAUNCH_VT(axpby_kernel, numel, alv(x, dt) && alv(y, dt) && alv(out, dt), dt, (const T*)x, (const T*)y, (T*)out, numel, a, b)
}
extern "C" int cn_mul(const void* x, const void* y, void* out, size_t numel, int dt, void* stream) {
    CN_CHECK_ARG(x && y && out, "mul: NULL");
    if (!numel) return CN_OK;
    EW_LAUNCH_VT(mul_kernel, numel, alv(x, dt) && alv(y, dt) && alv(out, dt), dt, (const T*)x, (const T*)y, (T*)out, numel)
}
extern "C" int cn_cast(const void* src, int src_dt, void* dst, int dst_dt, size_t numel, void* stream) {
    CN_CHECK_ARG(src && dst && (src_dt == CN_F32 || src_dt == CN_BF16) && (dst_dt == CN_F32 || dst_dt == CN_BF16) && src_dt != dst_dt,
                 "cast: bad args");
    if (!numel) return CN_OK;
    const int vec = alv(src, src_dt) && alv(dst, dst_dt);
    const dim3 grid(ew_blocks(vec ? (numel + 3) / 4 : numel));
    if (src_dt == CN_F32) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, (bf16_t*)dst, numel, vec);
    else hipLaunchKernelGGL((cast_kernel<bf16_t, float>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (float*)dst, numel, vec);
    CN_LAUNCH_CHECK();
    return CN_OK;
}
extern "C" int cn_sqdiff_sum(const void* a, const void* b, float* out, size_t numel, float scale, int dt, void* stream) {
    CN_CHECK_ARG(a && b && out && (dt == CN_F32 || dt == CN_BF16), "sqdiff_sum: bad args");
    if (!numel) return CN_OK;
    const int vec = alv(a, dt) && alv(b, dt);
    const int blocks = ew_blocks(numel / (vec ? 4 : 1) + 1) > 512 ? 512 : ew_blocks(numel / (vec ? 4 : 1) + 1);   // one atomic each
    float* parts = nullptr;
    if (cn_det()) {
        parts = cn_det_ws((hipStream_t)stream, (size_t)blocks);
        if (!parts) return CN_EINVAL;
    }
    CN_DISPATCH_DT(dt, hipLaunchKernelGGL((sqdiff_sum_kernel<T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const T*)a, (const T*)b, out, numel, scale, vec, parts));
    CN_LAUNCH_CHECK();
    if (parts) return cn_sum_parts(parts, out, blocks, 1, 1, scale, (hipStream_t)stream);    // out += scale * sum (out is the caller's accumulator)
    return CN_OK;
}
extern "C" int cn_row_sumsq(const float* x, float* out, int n, size_t row, void* stream) {
    CN_CHECK_ARG(x && out && n > 0 && row > 0, "row_sumsq: bad args");
    const int vec = al16(x) && row % 4 == 0;
    int bpr = (int)((row + 256 * 32 - 1) / (256 * 32));
    if (bpr > 64) bpr = 64;
    if (cn_det()) {
        float* parts = cn_det_ws((hipStream_t)stream, (size_t)bpr * n);
        if (!parts) return CN_EINVAL;
        hipLaunchKernelGGL(row_sumsq_kernel, dim3(bpr, n), dim3(256), 0, (hipStream_t)stream, x, out, row, vec, parts);
        CN_LAUNCH_CHECK();
        return cn_sum_parts(parts, out, bpr, n, 0, 1.f, (hipStream_t)stream);
    }
    if (int ez__ = cn_zero_async(out, sizeof(float) * n, (hipStream_t)stream)) return ez__;
    hipLaunchKernelGGL(row_sumsq_kernel, dim3(bpr, n), dim3(256), 0, (hipStream_t)stream, x, out, row, vec, (float*)nullptr);
    CN_LAUNCH_CHECK();
    return CN_OK;
}
extern "C" int cn_row_scale(const void* x, const float* s, void* out, int n, size_t row, float k, int dt, void* stream) {
    CN_CHECK_ARG(x && s && out && n > 0 && row > 0 && (dt == CN_F32 || dt == CN_BF16), "row_scale: bad args");
    const int vec = alv(x, dt) && alv(out, dt) && row % 4 == 0;
    int bpr = (int)((row + 256 * 16 - 1) / (256 * 16));
    if (bpr > 512) bpr = 512;
    CN_DISPATCH_DT(dt, hipLaunchKernelGGL((row_scale_kernel<T>), dim3(bpr, n), dim3(256), 0, (hipStream_t)stream, (const T*)x, s, (T*)out, row, k, vec));
    CN_LAUNCH_CHECK();
    return CN_OK;
}
extern "C" int cn_row_scale_diff(const void* x, const void* x2, const float* s, void* out, int n, size_t row, float k, int dt, void* stream) {
    CN_CHECK_ARG(x && x2 && s && out && n > 0 && row > 0 && (dt == CN_F32 || dt == CN_BF16), "row_scale_diff: bad args");
    const int vec = alv(x, dt) && alv(x2, dt) && alv(out, dt) && row % 4 == 0;
    int bpr = (int)((row + 256 * 16 - 1) / (256 * 16));
    if (bpr > 512) bpr = 512;
    CN_DISPATCH_DT(dt, hipLaunchKernelGGL((row_scale_kernel<T>), dim3(bpr, n), dim3(256), 0, (hipStream_t)stream, (const T*)x, s, (T*)out, row, k, vec, (const T*)x2));
    CN_LAUNCH_CHECK();
    return CN_OK;
}
extern "C" int cn_tap_bwd(const void* y, const void* target, const void* g, const float* s, void* out, int n, size_t row, float k,
                          int act, float slope, int dt, void* stream) {
    CN_CHECK_ARG(y && target && s && out && n > 0 && row > 0 && (dt == CN_F32 || dt == CN_BF16), "tap_bwd: bad args");
    const int vec = alv(y, dt) && alv(target, dt) && alv(out, dt) && (!g || alv(g, dt)) && row % 4 == 0;
    int bpr = (int)((row + 256 * 16 - 1) / (256 * 16));
    if (bpr > 512) bpr = 512;
    CN_DISPATCH_DT(dt, hipLaunchKernelGGL((tap_bwd_kernel<T>), dim3(bpr, n), dim3(256), 0, (hipStream_t)stream, (const T*)y, (const T*)target,
                                          (const T*)g, s, (T*)out, row, k, vec, act, slope));
    CN_LAUNCH_CHECK();
    return CN_OK;
}
extern "C" int cn_masked_diff(const float* a, const float* b, const uint8_t* mask, float* out, size_t pixels, int c, void* stream) {
    CN_CHECK_ARG(a && mask && out && c > 0, "masked_diff: bad args");
    if (!pixels) return CN_OK;
    EW_LAUNCH(masked_diff_kernel, pixels, a, b, mask, out, pixels, c)
}
extern "C" int cn_maxpool_fwd(const void* x, void* y, int n, int h, int w, int c, int k, int s, int pad, int dt, void* stream) {
    CN_CHECK_ARG(x && y && n > 0 && h > 0 && w > 0 && c > 0 && k > 0 && s > 0 && pad >= 0 && (dt == CN_F32 || dt == CN_BF16), "maxpool: bad args");
    const int oh = (h + 2 * pad - k) / s + 1, ow = (w + 2 * pad - k) / s + 1;
    const size_t total = (size_t)n * oh * ow * c;
    CN_DISPATCH_DT(dt, hipLaunchKernelGGL((maxpool_fwd_kernel<T>), dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, n, h, w, c, oh, ow, k, s, pad));
    CN_LAUNCH_CHECK();
    return CN_OK;
}
extern "C" int cn_avgpool3_same(const void* x, void* y, int n, int h, int w, int c, int dt, void* stream) {
    CN_CHECK_ARG(x && y && n > 0 && h > 0 && w > 0 && c > 0 && (dt == CN_F32 || dt == CN_BF16), "avgpool3_same: bad args");
    const bool v4 = c % 4 == 0 && alv(x, dt) && alv(y, dt);
    const size_t total = (size_t)n * h * w * (v4 ? c / 4 : c);
    CN_DISPATCH_DT(dt, {
        if (v4) hipLaunchKernelGGL((avgpool3_same_kernel<4, T>), dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, n, h, w, c);
        else hipLaunchKernelGGL((avgpool3_same_kernel<1, T>), dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, n, h, w, c);
    });
    CN_LAUNCH_CHECK();
    return CN_OK;
}
extern "C" int cn_maxpool_bwd(const void* x, const void* gy, void* gx, int n, int h, int w, int c, int k, int s, int pad, int dt, void* stream) {
    CN_CHECK_ARG(x && gy && gx && n > 0 && h > 0 && w > 0 && c > 0 && k > 0 && s > 0 && pad >= 0 && (dt == CN_F32 || dt == CN_BF16), "maxpool_bwd: bad args");
    const int oh = (h + 2 * pad - k) / s + 1, ow = (w + 2 * pad - k) / s + 1;
    const size_t total = (size_t)n * h * w * c;
    if (k == 2 && s == 2 && pad == 0 && h % 2 == 0 && w % 2 == 0 && c % 4 == 0 && alv(x, dt) && alv(gy, dt) && alv(gx, dt)) {
        const long windows4 = (long)n * oh * ow * (c / 4);
        CN_DISPATCH_DT(dt, hipLaunchKernelGGL((maxpool2_bwd_kernel<T>), dim3(ew_blocks((size_t)windows4)), dim3(256), 0, (hipStream_t)stream,
                                              (const T*)x, (const T*)gy, (T*)gx, windows4, oh, ow, c / 4));
        CN_LAUNCH_CHECK();
        return CN_OK;
    }
    if (c % 4 == 0 && alv(x, dt) && alv(gy, dt) && alv(gx, dt) && total / 4 < 2147483647UL) {
        CN_DISPATCH_DT(dt, hipLaunchKernelGGL((maxpool_bwd4_kernel<T>), dim3(ew_blocks(total / 4)), dim3(256), 0, (hipStream_t)stream,
                                              (const T*)x, (const T*)gy, (T*)gx, n, h, w, c / 4, oh, ow, k, s, pad));
        CN_LAUNCH_CHECK();
        return CN_OK;
    }
    CN_DISPATCH_DT(dt, hipLaunchKernelGGL((maxpool_bwd_kernel<T>), dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)gy, (T*)gx, n, h, w, c, oh, ow, k, s, pad));
    CN_LAUNCH_CHECK();
    return CN_OK;
}
// Backward of ReLU -> MaxPooling2D(2, 2) in one pass (round 6): gx = (routed gy [+ (x - target) s[sample] k]) relu'(x), x = the
// ReLU output that was pooled (perceptual_loss.py:19-41: VGG conv1_2 / conv2_2 / conv3_4; conv1_2 is also a tap).  Windows of
// 2 x 2 / stride 2 on even extents with c % 4 == 0 and 16-byte aligned tensors; anything else: CN_EUNSUPPORTED, nothing launched.
extern "C" int cn_maxpool2_bwd_act(const void* x, const void* gy, const void* target, const float* s, float k, void* gx, int n, int h,
                                   int w, int c, int s_rows, int dt, void* stream) {
    CN_CHECK_ARG(x && gy && gx && n > 0 && h > 0 && w > 0 && c > 0 && (dt == CN_F32 || dt == CN_BF16) && (!target || (s && (s_rows == 1 || s_rows == n))),
                 "maxpool2_bwd_act: bad args");
    if (h % 2 || w % 2 || c % 4 || !alv(x, dt) || !alv(gy, dt) || !alv(gx, dt) || (target && !alv(target, dt))) return CN_EUNSUPPORTED;
    const int oh = h / 2, ow = w / 2;
    const long windows4 = (long)n * oh * ow * (c / 4);
    CN_DISPATCH_DT(dt, hipLaunchKernelGGL((maxpool2_bwd_kernel<T>), dim3(ew_blocks((size_t)windows4)), dim3(256), 0, (hipStream_t)stream,
                                          (const T*)x, (const T*)gy, (T*)gx, windows4, oh, ow, c / 4, 1, (const T*)target, s, s_rows, k));
    CN_LAUNCH_CHECK();
    return CN_OK;
}
extern "C" int cn_chan_affine3_fwd(const float* x, float* y, size_t pixels, const int* perm, float scale, const float* off, void* stream) {
    CN_CHECK_ARG(x && y && perm && off, "chan_affine3: NULL");
    for (int i = 0; i < 3; ++i) CN_CHECK_ARG(perm[i] >= 0 && perm[i] < 3, "chan_affine3: bad perm");
    EW_LAUNCH(chan_affine3_fwd_kernel, pixels, x, y, pixels, perm[0], perm[1], perm[2], scale, off[0], off[1], off[2])
}
extern "C" int cn_chan_affine3_bwd(const float* gy, float* gx, size_t pixels, const int* perm, float scale, void* stream) {
    CN_CHECK_ARG(gy && gx && perm, "chan_affine3_bwd: NULL");
    for (int i = 0; i < 3; ++i) CN_CHECK_ARG(perm[i] >= 0 && perm[i] < 3, "chan_affine3: bad perm");
    EW_LAUNCH(chan_affine3_bwd_kernel, pixels, gy, gx, pixels, perm[0], perm[1], perm[2], scale)
}
extern "C" int cn_gan_loss_fwd(const float* s, float* out, int n, float label, void* stream) {
    CN_CHECK_ARG(s && out && n > 0, "gan_loss: bad args");
    hipLaunchKernelGGL(gan_loss_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, s, out, n, label);
    CN_LAUNCH_CHECK();
    return CN_OK;
}
extern "C" int cn_gan_loss_bwd(const float* s, const float* gout, float* gs, int n, float label, void* stream) {
    CN_CHECK_ARG(s && gout && gs && n > 0, "gan_loss_bwd: bad args");
    hipLaunchKernelGGL(gan_loss_bwd_kernel, dim3(cn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, s, gout, gs, n, label);
    CN_LAUNCH_CHECK();
    return CN_OK;
}
// ---- the GAN losses of every head of one discriminator call in ONE launch (round 6): a workgroup per head -------------------
namespace {
constexpr int CN_GAN_GROUP = 16;
struct GanJobs {
    const float* s[CN_GAN_GROUP];
    float* out[CN_GAN_GROUP];            // forward: the head's scalar; backward: the gradient w.r.t. its scores
    const float* gout[CN_GAN_GROUP];     // backward: the scalar's cotangent (NULL: zero)
    int n[CN_GAN_GROUP];
    float label[CN_GAN_GROUP];
};
__global__ __launch_bounds__(256) void gan_loss_grouped_fwd_kernel(GanJobs J) {
    const int j = blockIdx.x;
    const float* __restrict__ s = J.s[j];
    const int n = J.n[j];
    const float label = J.label[j];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += label * softplus_f(-s[i]) + (1.f - label) * softplus_f(s[i]);
    const float t = block_sum(acc);
    if (threadIdx.x == 0) J.out[j][0] = t / (float)n;
}
__global__ __launch_bounds__(256) void gan_loss_grouped_bwd_kernel(GanJobs J) {
    const int j = blockIdx.x;
    const float* __restrict__ s = J.s[j];
    float* __restrict__ gs = J.out[j];
    const int n = J.n[j];
    const float label = J.label[j];
    const float g = J.gout[j] ? J.gout[j][0] : 0.f;
    for (int i = threadIdx.x; i < n; i += 256) gs[i] = g * (-label * sigmoid_f(-s[i]) + (1.f - label) * sigmoid_f(s[i])) / (float)n;
}
}  // namespace
// Per job the arithmetic of cn_gan_loss_fwd / cn_gan_loss_bwd (losses.py:7-11); `jobs` is a HOST array, at most 16 jobs per launch.
extern "C" int cn_gan_loss_grouped(const CnGanJob* jobs, int njobs, int backward, void* stream) {
    CN_CHECK_ARG(njobs >= 0 && (njobs == 0 || jobs), "cn_gan_loss_grouped: bad arguments");
    for (int first = 0; first < njobs; first += CN_GAN_GROUP) {
        GanJobs J{};
        const int cnt = njobs - first < CN_GAN_GROUP ? njobs - first : CN_GAN_GROUP;
        for (int q = 0; q < cnt; ++q) {
            const CnGanJob& d = jobs[first + q];
            CN_CHECK_ARG(d.s && d.out && d.n > 0, "cn_gan_loss_grouped: job %d: NULL tensor or n = %d", first + q, d.n);
            J.s[q] = d.s; J.out[q] = d.out; J.gout[q] = d.gout; J.n[q] = d.n; J.label[q] = d.label;
        }
        if (backward) hipLaunchKernelGGL(gan_loss_grouped_bwd_kernel, dim3(cnt), dim3(256), 0, (hipStream_t)stream, J);
        else hipLaunchKernelGGL(gan_loss_grouped_fwd_kernel, dim3(cnt), dim3(256), 0, (hipStream_t)stream, J);
        CN_LAUNCH_CHECK();
    }
    return CN_OK;
}
extern "C" int cn_adam_step(float* theta, const float* grad, float* m, float* v, float* ema, size_t numel,
                            const float* lr_t, float beta1, float beta2, float eps, float ema_alpha, void* stream) {
    CN_CHECK_ARG(theta && grad && m && v && lr_t, "adam: NULL");
    if (!numel) return CN_OK;
    EW_LAUNCH(adam_kernel, numel, theta, grad, m, v, ema, numel, lr_t, beta1, beta2, eps, ema_alpha)
}
extern "C" int cn_ema_step(float* ema, const float* theta, size_t numel, float alpha, void* stream) {
    CN_CHECK_ARG(ema && theta, "ema: NULL");
    if (!numel) return CN_OK;
    EW_LAUNCH(ema_kernel, numel, ema, theta, numel, alpha)
}
extern "C" int cn_gather_images_u8(const uint8_t* pool, const int64_t* idx, const uint8_t* flip, float* out, int n, int h,
                                   int w, int c, void* stream) {
    CN_CHECK_ARG(pool && idx && out && n > 0, "gather_images: bad args");
    const size_t total = (size_t)n * h * w * c;
    EW_LAUNCH(gather_u8_kernel, total, pool, idx, flip, out, n, h, w, c)
}
namespace {
// dst (packed) = src[segment] * a[column]: every convolution filter of a network scaled per output channel in ONE launch.
// seg: 5 ints per segment -- source offset, packed destination offset, element count, columns (cout), coefficient offset; all
// multiples of 4 floats.  A thread owns 4 consecutive destination elements and finds its segment by bisection.
__global__ void scale_columns_segments_kernel(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ seg,
                                              const float* __restrict__ a, int nseg, long total4) {
    for (long e4 = (long)blockIdx.x * blockDim.x + threadIdx.x; e4 < total4; e4 += (long)gridDim.x * blockDim.x) {
        const int e = (int)(e4 * 4);
        int lo = 0, hi = nseg - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (seg[5 * mid + 1] <= e) lo = mid; else hi = mid - 1;
        }
        const int* sg = seg + 5 * lo;
        const int local = e - sg[1];
        if (local >= sg[2]) continue;                       // (padding between packed segments)
        const int col = local % sg[3];
        const float4 v = *reinterpret_cast<const float4*>(src + sg[0] + local);
        const float4 c = *reinterpret_cast<const float4*>(a + sg[4] + col);
        *reinterpret_cast<float4*>(dst + e) = make_float4(v.x * c.x, v.y * c.y, v.z * c.z, v.w * c.w);
    }
}

// The adjoint of the folding above, for every listed filter in one launch (round 6).  The taped ResNet-50 runs on the folded
// filters w' = w * a[c], shift = beta + a (b - mean), a = gamma * rs, rs = rsqrt(var + eps); its backward pass leaves the folded
// filters' gradients g' (packed like w') and the shifts' gradients gs (concatenated like a) behind, and this kernel ADDS
//   d w[k][c] = g'[k][c] a[c],  d gamma[c] = rs[c] (sum_k g'[k][c] w[k][c] + gs[c] (b - mean)[c]),  d beta[c] = gs[c],  d b[c] = a[c] gs[c]
// into `gout`, which is laid out like the weight arena.  seg: 9 ints per segment -- the five of cn_scale_columns_segments, then
// the arena offsets of the layer's bias, gamma and beta, then the segment's first workgroup (one workgroup per 64 columns).
// A workgroup is 16 float4 column lanes x 16 row lanes; the row lanes' partial dot products are added in lane order (no atomics).
__global__ __launch_bounds__(256) void bn_fold_bwd_kernel(const int* __restrict__ seg, int nseg, const float* __restrict__ gwf,
                                                          const float* __restrict__ gshift, const float* __restrict__ arena,
                                                          const float* __restrict__ a, const float* __restrict__ rs,
                                                          const float* __restrict__ bm, float* __restrict__ gout) {
    const int b = blockIdx.x;
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (seg[9 * mid + 8] <= b) lo = mid; else hi = mid - 1;
    }
    const int* sg = seg + 9 * lo;
    const int cout = sg[3], K = sg[2] / cout;
    const int c4 = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int col = (b - sg[8]) * 64 + c4 * 4;
    const float4 av = *reinterpret_cast<const float4*>(a + sg[4] + col);
    float4 dot = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = rl; r < K; r += 16) {
        const long off = (long)r * cout + col;
        const float4 g = *reinterpret_cast<const float4*>(gwf + sg[1] + off);
        const float4 w = *reinterpret_cast<const float4*>(arena + sg[0] + off);
        float4* d = reinterpret_cast<float4*>(gout + sg[0] + off);
        float4 o = *d;
        o.x += g.x * av.x; o.y += g.y * av.y; o.z += g.z * av.z; o.w += g.w * av.w;
        *d = o;
        dot.x += g.x * w.x; dot.y += g.y * w.y; dot.z += g.z * w.z; dot.w += g.w * w.w;
    }
    __shared__ float4 red[16][16];
    red[rl][c4] = dot;
    __syncthreads();
    if (rl == 0) {
        float4 t = red[0][c4];
        for (int q = 1; q < 16; ++q) { const float4 u = red[q][c4]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        const float4 gs = *reinterpret_cast<const float4*>(gshift + sg[4] + col);
        const float4 m = *reinterpret_cast<const float4*>(bm + sg[4] + col);
        const float4 rv = *reinterpret_cast<const float4*>(rs + sg[4] + col);
        float4* gg = reinterpret_cast<float4*>(gout + sg[6] + col);
        float4* gbeta = reinterpret_cast<float4*>(gout + sg[7] + col);
        float4* gbias = reinterpret_cast<float4*>(gout + sg[5] + col);
        float4 o = *gg;
        o.x += rv.x * (t.x + gs.x * m.x); o.y += rv.y * (t.y + gs.y * m.y); o.z += rv.z * (t.z + gs.z * m.z); o.w += rv.w * (t.w + gs.w * m.w);
        *gg = o;
        o = *gbeta;
        o.x += gs.x; o.y += gs.y; o.z += gs.z; o.w += gs.w;
        *gbeta = o;
        o = *gbias;
        o.x += av.x * gs.x; o.y += av.y * gs.y; o.z += av.z * gs.z; o.w += av.w * gs.w;
        *gbias = o;
    }
}
}  // namespace

// The backward of cn_scale_columns_segments together with the BatchNorm coefficient algebra (real_encoder.py:13; see the kernel).
// seg: nseg x 9 ints on the device, `blocks` = the sum over the segments of cout / 64 (every cout a multiple of 64).
extern "C" int cn_bn_fold_bwd(const int* seg, int nseg, int blocks, const float* gwf, const float* gshift, const float* arena,
                              const float* a, const float* rs, const float* bm, float* gout, void* stream) {
    CN_CHECK_ARG(seg && gwf && gshift && arena && a && rs && bm && gout && nseg > 0 && blocks > 0, "bn_fold_bwd: bad args");
    hipLaunchKernelGGL(bn_fold_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, seg, nseg, gwf, gshift, arena, a, rs,
                       bm, gout);
    CN_LAUNCH_CHECK();
    return CN_OK;
}

// BatchNormalization (inference) folded into the preceding convolutions' filters: w'[k][c] = w[k][c] * a[c] for every listed
// filter of one weight arena in one launch (real_encoder.py:13: keras ResNet50 called without training=).
extern "C" int cn_scale_columns_segments(const float* src, float* dst, const int* seg, const float* a, int nseg, size_t total,
                                         void* stream) {
    CN_CHECK_ARG(src && dst && seg && a && nseg > 0 && total % 4 == 0, "scale_columns_segments: bad args");
    if (!total) return CN_OK;
    const long total4 = (long)(total / 4);
    hipLaunchKernelGGL(scale_columns_segments_kernel, dim3(ew_blocks((size_t)total4)), dim3(256), 0, (hipStream_t)stream, src, dst, seg, a,
                       nseg, total4);
    CN_LAUNCH_CHECK();
    return CN_OK;
}

extern "C" int cn_to_uint8(const float* x, uint8_t* out, size_t numel, void* stream) {
    CN_CHECK_ARG(x && out, "to_uint8: NULL");
    if (!numel) return CN_OK;
    EW_LAUNCH(to_uint8_kernel, numel, x, out, numel)
}
