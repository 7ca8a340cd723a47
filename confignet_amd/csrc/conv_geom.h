// conv_geom.h -- geometry of the implicit-GEMM convolution family shared by igemm_conv.hip (fp32 MFMA) and igemm_bf16.hip
// (bf16 MFMA): row decoding, the gather predicate (TF SAME padding, folded x2 upsample, zero-stuffed strided data gradients),
// parity-class row order, exact algorithmic FLOP count, argument checks.
#pragma once
#include "common.h"

namespace {

struct RowInfo {
    int nbase, vd, vh, vw;
    bool ok;
};

__device__ __forceinline__ RowInfo decode_row(const CnConvGeom& g, int m, int M) {
    RowInfo r;
    r.ok = m < M;
    if (!r.ok) m = 0;
    int ow, oh, od, n, t;
    divmod_pos(m, g.out_w, t, ow);
    divmod_pos(t, g.out_h, t, oh);
    divmod_pos(t, g.out_d, n, od);
    r.nbase = n * g.in_d;
    r.vd = od * g.s_d - g.p_d;
    r.vh = oh * g.s_h - g.p_h;
    r.vw = ow * g.s_w - g.p_w;
    return r;
}

// Parity-class row order for the data-gradient of a strided convolution (dl > 1, stride 1 on the output
// side, out % dl == 0): rows are enumerated class-major, class = (od % dl_d, oh % dl_h, ow % dl_w), so
// that all rows of a tile hit the zero-stuffed positions for the SAME taps and those taps are skipped as a
// whole (4x fewer MFMAs for the 2-D stride-2 discriminator blocks).  Returns the true output row.
__device__ __forceinline__ int par_row(const CnConvGeom& g, int mp, int M, int& cls) {
    int qd, qh, qw, z_;
    divmod_pos(g.out_d, g.dl_d, qd, z_);
    divmod_pos(g.out_h, g.dl_h, qh, z_);
    divmod_pos(g.out_w, g.dl_w, qw, z_);
    const int per = g.n * qd * qh * qw;
    if (mp >= M) { cls = -1; return M; }
    int rem, cw, ch, cd, xw, xh, xd, n, t;
    divmod_pos(mp, per, cls, rem);
    divmod_pos(cls, g.dl_w, t, cw);
    divmod_pos(t, g.dl_h, cd, ch);
    divmod_pos(rem, qw, rem, xw);
    divmod_pos(rem, qh, rem, xh);
    divmod_pos(rem, qd, n, xd);
    return ((n * g.out_d + xd * g.dl_d + cd) * g.out_h + xh * g.dl_h + ch) * g.out_w + xw * g.dl_w + cw;
}

__device__ __forceinline__ unsigned long long par_tap_mask(const CnConvGeom& g, int cls) {
    const int cw = cls % g.dl_w, ch = (cls / g.dl_w) % g.dl_h, cd = cls / (g.dl_w * g.dl_h);
    unsigned long long mask = 0ull;
    int tap = 0;
    for (int kd = 0; kd < g.k_d; ++kd)
        for (int kh = 0; kh < g.k_h; ++kh)
            for (int kw = 0; kw < g.k_w; ++kw, ++tap) {
                const int vd = cd - g.p_d + kd, vh = ch - g.p_h + kh, vw = cw - g.p_w + kw;
                const bool ok = ((vd % g.dl_d) == 0) && ((vh % g.dl_h) == 0) && ((vw % g.dl_w) == 0);
                if (ok) mask |= 1ull << tap;
            }
    return mask;
}

__device__ __forceinline__ bool map1(int v, int dl, int ext, int up, int& q) {
    if (v < 0) return false;
    if (dl > 1) {
        int q_, r_;
        divmod_pos(v, dl, q_, r_);
        if (r_) return false;
        v = q_;
    }
    if (v >= ext) return false;
    q = v >> up;
    return true;
}

// element offset (channel 0) of the stored input element read by row r at tap (kd,kh,kw), or -1
__device__ __forceinline__ int src_off(const CnConvGeom& g, const RowInfo& r, int kd, int kh, int kw) {
    int qd, qh, qw;
    if (!r.ok) return -1;
    if (!map1(r.vd + kd, g.dl_d, g.in_d << g.up, g.up, qd)) return -1;
    if (!map1(r.vh + kh, g.dl_h, g.in_h << g.up, g.up, qh)) return -1;
    if (!map1(r.vw + kw, g.dl_w, g.in_w << g.up, g.up, qw)) return -1;
    return (((r.nbase + qd) * g.in_h + qh) * g.in_w + qw) * g.cin;
}

__device__ __forceinline__ void tap_decode(const CnConvGeom& g, int tap, int& kd, int& kh, int& kw) {
    kw = tap % g.k_w;
    int t = tap / g.k_w;
    kh = t % g.k_h;
    kd = t / g.k_h;
}

// exact number of (output position, tap) pairs that touch a stored element, per spatial axis
inline double valid_pairs_1d(int out, int k, int s, int dl, int p, int in, int up) {
    long cnt = 0;
    for (int o = 0; o < out; ++o)
        for (int kk = 0; kk < k; ++kk) {
            int v = o * s - p + kk;
            if (v < 0 || v % dl) continue;
            if (v / dl >= (in << up)) continue;
            ++cnt;
        }
    return (double)cnt;
}

inline double conv_flops(const CnConvGeom& g) {
    return 2.0 * g.n * valid_pairs_1d(g.out_d, g.k_d, g.s_d, g.dl_d, g.p_d, g.in_d, g.up) *
           valid_pairs_1d(g.out_h, g.k_h, g.s_h, g.dl_h, g.p_h, g.in_h, g.up) *
           valid_pairs_1d(g.out_w, g.k_w, g.s_w, g.dl_w, g.p_w, g.in_w, g.up) * g.cin * g.cout;
}

// algorithmic HBM bytes of one launch: input (stored extent) and filter read once, output written once; eb = bytes per element
inline double conv_bytes(const CnConvGeom& g, double eb_in = 4.0, double eb_out = 4.0, double eb_w = 4.0) {
    return eb_in * g.n * g.in_d * g.in_h * g.in_w * g.cin + eb_w * g.k_d * g.k_h * g.k_w * g.cin * g.cout +
           eb_out * g.n * g.out_d * g.out_h * g.out_w * g.cout;
}

// parity-class row order applies to data-gradient geometries of strided convolutions
inline bool parity_ordered(const CnConvGeom& g) {
    if (g.s_d != 1 || g.s_h != 1 || g.s_w != 1 || g.up) return false;
    if (g.dl_d * g.dl_h * g.dl_w == 1) return false;
    return g.out_d % g.dl_d == 0 && g.out_h % g.dl_h == 0 && g.out_w % g.dl_w == 0;
}

inline int check_geom(const CnConvGeom* g) {
    CN_CHECK_ARG(g != nullptr, "geom is NULL");
    CN_CHECK_ARG(g->nd == 2 || g->nd == 3, "nd must be 2 or 3 (got %d)", g->nd);
    CN_CHECK_ARG(g->n > 0 && g->cin > 0 && g->cout > 0, "empty batch/channels");
    CN_CHECK_ARG(g->in_d > 0 && g->in_h > 0 && g->in_w > 0 && g->out_d > 0 && g->out_h > 0 && g->out_w > 0, "empty extent");
    CN_CHECK_ARG(g->k_d > 0 && g->k_h > 0 && g->k_w > 0 && g->s_d > 0 && g->s_h > 0 && g->s_w > 0, "bad kernel/stride");
    CN_CHECK_ARG(g->dl_d > 0 && g->dl_h > 0 && g->dl_w > 0, "bad dilation divisor");
    CN_CHECK_ARG(g->up == 0 || g->up == 1, "up must be 0/1");
    CN_CHECK_ARG(g->nd == 3 || (g->in_d == 1 && g->out_d == 1 && g->k_d == 1), "2-D geometry must have depth 1");
    const double in_el = (double)g->n * g->in_d * g->in_h * g->in_w * g->cin;
    const double out_el = (double)g->n * g->out_d * g->out_h * g->out_w * g->cout;
    CN_CHECK_ARG(in_el < 2147483647.0 && out_el < 2147483647.0, "tensor exceeds 2^31 elements (32-bit offsets)");
    return CN_OK;
}

}  // namespace
