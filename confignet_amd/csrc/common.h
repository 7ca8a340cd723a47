// common.h -- shared helpers of libconfignet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/confignet_hip.h"

void cn_set_error(const char* fmt, ...);

#define CN_CHECK_ARG(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            cn_set_error(__VA_ARGS__);     \
            return CN_EINVAL;              \
        }                                  \
    } while (0)

#define CN_HIP(expr)                                                              \
    do {                                                                          \
        hipError_t e__ = (expr);                                                  \
        if (e__ != hipSuccess) {                                                  \
            cn_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return CN_EHIP;                                                       \
        }                                                                         \
    } while (0)

#define CN_LAUNCH_CHECK()                                                         \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            cn_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
            return CN_EHIP;                                                       \
        }                                                                         \
    } while (0)

// Zero `bytes` (multiple of 4) bytes at p with an ordinary kernel.  hipMemsetAsync nodes were observed NOT to
// re-execute on HIP-graph replay (ROCm 7.2, gfx950: accumulators kept growing across replays), so every
// accumulate-into buffer is cleared by a kernel node instead.
int cn_zero_async(void* p, size_t bytes, hipStream_t s);

// Deterministic mode (cn_set_deterministic, prof.hip): reductions in a fixed order instead of fp32 atomics.  Kernels that
// need room for their partial results take it from a per-stream workspace (allocated when the mode is switched on).
int cn_det();
float* cn_det_ws(hipStream_t s, size_t need_floats);     // NULL (+ error string) if the request does not fit
constexpr size_t CN_DET_WS_FLOATS = (size_t)16 << 20;     // 64 MiB per stream, 16 streams (prof.hip: CN_DET_SLOTS; 1 GiB, allocated when the mode is first switched on)
// dst[i] (+)= scale * sum_{p < parts} src[p * count + i], parts added in index order (one thread per i)
int cn_sum_parts(const float* src, float* dst, int parts, long count, int accumulate, float scale, hipStream_t s);

// fwd2.hip: C[M][N] (+)= A[M][K] B for the 1x1 stride-1 convolutions (gp = NULL) and, with a geometry, the same main loop over the
// gathered rows of a vec convolution (par: parity-ordered rows); tile cfg 0 / 1 / 2 / 3 / 4 of the implicit-GEMM numbering, bt = B is the
// original filter [N][K] (data gradient), the split-K protocol of igemm_fwd_kernel; both operands go straight into LDS (LDS-DMA), NS
// stages deep; x_elems / w_elems size the buffer descriptors; CN_EUNSUPPORTED for anything it does not take
int cn_fwd2(const CnConvGeom* gp, int cfg, int bt, const float* A, const float* B, const float* bias, float* C, long M, int N, int K,
            int act, float slope, int splits, long part_stride, int par, hipStream_t s, const float* res, double x_elems, double w_elems,
            float* stats = nullptr, int stats_mode = 0, float stats_slope = 0.f, int srows = 1, int sper = 1);
void cn_fwd2_tune(int kb, int ns, int np);
int cn_fwd2_bf16(const CnConvGeom& g, int cfg, int flip, const void* x, const void* wb, const float* bias, void* y, int act, float slope,
                 int par, hipStream_t s);

// profiling hooks (prof.hip): bracket one launch of the dominant kernel class
// family: which kernel of the class is launched (cn_prof_collect_by_family); bytes: the launch's algorithmic HBM bytes
enum { CN_FAM_FWD_128x128 = 0, CN_FAM_FWD_128x64, CN_FAM_FWD_64x64, CN_FAM_FWD_128x32, CN_FAM_FWD_128x96, CN_FAM_WGRAD_128x128,
       CN_FAM_WGRAD_128x96, CN_FAM_WGRAD_64x64, CN_FAM_WGRAD_128x32, CN_FAM_WINO, CN_FAM_C3_FWD, CN_FAM_S2_IMAGE_DGRAD,
       CN_FAM_THIN, CN_FAM_C3_WGRAD, CN_FAM_BF16_FWD, CN_FAM_BF16_WGRAD, CN_FAM_WGRAD_256x64, CN_FAM_WGRAD_SLAB_SUM };
void cn_prof_begin(hipStream_t s, double flops, double bytes = 0.0, int family = 31);
void cn_prof_end(hipStream_t s);

// q = m / d, r = m % d for m >= 0, d > 0.  The extents of this path are powers of two almost everywhere (256 / 128 / ... / 4 pixels,
// 2 x 2 parity classes): a shift and a mask behind a wave-uniform test instead of the ~40-instruction software division (the target
// has no integer divide) -- a workgroup's prologue decodes its rows with 3 - 11 of them per row.
__device__ __forceinline__ void divmod_pos(int m, int d, int& q, int& r) {
    if ((d & (d - 1)) == 0) {
        q = m >> (__builtin_ctz(d));
        r = m & (d - 1);
    } else {
        q = m / d;
        r = m - q * d;
    }
}

__device__ __forceinline__ float cn_apply_act(float v, int act, float slope) {
    switch (act) {
        case CN_ACT_LRELU: return v > 0.f ? v : v * slope;
        case CN_ACT_RELU: return v > 0.f ? v : 0.f;
        case CN_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

static inline int cn_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// compute units of the current device (hipDeviceAttributeMultiprocessorCount, read once; 256 on an MI355X): the launch cost models
// count workgroup rounds with it
int cn_cu_count();
