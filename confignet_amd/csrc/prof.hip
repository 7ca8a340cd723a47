// prof.hip -- error string + HIP-event profiling of the dominant kernel class.
#include "common.h"
#include <stdarg.h>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "no error";

void cn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* cn_last_error_string(void) { return g_err; }
extern "C" int cn_version(void) { return 1; }

__global__ void spin_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

extern "C" int cn_spin(unsigned long long ticks, void* stream) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ticks);
    CN_LAUNCH_CHECK();
    return CN_OK;
}

namespace {
struct Rec { hipEvent_t a, b; double flops, bytes; int family; };
std::mutex g_mu;
bool g_on = false;
std::vector<Rec> g_pool;      // allocated event pairs
size_t g_used = 0;            // pairs used since the last reset
}  // namespace

void cn_prof_begin(hipStream_t s, double flops, double bytes, int family) {
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_used == g_pool.size()) {
        Rec r;
        r.flops = 0;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
        g_pool.push_back(r);
    }
    g_pool[g_used].flops = flops;
    g_pool[g_used].bytes = bytes;
    g_pool[g_used].family = family;
    (void)hipEventRecord(g_pool[g_used].a, s);
}

void cn_prof_end(hipStream_t s) {
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_used < g_pool.size()) {
        (void)hipEventRecord(g_pool[g_used].b, s);
        ++g_used;
    }
}

extern "C" int cn_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    return CN_OK;
}

extern "C" int cn_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_used = 0;
    return CN_OK;
}

extern "C" int cn_prof_collect(int* launches, double* total_ms, double* total_flops) {
    std::lock_guard<std::mutex> lk(g_mu);
    double ms = 0, fl = 0;
    for (size_t i = 0; i < g_used; ++i) {
        CN_HIP(hipEventSynchronize(g_pool[i].b));
        float t = 0;
        CN_HIP(hipEventElapsedTime(&t, g_pool[i].a, g_pool[i].b));
        ms += t;
        fl += g_pool[i].flops;
    }
    if (launches) *launches = (int)g_used;
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    return CN_OK;
}


// Per kernel family (CN_FAM_* of common.h; slot 31 collects anything above): launches, summed ms, issued flops and ALGORITHMIC
// bytes (every operand read once, the result written once) -- the byte model the PMC FETCH_SIZE / WRITE_SIZE of the same
// kernels are compared with (profiles/round3_pmc_traffic_by_kernel.txt).  Arrays of 32 entries each.
extern "C" int cn_prof_collect_by_family(int* launches, double* ms, double* flops, double* bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (int f = 0; f < 32; ++f) { launches[f] = 0; ms[f] = flops[f] = bytes[f] = 0.0; }
    for (size_t i = 0; i < g_used; ++i) {
        CN_HIP(hipEventSynchronize(g_pool[i].b));
        float t = 0;
        CN_HIP(hipEventElapsedTime(&t, g_pool[i].a, g_pool[i].b));
        const int f = g_pool[i].family < 0 || g_pool[i].family > 31 ? 31 : g_pool[i].family;
        ++launches[f];
        ms[f] += t;
        flops[f] += g_pool[i].flops;
        bytes[f] += g_pool[i].bytes;
    }
    return CN_OK;
}

// ---- deterministic mode ------------------------------------------------------------------------------------------
namespace {
int g_det = 0;
std::mutex g_det_mu;
constexpr int CN_DET_SLOTS = 16;
float* g_det_buf[CN_DET_SLOTS] = {};
hipStream_t g_det_stream[CN_DET_SLOTS];
bool g_det_pinned[CN_DET_SLOTS] = {};      // handed out during a stream capture: a graph may replay with this pointer baked in
unsigned long long g_det_last_use[CN_DET_SLOTS] = {};
unsigned long long g_det_tick = 0;
int g_det_used = 0;

__global__ void sum_parts_kernel(const float* __restrict__ src, float* __restrict__ dst, int parts, long count, int accumulate,
                                 float scale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float t = 0.f;
    for (int p = 0; p < parts; ++p) t += src[(long)p * count + i];
    t *= scale;
    if (accumulate) unsafeAtomicAdd(&dst[i], t);       // one writer per element and launch; concurrent launches may share dst
    else dst[i] = t;
}
}  // namespace

int cn_cu_count() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return cus;
}

extern "C" int cn_set_deterministic(int on) {
    std::lock_guard<std::mutex> lk(g_det_mu);
    if (on && !g_det_buf[0]) {
        for (int i = 0; i < CN_DET_SLOTS; ++i) CN_HIP(hipMalloc((void**)&g_det_buf[i], sizeof(float) * CN_DET_WS_FLOATS));
    }
    g_det = on ? 1 : 0;
    return CN_OK;
}

extern "C" int cn_get_deterministic(void) { return g_det; }

int cn_det() { return g_det; }

float* cn_det_ws(hipStream_t s, size_t need_floats) {
    std::lock_guard<std::mutex> lk(g_det_mu);
    if (!g_det_buf[0] || need_floats > CN_DET_WS_FLOATS) {
        cn_set_error("deterministic workspace: %zu floats requested, %zu available per stream", need_floats, g_det_buf[0] ? CN_DET_WS_FLOATS : (size_t)0);
        return nullptr;
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusActive;
    for (int i = 0; i < g_det_used; ++i)
        if (g_det_stream[i] == s) {
            g_det_last_use[i] = ++g_det_tick;
            g_det_pinned[i] |= capturing;
            return g_det_buf[i];
        }
    int slot = g_det_used;
    if (g_det_used == CN_DET_SLOTS) {
        // more streams than slots over the life of the process (models come and go): the least recently used binding that
        // NO CAPTURED GRAPH can still reference is handed to the new stream.  A binding handed out during a capture is pinned
        // until its owner calls cn_det_release_stream(): re-binding it would let a replayed graph share scratch with a live stream.
        slot = -1;
        for (int i = 0; i < CN_DET_SLOTS; ++i)
            if (!g_det_pinned[i] && (slot < 0 || g_det_last_use[i] < g_det_last_use[slot])) slot = i;
        if (slot < 0) {
            cn_set_error("deterministic workspace: all %d per-stream workspaces are referenced by captured graphs "
                         "(release their streams with cn_det_release_stream)", CN_DET_SLOTS);
            return nullptr;
        }
    } else {
        ++g_det_used;
    }
    g_det_stream[slot] = s;
    g_det_last_use[slot] = ++g_det_tick;
    g_det_pinned[slot] = capturing;
    return g_det_buf[slot];
}

extern "C" int cn_det_release_stream(void* stream) {
    std::lock_guard<std::mutex> lk(g_det_mu);
    for (int i = 0; i < g_det_used; ++i)
        if (g_det_stream[i] == (hipStream_t)stream) {
            g_det_pinned[i] = false;
            g_det_last_use[i] = 0;              // first in line for re-binding
            g_det_stream[i] = (hipStream_t)(~(uintptr_t)0 - (uintptr_t)i);   // matches no live stream handle
        }
    return CN_OK;
}

namespace {
// many parts: 16 lanes per output element take every 16th part (independent loads instead of one chain of `parts` dependent
// HBM latencies: 512 parts took 117 us that way), then the 16 sub-sums are added in lane order -- still a fixed order
__global__ __launch_bounds__(256) void sum_parts_wide_kernel(const float* __restrict__ src, float* __restrict__ dst, int parts,
                                                             long count, int accumulate, float scale) {
    __shared__ float red[16][17];
    const int o = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const long i = (long)blockIdx.x * 16 + o;
    float t = 0.f;
    if (i < count)
        for (int p = grp; p < parts; p += 16) t += src[(long)p * count + i];
    red[grp][o] = t;
    __syncthreads();
    if (grp == 0 && i < count) {
        float u = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) u += red[k][o];
        u *= scale;
        if (accumulate) unsafeAtomicAdd(&dst[i], u);
        else dst[i] = u;
    }
}
}  // namespace

int cn_sum_parts(const float* src, float* dst, int parts, long count, int accumulate, float scale, hipStream_t s) {
    if (parts >= 64 && count * 16 <= (long)256 * 65536) {
        hipLaunchKernelGGL(sum_parts_wide_kernel, dim3(cn_cdiv(count, 16)), dim3(256), 0, s, src, dst, parts, count, accumulate, scale);
        CN_LAUNCH_CHECK();
        return CN_OK;
    }
    hipLaunchKernelGGL(sum_parts_kernel, dim3(cn_cdiv(count, 256)), dim3(256), 0, s, src, dst, parts, count, accumulate, scale);
    CN_LAUNCH_CHECK();
    return CN_OK;
}

// ---- many ordered slab reductions in ONE launch (round 6: the filter gradients of a backward pass leave their row slices' slabs
// behind and the pass adds all of them at its join -- one launch instead of one per layer on the backward chain).  Every job is
// reduced exactly as cn_sum_parts would reduce it alone (the same two schemes, the same order): the results are the same bits.
namespace {
constexpr int CN_SUM_GROUP = 80;          // jobs per launch (the table travels in the kernel arguments: < 4 KB)
struct SumJobs {
    const float* src[CN_SUM_GROUP];
    float* dst[CN_SUM_GROUP];
    long count[CN_SUM_GROUP];
    int parts[CN_SUM_GROUP];
    int flags[CN_SUM_GROUP];              // bit 0: accumulate, bit 1: the wide scheme
    int blk0[CN_SUM_GROUP + 1];           // first workgroup of job j
    int n;
};

__global__ __launch_bounds__(256) void sum_parts_grouped_kernel(SumJobs J) {
    __shared__ float red[16][17];
    const int b = blockIdx.x;
    int lo = 0, hi = J.n;                 // job of this workgroup: blk0[lo] <= b < blk0[lo + 1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (J.blk0[mid] <= b) lo = mid;
        else hi = mid;
    }
    const int j = __builtin_amdgcn_readfirstlane(lo), lb = b - J.blk0[j];
    const float* __restrict__ src = J.src[j];
    float* __restrict__ dst = J.dst[j];
    const long count = J.count[j];
    const int parts = J.parts[j], accumulate = J.flags[j] & 1;
    if (J.flags[j] & 2) {                 // (sum_parts_wide_kernel)
        const int o = threadIdx.x & 15, grp = threadIdx.x >> 4;
        const long i = (long)lb * 16 + o;
        float t = 0.f;
        if (i < count)
            for (int p = grp; p < parts; p += 16) t += src[(long)p * count + i];
        red[grp][o] = t;
        __syncthreads();
        if (grp == 0 && i < count) {
            float u = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) u += red[k][o];
            if (accumulate) unsafeAtomicAdd(&dst[i], u);
            else dst[i] = u;
        }
    } else {                              // (sum_parts_kernel)
        const long i = (long)lb * 256 + threadIdx.x;
        if (i >= count) return;
        float t = 0.f;
        for (int p = 0; p < parts; ++p) t += src[(long)p * count + i];
        if (accumulate) unsafeAtomicAdd(&dst[i], t);
        else dst[i] = t;
    }
}
}  // namespace

extern "C" int cn_sum_parts_grouped(const CnSumJob* jobs, int njobs, void* stream) {
    CN_CHECK_ARG(njobs >= 0 && (njobs == 0 || jobs), "cn_sum_parts_grouped: bad arguments");
    for (int first = 0; first < njobs; first += CN_SUM_GROUP) {
        SumJobs J{};
        const int n = njobs - first < CN_SUM_GROUP ? njobs - first : CN_SUM_GROUP;
        long blocks = 0;
        for (int k = 0; k < n; ++k) {
            const CnSumJob& q = jobs[first + k];
            CN_CHECK_ARG(q.src && q.dst && q.count > 0 && q.parts > 0, "cn_sum_parts_grouped: job %d is empty", first + k);
            const bool wide = !(q.accumulate & 2) && q.parts >= 64 && q.count * 16 <= (long)256 * 65536;      // (cn_sum_parts' rule)
            J.src[k] = q.src; J.dst[k] = q.dst; J.count[k] = q.count; J.parts[k] = q.parts;
            J.flags[k] = ((q.accumulate & 1) ? 1 : 0) | (wide ? 2 : 0);
            J.blk0[k] = (int)blocks;
            blocks += cn_cdiv(q.count, wide ? 16 : 256);
            CN_CHECK_ARG(blocks < 0x7fffffffL, "cn_sum_parts_grouped: too many workgroups");
        }
        J.blk0[n] = (int)blocks;
        J.n = n;
        // (part of the convolution class' time: these are the row slices' reductions of the filter gradients -- no FLOP, the slabs
        // read + the filters written as algorithmic-free bytes, i.e. counted as time only)
        cn_prof_begin((hipStream_t)stream, 0.0, 0.0, CN_FAM_WGRAD_SLAB_SUM);
        hipLaunchKernelGGL(sum_parts_grouped_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, J);
        cn_prof_end((hipStream_t)stream);
        CN_LAUNCH_CHECK();
    }
    return CN_OK;
}

namespace {
__global__ void zero_kernel(float* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.f;
}
}  // namespace

namespace {
__global__ void zero4_kernel(float4* __restrict__ p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
}  // namespace

// Clears a buffer (bytes % 4 == 0) with a kernel: a whole gradient arena at the start of a backward pass.
extern "C" int cn_zero(void* p, size_t bytes, void* stream) {
    if (!bytes) return CN_OK;
    CN_CHECK_ARG(p && bytes % 4 == 0, "cn_zero: bad buffer");
    hipStream_t s = (hipStream_t)stream;
    if (((uintptr_t)p & 15) == 0 && bytes >= 65536) {
        const size_t n4 = bytes / 16;
        size_t blocks = (n4 + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(zero4_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (float4*)p, n4);
        CN_LAUNCH_CHECK();
        p = (char*)p + n4 * 16;
        bytes -= n4 * 16;
        if (!bytes) return CN_OK;
    }
    return cn_zero_async(p, bytes, s);
}

int cn_zero_async(void* p, size_t bytes, hipStream_t s) {
    if (!bytes) return CN_OK;
    CN_CHECK_ARG(p && bytes % 4 == 0, "cn_zero_async: bad buffer");
    const size_t n = bytes / 4;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (float*)p, n);
    CN_LAUNCH_CHECK();
    return CN_OK;
}
