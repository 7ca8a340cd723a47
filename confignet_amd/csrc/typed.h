// typed.h -- storage-type helpers: activations live in HBM as fp32 (CN_F32) or bf16 (CN_BF16); arithmetic,
// statistics, coefficients, loss sums, gradients of parameters and optimizer state are always fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;                  // raw bfloat16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float x) {       // round to nearest even (NaN stays NaN)
    const unsigned u = __float_as_uint(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);
    return (bf16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
}

// scalar / 4-wide loads and stores of a tensor stored as T (float or bf16_t); the 4-wide forms need
// 4*sizeof(T)-byte alignment
template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }

template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

template <typename T> __device__ __forceinline__ float4 ld4(const T* p);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 ld4<bf16_t>(const bf16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}

template <typename T> __device__ __forceinline__ void st4(T* p, float4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, float4 v) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
}

// runtime dispatch on the dtype code of the C ABI
#define CN_DISPATCH_DT(dt, ...)                                     \
    do {                                                            \
        if ((dt) == CN_BF16) { typedef bf16_t T; __VA_ARGS__; }     \
        else { typedef float T; __VA_ARGS__; }                      \
    } while (0)
