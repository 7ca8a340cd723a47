// slots.h -- helpers of the hand-scheduled LDS-DMA loops (fwd2.hip round 5, wgrad2.hip round 6).
//
// A wave issues in order and a v_mfma_f32_32x32x2_f32 holds the matrix pipe 64 cycles: whatever the wave issues between two
// MFMAs runs in that shadow for free as long as it issues in < 64 cycles, and whatever is NOT placed between two MFMAs is paid
// in full (fwd2.hip's ablation: a loop whose parts sat in blocks ran at the SUM of their times).  So these loops are written as
// a sequence of MFMA "slots" with a compile-time index, and every other instruction of a step -- operand reads, address
// arithmetic, LDS-DMA issue -- is assigned to a slot by a constant expression.  sl_static_for gives the loop index as a type so
// that it can feed asm immediates and `if constexpr`.
#pragma once

template <int N>
struct SlInt {
    static constexpr int value = N;
};

template <int I, int N, class F>
__device__ __forceinline__ void sl_static_for_impl(F&& f) {
    if constexpr (I < N) {
        f(SlInt<I>{});
        sl_static_for_impl<I + 1, N>(f);
    }
}

// f(SlInt<0>) ... f(SlInt<N - 1>)
template <int N, class F>
__device__ __forceinline__ void sl_static_for(F&& f) {
    sl_static_for_impl<0, N>(f);
}

template <int N>
__device__ __forceinline__ void sl_wait_vmcnt() {
    // s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt = imm[3:0] | imm[15:14] << 4, expcnt imm[6:4], lgkmcnt imm[11:8])
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

template <int N>
__device__ __forceinline__ void sl_wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N));
    __builtin_amdgcn_sched_barrier(0);
}
