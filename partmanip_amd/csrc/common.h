// Shared device/host helpers for libpartmanip_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/partmanip_hip.h"

#define PM_WAVE 64

#define PM_CHECK_LAUNCH()                                   \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return -(1000 + (int)e__);   \
    } while (0)

#define PM_REQUIRE(cond) \
    do {                 \
        if (!(cond)) return PM_EINVAL; \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline hipStream_t pm_stream(void* s) { return (hipStream_t)s; }

// ---- wave / block reductions (64-lane waves) -------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T w = __shfl_xor(v, o, 64);
        v = w > v ? w : v;
    }
    return v;
}

// Sum across a work-group of NT threads (NT multiple of 64, <= 1024). `red` needs NT/64 slots.
// Every thread gets the total; deterministic order.
template <typename T, int NT>
__device__ __forceinline__ T block_sum(T v, T* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    T t = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}

// exact-rounding helpers: keep hipcc from contracting a*b+c into an FMA where the
// reference's op-by-op rounding must be reproduced bit for bit.
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }
