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

// ---- tanh -------------------------------------------------------------------------------
// Branch-free fp32 tanh, ~17 VALU ops (2 transcendental) instead of ocml tanhf's ~100 with
// divergent branches -- the encoder evaluates 384 tanh per point, so this is hot.
//   |x| <  0.35 : odd Taylor polynomial through x^13 (truncation < 1e-9 relative)
//   |x| >= 0.35 : 1 - 2/(exp(2|x|)+1) on v_exp_f32 / v_rcp_f32 (<= ~4 ulp; -> 1 for large |x|)
// Max observed error vs fp64 tanh: see tests/test_gpu_kernels.py::test_fast_tanh_accuracy.
__device__ __forceinline__ float pm_tanh(float x) {
    const float ax = fabsf(x);
    const float e = __builtin_amdgcn_exp2f(ax * 2.8853900817779268f);      // exp(2|x|)
    const float big = fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
    const float x2 = ax * ax;
    float p = 0.0035921280365724810f;                                        // 21844/6081075
    p = fmaf(p, x2, -0.0088632355299021966f);                                // -1382/155925
    p = fmaf(p, x2, 0.021869488536155203f);                                  // 62/2835
    p = fmaf(p, x2, -0.053968253968253971f);                                 // -17/315
    p = fmaf(p, x2, 0.13333333333333333f);                                   // 2/15
    p = fmaf(p, x2, -0.33333333333333331f);                                  // -1/3
    const float small = fmaf(ax * x2, p, ax);
    return copysignf(ax < 0.35f ? small : big, x);
}
