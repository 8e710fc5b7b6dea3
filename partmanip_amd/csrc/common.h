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

// Compute units of the current device, asked of the runtime on every call (an attribute lookup, no synchronisation): the
// library keeps NO state between calls -- no function-static caches, no environment reads (include/partmanip_hip.h preamble).
static inline int pm_cu_count() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
        n = 256;
    return n;
}

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
// fp32 tanh as the odd rational x * P(x^2) / Q(x^2) (degree 13 / 6 minimax, the coefficients of Eigen's float
// tanh), |x| clamped to 7.905 where it rounds to 1: branch-free, one transcendental (v_rcp_f32 + one Newton
// step), <= ~5 ulp against fp64 tanh everywhere incl. denormal inputs (tests/test_gpu_kernels.py::
// test_fast_tanh_accuracy) -- ocml tanhf is ~100 instructions with divergent branches, and the encoder evaluates
// 384 tanh per point.  On gfx950 fp32 MFMA and VALU instructions of different waves do NOT overlap on a SIMD
// (tools/ubench/mfma_valu_overlap.hip: two MFMA waves + two VALU waves take the SUM of their times), so every VALU
// slot here is paid for in MFMA time.  pm_tanh2 evaluates two values on the packed-fp32 pipe (v_pk_fma_f32 /
// v_pk_mul_f32: 14 packed + 2 clamp + 2 rcp instructions per pair = 12 issue slots per value with v_rcp at quarter
// rate; the previous exp2/rcp + Taylor-seam version cost ~22).  Both forms round identically (same fma chain).
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define PM_TANH_CLAMP 7.90531110763549805f
#define PM_TANH_A13 -2.76076847742355e-16f
#define PM_TANH_A11 2.00018790482477e-13f
#define PM_TANH_A9 -8.60467152213735e-11f
#define PM_TANH_A7 5.12229709037114e-08f
#define PM_TANH_A5 1.48572235717979e-05f
#define PM_TANH_A3 6.37261928875436e-04f
#define PM_TANH_A1 4.89352455891786e-03f
#define PM_TANH_B6 1.19825839466702e-06f
#define PM_TANH_B4 1.18534705686654e-04f
#define PM_TANH_B2 2.26843463243900e-03f
#define PM_TANH_B0 4.89352518554385e-03f
// v_med3_f32 drops a NaN operand (a NaN pre-activation would come out as -1): both forms select the input back when it
// is NaN so that a NaN observation / a diverged network stays visible in the features and losses, as with torch.tanh
// (two more VALU instructions per value; the encoder forward's pooling then turns a channel whose column holds a NaN
// into NaN, as torch.max does -- pn_fwd_kernel's epilogue).
__device__ __forceinline__ float pm_tanh(float x0) {
    float x = __builtin_amdgcn_fmed3f(x0, -PM_TANH_CLAMP, PM_TANH_CLAMP);
    const float x2 = x * x;
    float p = fmaf(PM_TANH_A13, x2, PM_TANH_A11);
    p = fmaf(p, x2, PM_TANH_A9);
    p = fmaf(p, x2, PM_TANH_A7);
    p = fmaf(p, x2, PM_TANH_A5);
    p = fmaf(p, x2, PM_TANH_A3);
    p = fmaf(p, x2, PM_TANH_A1);
    float q = fmaf(PM_TANH_B6, x2, PM_TANH_B4);
    q = fmaf(q, x2, PM_TANH_B2);
    q = fmaf(q, x2, PM_TANH_B0);
    float r = __builtin_amdgcn_rcpf(q);
    r = fmaf(fmaf(-q, r, 1.0f), r, r);                     // one Newton step: 1/q to ~0.5 ulp
    const float t = x * (p * r);                            // P/Q ~ 1 first, then times x: no denormal intermediates
    return x0 != x0 ? x0 : t;
}
__device__ __forceinline__ f32x2 pm_tanh2(f32x2 x0) {
#define PM_S2(v) ((f32x2){(v), (v)})
    f32x2 x;
    x.x = __builtin_amdgcn_fmed3f(x0.x, -PM_TANH_CLAMP, PM_TANH_CLAMP);
    x.y = __builtin_amdgcn_fmed3f(x0.y, -PM_TANH_CLAMP, PM_TANH_CLAMP);
    const f32x2 x2 = x * x;
    f32x2 p = __builtin_elementwise_fma(PM_S2(PM_TANH_A13), x2, PM_S2(PM_TANH_A11));
    p = __builtin_elementwise_fma(p, x2, PM_S2(PM_TANH_A9));
    p = __builtin_elementwise_fma(p, x2, PM_S2(PM_TANH_A7));
    p = __builtin_elementwise_fma(p, x2, PM_S2(PM_TANH_A5));
    p = __builtin_elementwise_fma(p, x2, PM_S2(PM_TANH_A3));
    p = __builtin_elementwise_fma(p, x2, PM_S2(PM_TANH_A1));
    f32x2 q = __builtin_elementwise_fma(PM_S2(PM_TANH_B6), x2, PM_S2(PM_TANH_B4));
    q = __builtin_elementwise_fma(q, x2, PM_S2(PM_TANH_B2));
    q = __builtin_elementwise_fma(q, x2, PM_S2(PM_TANH_B0));
    f32x2 r;
    r.x = __builtin_amdgcn_rcpf(q.x);
    r.y = __builtin_amdgcn_rcpf(q.y);
    r = __builtin_elementwise_fma(__builtin_elementwise_fma(-q, r, PM_S2(1.0f)), r, r);
    f32x2 t = x * (p * r);
#ifndef PM_TANH2_DROPS_NAN                                   // A/B builds only: what the two selects cost
    t.x = x0.x != x0.x ? x0.x : t.x;                        // NaN in -> NaN out, as torch.tanh (v_cmp_u_f32 + v_cndmask_b32 per value)
    t.y = x0.y != x0.y ? x0.y : t.y;
#endif
    return t;
#undef PM_S2
}
__device__ __forceinline__ f32x2 pm_tanh2(float a, float b) { return pm_tanh2((f32x2){a, b}); }

// ---- the activation set of network.py:7-24 (Linear epilogues) ------------------------------------------------------------
#define PM_SELU_L 1.0507009873554804934193349852946f
#define PM_SELU_A 1.6732632423543772848170429916717f
__device__ __forceinline__ float pm_act(float z, int act) {
    switch (act) {
        case PM_ACT_TANH: return pm_tanh(z);
        case PM_ACT_RELU: return fmaxf(z, 0.0f);
        case PM_ACT_LRELU: return z > 0.0f ? z : 0.01f * z;
        case PM_ACT_ELU: return z > 0.0f ? z : expm1f(z);
        case PM_ACT_SELU: return PM_SELU_L * (z > 0.0f ? z : PM_SELU_A * expm1f(z));
        case PM_ACT_SIGMOID: return 1.0f / (1.0f + expf(-z));
        default: return z;
    }
}
// derivative of the activation expressed through its OUTPUT h
__device__ __forceinline__ float pm_dact(float h, int act) {
    switch (act) {
        case PM_ACT_TANH: return 1.0f - h * h;
        case PM_ACT_RELU: return h > 0.0f ? 1.0f : 0.0f;
        case PM_ACT_LRELU: return h > 0.0f ? 1.0f : 0.01f;
        case PM_ACT_ELU: return h > 0.0f ? 1.0f : h + 1.0f;
        case PM_ACT_SELU: return h > 0.0f ? PM_SELU_L : h + PM_SELU_L * PM_SELU_A;
        case PM_ACT_SIGMOID: return h * (1.0f - h);
        default: return 1.0f;
    }
}

// Work-groups b = x (mod 8) of a launch land on XCD x when the dispatcher round-robins (MI355X_MICROARCH.md, work-group dispatch):
// this maps block b of n to a position such that every XCD owns a CONTIGUOUS range of positions (a bijection for any n), so that
// neighbouring tiles / rows meet in one XCD's L2 instead of being fetched by all eight.  Speed and traffic only.
__device__ __forceinline__ int pm_xcd_contiguous(int b, int n) {
    const int q = n >> 3, r = n & 7, x = b & 7;
    return x * q + min(x, r) + (b >> 3);
}
