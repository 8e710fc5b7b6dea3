// Row-wise forms of a skinny Linear layer (N <= 16 outputs: the policy / value heads), shared by the stand-alone kernels of
// gemm_f32.hip and the fused head kernels of losses.hip -- ONE definition of the summation order, so that the fused and the
// separate launch chains give bit-identical results.
#pragma once
#include "common.h"

#define SK_MAXN 16
#define SK_T 256

// W (N x K, row stride ldw) -> LDS image [N][K]; K % 4 == 0, 16-byte aligned rows
template <int NT = SK_T>
__device__ __forceinline__ void sk_fill_w(float* __restrict__ sW, const float* __restrict__ W, long ldw, int N, int K) {
    for (int e = threadIdx.x * 4; e < N * K; e += NT * 4) {
        const int n = e / K, k = e - n * K;
        *(float4*)(sW + e) = *(const float4*)(W + (long)n * ldw + k);
    }
}
// this lane's partial sums of one row's N dot products (lanes over k, a float4 per lane per 256-wide chunk);
// the caller finishes each with wave_sum()
__device__ __forceinline__ void sk_row_dot(const float* __restrict__ xrow, const float* __restrict__ sW, int N, int K, int lane,
                                           float (&acc)[SK_MAXN]) {
#pragma unroll
    for (int n = 0; n < SK_MAXN; ++n) acc[n] = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const float4 x = *(const float4*)(xrow + k);
#pragma unroll
        for (int n = 0; n < SK_MAXN; ++n)
            if (n < N) {
                const float4 w = *(const float4*)(sW + n * K + k);
                acc[n] = fmaf(x.w, w.w, fmaf(x.z, w.z, fmaf(x.y, w.y, fmaf(x.x, w.x, acc[n]))));
            }
    }
}
// four columns k..k+3 of one row of dX = (dY W) .* act'(H); g(n) = dY[row][n]
template <typename G>
__device__ __forceinline__ float4 sk_dgrad4(G g, const float* __restrict__ sW, int N, int K, int k, const float* __restrict__ hrow, int act) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int n = 0; n < SK_MAXN; ++n)
        if (n < N) {
            const float gn = g(n);
            const float4 w = *(const float4*)(sW + n * K + k);
            s.x = fmaf(gn, w.x, s.x); s.y = fmaf(gn, w.y, s.y); s.z = fmaf(gn, w.z, s.z); s.w = fmaf(gn, w.w, s.w);
        }
    if (act != PM_ACT_NONE) {
        const float4 h = *(const float4*)(hrow + k);
        s.x *= pm_dact(h.x, act); s.y *= pm_dact(h.y, act); s.z *= pm_dact(h.z, act); s.w *= pm_dact(h.w, act);
    }
    return s;
}
static inline bool sk_aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
// can (N, K) with these operands take the row-wise kernels?  a, b: the two 16-byte-loaded operands and their row strides
static inline bool skinny_ok(int N, int K, const void* a, long lda, const void* b, long ldb) {
    return N <= SK_MAXN && K % 4 == 0 && (long)N * K * 4 <= 64 * 1024 && lda % 4 == 0 && ldb % 4 == 0 && sk_aligned16(a) && sk_aligned16(b);
}
