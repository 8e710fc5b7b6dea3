// fp32 MFMA building blocks shared by the fused encoder kernels (pointnet_enc.hip, sa_fused.hip).
// v_mfma_f32_32x32x2_f32 with the k-split convention: lanes 0-31 own k in [0,K/2), lanes 32-63 own
// k in [K/2,K); a lane's 16 accumulator registers are rows (r&3) + 8*(r>>2) + 4*(lane>>5), column lane&31.
#pragma once
#include "common.h"

#ifndef PN_ABLATE
#define PN_ABLATE 0
#endif

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// ---- MFMA operand streaming without register copies ------------------------------------------
// acc[mb][nb] += A(MB*32 x K, LDS, row stride lda) * B(K x NB*32, packed weights in L2).  Two named
// operand sets (ping / pong): the loads of k-group g+1 are issued before the 4*MB*NB MFMAs of group g
// and are first waited for a full group (>= 500 cycles) later.  (A "next -> current" register copy at
// the loop top makes hipcc wait for the just-issued loads in the same iteration, exposing the whole
// L2 round trip every group: measured -27 % on the forward's layer-3 loop.)
template <int MB, int NB>
struct OperandSet {
    float4 a[MB], b[NB];
};
template <int MB, int NB>
__device__ __forceinline__ void load_set(OperandSet<MB, NB>& o, const float* __restrict__ A, int lda,
                                         const float4* __restrict__ Bp, int bstride, int g) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) o.b[nb] = Bp[(size_t)(nb * bstride + g) * 64];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) o.a[mb] = *(const float4*)(A + mb * 32 * lda + g * 4);
}
template <int MB, int NB>
__device__ __forceinline__ void mfma_set(const OperandSet<MB, NB>& o, f32x16 (&acc)[MB][NB]) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[mb][nb] = MFMA(o.a[mb][e], o.b[nb][e], acc[mb][nb]);
}
// A points at this lane's first row/k (row li, k = lh*K/2); NG = 4-step k-groups per lane half (K/8);
// packed B layout [nb][NG][lane][4] (Bp already offset to this wave's first N-block and this lane).
template <int MB, int NB, int NG>
__device__ __forceinline__ void mfma_stream(const float* __restrict__ A, int lda, const float4* __restrict__ Bp,
                                            f32x16 (&acc)[MB][NB]) {
    OperandSet<MB, NB> ping, pong;
    load_set<MB, NB>(ping, A, lda, Bp, NG, 0);
#pragma unroll 1
    for (int g = 0; g < NG; g += 2) {
        // sched_barrier(0) pins the issue order "loads of the NEXT group, then this group's MFMAs":
        // left alone, hipcc's scheduler sinks each load_set down to its first use (register pressure
        // heuristic) and the loop degenerates to load -> s_waitcnt -> MFMA with the L2 latency exposed.
#if (PN_ABLATE & 4)
        pong = ping;                              // profiling only: no operand traffic inside the loop
#else
        load_set<MB, NB>(pong, A, lda, Bp, NG, g + 1);
#endif
        __builtin_amdgcn_sched_barrier(0);
        mfma_set<MB, NB>(ping, acc);
        __builtin_amdgcn_sched_barrier(0);
        // UNCONDITIONAL: on the last trip this reads one k-group past the slice (the LDS row padding /
        // the next packed block or the packed buffer's 4 KB tail pad) and discards it.  A conditional
        // load gives the two loop paths different outstanding-load counts and hipcc then waits
        // vmcnt(1)/(0) for the just-issued loads as well.
#if !(PN_ABLATE & 4)
        load_set<MB, NB>(ping, A, lda, Bp, NG, g + 2);
#endif
        __builtin_amdgcn_sched_barrier(0);
        mfma_set<MB, NB>(pong, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int MB, int NB>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[MB][NB]) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
}


// Same pipeline with the A operand produced by a functor `aload(mb, g) -> float4` (4 consecutive k of this lane's
// row in M-block mb) instead of an LDS tile -- used where A is generated on the fly.
template <int MB, int NB, int NG, class AF>
__device__ __forceinline__ void mfma_stream_fn(AF aload, const float4* __restrict__ Bp, f32x16 (&acc)[MB][NB]) {
    OperandSet<MB, NB> ping, pong;
#define PM_LOAD_FN(o, g_)                                                                  \
    _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) o.b[nb] = Bp[(size_t)(nb * NG + (g_)) * 64]; \
    _Pragma("unroll") for (int mb = 0; mb < MB; ++mb) o.a[mb] = aload(mb, (g_));
    PM_LOAD_FN(ping, 0)
#pragma unroll 1
    for (int g = 0; g < NG; g += 2) {
        PM_LOAD_FN(pong, g + 1)
        __builtin_amdgcn_sched_barrier(0);
        mfma_set<MB, NB>(ping, acc);
        __builtin_amdgcn_sched_barrier(0);
        PM_LOAD_FN(ping, g + 2)                    // unconditional: one group past the end, discarded
        __builtin_amdgcn_sched_barrier(0);
        mfma_set<MB, NB>(pong, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef PM_LOAD_FN
}
