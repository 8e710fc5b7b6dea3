// K15 fused: one PointNet++ set-abstraction level (ball-query groups of 32 -> shared 3-layer MLP,
// tanh after every layer -> max over the group) as ONE forward and ONE backward kernel.  Absent from the
// reference snapshot (README.md:23,30); mandated by BASELINE.json's north_star; the structure follows
// partmanip_amd.algo_utils.network.PointNet2 (rows = [xyz[idx]-centre | feat[idx] | 0-pad]).
//
// What is never materialised: the grouped rows and the three per-row activations
// ((B*S*32) x (C0+C1+C2+C3) floats: 17 GB + 11 GB per net at B = 2048 for the default two levels).
//
// Layer 1 is split by linearity:  z1[row] = W1[:, :3] (xyz[idx]-centre) + b1  +  Y[b, idx[row], :]
// with Y = feat * W1[:, 3:3+Cf]^T computed ONCE PER SOURCE POINT by the Linear kernel (K4) instead of
// once per grouped row (every point sits in ~S*32/P = 8 groups): 8x fewer MACs for that layer and for its
// two backward GEMMs, and the kernels here become the same for every level (VALU layer 1 + gather).
//
// Forward, per tile of TM rows (TM/32 groups), NW waves:
//   gather idx -> rel. xyz (LDS) -> H1 = tanh(z1) (VALU, Y gathered coalesced) -> LDS
//   H2 = tanh(H1 W2^T + b2)   fp32 MFMA, A from LDS (ds_read_b128), B streamed from L2 in operand order
//   Z3 = H2 W3^T              fp32 MFMA; a group's 32 rows are exactly one MFMA M-block, so the max-pool
//                             is an in-lane reduction over the 16 accumulator registers + one lane^32
//                             exchange;  pooled = tanh(max + b3)  (tanh is monotone: max commutes with it;
//                             arg = the lowest row attaining the max of the PRE-activation)
// Backward (structured max-pool gradient, see sa_bwd_kernel below).
#include "common.h"
#include "mfma_f32.h"

#define SA_NS 32
#ifndef SA_ABLATE
#define SA_ABLATE 0   // profiling only (wrong results): 1 dH2 atomics, 2 dW3, 4 dW2 MFMA, 8 dH1 MFMA, 16 P7, 32 layer-2 recompute, 64 layer 1, 128 dZ2
#endif

struct SaArgs {
    const float* xyz;        // (B*P, 3)
    const float* centers;    // (G, 3)      G = B*S groups
    const int32_t* idx;      // (G, 32)     neighbour index inside the cloud
    const float* Y;          // (B*P, C1)   feature part of layer 1 (pre-bias), or null
    const float* W1;         // (C1, ldw1)  columns 0..2 = xyz weights
    long ldw1;
    const float* b1;
    const float* b2;
    const float* b3;
    const float* packed;     // pm_sa_pack_weights_f32
    float* pooled;           // (G, ldp)
    long ldp;
    int32_t* arg;            // (G, C3)
    long G;
    int S, P;
    // backward only
    const float* W3;         // (C3, C2) plain
    const float* dpooled;    // (G, lddp)
    long lddp;
    float* dY;               // (B*P, C1) zero-filled by the caller, or null (fp32 atomics: run-dependent last bits; A/B only)
    float* dz1;              // packed kernels: (R, C1) layer-1 pre-activation gradient PER PACKED ROW, plain stores -- summed per source
                             // point in a fixed order by pm_sa_dy_segsum_f32 / pm_sa_dy_consume_f32 (the deterministic path)
    float* parts;            // per-work-group partial sums
    // layer-2 activations (G*32, C2): written by the training forward, read back by the backward instead of a
    // recompute (C2*4 B per row against 2*C1*C2 FLOP: 32-64 FLOP/B, machine balance ~20); null = recompute
    float* h2;
    int tail_cols;           // packed forward with `centers`: columns [C3, C3 + tail_cols) of a pooled row = (centre x, y, z, 0 ...)
};

extern "C" size_t pm_sa_packed_elems(int C1, int C2, int C3) {
    return (size_t)C1 * C2 * 2 + (size_t)C2 * C3 * 2 + 1024;   // W2 fwd | W3 fwd | W2 bwd | W3 bwd | 4 KB tail pad
}

__global__ __launch_bounds__(256) void sa_pack_kernel(const float* __restrict__ W2, const float* __restrict__ W3,
                                                       int C1, int C2, int C3, float* __restrict__ packed) {
    const long n2 = (long)C1 * C2, n3 = (long)C2 * C3, total = 2 * n2 + 2 * n3 + 1024;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int e = i & 3, lane = (i >> 2) & 63, li = lane & 31, lh = lane >> 5;
    if (i < n2) {                                          // B[k=in][n=out] = W2[out][in], K = C1
        const int ng = C1 / 8, g = (int)((i >> 8) % ng), nb = (int)((i >> 8) / ng);
        packed[i] = W2[(nb * 32 + li) * C1 + lh * (C1 / 2) + g * 4 + e];
    } else if (i < n2 + n3) {                              // W3, K = C2
        const long j = i - n2;
        const int ng = C2 / 8, g = (int)((j >> 8) % ng), nb = (int)((j >> 8) / ng);
        packed[i] = W3[(nb * 32 + li) * C2 + lh * (C2 / 2) + g * 4 + e];
    } else if (i < 2 * n2 + n3) {                          // dH1 = dZ2 * W2:  B[k=out][n=in] = W2[out][in], K = C2
        const long j = i - n2 - n3;
        const int ng = C2 / 8, g = (int)((j >> 8) % ng), nb = (int)((j >> 8) / ng);
        packed[i] = W2[(lh * (C2 / 2) + g * 4 + e) * C1 + nb * 32 + li];
    } else if (i < 2 * n2 + 2 * n3) {                      // dH2 = dZ3 * W3:  B[k=out][n=in] = W3[out][in], K = C3
        const long j = i - 2 * n2 - n3;
        const int ng = C3 / 8, g = (int)((j >> 8) % ng), nb = (int)((j >> 8) / ng);
        packed[i] = W3[(lh * (C3 / 2) + g * 4 + e) * C2 + nb * 32 + li];
    } else {
        packed[i] = 0.f;
    }
}

extern "C" int pm_sa_pack_weights_f32(const float* W2, const float* W3, int C1, int C2, int C3, float* packed,
                                      void* stream) {
    PM_REQUIRE(W2 && W3 && packed && C1 > 0 && C2 > 0 && C3 > 0 && C1 % 32 == 0 && C2 % 32 == 0 && C3 % 32 == 0);
    const long total = (long)pm_sa_packed_elems(C1, C2, C3);
    hipLaunchKernelGGL(sa_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, pm_stream(stream), W2, W3,
                       C1, C2, C3, packed);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ---- shared pieces -------------------------------------------------------------------------
// wave -> (M part, N part) for a TM x N output over NW waves: as many waves along N as there are 32-column
// blocks (each streamed weight block is then reused by MB row blocks), the rest along M.
template <int TM, int N, int NW>
struct WaveMap {
    static constexpr int NBLK = N / 32, MBLK = TM / 32;
    static constexpr int NBW = NBLK < NW ? NBLK : NW;
    static constexpr int MW = NW / NBW;
    static constexpr int NB = NBLK / NBW, MB = MBLK / MW;
    static_assert(NBW * MW == NW && NB * NBW == NBLK && MB * MW == MBLK && MB >= 1, "tile/wave mapping");
};

// gather one tile: relative xyz -> Xz[TM][4], flat source point -> Src[TM]
template <int TM, int NT>
__device__ __forceinline__ void sa_stage(const SaArgs& a, long tile, int tid, float* __restrict__ Xz,
                                         int* __restrict__ Src) {
    for (int t = tid; t < TM; t += NT) {
        long g = tile * (TM / SA_NS) + (t >> 5);
        if (g >= a.G) g = a.G - 1;                          // ragged last tile: recompute the last group, never stored
        const long row = g * SA_NS + (t & 31);
        const long sp = (long)(g / a.S) * a.P + a.idx[row];
        float4 v;
        v.x = sub_rn(a.xyz[sp * 3], a.centers[g * 3]);
        v.y = sub_rn(a.xyz[sp * 3 + 1], a.centers[g * 3 + 1]);
        v.z = sub_rn(a.xyz[sp * 3 + 2], a.centers[g * 3 + 2]);
        v.w = 0.f;
        *(float4*)(Xz + t * 4) = v;
        Src[t] = (int)sp;
    }
}

// layer 1: thread (c = tid % C1, part = tid / C1) computes its rows of H1 = tanh(W1z . xyz + b1 + Y[src])
template <int C1, int TM, int NT>
__device__ __forceinline__ void sa_layer1(const SaArgs& a, int tid, const float* __restrict__ Xz,
                                          const int* __restrict__ Src, float* __restrict__ H1) {
    static_assert(NT % C1 == 0 && TM % (NT / C1) == 0, "layer-1 thread mapping");
    constexpr int PARTS = NT / C1, RPT = TM / PARTS, LD1 = C1 + 4;
    const int c = tid % C1, p0 = (tid / C1) * RPT;
    const float w0 = a.W1[c * a.ldw1], w1 = a.W1[c * a.ldw1 + 1], w2 = a.W1[c * a.ldw1 + 2], bb = a.b1[c];
    if (a.Y) {
        static_assert(RPT % 2 == 0, "layer 1 pairs rows for the packed tanh");
        // The Y rows of a tile are a gather (one 4*C1-byte row per source point, wherever it lies): ALL of a batch's loads are
        // issued before the first is used -- with the loads inside the row loop (four in flight per thread) this phase was a chain
        // of HBM / L2 round trips: 0.28 ms of the 1.67 ms SA2 backward (profiles/round4_f_sa_packed_phase_ablation.txt).
        constexpr int CH = RPT < 16 ? RPT : 16;
        static_assert(RPT % CH == 0 && CH % 2 == 0, "layer-1 load batches");
#pragma unroll 1
        for (int p = p0; p < p0 + RPT; p += CH) {
            float y[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) y[j] = a.Y[(long)Src[p + j] * C1 + c];
#pragma unroll
            for (int j = 0; j < CH; j += 2) {
                float z[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float4 x = *(const float4*)(Xz + (p + j + i) * 4);
                    float s = fmaf(w0, x.x, bb);
                    s = fmaf(w1, x.y, s);
                    s = fmaf(w2, x.z, s);
                    z[i] = s + y[j + i];
                }
                const f32x2 t = pm_tanh2(z[0], z[1]);
                H1[(p + j) * LD1 + c] = t.x;
                H1[(p + j + 1) * LD1 + c] = t.y;
            }
        }
    } else {
#pragma unroll 4
        for (int p = p0; p < p0 + RPT; p += 2) {
            float z[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4 x = *(const float4*)(Xz + (p + j) * 4);
                float s = fmaf(w0, x.x, bb);
                s = fmaf(w1, x.y, s);
                z[j] = fmaf(w2, x.z, s);
            }
            const f32x2 t = pm_tanh2(z[0], z[1]);
            H1[p * LD1 + c] = t.x;
            H1[(p + 1) * LD1 + c] = t.y;
        }
    }
}

// H2 = tanh(H1 * W2^T + b2): this wave's MB x NB blocks.  ALIAS: H2 overwrites H1 (same buffer), so every
// wave must have finished its MFMA reads of H1 before the first store.
template <int C1, int C2, int TM, int NW, bool ALIAS>
__device__ __forceinline__ void sa_layer2(const float* __restrict__ H1, const float4* __restrict__ P2v,
                                          const float* __restrict__ b2, int wave, int lane, float* __restrict__ H2) {
    using M = WaveMap<TM, C2, NW>;
    constexpr int LD1 = C1 + 4, LD2 = C2 + 4, NG = C1 / 8;
    const int li = lane & 31, lh = lane >> 5, wn = wave % M::NBW, wm = wave / M::NBW;
    f32x16 acc[M::MB][M::NB];
    zero_acc<M::MB, M::NB>(acc);
    if (!(SA_ABLATE & 32))
        mfma_stream<M::MB, M::NB, NG>(H1 + (wm * M::MB * 32 + li) * LD1 + lh * (C1 / 2), LD1,
                                      P2v + (size_t)(wn * M::NB) * NG * 64 + lane, acc);
    if (ALIAS) __syncthreads();
#pragma unroll
    for (int nb = 0; nb < M::NB; ++nb) {
        const int col = (wn * M::NB + nb) * 32 + li;
        const float bv = b2[col];
#pragma unroll
        for (int mb = 0; mb < M::MB; ++mb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 t = pm_tanh2(acc[mb][nb][r] + bv, acc[mb][nb][r + 1] + bv);
                const int row = (wm * M::MB + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;     // r even: row(r+1) = row(r) + 1
                H2[row * LD2 + col] = t.x;
                H2[(row + 1) * LD2 + col] = t.y;
            }
    }
}

// ==================================================================================== forward
// WPE = waves per SIMD the register budget is sized for (HIP's second __launch_bounds__ argument).  H1 and H2
// share one LDS buffer (H2 is written after a barrier that ends the layer-2 MFMA reads), which brings a
// work-group to ~36 KB: four 4-wave groups per CU, so the gather / tanh / barrier phases of one group hide
// under the MFMA phases of the others.
template <int C1, int C2, int C3, int TM, int NW, int WPE>
__global__ __launch_bounds__(NW * 64, WPE) void sa_fwd_kernel(SaArgs a) {
    constexpr int NT = NW * 64, LD1 = C1 + 4, LD2 = C2 + 4, LDM = LD1 > LD2 ? LD1 : LD2;
    __shared__ __attribute__((aligned(16))) float smem[TM * (LDM + 4) + TM];
    float* H1 = smem;
    float* H2 = smem;
    float* Xz = smem + TM * LDM;
    int* Src = (int*)(Xz + TM * 4);
    using M3 = WaveMap<TM, C3, NW>;
    constexpr int NG3 = C2 / 8;

    const int tid0 = threadIdx.x;
    const float4* P2v = (const float4*)a.packed;
    const float4* P3v = (const float4*)(a.packed + (size_t)C1 * C2);
    const long ntiles = (a.G + TM / SA_NS - 1) / (TM / SA_NS);

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // Launder the thread id once per tile: every thread-derived address below is then recomputed inside the
        // tile (a few VALU ops) instead of being hoisted to kernel entry as dozens of loop-invariant VGPRs
        // that do not fit next to the accumulators and get spilled.
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int li = lane & 31, lh = lane >> 5;
        sa_stage<TM, NT>(a, tile, tid, Xz, Src);
        __syncthreads();
        sa_layer1<C1, TM, NT>(a, tid, Xz, Src, H1);
        __syncthreads();
        sa_layer2<C1, C2, TM, NW, true>(H1, P2v, a.b2, wave, lane, H2);
        __syncthreads();
        // ---- layer 3 + max-pool over each 32-row block --------------------------------------
        const int wn = wave % M3::NBW, wm = wave / M3::NBW;
        f32x16 acc[M3::MB][M3::NB];
        zero_acc<M3::MB, M3::NB>(acc);
        mfma_stream<M3::MB, M3::NB, NG3>(H2 + (wm * M3::MB * 32 + li) * LD2 + lh * (C2 / 2), LD2,
                                         P3v + (size_t)(wn * M3::NB) * NG3 * 64 + lane, acc);
#pragma unroll
        for (int nb = 0; nb < M3::NB; ++nb) {
            const int ch = (wn * M3::NB + nb) * 32 + li;
            const float bv = a.b3[ch];
#pragma unroll
            for (int mb = 0; mb < M3::MB; ++mb) {
                float best = acc[mb][nb][0];
                int bi = 4 * lh;
#pragma unroll
                for (int r = 1; r < 16; ++r) {
                    const float v = acc[mb][nb][r];
                    if (v > best) {
                        best = v;
                        bi = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    }
                }
                const float ov = __shfl_xor(best, 32, 64);
                const int oi = __shfl_xor(bi, 32, 64);
                if (ov > best || (ov == best && oi < bi)) {
                    best = ov;
                    bi = oi;
                }
                const long g = tile * (TM / SA_NS) + wm * M3::MB + mb;
                if (lh == 0 && g < a.G) {
                    a.pooled[g * a.ldp + ch] = pm_tanh(best + bv);
                    a.arg[g * C3 + ch] = bi;
                }
            }
        }
        if (a.h2) {                                          // training forward: keep H2 for the backward
            const long row0 = tile * TM, nrows = a.G * SA_NS;
#pragma unroll 2
            for (int q = tid; q < TM * C2 / 4; q += NT) {
                const int row = q / (C2 / 4), c4 = q % (C2 / 4);
                if (row0 + row < nrows)
                    *(float4*)(a.h2 + (row0 + row) * C2 + 4 * c4) = *(const float4*)(H2 + row * LD2 + 4 * c4);
            }
        }
    }
}

#define SA_CFG_A(C1, C2, C3) ((C1) == 64 && (C2) == 64 && (C3) == 128)
#define SA_CFG_B(C1, C2, C3) ((C1) == 128 && (C2) == 128 && (C3) == 256)

extern "C" int pm_sa_supported(int C1, int C2, int C3, int nsample) {
    return nsample == SA_NS && (SA_CFG_A(C1, C2, C3) || SA_CFG_B(C1, C2, C3));
}

static int sa_cu_count() { return pm_cu_count(); }

extern "C" int pm_sa_fwd_f32(const float* xyz, const float* centers, const int32_t* idx, const float* Y, int B, int P,
                             int S, int nsample, const float* W1, long ldw1, const float* b1, const float* b2,
                             const float* b3, const float* packed, int C1, int C2, int C3, float* pooled, long ldp,
                             int32_t* arg, float* h2_save, void* stream) {
    PM_REQUIRE(xyz && centers && idx && W1 && b1 && b2 && b3 && packed && pooled && arg);
    PM_REQUIRE(B > 0 && P > 0 && S > 0 && ldw1 >= 3 && ldp >= C3);
    if (!pm_sa_supported(C1, C2, C3, nsample)) return PM_EUNSUPPORTED;
    if (((uintptr_t)packed & 15) != 0) return PM_EALIGN;
    SaArgs a = {};
    a.xyz = xyz; a.centers = centers; a.idx = idx; a.Y = Y; a.W1 = W1; a.ldw1 = ldw1; a.b1 = b1; a.b2 = b2; a.b3 = b3;
    a.packed = packed; a.pooled = pooled; a.ldp = ldp; a.arg = arg; a.G = (long)B * S; a.S = S; a.P = P;
    a.h2 = h2_save;
    const int ncu = sa_cu_count();
#define SA_FWD_LAUNCH(C1_, C2_, C3_, TM_, NW_, WGCU_)                                                          \
    {                                                                                                          \
        const long ntiles = (a.G + (TM_) / SA_NS - 1) / ((TM_) / SA_NS);                                       \
        const long grid = ntiles < (long)ncu * (WGCU_) ? ntiles : (long)ncu * (WGCU_);                         \
        hipLaunchKernelGGL((sa_fwd_kernel<C1_, C2_, C3_, TM_, NW_, WGCU_>), dim3((unsigned)grid), dim3((NW_) * 64), 0, \
                           pm_stream(stream), a);                                                              \
    }
    if (SA_CFG_A(C1, C2, C3)) SA_FWD_LAUNCH(64, 64, 128, 128, 4, 4)
    else SA_FWD_LAUNCH(128, 128, 256, 64, 4, 4)
#undef SA_FWD_LAUNCH
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// =================================================================================== backward
// The gradient w.r.t. the layer-3 pre-activation is structured: per (group, channel) ONE non-zero,
//   val[g,c] = dpooled[g,c] * (1 - pooled[g,c]^2)   at row arg[g,c],
// so layer 3 needs no dense backward:
//   dW3[c,:] += val * H2[row,:]      (VALU; thread (c, k-slice) keeps its slice of dW3 in registers for the
//                                     whole kernel)
//   dH2 = dZ3 * W3                   (MFMA; the sparse A operand is generated in registers from (val, arg) --
//                                     one compare-select per element, no dZ3 tile anywhere.  A scatter with
//                                     ds_add_f32 does 32x fewer FLOPs but measured 4x slower than this whole
//                                     kernel: LDS float atomics retire ~1 lane per 12 clocks)
// Layers 1-2 are recomputed per tile exactly as in the forward, then
//   dZ2 = dH2 .* (1-H2^2)            (in place; db2 on the way)
//   dW2 += dZ2^T * H1                (MFMA, both operands from LDS, accumulators persistent in registers)
//   dH1  = dZ2 * W2                  (MFMA, B streamed from L2)   dZ1 = dH1 .* (1-H1^2) (in place of H1)
//   dW1[:, :3], db1 += dZ1^T [xyz 1] (VALU)      dY[src[row], :] += dZ1[row, :]  (global fp32 atomics)
// Work-groups are persistent over tiles; per-work-group partial sums are reduced by sa_bwd_reduce_kernel.
// The global fp32 atomics of the dY scatter make its summation order, hence its last bits, run-dependent.
template <int C1, int C2, int C3>
struct SaPart {
    static constexpr int O_DW2 = 0, O_DB2 = C2 * C1, O_DW3 = O_DB2 + C2, O_DB3 = O_DW3 + C3 * C2, O_DW1 = O_DB3 + C3,
                         N = O_DW1 + C1 * 4;
};
#define SA_BWD_MAXGRID 1024
#ifndef SA_A_BWD_WPE
#define SA_A_BWD_WPE 4        // SA1-shaped level: 4-wave work-groups, this many per CU
#endif
#ifndef SA_B_BWD_NW
#define SA_B_BWD_NW 16        // SA2-shaped level: one work-group per CU of 16 waves x 128 rows (or 8 x 64)
#endif
#ifndef SA_B_BWD_TM
#define SA_B_BWD_TM (SA_B_BWD_NW * 8)
#endif

template <int C1, int C2, int C3, int TM, int NW, int WPE>
__global__ __launch_bounds__(NW * 64, WPE) void sa_bwd_kernel(SaArgs a) {
    constexpr int NT = NW * 64, LD1 = C1 + 4, LD2 = C2 + 4, NGRP = TM / SA_NS;
    __shared__ __attribute__((aligned(16))) float smem[TM * (LD1 + LD2 + 4) + TM + 2 * (NGRP * C3 + 4)];
    float* H1 = smem;                        // H1, later dZ1 in place
    float* H2 = H1 + TM * LD1;
    float* D = H2;                           // dZ2 overwrites H2 in place (after a barrier, element by element)
    float* Xz = H2 + TM * LD2;
    int* Src = (int*)(Xz + TM * 4);
    float* Val = (float*)(Src + TM);
    int* Arg = (int*)(Val + NGRP * C3 + 4);      // +4: the operand stream reads one k-group past the end
    using P = SaPart<C1, C2, C3>;
    using M2 = WaveMap<TM, C2, NW>;          // dH2 output mapping
    using MH = WaveMap<TM, C1, NW>;          // dH1 output mapping
    constexpr int NGT = C2 / 8;

    const int tid0 = threadIdx.x, lane0 = tid0 & 63, wave0 = tid0 >> 6;
    const float4* P2v = (const float4*)a.packed;
    const float4* P2Tv = (const float4*)(a.packed + (size_t)C1 * C2 + (size_t)C2 * C3);
    const float4* P3Tv = (const float4*)(a.packed + (size_t)C1 * C2 * 2 + (size_t)C2 * C3);
    const long ntiles = (a.G + NGRP - 1) / NGRP;

    // persistent accumulators
    constexpr int TPC = NT / C3, KS = C2 / TPC;             // dW3: thread (c3 = tid % C3, ks = tid / C3) owns KS k's
    static_assert(NT % C3 == 0 && C2 % TPC == 0 && KS % 4 == 0, "dW3 thread mapping");
    float accW3[KS];
#pragma unroll
    for (int j = 0; j < KS; ++j) accW3[j] = 0.f;
    float accb3 = 0.f, accW1[4] = {0.f, 0.f, 0.f, 0.f};
    float accb2[M2::NB];                                    // column sums of dZ2 over this lane's rows
#pragma unroll
    for (int nb = 0; nb < M2::NB; ++nb) accb2[nb] = 0.f;
    constexpr int WBLK = (C2 / 32) * (C1 / 32), NBK = WBLK / NW;   // dW2 blocks per wave (same c2 block, NBK c1 blocks)
    static_assert(NBK * NW == WBLK && (C1 / 32) % NBK == 0, "dW2 wave mapping");
    f32x16 accW2[NBK];
#pragma unroll
    for (int j = 0; j < NBK; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW2[j][r] = 0.f;

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));                    // see sa_fwd_kernel: recompute, don't hoist
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int li = lane & 31, lh = lane >> 5;
        const int w2_m = wave % (C2 / 32), w2_n0 = (wave / (C2 / 32)) * NBK;
        // ---- P0: gather ------------------------------------------------------------------
        sa_stage<TM, NT>(a, tile, tid, Xz, Src);
        for (int i = tid; i < NGRP * C3; i += NT) {
            const long g = tile * NGRP + i / C3;
            const int c = i % C3;
            float v = 0.f;
            int r = 0;
            if (g < a.G) {
                const float p = a.pooled[g * a.ldp + c];
                v = a.dpooled[g * a.lddp + c] * (1.0f - p * p);
                r = a.arg[g * C3 + c];
            }
            Val[i] = v;
            Arg[i] = r;
        }
        __syncthreads();
        // ---- P1/P2: recompute H1, H2 ------------------------------------------------------------
        if (!(SA_ABLATE & 64)) sa_layer1<C1, TM, NT>(a, tid, Xz, Src, H1);
        __syncthreads();
        if (a.h2) {                                          // saved by the forward: 1 coalesced pass instead of a GEMM
            const long row0 = tile * TM, nrows = a.G * SA_NS;
#pragma unroll 2
            for (int q = tid; q < TM * C2 / 4; q += NT) {
                const int row = q / (C2 / 4), c4 = q % (C2 / 4);
                long gr = row0 + row;
                if (gr >= nrows) gr = nrows - 1;             // ragged last tile (its Val entries are 0)
                *(float4*)(H2 + row * LD2 + 4 * c4) = *(const float4*)(a.h2 + gr * C2 + 4 * c4);
            }
        } else {
            sa_layer2<C1, C2, TM, NW, false>(H1, P2v, a.b2, wave, lane, H2);
        }
        __syncthreads();
        // ---- P3: structured layer-3 backward ---------------------------------------------------
        {
            const int c = tid % C3, ks = tid / C3;
#pragma unroll
            for (int gi = 0; gi < ((SA_ABLATE & 2) ? 0 : NGRP); ++gi) {
                const float v = Val[gi * C3 + c];
                const float* hrow = H2 + (gi * SA_NS + Arg[gi * C3 + c]) * LD2 + ks * KS;
#pragma unroll
                for (int k4 = 0; k4 < KS / 4; ++k4) {
                    const float4 h = *(const float4*)(hrow + 4 * k4);
                    accW3[4 * k4] = fmaf(v, h.x, accW3[4 * k4]);
                    accW3[4 * k4 + 1] = fmaf(v, h.y, accW3[4 * k4 + 1]);
                    accW3[4 * k4 + 2] = fmaf(v, h.z, accW3[4 * k4 + 2]);
                    accW3[4 * k4 + 3] = fmaf(v, h.w, accW3[4 * k4 + 3]);
                    // keep at most 4 row reads in flight: hipcc otherwise hoists all KS/4 ds_read_b128 of both
                    // groups (up to 128 VGPRs) above the FMAs and spills the persistent accumulators
                    if ((k4 & 3) == 3) {
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (ks == 0) accb3 += v;
            }
        }
        // A real barrier, not just a scheduling fence: without it hipcc overlaps the dW3 row reads with the
        // operand stream below and the live ranges of both add up (256 VGPRs + 94-212 spilled; 128-170 with it).
        __syncthreads();
        // ---- P3b/P4: dH2 = dZ3 * W3 (sparse A built in registers), dZ2 = dH2 .* (1 - H2^2) -> D, db2 ----
        {
            const int wn = wave % M2::NBW, wm = wave / M2::NBW;
            f32x16 acc[M2::MB][M2::NB];
            zero_acc<M2::MB, M2::NB>(acc);
            auto asel = [&](int mb, int g) -> float4 {
                const int base = (wm * M2::MB + mb) * C3 + lh * (C3 / 2) + g * 4;
                const int4 ar = *(const int4*)(Arg + base);
                const float4 vv = *(const float4*)(Val + base);
                float4 o;
                o.x = ar.x == li ? vv.x : 0.f;
                o.y = ar.y == li ? vv.y : 0.f;
                o.z = ar.z == li ? vv.z : 0.f;
                o.w = ar.w == li ? vv.w : 0.f;
                return o;
            };
            if (!(SA_ABLATE & 1))
                mfma_stream_fn<M2::MB, M2::NB, C3 / 8>(asel, P3Tv + (size_t)(wn * M2::NB) * (C3 / 8) * 64 + lane, acc);
            __syncthreads();                         // every wave is done with its dW3 reads of H2 rows
#pragma unroll
            for (int nb = 0; nb < M2::NB; ++nb) {
                const int col = (wn * M2::NB + nb) * 32 + li;
#pragma unroll
                for (int mb = 0; mb < M2::MB; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (wm * M2::MB + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const float h = H2[row * LD2 + col];
                        const float z = acc[mb][nb][r] * (1.0f - h * h);
                        D[row * LD2 + col] = z;
                        accb2[nb] += z;
                    }
            }
        }
        __syncthreads();
        // ---- P5: dW2 += dZ2^T * H1  (K = TM rows; lanes < 32: rows [0,TM/2), lanes >= 32: the rest) ----
        {
            const float* Ap = D + (lh * (TM / 2)) * LD2 + w2_m * 32 + li;       // A[m = c2][k = row]
            const float* Bp = H1 + (lh * (TM / 2)) * LD1 + w2_n0 * 32 + li;     // B[k = row][n = c1]
            float ap, bp[NBK], aq, bq[NBK];
#define SA_DW2_LOAD(a_, b_, s_)  \
    a_ = Ap[(s_) * LD2];         \
    _Pragma("unroll") for (int j = 0; j < NBK; ++j) b_[j] = Bp[(s_) * LD1 + j * 32];
#define SA_DW2_MMA(a_, b_) _Pragma("unroll") for (int j = 0; j < NBK; ++j) accW2[j] = MFMA(a_, b_[j], accW2[j]);
            SA_DW2_LOAD(ap, bp, 0)
#pragma unroll 1
            for (int s = 0; s < ((SA_ABLATE & 4) ? 0 : TM / 2); s += 2) {
                SA_DW2_LOAD(aq, bq, s + 1)
                SA_DW2_MMA(ap, bp)
                SA_DW2_LOAD(ap, bp, s + 2)          // last trip reads one row past this half: discarded
                SA_DW2_MMA(aq, bq)
            }
#undef SA_DW2_LOAD
#undef SA_DW2_MMA
        }
        // ---- P6: dH1 = dZ2 * W2 -> dZ1 = dH1 .* (1 - H1^2), in place of H1 ----------------------------
        {
            const int wn = wave % MH::NBW, wm = wave / MH::NBW;
            f32x16 accH[MH::MB][MH::NB];
            zero_acc<MH::MB, MH::NB>(accH);
            if (!(SA_ABLATE & 8))
                mfma_stream<MH::MB, MH::NB, NGT>(D + (wm * MH::MB * 32 + li) * LD2 + lh * (C2 / 2), LD2,
                                                 P2Tv + (size_t)(wn * MH::NB) * NGT * 64 + lane, accH);
            __syncthreads();                         // every wave is done reading H1 (P5) and D
#pragma unroll
            for (int nb = 0; nb < MH::NB; ++nb)
#pragma unroll
                for (int mb = 0; mb < MH::MB; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (wm * MH::MB + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const int col = (wn * MH::NB + nb) * 32 + li;
                        const float h = H1[row * LD1 + col];
                        H1[row * LD1 + col] = accH[mb][nb][r] * (1.0f - h * h);
                    }
        }
        __syncthreads();
        // ---- P7: dW1[:, :3], db1, scatter dZ1 to the source points -------------------------------
        {
            constexpr int PARTS = NT / C1, RPT = TM / PARTS;
            const int c = tid % C1, p0 = (tid / C1) * RPT;
#pragma unroll 4
            for (int p = p0; p < ((SA_ABLATE & 16) ? p0 : p0 + RPT); ++p) {
                const float z = H1[p * LD1 + c];
                const float4 x = *(const float4*)(Xz + p * 4);
                accW1[0] = fmaf(z, x.x, accW1[0]);
                accW1[1] = fmaf(z, x.y, accW1[1]);
                accW1[2] = fmaf(z, x.z, accW1[2]);
                accW1[3] += z;
                if (a.dY) unsafeAtomicAdd(a.dY + (long)Src[p] * C1 + c, z);
            }
        }
        __syncthreads();
    }

    // ---- write this work-group's partial sums ---------------------------------------------------------
    float* part = a.parts + (size_t)blockIdx.x * P::N;
    const int tid = tid0, wave = wave0, lh0 = lane0 >> 5, li0 = lane0 & 31;
    const int w2_m = wave % (C2 / 32), w2_n0 = (wave / (C2 / 32)) * NBK;
#pragma unroll
    for (int j = 0; j < NBK; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c2 = w2_m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh0;
            part[P::O_DW2 + c2 * C1 + (w2_n0 + j) * 32 + li0] = accW2[j][r];
        }
    {
        const int c = tid % C3, ks = tid / C3;
#pragma unroll
        for (int j = 0; j < KS; ++j) part[P::O_DW3 + c * C2 + ks * KS + j] = accW3[j];
        if (ks == 0) part[P::O_DB3 + c] = accb3;
    }
    // db2 and dW1z/db1: reduce the per-lane / per-row-part sums through LDS (H1/H2 are free now)
    __syncthreads();
    {
        constexpr int PARTS1 = NT / C1;
        float* s2 = H1;                              // [M2::MW][C2]
        float* s1 = H2;                              // [PARTS1][C1][4]
        static_assert(M2::MW * C2 <= TM * LD1 && PARTS1 * C1 * 4 <= TM * LD2, "reduction scratch");
        const int wn = wave % M2::NBW, wm = wave / M2::NBW;
#pragma unroll
        for (int nb = 0; nb < M2::NB; ++nb) {
            const float v = accb2[nb] + __shfl_xor(accb2[nb], 32, 64);
            if (lh0 == 0) s2[wm * C2 + (wn * M2::NB + nb) * 32 + li0] = v;
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) s1[tid * 4 + d] = accW1[d];       // tid = part * C1 + c
        __syncthreads();
        if (tid < C2) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < M2::MW; ++q) s += s2[q * C2 + tid];
            part[P::O_DB2 + tid] = s;
        }
        if (tid < C1 * 4) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < PARTS1; ++q) s += s1[q * C1 * 4 + tid];
            part[P::O_DW1 + tid] = s;                // [c][x,y,z,bias]
        }
    }
}

// (a work-group owns 64 outputs; its 8 waves take the partial sums w, w + 8, ... and add up in wave order through LDS: with one
// thread per output walking all <= 1024 partials the SA1 reduction took 220 us on 50 work-groups -- 2 % of a PointNet++ step)
#define SA_RED_G 8
template <int C1, int C2, int C3>
__global__ __launch_bounds__(64 * SA_RED_G) void sa_bwd_reduce_kernel(const float* __restrict__ parts, int nparts,
                                                                      float* __restrict__ dW1, long lddw1,
                                                                      float* __restrict__ db1, float* __restrict__ dW2,
                                                                      float* __restrict__ db2, float* __restrict__ dW3,
                                                                      float* __restrict__ db3, int dw1_zero_end) {
    using P = SaPart<C1, C2, C3>;
    __shared__ float red[SA_RED_G][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (i < P::N) {
#pragma unroll 4
        for (int w = wv; w < nparts; w += SA_RED_G) s += parts[(size_t)w * P::N + i];
    }
    red[wv][lane] = s;
    __syncthreads();
    if (wv != 0 || i >= P::N) return;
    s = red[0][lane];
#pragma unroll
    for (int q = 1; q < SA_RED_G; ++q) s += red[q][lane];
    if (i < P::O_DB2) dW2[i] = s;
    else if (i < P::O_DW3) db2[i - P::O_DB2] = s;
    else if (i < P::O_DB3) dW3[i - P::O_DW3] = s;
    else if (i < P::O_DW1) db3[i - P::O_DB3] = s;
    else {
        const int j = i - P::O_DW1, c = j >> 2, d = j & 3;
        if (d < 3) dW1[c * lddw1 + d] = s;
        else {
            db1[c] = s;
            // a level without input features: columns [3, dw1_zero_end) of its first weight are padding and never receive data
            for (int z = 3; z < dw1_zero_end; ++z) dW1[c * lddw1 + z] = 0.f;
        }
    }
}

extern "C" size_t pm_sa_bwd_workspace_bytes(int C1, int C2, int C3) {
    const size_t n = (size_t)C2 * C1 + C2 + (size_t)C3 * C2 + C3 + (size_t)C1 * 4;
    return n * SA_BWD_MAXGRID * sizeof(float);
}

extern "C" int pm_sa_bwd_f32(const float* xyz, const float* centers, const int32_t* idx, const float* Y, int B, int P,
                             int S, int nsample, const float* W1, long ldw1, const float* b1, const float* b2,
                             const float* W3, const float* packed, int C1, int C2, int C3, const float* pooled,
                             long ldp, const int32_t* arg, const float* dpooled, long lddp, float* dW1, long lddw1,
                             float* db1, float* dW2, float* db2, float* dW3, float* db3, float* dY,
                             const float* h2_saved, void* workspace, size_t workspace_bytes, void* stream) {
    PM_REQUIRE(xyz && centers && idx && W1 && b1 && b2 && W3 && packed && pooled && arg && dpooled);
    PM_REQUIRE(dW1 && db1 && dW2 && db2 && dW3 && db3 && workspace);
    PM_REQUIRE(B > 0 && P > 0 && S > 0 && ldw1 >= 3 && lddw1 >= 3 && ldp >= C3 && lddp >= C3);
    if (!pm_sa_supported(C1, C2, C3, nsample)) return PM_EUNSUPPORTED;
    if (((uintptr_t)packed & 15) != 0) return PM_EALIGN;
    if (workspace_bytes < pm_sa_bwd_workspace_bytes(C1, C2, C3)) return PM_EWORKSPACE;
    SaArgs a = {};
    a.xyz = xyz; a.centers = centers; a.idx = idx; a.Y = Y; a.W1 = W1; a.ldw1 = ldw1; a.b1 = b1; a.b2 = b2;
    a.packed = packed; a.pooled = const_cast<float*>(pooled); a.ldp = ldp; a.arg = const_cast<int32_t*>(arg);
    a.G = (long)B * S; a.S = S; a.P = P; a.W3 = W3; a.dpooled = dpooled; a.lddp = lddp; a.dY = dY;
    a.parts = (float*)workspace;
    a.h2 = const_cast<float*>(h2_saved);
    const int ncu = sa_cu_count();
#define SA_BWD_LAUNCH(C1_, C2_, C3_, TM_, NW_, WPE_, WGCU_)                                                        \
    {                                                                                                              \
        const long ntiles = (a.G + (TM_) / SA_NS - 1) / ((TM_) / SA_NS);                                           \
        long grid = (long)ncu * (WGCU_);                                                                           \
        if (grid > SA_BWD_MAXGRID) grid = SA_BWD_MAXGRID;                                                          \
        if (grid > ntiles) grid = ntiles;                                                                          \
        hipLaunchKernelGGL((sa_bwd_kernel<C1_, C2_, C3_, TM_, NW_, WPE_>), dim3((unsigned)grid), dim3((NW_) * 64), \
                           0, pm_stream(stream), a);                                                               \
        constexpr int n = SaPart<C1_, C2_, C3_>::N;                                                                \
        hipLaunchKernelGGL((sa_bwd_reduce_kernel<C1_, C2_, C3_>), dim3((n + 63) / 64), dim3(64 * SA_RED_G), 0,             \
                           pm_stream(stream), a.parts, (int)grid, dW1, lddw1, db1, dW2, db2, dW3, db3, 0);         \
    }
    if (SA_CFG_A(C1, C2, C3)) SA_BWD_LAUNCH(64, 64, 128, 64, 4, SA_A_BWD_WPE, SA_A_BWD_WPE)
    else SA_BWD_LAUNCH(128, 128, 256, SA_B_BWD_TM, SA_B_BWD_NW, (SA_B_BWD_NW / 4), 1)
#undef SA_BWD_LAUNCH
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ============================================================================================================
// Duplicate-free ("packed") form of the same level.
//
// Ball query pads a group whose ball holds fewer than 32 points with copies of its FIRST hit (the canonical
// PointNet++ rule, pm_ball_query_f32).  A padded row is the same source point against the same
// centre as row 0 of its group, so its three activations are bit-identical to row 0's; the max-pool takes the
// LOWEST row attaining the maximum, so a padding row never wins, never receives a gradient, and the level's
// outputs (pooled, arg) and every gradient are exactly those of the group's DISTINCT rows.  On the SURVEY's
// synthetic clouds (1024 points uniform in a 2 m cube, radius 0.2 / 0.4) a group has 3.9 / 5.4 distinct rows of
// 32: the dense kernels above spend 88 % / 83 % of their rows on copies.
//
// pm_sa_plan_i32 (once per neighbourhood table) lists the distinct rows back to back (rowmap: source point and
// group per packed row; grow: first packed row of every group) and cuts them into tiles of whole groups
// (<= tile_rows rows, <= tile_groups groups, never across clouds).  The kernels below run the SAME per-row
// arithmetic as sa_fwd_kernel / sa_bwd_kernel on those tiles:
//   forward   layers 1-3 as above on the tile's rows; the pre-activations of layer 3 go through LDS and one thread
//             per (group, channel) takes the max over the group's rows (same strict `>` in row order => same arg).
//   backward  dZ3 has one non-zero per (group, channel) at packed row grow[g] + arg: the MFMA A operand is built
//             per row from ITS group's (val, arg) entries, so dH2 = dZ3 * W3 is ONE rows x C3 x C2 product per
//             tile whatever the number of groups in it (the dense kernel runs it per group on 32 rows).
// Results are bit-identical to the dense kernels wherever those are deterministic (pooled, arg, the saved layer 2);
// the weight gradients differ by fp32 summation order only (fewer, exactly-zero terms dropped).
// The kernels read the tile count from device memory (persistent work-groups): no host synchronisation anywhere.
#ifndef SA_PK_ABLATE
#define SA_PK_ABLATE 0   // timing probes (wrong results): bwd 1 = no dW3, 2 = no dH2 MFMA, 4 = no layer-1 recompute, 8 = no P7, 16 = no dW2 / dH1 MFMA; fwd 32 = no pooling pass, 64 = no layer 2 / 3 MFMA
#endif
#ifndef SA_A_PK_BWD_WPE
#define SA_A_PK_BWD_WPE 3     // work-groups per CU of the SA1-shaped packed backward (49.6 KB of LDS each; A/B: 2)
#endif
#ifndef SA_B_PK_BWD_NW
#define SA_B_PK_BWD_NW 16     // waves of the SA2-shaped packed backward's work-group (one per CU); 8: two row blocks per wave (A/B)
#endif
#ifndef SA_PK_DZ3_LDS
#define SA_PK_DZ3_LDS 1   // 1: the layer-3 gradient rows of a channel window are BUILT ONCE PER TILE in LDS (zero-fill + one store per
                          // (group, channel)) and the dH2 product streams its A operand from there; 0: every lane builds its operand
                          // in registers from (val, arg) per k-group (round 4: ~14 VALU per 4 MFMAs, 0.53 ms of the 1.67 ms SA2 backward
                          // against an MFMA floor of 0.30)
#endif
struct SaPk {
    const int32_t* grow;     // (G + 1)   first packed row of each group
    const int2* rowmap;      // (R)       {flat source point b*P + idx, local group << 8 | row inside the group}
    const float4* relxyz;    // (R)       xyz[source point] - centre[group]
    const int4* tiles;       // (T)       {first packed row, first group, groups, rows}
    const int32_t* totals;   // [0] = R, [1] = T
};

extern "C" int pm_sa_packed_tile(int C1, int C2, int C3, int* tile_rows, int* tile_groups) {
    PM_REQUIRE(tile_rows && tile_groups);
    if (SA_CFG_A(C1, C2, C3)) { *tile_rows = 64; *tile_groups = 20; return PM_OK; }
    if (SA_CFG_B(C1, C2, C3)) { *tile_rows = 128; *tile_groups = 28; return PM_OK; }
    return PM_EUNSUPPORTED;
}

// ---- plan ---------------------------------------------------------------------------------------------------
// one work-group per cloud: distinct rows per group (entries j >= 1 equal to entry 0 are padding: real hits are
// ascending and distinct), their prefix inside the cloud, and the greedy cut into tiles of whole groups
__global__ __launch_bounds__(256) void sa_plan_count_kernel(const int32_t* __restrict__ idx, int S, int ns, int tile_rows,
                                                             int tile_groups, int32_t* __restrict__ lrow,
                                                             int32_t* __restrict__ lstart, int4* __restrict__ ltiles,
                                                             int32_t* __restrict__ crows, int32_t* __restrict__ ctiles) {
    __shared__ int cnt[1024];
    const long b = blockIdx.x;
    for (int s = threadIdx.x; s < S; s += 256) {
        const int32_t* p = idx + (b * S + s) * ns;
        const int first = p[0];
        int c = 1;
        for (int j = 1; j < ns; ++j) c += p[j] != first;
        cnt[s] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int rows = 0, start = 0, prefix = 0, nt = 0, row0 = 0;
        for (int s = 0; s < S; ++s) {
            const int c = cnt[s];
            if (s > start && (rows + c > tile_rows || s - start == tile_groups)) {
                ltiles[b * S + nt++] = make_int4(row0, start, s - start, rows);
                start = s;
                row0 = prefix;
                rows = 0;
            }
            lrow[b * S + s] = prefix;
            lstart[b * S + s] = start;                   // first group of the tile this group belongs to
            prefix += c;
            rows += c;
        }
        ltiles[b * S + nt++] = make_int4(row0, start, S - start, rows);
        crows[b] = prefix;
        ctiles[b] = nt;
    }
}

// exclusive prefix over the clouds (one work-group)
__global__ __launch_bounds__(1024) void sa_plan_scan_kernel(const int32_t* __restrict__ crows, const int32_t* __restrict__ ctiles,
                                                             int B, int32_t* __restrict__ rbase, int32_t* __restrict__ tbase,
                                                             int32_t* __restrict__ totals) {
    __shared__ int sr[1024], st[1024];
    __shared__ int carry[2];
    if (threadIdx.x == 0) carry[0] = carry[1] = 0;
    __syncthreads();
    for (int lo = 0; lo < B; lo += 1024) {
        const int i = lo + threadIdx.x;
        const int r = i < B ? crows[i] : 0, t = i < B ? ctiles[i] : 0;
        sr[threadIdx.x] = r;
        st[threadIdx.x] = t;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int ar = threadIdx.x >= o ? sr[threadIdx.x - o] : 0, at = threadIdx.x >= o ? st[threadIdx.x - o] : 0;
            __syncthreads();
            sr[threadIdx.x] += ar;
            st[threadIdx.x] += at;
            __syncthreads();
        }
        if (i < B) {
            rbase[i] = carry[0] + sr[threadIdx.x] - r;
            tbase[i] = carry[1] + st[threadIdx.x] - t;
        }
        __syncthreads();
        if (threadIdx.x == 1023) {
            carry[0] += sr[1023];
            carry[1] += st[1023];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        totals[0] = carry[0];
        totals[1] = carry[1];
    }
}

// rowmap[r] = {flat source point, (group - its tile's first group) << 8 | row inside the group}; relxyz[r] = xyz[source] -
// centre[group] (the sub_rn the dense kernels evaluate per row and tile): both depend on the coordinates only, so the level
// kernels' staging is two coalesced loads per row instead of a chain row -> point -> coordinates.
__global__ __launch_bounds__(256) void sa_plan_fill_kernel(const int32_t* __restrict__ idx, const float* __restrict__ xyz,
                                                            const float* __restrict__ centers, int B, int P, int S, int ns,
                                                            const int32_t* __restrict__ lrow, const int32_t* __restrict__ lstart,
                                                            const int4* __restrict__ ltiles,
                                                            const int32_t* __restrict__ ctiles, const int32_t* __restrict__ rbase,
                                                            const int32_t* __restrict__ tbase, const int32_t* __restrict__ totals,
                                                            int32_t* __restrict__ grow, int2* __restrict__ rowmap,
                                                            float4* __restrict__ relxyz, int4* __restrict__ tiles) {
    const long b = blockIdx.x;
    const int rb = rbase[b], tb = tbase[b];
    for (int s = threadIdx.x; s < S; s += 256) {
        const long g = b * S + s;
        const int r0 = rb + lrow[g];
        grow[g] = r0;
        const int lg = (s - lstart[g]) << 8;
        const float cx = centers[g * 3], cy = centers[g * 3 + 1], cz = centers[g * 3 + 2];
        const int32_t* p = idx + g * ns;
        const int first = p[0];
        int k = 0;
        for (int j = 0; j < ns; ++j) {
            const int v = p[j];
            if (j > 0 && v == first) continue;
            const long sp = b * P + v;
            rowmap[r0 + k] = make_int2((int)sp, lg | k);
            relxyz[r0 + k] = make_float4(sub_rn(xyz[sp * 3], cx), sub_rn(xyz[sp * 3 + 1], cy), sub_rn(xyz[sp * 3 + 2], cz), 0.f);
            ++k;
        }
    }
    const int nt = ctiles[b];
    for (int t = threadIdx.x; t < nt; t += 256) {
        const int4 lt = ltiles[b * S + t];
        tiles[tb + t] = make_int4(rb + lt.x, (int)(b * S) + lt.y, lt.z, lt.w);
    }
    if (b == B - 1 && threadIdx.x == 0) grow[(long)B * S] = totals[0];
}

extern "C" size_t pm_sa_plan_workspace_bytes(int B, int S) {
    return ((size_t)B * S * 6 + (size_t)B * 4) * sizeof(int32_t) + 64;
}

extern "C" int pm_sa_plan_i32(const int32_t* idx, const float* xyz, const float* centers, int B, int P, int S, int nsample,
                              int tile_rows, int tile_groups, int32_t* grow, int32_t* rowmap, float* relxyz, int32_t* tiles,
                              int32_t* totals, void* workspace, size_t workspace_bytes, void* stream) {
    PM_REQUIRE(idx && xyz && centers && grow && rowmap && relxyz && tiles && totals && workspace);
    PM_REQUIRE(B > 0 && P > 0 && S > 0 && S <= 1024 && nsample > 0 && nsample <= tile_rows && nsample <= 255 && tile_groups > 0);
    PM_REQUIRE((long)B * P < (1L << 31) && (long)B * S * nsample < (1L << 31));
    if (workspace_bytes < pm_sa_plan_workspace_bytes(B, S)) return PM_EWORKSPACE;
    if ((((uintptr_t)workspace) & 15) || (((uintptr_t)tiles) & 15) || (((uintptr_t)rowmap) & 7) || (((uintptr_t)relxyz) & 15)) return PM_EALIGN;
    const size_t G = (size_t)B * S;
    int4* ltiles = (int4*)workspace;
    int32_t* lrow = (int32_t*)(ltiles + G);
    int32_t* lstart = lrow + G;
    int32_t* crows = lstart + G;
    int32_t *ctiles = crows + B, *rbase = ctiles + B, *tbase = rbase + B;
    hipStream_t st = pm_stream(stream);
    hipLaunchKernelGGL(sa_plan_count_kernel, dim3(B), dim3(256), 0, st, idx, S, nsample, tile_rows, tile_groups, lrow, lstart, ltiles,
                       crows, ctiles);
    hipLaunchKernelGGL(sa_plan_scan_kernel, dim3(1), dim3(1024), 0, st, crows, ctiles, B, rbase, tbase, totals);
    hipLaunchKernelGGL(sa_plan_fill_kernel, dim3(B), dim3(256), 0, st, idx, xyz, centers, B, P, S, nsample, lrow, lstart, ltiles, ctiles,
                       rbase, tbase, totals, grow, (int2*)rowmap, (float4*)relxyz, (int4*)tiles);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ---- inverse of the plan: source point -> its packed rows (CSR, ascending) --------------------------------------
// The layer-1 gradient of a level with input features leaves the backward PER PACKED ROW (dz1, plain stores); a source point
// sits in several groups, so dY[point] = sum of its rows.  With the rows of every point listed in ascending order the sum has
// ONE order whatever the launch does: no floating-point atomics anywhere (they were the build's only run-dependent bits).
// One work-group per cloud (its rows are contiguous: [grow[b*S], grow[(b+1)*S])): integer counts by LDS atomics (order-free),
// prefix, fill through per-point cursors (slot order run-dependent), then every point's short list is sorted.
__global__ __launch_bounds__(256) void sa_plan_inverse_kernel(const int2* __restrict__ rowmap, const int32_t* __restrict__ grow, int B,
                                                               int P, int S, int32_t* __restrict__ inv_start,
                                                               int32_t* __restrict__ inv_rows) {
    extern __shared__ int sh[];                          // cnt[P + 1] | cur[P] | part[256]
    int* cnt = sh;
    int* cur = sh + P + 1;
    int* part = cur + P;
    const long b = blockIdx.x;
    const int rb = grow[b * S], re = grow[(b + 1) * S];
    const int base = (int)(b * P);
    for (int p = threadIdx.x; p <= P; p += 256) cnt[p] = 0;
    __syncthreads();
    for (int r = rb + threadIdx.x; r < re; r += 256) atomicAdd(&cnt[rowmap[r].x - base], 1);
    __syncthreads();
    // exclusive prefix over the P counts: a contiguous chunk per thread, then the 256 chunk sums
    const int per = (P + 255) / 256, lo = threadIdx.x * per, hi = lo + per < P ? lo + per : P;
    int sum = 0;
    for (int p = lo; p < hi; ++p) sum += cnt[p];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;
    for (int p = lo; p < hi; ++p) {
        const int c = cnt[p];
        cur[p] = run;
        inv_start[base + p] = rb + run;
        run += c;
    }
    if (b == B - 1 && threadIdx.x == 0) inv_start[(long)B * P] = re;
    __syncthreads();
    for (int r = rb + threadIdx.x; r < re; r += 256) {
        const int p = rowmap[r].x - base;
        inv_rows[rb + atomicAdd(&cur[p], 1)] = r;
    }
    __threadfence_block();
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += 256) {          // cur[p] is now the END of p's list
        const int e = rb + cur[p], s = e - cnt[p];
        for (int i = s + 1; i < e; ++i) {
            const int v = inv_rows[i];
            int j = i - 1;
            while (j >= s && inv_rows[j] > v) {
                inv_rows[j + 1] = inv_rows[j];
                --j;
            }
            inv_rows[j + 1] = v;
        }
    }
}

extern "C" int pm_sa_plan_inverse_i32(const int32_t* rowmap, const int32_t* grow, int B, int P, int S, int32_t* inv_start,
                                      int32_t* inv_rows, void* stream) {
    PM_REQUIRE(rowmap && grow && inv_start && inv_rows && B > 0 && P > 0 && S > 0);
    PM_REQUIRE(P <= 7000);                               // cnt + cur + part in 64 KB of LDS
    PM_REQUIRE((long)B * P < (1L << 31));
    if (((uintptr_t)rowmap) & 7) return PM_EALIGN;
    const size_t lds = ((size_t)2 * P + 1 + 256) * sizeof(int);
    hipLaunchKernelGGL(sa_plan_inverse_kernel, dim3(B), dim3(256), lds, pm_stream(stream), (const int2*)rowmap, grow, B, P, S, inv_start,
                       inv_rows);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// dY[point, :] = sum over the point's packed rows (ascending) of dz1[row, :]; points in no group get zeros (no zero-fill pass).
// One thread per (point, 16-byte piece): the C1/4 threads of a point read whole 4*C1-byte rows.
__global__ __launch_bounds__(256) void sa_dy_segsum_kernel(const float4* __restrict__ dz1, const int32_t* __restrict__ inv_start,
                                                            const int32_t* __restrict__ inv_rows, long npoints, int c4n,
                                                            float* __restrict__ dY, long lddy) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long p = i / c4n;
    const int c4 = (int)(i % c4n);
    if (p >= npoints) return;
    const int s = inv_start[p], e = inv_start[p + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = s; j < e; ++j) {
        const float4 v = dz1[(long)inv_rows[j] * c4n + c4];
        acc.x += v.x;
        acc.y += v.y;
        acc.z += v.z;
        acc.w += v.w;
    }
    *(float4*)(dY + p * lddy + 4 * c4) = acc;
}

extern "C" int pm_sa_dy_segsum_f32(const float* dz1, const int32_t* inv_start, const int32_t* inv_rows, long npoints, int C1, float* dY,
                                   long lddy, void* stream) {
    PM_REQUIRE(dz1 && inv_start && inv_rows && dY && npoints > 0 && C1 > 0 && C1 % 4 == 0 && lddy >= C1 && lddy % 4 == 0);
    if ((((uintptr_t)dz1) & 15) || (((uintptr_t)dY) & 15)) return PM_EALIGN;
    const long n = npoints * (C1 / 4);
    hipLaunchKernelGGL(sa_dy_segsum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, pm_stream(stream), (const float4*)dz1, inv_start,
                       inv_rows, npoints, C1 / 4, dY, lddy);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ---- shared staging of a packed tile -------------------------------------------------------------------------
// Xz / Src as sa_stage, from the plan's per-row tables (two coalesced loads per row); Lgr[t] = local group << 8 | local row
// (rows past the tile's end: row 255, which no arg-max equals); Ls[j] = first tile-local row of group j (j <= groups).  Rows
// past the end repeat the tile's first row (finite activations that nothing reads back).
template <int TM, int NT>
__device__ __forceinline__ void sa_stage_pk(const SaPk& k, const int4 td, int tid, float* __restrict__ Xz,
                                            int* __restrict__ Src, int* __restrict__ Lgr, int* __restrict__ Ls) {
    for (int t = tid; t < TM; t += NT) {
        const bool live = t < td.w;
        const long r = td.x + (live ? t : 0);
        const int2 rm = k.rowmap[r];
        *(float4*)(Xz + t * 4) = k.relxyz[r];
        Src[t] = rm.x;
        if (Lgr) Lgr[t] = live ? rm.y : 255;
    }
    for (int j = tid; j <= td.z; j += NT) Ls[j] = k.grow[td.y + j] - td.x;
}

// ==================================================================================== packed forward
// NBW3 waves side by side along the C3 axis (NW / NBW3 along the rows); a staging round holds NBW3*32 channels of Z3.
template <int C1, int C2, int C3, int TM, int NW, int WPE, int NGMAX, int NBW3>
__global__ __launch_bounds__(NW * 64, WPE) void sa_fwd_pk_kernel(SaArgs a, SaPk k) {
    constexpr int NT = NW * 64, LD1 = C1 + 4, LD2 = C2 + 4, LDM = LD1 > LD2 ? LD1 : LD2;
    constexpr int MW3 = NW / NBW3, MB3 = TM / 32 / MW3, NB3 = C3 / 32 / NBW3, W3S = NBW3 * 32, LDZ = W3S + 4;
    static_assert(MW3 * NBW3 == NW && MB3 * MW3 * 32 == TM && NB3 * NBW3 * 32 == C3 && NT % W3S == 0, "layer-3 wave mapping");
    constexpr int HSZ = TM * LDM, ZSZ = TM * LDZ, BUF = HSZ > ZSZ ? HSZ : ZSZ;
    __shared__ __attribute__((aligned(16))) float smem[BUF + TM * 4 + TM + (NGMAX + 4)];
    float* H1 = smem;
    float* H2 = smem;
    float* Zs = smem;                                    // layer-3 pre-activations of a staging round (after H2 is dead)
    float* Xz = smem + BUF;
    int* Src = (int*)(Xz + TM * 4);
    int* Ls = Src + TM;
    constexpr int NG3 = C2 / 8;

    const int tid0 = threadIdx.x;
    const float4* P2v = (const float4*)a.packed;
    const float4* P3v = (const float4*)(a.packed + (size_t)C1 * C2);
    const int ntiles = k.totals[1];

    int4 td = k.tiles[(int)blockIdx.x < ntiles ? blockIdx.x : 0];
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));                    // see sa_fwd_kernel: recompute, don't hoist
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int li = lane & 31, lh = lane >> 5;
        const int nxt = tile + (int)gridDim.x;
        const int4 tdn = k.tiles[nxt < ntiles ? nxt : tile];       // the next tile's descriptor: one hop off the next staging
        sa_stage_pk<TM, NT>(k, td, tid, Xz, Src, nullptr, Ls);
        __syncthreads();
        sa_layer1<C1, TM, NT>(a, tid, Xz, Src, H1);
        __syncthreads();
        sa_layer2<C1, C2, TM, NW, true>(H1, P2v, a.b2, wave, lane, H2);
        __syncthreads();
        // ---- layer 3 ---------------------------------------------------------------------------------
        const int wn = wave % NBW3, wm = wave / NBW3;
        f32x16 acc[MB3][NB3];
        zero_acc<MB3, NB3>(acc);
        if (!(SA_PK_ABLATE & 64))
            mfma_stream<MB3, NB3, NG3>(H2 + (wm * MB3 * 32 + li) * LD2 + lh * (C2 / 2), LD2,
                                       P3v + (size_t)(wn * NB3) * NG3 * 64 + lane, acc);
        if (a.h2) {                                          // training forward: keep H2 (packed rows) for the backward
#pragma unroll 2
            for (int q = tid; q < TM * C2 / 4; q += NT) {
                const int row = q / (C2 / 4), c4 = q % (C2 / 4);
                if (row < td.w)
                    *(float4*)(a.h2 + (long)(td.x + row) * C2 + 4 * c4) = *(const float4*)(H2 + row * LD2 + 4 * c4);
            }
        }
        // ---- max over each group's rows: Z3 through LDS, one thread per (group, channel) ------------------
#pragma unroll
        for (int nb = 0; nb < NB3; ++nb) {
            __syncthreads();                             // H2 (or the previous round of Zs) is no longer read
#pragma unroll
            for (int mb = 0; mb < MB3; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (wm * MB3 + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    Zs[row * LDZ + wn * 32 + li] = acc[mb][nb][r];
                }
            __syncthreads();
            const int s = tid % W3S, ch = ((s >> 5) * NB3 + nb) * 32 + (s & 31);
            const float bv = a.b3[ch];
            for (int j = tid / W3S; j < ((SA_PK_ABLATE & 32) ? 0 : td.z); j += NT / W3S) {
                const int r0 = Ls[j], r1 = Ls[j + 1];
                float best = Zs[r0 * LDZ + s];
                int bi = 0;
                for (int r = r0 + 1; r < r1; ++r) {
                    const float v = Zs[r * LDZ + s];
                    if (v > best) {
                        best = v;
                        bi = r - r0;
                    }
                }
                const long g = td.y + j;
                a.pooled[g * a.ldp + ch] = pm_tanh(best + bv);
                a.arg[g * C3 + ch] = bi;
                // the rows feed a group-all level directly ([features | xyz | 0], network.py): their tail in the same pass
                if (a.centers && nb == 0 && s < a.tail_cols) a.pooled[g * a.ldp + C3 + s] = s < 3 ? a.centers[g * 3 + s] : 0.f;
            }
        }
        __syncthreads();
        td = tdn;
    }
}

static long sa_pk_grid(long G, int ncu, int wgcu, long cap) {
    long grid = (long)ncu * wgcu;
    if (cap > 0 && grid > cap) grid = cap;
    if (grid > G) grid = G;                              // a tile holds at least one group
    return grid;
}

extern "C" int pm_sa_fwd_packed_f32(const float* Y, int B, int P, int S, const int32_t* grow, const int32_t* rowmap,
                                    const float* relxyz, const int32_t* tiles, const int32_t* totals,
                                    const float* W1, long ldw1, const float* b1, const float* b2, const float* b3,
                                    const float* packed, int C1, int C2, int C3, float* pooled, long ldp, int32_t* arg,
                                    float* h2_save, const float* tail_xyz, int tail_cols, void* stream) {
    PM_REQUIRE(grow && rowmap && relxyz && tiles && totals && W1 && b1 && b2 && b3 && packed && pooled && arg);
    PM_REQUIRE(B > 0 && P > 0 && S > 0 && ldw1 >= 3 && ldp >= C3);
    PM_REQUIRE(!tail_xyz || (tail_cols >= 3 && tail_cols <= 128 && C3 + tail_cols <= ldp));
    if (!pm_sa_supported(C1, C2, C3, SA_NS)) return PM_EUNSUPPORTED;
    if (((uintptr_t)packed & 15) != 0 || ((uintptr_t)tiles & 15) != 0 || ((uintptr_t)relxyz & 15) != 0) return PM_EALIGN;
    SaArgs a = {};
    a.Y = Y; a.W1 = W1; a.ldw1 = ldw1; a.b1 = b1; a.b2 = b2; a.b3 = b3;
    a.packed = packed; a.pooled = pooled; a.ldp = ldp; a.arg = arg; a.G = (long)B * S; a.S = S; a.P = P;
    a.h2 = h2_save;
    a.centers = tail_xyz;
    a.tail_cols = tail_xyz ? tail_cols : 0;
    SaPk k = {grow, (const int2*)rowmap, (const float4*)relxyz, (const int4*)tiles, totals};
    const int ncu = sa_cu_count();
    if (SA_CFG_A(C1, C2, C3))
        hipLaunchKernelGGL((sa_fwd_pk_kernel<64, 64, 128, 64, 4, 4, 20, 4>), dim3((unsigned)sa_pk_grid(a.G, ncu, 4, 0)),
                           dim3(256), 0, pm_stream(stream), a, k);
    else
        hipLaunchKernelGGL((sa_fwd_pk_kernel<128, 128, 256, 128, 8, 2, 28, 4>), dim3((unsigned)sa_pk_grid(a.G, ncu, 2, 0)),
                           dim3(512), 0, pm_stream(stream), a, k);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// =================================================================================== packed backward
// as mfma_stream_fn, for a window of NGC k-groups out of a packed operand with NGS k-groups per N-block
template <int MB, int NB, int NGS, int NGC, class AF>
__device__ __forceinline__ void mfma_stream_fn_win(AF aload, const float4* __restrict__ Bp, f32x16 (&acc)[MB][NB]) {
    OperandSet<MB, NB> ping, pong;
#define PM_LOAD_FN(o, g_)                                                                   \
    _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) o.b[nb] = Bp[(size_t)(nb * NGS + (g_)) * 64]; \
    _Pragma("unroll") for (int mb = 0; mb < MB; ++mb) o.a[mb] = aload(mb, (g_));
    PM_LOAD_FN(ping, 0)
#pragma unroll 1
    for (int g = 0; g < NGC; g += 2) {
        PM_LOAD_FN(pong, g + 1)
        __builtin_amdgcn_sched_barrier(0);
        mfma_set<MB, NB>(ping, acc);
        __builtin_amdgcn_sched_barrier(0);
        PM_LOAD_FN(ping, g + 2)                    // unconditional: one group past the window, discarded
        __builtin_amdgcn_sched_barrier(0);
        mfma_set<MB, NB>(pong, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef PM_LOAD_FN
}

template <int C1, int C2, int C3, int TM, int NW, int WPE, int NGMAX>
__global__ __launch_bounds__(NW * 64, WPE) void sa_bwd_pk_kernel(SaArgs a, SaPk k) {
    constexpr int NT = NW * 64, LD1 = C1 + 4, LD2 = C2 + 4;
    // channels are staged 128 at a time (slots 0-63: channels [64q, 64q+64), slots 64-127: C3/2 + the same -- the two
    // k halves the MFMA lanes own); Val rows padded to 132 floats, Arg rows to 132 bytes (bank spread across groups)
    constexpr int NCH = C3 / 128, LDV = 132, LDA = 33;
    // the LDS-built layer-3 gradient operand needs a 128-channel window to fit H1's buffer in ONE piece (C1 = 128: the SA2 shape);
    // at C1 = 64 it takes two windows with their own zero / scatter / barrier rounds and measured SLOWER than the register build
    // (SA1 backward 1.09 -> 1.17 ms per 2048 clouds, round 5 call B), so that shape keeps the per-lane build
    constexpr bool ZL = SA_PK_DZ3_LDS && C1 == 128;
    __shared__ __attribute__((aligned(16))) float smem[TM * (LD1 + LD2 + 4) + 2 * TM + (NGMAX + 4) + NGMAX * (LDV + LDA)];
    float* H1 = smem;                        // H1, later dZ1 in place
    float* H2 = H1 + TM * LD1;
    float* D = H2;                           // dZ2 overwrites H2 in place
    float* Xz = H2 + TM * LD2;
    int* Src = (int*)(Xz + TM * 4);
    int* Lgr = Src + TM;
    int* Ls = Lgr + TM;
    float* Val = (float*)(Ls + NGMAX + 4);
    uint32_t* ArgW = (uint32_t*)(Val + NGMAX * LDV);
    using P = SaPart<C1, C2, C3>;
    using M2 = WaveMap<TM, C2, NW>;          // dH2 output mapping
    using MH = WaveMap<TM, C1, NW>;          // dH1 output mapping
    constexpr int NGT = C2 / 8;

    const int tid0 = threadIdx.x, lane0 = tid0 & 63, wave0 = tid0 >> 6;
    const float4* P2v = (const float4*)a.packed;
    const float4* P2Tv = (const float4*)(a.packed + (size_t)C1 * C2 + (size_t)C2 * C3);
    const float4* P3Tv = (const float4*)(a.packed + (size_t)C1 * C2 * 2 + (size_t)C2 * C3);
    const int ntiles = k.totals[1];

    // persistent accumulators
    constexpr int TPC = NT / 128, KS = C2 / TPC;            // dW3: thread (slot = tid % 128, ks = tid / 128) owns KS k's per chunk
    static_assert(NT % 128 == 0 && C2 % TPC == 0 && KS % 4 == 0 && NCH * 128 == C3, "dW3 thread mapping");
    float accW3[NCH][KS], accb3[NCH];
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
        accb3[q] = 0.f;
#pragma unroll
        for (int j = 0; j < KS; ++j) accW3[q][j] = 0.f;
    }
    float accW1[4] = {0.f, 0.f, 0.f, 0.f};
    float accb2[M2::NB];
#pragma unroll
    for (int nb = 0; nb < M2::NB; ++nb) accb2[nb] = 0.f;
    constexpr int WBLK = (C2 / 32) * (C1 / 32), NBK = WBLK / NW;
    static_assert(NBK * NW == WBLK && (C1 / 32) % NBK == 0, "dW2 wave mapping");
    f32x16 accW2[NBK];
#pragma unroll
    for (int j = 0; j < NBK; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW2[j][r] = 0.f;

    constexpr int EPT = TM * C2 / 4 / NT, NV = (NGMAX * 128 + NT - 1) / NT;
    static_assert(EPT * NT * 4 == TM * C2, "H2 tile pieces per thread");
    int4 td = k.tiles[(int)blockIdx.x < ntiles ? blockIdx.x : 0];
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int li = lane & 31, lh = lane >> 5;
        const int w2_m = wave % (C2 / 32), w2_n0 = (wave / (C2 / 32)) * NBK;
        const int nxt = tile + (int)gridDim.x;
        const int4 tdn = k.tiles[nxt < ntiles ? nxt : tile];
        // ---- P0: gather; the tile's saved H2 rows and the first channel chunk's (pooled, dpooled, arg) entries are REQUESTED
        // here and stored to LDS behind the layer-1 recompute, which needs none of them: their round trips run under it
        sa_stage_pk<TM, NT>(k, td, tid, Xz, Src, Lgr, Ls);
        // (ZL with the saved layer 2: nothing runs between these requests and their LDS stores any more -- layer 1 moved behind
        // P4 -- so both go straight through: held across the barrier the sixteen H2 registers were SPILLED at this register budget,
        // 0.8 GB of scratch traffic per launch, profiles/hbm_traffic.json round 5: 2.50 -> 2.81 GB)
        const bool direct = ZL && a.h2 != nullptr;
        float4 hq[EPT];
        if (a.h2) {
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const int q = tid + i * NT, row = q / (C2 / 4), c4 = q % (C2 / 4);
                hq[i] = make_float4(0.f, 0.f, 0.f, 0.f);             // rows past the end: zero (their dZ2 is 0 * (1 - 0))
                if (row < td.w) hq[i] = *(const float4*)(a.h2 + (long)(td.x + row) * C2 + 4 * c4);
            }
            if (direct) {
#pragma unroll
                for (int i = 0; i < EPT; ++i) {
                    const int q = tid + i * NT, row = q / (C2 / 4), c4 = q % (C2 / 4);
                    *(float4*)(H2 + row * LD2 + 4 * c4) = hq[i];
                }
            }
        }
        float pv[NV], dv[NV];
        int av[NV];
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int i = tid + u * NT, j = i >> 7, sl = i & 127;
            pv[u] = dv[u] = 0.f;
            av[u] = 0;
            if (j < td.z) {
                const int c = (sl < 64 ? sl : C3 / 2 + (sl - 64));
                const long g = td.y + j;
                pv[u] = a.pooled[g * a.ldp + c];
                dv[u] = a.dpooled[g * a.lddp + c];
                av[u] = a.arg[g * C3 + c];
            }
        }
        if (direct) {
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int i = tid + u * NT, j = i >> 7, sl = i & 127;
                if (j < td.z) {
                    Val[j * LDV + sl] = dv[u] * (1.0f - pv[u] * pv[u]);
                    ((uint8_t*)ArgW)[j * (LDA * 4) + sl] = (uint8_t)av[u];
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        // ---- P1/P2: H1 recomputed, H2 stored (or recomputed) -----------------------------------------
        // (SA_PK_DZ3_LDS: H1's buffer holds the layer-3 gradient rows during P3; H1 itself is recomputed behind P4 -- with the
        // saved layer 2 nothing needs it before the dW2 product.  Without the saved layer 2 it is computed here too, for layer 2.)
        if (!(SA_PK_ABLATE & 4) && (!ZL || !a.h2)) sa_layer1<C1, TM, NT>(a, tid, Xz, Src, H1);
        __builtin_amdgcn_sched_barrier(0);
        if (a.h2 && !direct) {
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const int q = tid + i * NT, row = q / (C2 / 4), c4 = q % (C2 / 4);
                *(float4*)(H2 + row * LD2 + 4 * c4) = hq[i];
            }
        }
        if (!direct) {
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int i = tid + u * NT, j = i >> 7, sl = i & 127;
                if (j < td.z) {
                    Val[j * LDV + sl] = dv[u] * (1.0f - pv[u] * pv[u]);
                    ((uint8_t*)ArgW)[j * (LDA * 4) + sl] = (uint8_t)av[u];
                }
            }
        }
        if (!a.h2) {
            __syncthreads();
            sa_layer2<C1, C2, TM, NW, false>(H1, P2v, a.b2, wave, lane, H2);
        }
        // ---- P3: structured layer-3 backward, 128 channels at a time ---------------------------------------
        const int wn2 = wave % M2::NBW, wm2 = wave / M2::NBW;
        f32x16 acc[M2::MB][M2::NB];
        zero_acc<M2::MB, M2::NB>(acc);
        // Z = the tile's dZ3 rows for a window of ZW channels (ZW / 2 from each k half), in H1's buffer (same row stride)
        constexpr int ZW = C1, LDZ = LD1, NSUB = 128 / ZW, NGW = ZW / 8;
        static_assert(NSUB * ZW == 128 && NGW % 2 == 0, "layer-3 gradient window");
        float* Z = H1;
        int lgv[M2::MB], lrv[M2::MB];
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            if (q == 0) {
                if constexpr (!ZL) {
#pragma unroll
                    for (int mb = 0; mb < M2::MB; ++mb) {
                        const int lgr = Lgr[(wm2 * M2::MB + mb) * 32 + li];
                        lgv[mb] = lgr >> 8;
                        lrv[mb] = lgr & 255;
                    }
                }
            } else {
                __syncthreads();             // the previous chunk's Val, Arg are no longer read
                for (int i = tid; i < td.z * 128; i += NT) {
                    const int j = i >> 7, s = i & 127;
                    const int c = (s < 64 ? 64 * q + s : C3 / 2 + 64 * q + (s - 64));
                    const long g = td.y + j;
                    const float p = a.pooled[g * a.ldp + c];
                    Val[j * LDV + s] = a.dpooled[g * a.lddp + c] * (1.0f - p * p);
                    ((uint8_t*)ArgW)[j * (LDA * 4) + s] = (uint8_t)a.arg[g * C3 + c];
                }
            }
            __syncthreads();                 // H1, H2 and this chunk's Val, Arg complete (and nobody reads Z any more)
            auto z_build = [&](int h) {      // window h of this chunk: zero, then ONE store per (group, channel)
                for (int q4 = tid; q4 < TM * ZW / 4; q4 += NT)
                    *(float4*)(Z + (q4 / (ZW / 4)) * LDZ + 4 * (q4 % (ZW / 4))) = make_float4(0.f, 0.f, 0.f, 0.f);
                __syncthreads();
                for (int i = tid; i < td.z * ZW; i += NT) {
                    const int j = i / ZW, sl = i % ZW;
                    const int s128 = sl < ZW / 2 ? (ZW / 2) * h + sl : 64 + (ZW / 2) * h + (sl - ZW / 2);
                    const int row = Ls[j] + ((const uint8_t*)ArgW)[j * (LDA * 4) + s128];
                    Z[row * LDZ + sl] = Val[j * LDV + s128];
                }
            };
            if constexpr (ZL) {
                if (!(SA_PK_ABLATE & 2)) z_build(0);
            }
            {   // dW3[c, :] += val * H2[row of the arg-max, :]   (VALU; the slice stays in registers for the whole kernel)
                const int s = tid & 127, ks = tid >> 7;
                for (int j = 0; j < ((SA_PK_ABLATE & 1) ? 0 : td.z); ++j) {
                    const float v = Val[j * LDV + s];
                    const int row = Ls[j] + ((const uint8_t*)ArgW)[j * (LDA * 4) + s];
                    const float* hrow = H2 + row * LD2 + ks * KS;
#pragma unroll
                    for (int k4 = 0; k4 < KS / 4; ++k4) {
                        const float4 h = *(const float4*)(hrow + 4 * k4);
                        accW3[q][4 * k4] = fmaf(v, h.x, accW3[q][4 * k4]);
                        accW3[q][4 * k4 + 1] = fmaf(v, h.y, accW3[q][4 * k4 + 1]);
                        accW3[q][4 * k4 + 2] = fmaf(v, h.z, accW3[q][4 * k4 + 2]);
                        accW3[q][4 * k4 + 3] = fmaf(v, h.w, accW3[q][4 * k4 + 3]);
                        if ((k4 & 3) == 3) {
                            asm volatile("" ::: "memory");
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    if (ks == 0) accb3[q] += v;
                }
            }
            __syncthreads();                 // a real barrier (register live ranges, see sa_bwd_kernel); Z complete
            if constexpr (ZL) {
            // dH2 += dZ3[:, window] * W3[window, :]: the A operand streams from Z like any LDS tile
#pragma unroll
            for (int h = 0; h < NSUB; ++h) {
                if (h > 0) {
                    __syncthreads();         // the previous window's operand reads are done
                    if (!(SA_PK_ABLATE & 2)) z_build(h);
                    __syncthreads();
                }
                auto zload = [&](int mb, int g) -> float4 {
                    return *(const float4*)(Z + ((wm2 * M2::MB + mb) * 32 + li) * LDZ + lh * (ZW / 2) + g * 4);
                };
                if (!(SA_PK_ABLATE & 2))
                    mfma_stream_fn_win<M2::MB, M2::NB, C3 / 8, NGW>(
                        zload, P3Tv + ((size_t)(wn2 * M2::NB) * (C3 / 8) + 16 * q + NGW * h) * 64 + lane, acc);
            }
            } else {
            // dH2 += dZ3[:, chunk] * W3[chunk, :]: A built per ROW from its own group's entries
            auto asel = [&](int mb, int g) -> float4 {
                const int slot = lh * 64 + g * 4;
                const float4 vv = *(const float4*)(Val + lgv[mb] * LDV + slot);
                const uint32_t ab = ArgW[lgv[mb] * LDA + (slot >> 2)];
                const int lr = lrv[mb];
                float4 o;
                o.x = (int)(ab & 255u) == lr ? vv.x : 0.f;
                o.y = (int)((ab >> 8) & 255u) == lr ? vv.y : 0.f;
                o.z = (int)((ab >> 16) & 255u) == lr ? vv.z : 0.f;
                o.w = (int)(ab >> 24) == lr ? vv.w : 0.f;
                return o;
            };
            if (!(SA_PK_ABLATE & 2))
                mfma_stream_fn_win<M2::MB, M2::NB, C3 / 8, 16>(asel, P3Tv + ((size_t)(wn2 * M2::NB) * (C3 / 8) + 16 * q) * 64 + lane, acc);
            }
        }
        __syncthreads();                     // every wave is done with its dW3 reads of H2 rows (and with Z)
        // ---- P4: dZ2 = dH2 .* (1 - H2^2) -> D, db2 ---------------------------------------------------
#pragma unroll
        for (int nb = 0; nb < M2::NB; ++nb) {
            const int col = (wn2 * M2::NB + nb) * 32 + li;
#pragma unroll
            for (int mb = 0; mb < M2::MB; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (wm2 * M2::MB + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float h = H2[row * LD2 + col];
                    const float z = acc[mb][nb][r] * (1.0f - h * h);
                    D[row * LD2 + col] = z;
                    accb2[nb] += z;
                }
        }
        if constexpr (ZL) {
            if (!(SA_PK_ABLATE & 4)) sa_layer1<C1, TM, NT>(a, tid, Xz, Src, H1); // Z is dead: H1 for the dW2 / dH1 phases
        }
        __syncthreads();
        // ---- P5: dW2 += dZ2^T * H1 ------------------------------------------------------------------
        {
            const float* Ap = D + (lh * (TM / 2)) * LD2 + w2_m * 32 + li;
            const float* Bp = H1 + (lh * (TM / 2)) * LD1 + w2_n0 * 32 + li;
            float ap, bp[NBK], aq, bq[NBK];
#define SA_DW2_LOAD(a_, b_, s_)  \
    a_ = Ap[(s_) * LD2];         \
    _Pragma("unroll") for (int j = 0; j < NBK; ++j) b_[j] = Bp[(s_) * LD1 + j * 32];
#define SA_DW2_MMA(a_, b_) _Pragma("unroll") for (int j = 0; j < NBK; ++j) accW2[j] = MFMA(a_, b_[j], accW2[j]);
            SA_DW2_LOAD(ap, bp, 0)
#pragma unroll 1
            for (int s = 0; s < ((SA_PK_ABLATE & 16) ? 0 : TM / 2); s += 2) {
                SA_DW2_LOAD(aq, bq, s + 1)
                SA_DW2_MMA(ap, bp)
                SA_DW2_LOAD(ap, bp, s + 2)          // last trip reads one row past this half: discarded
                SA_DW2_MMA(aq, bq)
            }
#undef SA_DW2_LOAD
#undef SA_DW2_MMA
        }
        // ---- P6: dH1 = dZ2 * W2 -> dZ1 = dH1 .* (1 - H1^2), in place of H1 ----------------------------
        {
            const int wn = wave % MH::NBW, wm = wave / MH::NBW;
            f32x16 accH[MH::MB][MH::NB];
            zero_acc<MH::MB, MH::NB>(accH);
            if (!(SA_PK_ABLATE & 16))
                mfma_stream<MH::MB, MH::NB, NGT>(D + (wm * MH::MB * 32 + li) * LD2 + lh * (C2 / 2), LD2,
                                                 P2Tv + (size_t)(wn * MH::NB) * NGT * 64 + lane, accH);
            __syncthreads();
#pragma unroll
            for (int nb = 0; nb < MH::NB; ++nb)
#pragma unroll
                for (int mb = 0; mb < MH::MB; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (wm * MH::MB + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const int col = (wn * MH::NB + nb) * 32 + li;
                        const float h = H1[row * LD1 + col];
                        H1[row * LD1 + col] = accH[mb][nb][r] * (1.0f - h * h);
                    }
        }
        __syncthreads();
        // ---- P7: dW1[:, :3], db1, scatter dZ1 to the source points (live rows only) ---------------------
        {
            constexpr int PARTS = NT / C1, RPT = TM / PARTS;
            const int c = tid % C1, p0 = (tid / C1) * RPT;
            const int p1 = (SA_PK_ABLATE & 8) ? p0 : (p0 + RPT < td.w ? p0 + RPT : td.w);
#pragma unroll 4
            for (int p = p0; p < p1; ++p) {
                const float z = H1[p * LD1 + c];
                const float4 x = *(const float4*)(Xz + p * 4);
                accW1[0] = fmaf(z, x.x, accW1[0]);
                accW1[1] = fmaf(z, x.y, accW1[1]);
                accW1[2] = fmaf(z, x.z, accW1[2]);
                accW1[3] += z;
                if (a.dz1) a.dz1[(long)(td.x + p) * C1 + c] = z;
                else if (a.dY) unsafeAtomicAdd(a.dY + (long)Src[p] * C1 + c, z);
            }
        }
        __syncthreads();
        td = tdn;
    }

    // ---- write this work-group's partial sums (layout of SaPart, reduced by sa_bwd_reduce_kernel) ----------------
    float* part = a.parts + (size_t)blockIdx.x * P::N;
    const int tid = tid0, wave = wave0, lh0 = lane0 >> 5, li0 = lane0 & 31;
    const int w2_m = wave % (C2 / 32), w2_n0 = (wave / (C2 / 32)) * NBK;
#pragma unroll
    for (int j = 0; j < NBK; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c2 = w2_m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh0;
            part[P::O_DW2 + c2 * C1 + (w2_n0 + j) * 32 + li0] = accW2[j][r];
        }
    {
        const int s = tid & 127, ks = tid >> 7;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int c = (s < 64 ? 64 * q + s : C3 / 2 + 64 * q + (s - 64));
#pragma unroll
            for (int j = 0; j < KS; ++j) part[P::O_DW3 + c * C2 + ks * KS + j] = accW3[q][j];
            if (ks == 0) part[P::O_DB3 + c] = accb3[q];
        }
    }
    __syncthreads();
    {
        constexpr int PARTS1 = NT / C1;
        float* s2 = H1;                              // [M2::MW][C2]
        float* s1 = H2;                              // [PARTS1][C1][4]
        static_assert(M2::MW * C2 <= TM * LD1 && PARTS1 * C1 * 4 <= TM * LD2, "reduction scratch");
        const int wn = wave % M2::NBW, wm = wave / M2::NBW;
#pragma unroll
        for (int nb = 0; nb < M2::NB; ++nb) {
            const float v = accb2[nb] + __shfl_xor(accb2[nb], 32, 64);
            if (lh0 == 0) s2[wm * C2 + (wn * M2::NB + nb) * 32 + li0] = v;
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) s1[tid * 4 + d] = accW1[d];
        __syncthreads();
        if (tid < C2) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < M2::MW; ++q) s += s2[q * C2 + tid];
            part[P::O_DB2 + tid] = s;
        }
        if (tid < C1 * 4) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < PARTS1; ++q) s += s1[q * C1 * 4 + tid];
            part[P::O_DW1 + tid] = s;
        }
    }
}

extern "C" int pm_sa_bwd_packed_f32(const float* Y, int B, int P, int S, const int32_t* grow, const int32_t* rowmap,
                                    const float* relxyz, const int32_t* tiles, const int32_t* totals,
                                    const float* W1, long ldw1, const float* b1, const float* b2, const float* W3,
                                    const float* packed, int C1, int C2, int C3, const float* pooled, long ldp,
                                    const int32_t* arg, const float* dpooled, long lddp, float* dW1, long lddw1, float* db1,
                                    float* dW2, float* db2, float* dW3, float* db3, float* dY, float* dz1_rows,
                                    int dw1_zero_end, const float* h2_saved, void* workspace, size_t workspace_bytes, void* stream) {
    PM_REQUIRE(grow && rowmap && relxyz && tiles && totals && W1 && b1 && b2 && W3 && packed && pooled && arg && dpooled);
    PM_REQUIRE(dW1 && db1 && dW2 && db2 && dW3 && db3 && workspace);
    PM_REQUIRE(B > 0 && P > 0 && S > 0 && ldw1 >= 3 && lddw1 >= 3 && ldp >= C3 && lddp >= C3 && dw1_zero_end <= lddw1);
    if (!pm_sa_supported(C1, C2, C3, SA_NS)) return PM_EUNSUPPORTED;
    if (((uintptr_t)packed & 15) != 0 || ((uintptr_t)tiles & 15) != 0 || ((uintptr_t)relxyz & 15) != 0) return PM_EALIGN;
    if (workspace_bytes < pm_sa_bwd_workspace_bytes(C1, C2, C3)) return PM_EWORKSPACE;
    SaArgs a = {};
    a.Y = Y; a.W1 = W1; a.ldw1 = ldw1; a.b1 = b1; a.b2 = b2;
    a.packed = packed; a.pooled = const_cast<float*>(pooled); a.ldp = ldp; a.arg = const_cast<int32_t*>(arg);
    a.G = (long)B * S; a.S = S; a.P = P; a.W3 = W3; a.dpooled = dpooled; a.lddp = lddp; a.dY = dY; a.dz1 = dz1_rows;
    a.parts = (float*)workspace;
    a.h2 = const_cast<float*>(h2_saved);
    SaPk k = {grow, (const int2*)rowmap, (const float4*)relxyz, (const int4*)tiles, totals};
    const int ncu = sa_cu_count();
    if (SA_CFG_A(C1, C2, C3)) {
        const long grid = sa_pk_grid(a.G, ncu, SA_A_PK_BWD_WPE, SA_BWD_MAXGRID);
        hipLaunchKernelGGL((sa_bwd_pk_kernel<64, 64, 128, 64, 4, SA_A_PK_BWD_WPE, 20>), dim3((unsigned)grid), dim3(256), 0, pm_stream(stream), a, k);
        constexpr int n = SaPart<64, 64, 128>::N;
        hipLaunchKernelGGL((sa_bwd_reduce_kernel<64, 64, 128>), dim3((n + 63) / 64), dim3(64 * SA_RED_G), 0, pm_stream(stream), a.parts,
                           (int)grid, dW1, lddw1, db1, dW2, db2, dW3, db3, dw1_zero_end);
    } else {
        const long grid = sa_pk_grid(a.G, ncu, 1, SA_BWD_MAXGRID);
        hipLaunchKernelGGL((sa_bwd_pk_kernel<128, 128, 256, 128, SA_B_PK_BWD_NW, SA_B_PK_BWD_NW / 4, 28>), dim3((unsigned)grid),
                           dim3(SA_B_PK_BWD_NW * 64), 0, pm_stream(stream), a, k);
        constexpr int n = SaPart<128, 128, 256>::N;
        hipLaunchKernelGGL((sa_bwd_reduce_kernel<128, 128, 256>), dim3((n + 63) / 64), dim3(64 * SA_RED_G), 0, pm_stream(stream), a.parts,
                           (int)grid, dW1, lddw1, db1, dW2, db2, dW3, db3, dw1_zero_end);
    }
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ============================================================================================================
// The consumer of a level's layer-1 gradient rows (levels with input features: Y = feat * W1f^T per SOURCE point).
//
//   dY[p, :]   = sum over p's packed rows (ascending: the plan's inverse table) of dz1[row, :]      -- never written to HBM
//   dfeat[p,:] = dY[p, :] * W1f                  (the gradient the level below receives as its dpooled)
//   dW1f      += dY^T * feat                     (W1f = columns 3 .. 3+CF of the level's first weight)
//
// ONE kernel instead of zero-fill + atomics (or the segmented-sum pass) + two Linear launches + a slab reduction + a column copy:
// a work-group owns tiles of TM source points; every wave sums the rows of its TM/NW points into an LDS tile in the fixed order
// (bit-identical to pm_sa_dy_segsum_f32: a sequential chain starting from the first row), the feature tile lands beside it, the
// two products run on fp32 MFMA (dfeat: A from LDS, B = W1f streamed from its operand-order copy in L2; dW1f: both operands from
// LDS, accumulators persistent in registers).  Two work-groups per CU.  Measured (round 6, tools/time_sa.py, -DSA_DYC_ABLATE builds):
// 0.42 ms per launch = 0.30 ms with every load / store removed (MFMA floor 0.22) + 0.18 ms with the MFMAs removed, minus 0.05 of
// overlap: VMEM issue slots and MFMAs share a SIMD's issue port, so they add up; 32-point tiles x 4 waves x 4 work-groups per CU,
// 8 rows in flight per wave and half of the work-groups started half a tile late all measure the same (0.415-0.434).
// HBM per point: 1.35 rows of 4*C1 B in + 4*CF B in + 4*CF B out (SA2 of the bench: 0.9 GB per launch for 34.4 GFLOP).
#define SA_DYC_MAXGRID 1024
#ifndef SA_DYC_TM
#define SA_DYC_TM 64          // source points per tile
#endif
#ifndef SA_DYC_NW
#define SA_DYC_NW 8           // waves per work-group
#endif
#ifndef SA_DYC_WPC
#define SA_DYC_WPC 2          // work-groups per CU (persistent grid = CUs x this)
#endif
#ifndef SA_DYC_BATCH
#define SA_DYC_BATCH 4        // dz1 rows a wave has in flight
#endif
#ifndef SA_DYC_ABLATE
#define SA_DYC_ABLATE 0       // timing probes (wrong results): 1 no row loads, 2 no dfeat MFMA, 4 no dW1f MFMA, 8 no dfeat stores, 16 no feature tile
#endif
template <int C1, int CF, int TM, int NW>
__global__ __launch_bounds__(NW * 64, SA_DYC_WPC) void sa_dyc_kernel(const float* __restrict__ dz1, const int32_t* __restrict__ inv_start,
                                                             const int32_t* __restrict__ inv_rows, long npoints,
                                                             const float* __restrict__ feat, long ldf, const float* __restrict__ packedW,
                                                             float* __restrict__ dfeat, long lddf, float* __restrict__ dY, long lddy,
                                                             float* __restrict__ parts) {
    constexpr int NT = NW * 64, LDA = C1 + 4, LDF = CF + 4, PPW = TM / NW, VPL = C1 / 64;
    static_assert(C1 % 64 == 0 && CF % 32 == 0 && TM % NW == 0 && PPW < 63 && (TM * CF / 4) % NT == 0, "consumer tile mapping");
    __shared__ __attribute__((aligned(16))) float smem[TM * (LDA + LDF)];
    float* A = smem;                         // dY tile [TM][C1]
    float* F = smem + TM * LDA;              // feature tile [TM][CF]
    using MD = WaveMap<TM, CF, NW>;          // dfeat output mapping
    constexpr int NG = C1 / 8;
    constexpr int WBLK = (C1 / 32) * (CF / 32), NBK = WBLK / NW;       // dW1f blocks per wave: same c1 block, NBK feature blocks
    static_assert(NBK * NW == WBLK && (CF / 32) % NBK == 0, "dW1f wave mapping");
    constexpr int FPT = TM * CF / 4 / NT;
    f32x16 accW[NBK];
#pragma unroll
    for (int j = 0; j < NBK; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW[j][r] = 0.f;
    const int tid0 = threadIdx.x;
    const long ntiles = (npoints + TM - 1) / TM;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int tid = tid0;
        asm volatile("" : "+v"(tid));                    // see sa_fwd_kernel: recompute, don't hoist
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int li = lane & 31, lh = lane >> 5;
        const long p0 = tile * TM;
        // ---- feature tile: requested first, stored behind the row sums (their index chain needs none of it) ----
        float4 fq[FPT];
#pragma unroll
        for (int i = 0; i < FPT; ++i) {
            const int q = tid + i * NT, row = q / (CF / 4), c4 = q % (CF / 4);
            long p = p0 + row;
            if (p >= npoints) p = npoints - 1;           // ragged last tile: its dY rows are zero, whatever is multiplied
            fq[i] = (SA_DYC_ABLATE & 16) ? make_float4(1.f, 1.f, 1.f, 1.f) : *(const float4*)(feat + p * ldf + 4 * c4);
        }
        // ---- dY tile: this wave's PPW points, rows in ascending order -----------------------------------
        {
            long ps = p0 + (long)wave * PPW + (lane <= PPW ? lane : PPW);
            if (ps > npoints) ps = npoints;
            const int st = inv_start[ps];
            const int s0 = __builtin_amdgcn_readlane(st, 0), e1 = __builtin_amdgcn_readlane(st, PPW);
            int pt = 0, bound = __builtin_amdgcn_readlane(st, 1);
            float acc[VPL];
#pragma unroll
            for (int v = 0; v < VPL; ++v) acc[v] = 0.f;
            float* Arow = A + (wave * PPW) * LDA + lane * VPL;
            for (int j0 = s0; j0 < ((SA_DYC_ABLATE & 1) ? s0 : e1); j0 += 64) {
                const int nrow = e1 - j0 < 64 ? e1 - j0 : 64;
                const int myrow = inv_rows[j0 + (lane < nrow ? lane : 0)];
                for (int j = 0; j < nrow; j += SA_DYC_BATCH) {
                    float v4[SA_DYC_BATCH][VPL];
#pragma unroll
                    for (int k = 0; k < SA_DYC_BATCH; ++k) {
                        const int jj = j + k < nrow ? j + k : nrow - 1;
                        const int r = __builtin_amdgcn_readlane(myrow, jj);
                        const float* src = dz1 + (long)r * C1 + lane * VPL;
#pragma unroll
                        for (int v = 0; v < VPL; ++v) v4[k][v] = src[v];
                    }
#pragma unroll
                    for (int k = 0; k < SA_DYC_BATCH; ++k) {
                        if (j + k < nrow) {
                            while (j0 + j + k >= bound) {            // (uniform) the previous point is complete
#pragma unroll
                                for (int v = 0; v < VPL; ++v) {
                                    Arow[pt * LDA + v] = acc[v];
                                    acc[v] = 0.f;
                                }
                                ++pt;
                                bound = __builtin_amdgcn_readlane(st, pt + 1);
                            }
#pragma unroll
                            for (int v = 0; v < VPL; ++v) acc[v] += v4[k][v];
                        }
                    }
                }
            }
            for (; pt < PPW; ++pt) {                                 // the last point with rows, then the points without any
#pragma unroll
                for (int v = 0; v < VPL; ++v) {
                    Arow[pt * LDA + v] = acc[v];
                    acc[v] = 0.f;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < FPT; ++i) {
            const int q = tid + i * NT, row = q / (CF / 4), c4 = q % (CF / 4);
            *(float4*)(F + row * LDF + 4 * c4) = fq[i];
        }
        __syncthreads();
        if (dY) {                                                    // optional copy of the sums (tests, A/B)
            for (int q = tid; q < TM * C1 / 4; q += NT) {
                const int row = q / (C1 / 4), c4 = q % (C1 / 4);
                if (p0 + row < npoints) *(float4*)(dY + (p0 + row) * lddy + 4 * c4) = *(const float4*)(A + row * LDA + 4 * c4);
            }
        }
        // ---- dfeat = dY * W1f --------------------------------------------------------------------------
        if (dfeat) {
            const int wn = wave % MD::NBW, wm = wave / MD::NBW;
            f32x16 acc[MD::MB][MD::NB];
            zero_acc<MD::MB, MD::NB>(acc);
            if (!(SA_DYC_ABLATE & 2))
                mfma_stream<MD::MB, MD::NB, NG>(A + (wm * MD::MB * 32 + li) * LDA + lh * (C1 / 2), LDA,
                                                (const float4*)packedW + (size_t)(wn * MD::NB) * NG * 64 + lane, acc);
#pragma unroll
            for (int nb = 0; nb < MD::NB; ++nb)
#pragma unroll
                for (int mb = 0; mb < MD::MB; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (wm * MD::MB + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const int col = (wn * MD::NB + nb) * 32 + li;
                        if (p0 + row < ((SA_DYC_ABLATE & 8) ? 0 : npoints)) dfeat[(p0 + row) * lddf + col] = acc[mb][nb][r];
                    }
        }
        // ---- dW1f += dY^T * feat  (K = the tile's TM points; lanes < 32: points [0, TM/2), lanes >= 32: the rest) ----
        {
            const int w_m = wave % (C1 / 32), w_n0 = (wave / (C1 / 32)) * NBK;
            const float* Ap = A + (lh * (TM / 2)) * LDA + w_m * 32 + li;          // A[m = c1][k = point]
            const float* Bp = F + (lh * (TM / 2)) * LDF + w_n0 * 32 + li;         // B[k = point][n = feature]
            float ap, bp[NBK], aq, bq[NBK];
#define SA_DYC_LOAD(a_, b_, s_)  \
    a_ = Ap[(s_) * LDA];         \
    _Pragma("unroll") for (int j = 0; j < NBK; ++j) b_[j] = Bp[(s_) * LDF + j * 32];
#define SA_DYC_MMA(a_, b_) _Pragma("unroll") for (int j = 0; j < NBK; ++j) accW[j] = MFMA(a_, b_[j], accW[j]);
            SA_DYC_LOAD(ap, bp, 0)
#pragma unroll 1
            for (int s = 0; s < ((SA_DYC_ABLATE & 4) ? 0 : TM / 2); s += 2) {
                SA_DYC_LOAD(aq, bq, s + 1)
                SA_DYC_MMA(ap, bp)
                SA_DYC_LOAD(ap, bp, (s + 2 < TM / 2 ? s + 2 : 0))        // last trip: a valid row, discarded
                SA_DYC_MMA(aq, bq)
            }
#undef SA_DYC_LOAD
#undef SA_DYC_MMA
        }
        __syncthreads();                                             // A, F are free for the next tile
    }
    // ---- this work-group's partial dW1f ------------------------------------------------------------------
    float* part = parts + (size_t)blockIdx.x * (C1 * CF);
    const int lane0 = tid0 & 63, wave0 = tid0 >> 6, lh0 = lane0 >> 5, li0 = lane0 & 31;
    const int w_m = wave0 % (C1 / 32), w_n0 = (wave0 / (C1 / 32)) * NBK;
#pragma unroll
    for (int j = 0; j < NBK; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c1 = w_m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh0;
            part[c1 * CF + (w_n0 + j) * 32 + li0] = accW[j][r];
        }
}

// dW1[c1][3 + cf] = sum of the partials in work-group order (8 waves per 64 outputs, as sa_bwd_reduce_kernel); the pad columns
// [3 + CF, lddw1) of every row are zeroed (they never receive data)
__global__ __launch_bounds__(64 * SA_RED_G) void sa_dyc_reduce_kernel(const float* __restrict__ parts, int nparts, int C1, int CF,
                                                                      float* __restrict__ dW1, long lddw1, int pad_end) {
    __shared__ float red[SA_RED_G][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane, n = C1 * CF;
    float s = 0.f;
    if (i < n) {
#pragma unroll 4
        for (int w = wv; w < nparts; w += SA_RED_G) s += parts[(size_t)w * n + i];
    }
    red[wv][lane] = s;
    __syncthreads();
    if (wv != 0 || i >= n) return;
    s = red[0][lane];
#pragma unroll
    for (int q = 1; q < SA_RED_G; ++q) s += red[q][lane];
    const int c1 = i / CF, cf = i % CF;
    dW1[c1 * lddw1 + 3 + cf] = s;
    if (cf == 0)
        for (int c = 3 + CF; c < pad_end; ++c) dW1[c1 * lddw1 + c] = 0.f;
}

__global__ __launch_bounds__(256) void sa_dyc_pack_kernel(const float* __restrict__ W1, long ldw1, int C1, int CF, float* __restrict__ packed) {
    const long n = (long)C1 * CF, total = n + 1024;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    if (i >= n) {
        packed[i] = 0.f;
        return;
    }
    const int e = i & 3, lane = (i >> 2) & 63, li = lane & 31, lh = lane >> 5;
    const int ng = C1 / 8, g = (int)((i >> 8) % ng), nb = (int)((i >> 8) / ng);      // B[k = c1][n = feature] = W1[c1][3 + feature]
    packed[i] = W1[(long)(lh * (C1 / 2) + g * 4 + e) * ldw1 + 3 + nb * 32 + li];
}

extern "C" int pm_sa_dy_consume_supported(int C1, int CF) { return C1 == 128 && CF == 128; }
extern "C" size_t pm_sa_dy_consume_packed_elems(int C1, int CF) { return (size_t)C1 * CF + 1024; }
extern "C" size_t pm_sa_dy_consume_workspace_bytes(int C1, int CF) { return (size_t)C1 * CF * SA_DYC_MAXGRID * sizeof(float); }

extern "C" int pm_sa_dy_consume_pack_f32(const float* W1, long ldw1, int C1, int CF, float* packed, void* stream) {
    PM_REQUIRE(W1 && packed && C1 > 0 && CF > 0 && C1 % 32 == 0 && CF % 32 == 0 && ldw1 >= 3 + CF);
    const long total = (long)pm_sa_dy_consume_packed_elems(C1, CF);
    hipLaunchKernelGGL(sa_dyc_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, pm_stream(stream), W1, ldw1, C1, CF, packed);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

extern "C" int pm_sa_dy_consume_f32(const float* dz1, const int32_t* inv_start, const int32_t* inv_rows, long npoints, int C1, int CF,
                                    const float* feat, long ldf, const float* packedW, float* dfeat, long lddf, float* dW1, long lddw1,
                                    int dw1_cols, float* dY, long lddy, void* workspace, size_t workspace_bytes, void* stream) {
    PM_REQUIRE(dz1 && inv_start && inv_rows && feat && packedW && dW1 && workspace && npoints > 0);
    PM_REQUIRE(ldf >= CF && ldf % 4 == 0 && dw1_cols >= 3 + CF && lddw1 >= dw1_cols && (!dfeat || lddf >= CF) &&
               (!dY || (lddy >= C1 && lddy % 4 == 0)));
    if (!pm_sa_dy_consume_supported(C1, CF)) return PM_EUNSUPPORTED;
    if ((((uintptr_t)dz1) & 15) || (((uintptr_t)feat) & 15) || (((uintptr_t)packedW) & 15) || (((uintptr_t)dY) & 15)) return PM_EALIGN;
    if (workspace_bytes < pm_sa_dy_consume_workspace_bytes(C1, CF)) return PM_EWORKSPACE;
    const long ntiles = (npoints + SA_DYC_TM - 1) / SA_DYC_TM;
    long grid = (long)sa_cu_count() * SA_DYC_WPC;
    if (grid > SA_DYC_MAXGRID) grid = SA_DYC_MAXGRID;
    if (grid > ntiles) grid = ntiles;
    hipLaunchKernelGGL((sa_dyc_kernel<128, 128, SA_DYC_TM, SA_DYC_NW>), dim3((unsigned)grid), dim3(SA_DYC_NW * 64), 0, pm_stream(stream), dz1, inv_start, inv_rows,
                       npoints, feat, ldf, packedW, dfeat, lddf, dY, lddy, (float*)workspace);
    hipLaunchKernelGGL(sa_dyc_reduce_kernel, dim3((unsigned)((C1 * CF + 63) / 64)), dim3(64 * SA_RED_G), 0, pm_stream(stream),
                       (const float*)workspace, (int)grid, C1, CF, dW1, lddw1, dw1_cols);
    PM_CHECK_LAUNCH();
    return PM_OK;
}
