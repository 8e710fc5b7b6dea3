// Grouped, software-pipelined fp32-MFMA GEMM (v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation, bitwise an
// fmaf chain per output element) behind the Linear entry points of gemm_f32.hip.
//
// Why a second GEMM kernel: the Linear layers of the small-step regime are 2048 x 512 x 512 -- 1024 MFMA blocks, one
// per SIMD.  The first kernel (gemm_f32_kernel: 8 waves in lock-step on a 64 x 64 tile, two barriers per K-step, every
// wave staging at the same time and multiplying at the same time) reaches 57 TFLOP/s there, the library (hipBLASLt)
// the same.  This one keeps the matrix pipe of a SIMD fed from ONE wave:
//   * a work-group is 4 waves (2 x 2) -- one per SIMD -- on a (64 WM) x (64 WN) tile, each wave WM x WN blocks of 32 x 32;
//   * K-step 32, LDS double-buffered, ONE barrier per K-step.  A K-step is two half-steps of 8 k-values per lane half;
//     the fragments of a half-step live in two fixed register sets X / Y:
//         [barrier]  read X(k+1) <- LDS[next] | MFMA Y(k)  ||  read Y(k+1) .. | MFMA X(k+1) | staging -> LDS | global loads
//     so every LDS read, the LDS write of tile k+1 and the global loads of tile k+2 are issued between MFMAs of a
//     half-step whose operands are already in registers, and no LDS latency follows the barrier;
//   * global -> register -> LDS staging (T14: issue early, write late); out-of-range k is zero-filled by a select on
//     the loaded value, addresses are clamped instead of predicated (a predicated load makes hipcc branch around it and
//     drain the whole queue);
//   * k-major operands (the weight gradient's dY^T and X, the data gradient's W) stay k-major in LDS [k][row]; with two
//     blocks per wave the tile rows are interleaved (row = 2 i + block) so that one ds_read_b64 feeds both blocks;
//   * several problems per launch (Gemm2Group): block -> (problem, split-K slab, tile), XCD-aware (T1) inside a problem.
// Split-K slabs are written to slab z of C; the caller reduces them in fixed order (or hands them to the optimiser
// kernels, which sum the slabs while they compute the gradient norm: adam.hip).
#include "gemm2.h"
#include <cstdlib>

#ifndef G2_ABLATE
#define G2_ABLATE 0                     // profiling builds only: 1 = no global loads, 2 = no LDS writes in the K loop
#endif
#define G2_TK 32
#define G2_LDK 36                       // k-contiguous LDS row stride (floats): 16 lanes of a ds_read_b128 group hit 16 distinct 4-bank slots
#define G2_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int W>
struct G2Frag {
    float v[W][8];
};
template <int T>
struct G2Stage {
    float s[T / 8];                     // T x 32 floats over 256 threads
};

// (The zero fill of k >= kend is applied when the registers are WRITTEN to LDS, one K-step after the load was issued:
// a select next to the load makes the compiler wait for the load right there.)
// ---- global -> registers --------------------------------------------------------------------------------------------
template <bool KM, int T, bool VEC>
__device__ __forceinline__ void g2_stage_load(const float* __restrict__ P, long ld, int r0, int nrows, int k0, int kend, int Kfull,
                                              G2Stage<T>& st) {
    const int tid = threadIdx.x;
    if (VEC) {
#pragma unroll
        for (int j = 0; j < T / 32; ++j) {
            const int id = tid + 256 * j;
            int row, k;
            if (KM) { k = id / (T / 4); row = (id % (T / 4)) * 4; } else { row = id >> 3; k = (id & 7) * 4; }
            const int gk = k0 + k;
            float4 v;
            if (KM) v = *(const float4*)(P + (long)min(gk, Kfull - 1) * ld + min(r0 + row, nrows - 4));
            else v = *(const float4*)(P + (long)min(r0 + row, nrows - 1) * ld + min(gk, Kfull - 4));
            st.s[4 * j] = v.x; st.s[4 * j + 1] = v.y; st.s[4 * j + 2] = v.z; st.s[4 * j + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < T / 8; ++j) {
            const int id = tid + 256 * j;
            int row, k;
            if (KM) { k = id / T; row = id % T; } else { row = id >> 5; k = id & 31; }
            const int gk = k0 + k;
            const int gr = min(r0 + row, nrows - 1), gkc = min(gk, Kfull - 1);
            st.s[j] = KM ? P[(long)gkc * ld + gr] : P[(long)gr * ld + gkc];
        }
    }
}
// ---- registers -> LDS -----------------------------------------------------------------------------------------------
template <bool KM, int T, bool VEC>
__device__ __forceinline__ void g2_stage_store(float* __restrict__ S, const G2Stage<T>& st, int k0, int kend) {
    const int tid = threadIdx.x;
    if (VEC) {
#pragma unroll
        for (int j = 0; j < T / 32; ++j) {
            const int id = tid + 256 * j;
            const bool ok = k0 + (KM ? id / (T / 4) : (id & 7) * 4) < kend;
            const float4 v = make_float4(ok ? st.s[4 * j] : 0.f, ok ? st.s[4 * j + 1] : 0.f, ok ? st.s[4 * j + 2] : 0.f,
                                         ok ? st.s[4 * j + 3] : 0.f);
            if (KM) *(float4*)(S + (id / (T / 4)) * T + (id % (T / 4)) * 4) = v;
            else *(float4*)(S + (id >> 3) * G2_LDK + (id & 7) * 4) = v;
        }
    } else {
#pragma unroll
        for (int j = 0; j < T / 8; ++j) {
            const int id = tid + 256 * j;
            const float v = (k0 + (KM ? id / T : (id & 31)) < kend) ? st.s[j] : 0.f;
            if (KM) S[(id / T) * T + (id % T)] = v;
            else S[(id >> 5) * G2_LDK + (id & 31)] = v;
        }
    }
}
// ---- LDS -> fragments of half-step h: lane (li, lh) holds k = 16 h + 8 lh + s, s = 0..7, of its row in each block ----
template <bool KM, int W, int T>
__device__ __forceinline__ void g2_read_frag(const float* __restrict__ S, int woff, int li, int lh, int h, G2Frag<W>& f) {
    if (KM) {
        const float* p = S + (h * 16 + lh * 8) * T + woff + li * W;      // rows interleaved: tile row = woff + W * i + block
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (W == 1) {
                f.v[0][s] = p[s * T];
            } else {
                const float2 v = *(const float2*)(p + s * T);
                f.v[0][s] = v.x;
                f.v[W - 1][s] = v.y;
            }
        }
    } else {
#pragma unroll
        for (int b = 0; b < W; ++b) {
            const float* p = S + (woff + b * 32 + li) * G2_LDK + h * 16 + lh * 8;
            const float4 v0 = *(const float4*)p, v1 = *(const float4*)(p + 4);
            f.v[b][0] = v0.x; f.v[b][1] = v0.y; f.v[b][2] = v0.z; f.v[b][3] = v0.w;
            f.v[b][4] = v1.x; f.v[b][5] = v1.y; f.v[b][6] = v1.z; f.v[b][7] = v1.w;
        }
    }
}
template <int WM, int WN, int S0, int S1>
__device__ __forceinline__ void g2_mfma_part(const G2Frag<WM>& a, const G2Frag<WN>& b, f32x16 (&acc)[WM][WN]) {
#pragma unroll
    for (int s = S0; s < S1; ++s)
#pragma unroll
        for (int bn = 0; bn < WN; ++bn)
#pragma unroll
            for (int bm = 0; bm < WM; ++bm) acc[bm][bn] = G2_MFMA(a.v[bm][s], b.v[bn][s], acc[bm][bn]);
}
template <int WM, int WN>
__device__ __forceinline__ void g2_mfma_half(const G2Frag<WM>& a, const G2Frag<WN>& b, f32x16 (&acc)[WM][WN]) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int bn = 0; bn < WN; ++bn)
#pragma unroll
            for (int bm = 0; bm < WM; ++bm) acc[bm][bn] = G2_MFMA(a.v[bm][s], b.v[bn][s], acc[bm][bn]);
}

// ---- epilogue (shared by the register-staged and the LDS-DMA kernel) ------------------------------------------------
typedef unsigned int g2_u32x4 __attribute__((ext_vector_type(4)));
// SC1: the tile leaves as write-through (sc1) 16-byte stores -- another work-group of the same launch reads it next
template <bool A_KM, bool B_KM, int WM, int WN, int NT, int TM = 64 * WM, int TN = 64 * WN, bool SC1 = false>
__device__ __forceinline__ void g2_epilogue(const Gemm2Prob& g, f32x16 (&acc)[WM][WN], float* __restrict__ lds, int m0, int n0, int z,
                                            int wmo, int wno, int li, int lh, bool writer) {
    const int tid = threadIdx.x;
    // ---- epilogue: acc[bm][bn][r] is block row i = (r&3) + 8 (r>>2) + 4 lh, block column li.  Written straight from the
    // accumulators a wave stores 16 x (2 rows x 128 B) per block -- store-issue-bound: 8.7 k cycles for a 64 x 64 tile, as long
    // as half its K loop.  The tile goes through LDS instead (the operand buffers are free now) and leaves as 16-byte stores of
    // whole rows; bias / activation / (1 - h^2) are applied on the way out, four columns at a time.
    constexpr int LDC = TN + 4;
    __syncthreads();                                                       // every wave is done with the operand buffers
    if (writer)
#pragma unroll
    for (int bn = 0; bn < WN; ++bn)
#pragma unroll
        for (int bm = 0; bm < WM; ++bm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
                lds[(wmo + (A_KM ? i * WM + bm : bm * 32 + i)) * LDC + wno + (B_KM ? li * WN + bn : bn * 32 + li)] = acc[bm][bn][r];
            }
    __syncthreads();
    float* C = g.C + (long)z * g.slab;
    const int epi = g.epi, act = g.act;
    if (g.vecC) {
#pragma unroll
        for (int j = 0; j < TM * TN / 4 / NT; ++j) {
            const int id = tid + NT * j, row = id / (TN / 4), c4 = (id % (TN / 4)) * 4;
            const int grow = m0 + row, gcol = n0 + c4;
            if (grow < g.M && gcol < g.N) {                                // N % 4 == 0: a float4 is inside or outside as a whole
                long coff = (long)grow * g.ldc + gcol, hoff = (long)grow * g.ldh + gcol;
                if (g.sidx) {                                              // scattered rows (sC % 4 == 0: a float4 stays inside one)
                    const int t = gcol / g.sC;
                    const int dst = g.sidx[(long)grow * g.sJ + t];
                    if (dst < 0) continue;
                    coff = hoff = (long)dst * g.sC + (gcol - t * g.sC);
                }
                float4 v = *(const float4*)(lds + row * LDC + c4);
                if (epi == G2_EPI_BIAS_ACT) {
                    if (g.bias) {
                        const float4 bb = *(const float4*)(g.bias + gcol);
                        v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
                    }
                    if (act == PM_ACT_TANH) { v.x = pm_tanh(v.x); v.y = pm_tanh(v.y); v.z = pm_tanh(v.z); v.w = pm_tanh(v.w); }
                    else if (act != PM_ACT_NONE) { v.x = pm_act(v.x, act); v.y = pm_act(v.y, act); v.z = pm_act(v.z, act); v.w = pm_act(v.w, act); }
                } else if (epi == G2_EPI_MUL_DACT && act != PM_ACT_NONE) {
                    const float4 hh = *(const float4*)(g.H + hoff);
                    if (act == PM_ACT_TANH) {
                        v.x *= 1.0f - hh.x * hh.x; v.y *= 1.0f - hh.y * hh.y; v.z *= 1.0f - hh.z * hh.z; v.w *= 1.0f - hh.w * hh.w;
                    } else {
                        v.x *= pm_dact(hh.x, act); v.y *= pm_dact(hh.y, act); v.z *= pm_dact(hh.z, act); v.w *= pm_dact(hh.w, act);
                    }
                }
                if constexpr (SC1) {
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(C, 0, 0x7fffffff, 0x00020000);
                    g2_u32x4 d;
                    d.x = __float_as_uint(v.x); d.y = __float_as_uint(v.y); d.z = __float_as_uint(v.z); d.w = __float_as_uint(v.w);
                    __builtin_amdgcn_raw_buffer_store_b128(d, rs, (int)(coff * 4), 0, 16);
                } else {
                    *(float4*)(C + coff) = v;
                }
            }
        }
    } else {
#pragma unroll 4
        for (int j = 0; j < TM * TN / NT; ++j) {
            const int id = tid + NT * j, row = id / TN, col = id % TN;
            const int grow = m0 + row, gcol = n0 + col;
            if (grow < g.M && gcol < g.N) {
                float v = lds[row * LDC + col];
                if (epi == G2_EPI_BIAS_ACT) {
                    if (g.bias) v += g.bias[gcol];
                    v = pm_act(v, act);
                } else if (epi == G2_EPI_MUL_DACT && act != PM_ACT_NONE) {
                    v *= pm_dact(g.H[(long)grow * g.ldh + gcol], act);
                }
                C[(long)grow * g.ldc + gcol] = v;
            }
        }
    }
}

// T1 (pm_xcd_contiguous, common.h): an XCD owns a contiguous range of a problem's tiles -- for ANY tile count (round 5: with the
// n % 8 == 0 condition this had, a 756-block weight gradient ran un-swizzled and the seven M-tiles of a split-K slab fetched their
// dY slab, index rows and source rows on seven XCDs: 4.9 -> 2.0 GB per launch); a problem's block0 is a multiple of 8
__device__ __forceinline__ int g2_xcd_local(int local, int n) { return pm_xcd_contiguous(local, n); }

template <bool A_KM, bool B_KM, int WM, int WN, bool VEC>
__global__ __launch_bounds__(256) void gemm2_kernel(Gemm2Group gg) {
    constexpr int TM = 64 * WM, TN = 64 * WN;
    constexpr int ABUF = TM * G2_LDK, BBUF = TN * G2_LDK;              // >= 32 * T floats of the k-major layout
    __shared__ __attribute__((aligned(16))) float lds[2 * ABUF + 2 * BBUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;

#ifdef G2_PROFILE
    const unsigned long long t_start = __builtin_readcyclecounter();
#define G2_STAMP(i) if (gg.prof && tid == 0) gg.prof[(size_t)blockIdx.x * 4 + (i)] = __builtin_readcyclecounter()
    if (gg.prof && tid == 0) gg.prof[(size_t)blockIdx.x * 4] = t_start;
#define G2_KSTAMP(i) if (gg.prof && tid == 0 && k == 4) gg.prof[(size_t)(65536 + blockIdx.x) * 4 * 2 + (i) - 4] = __builtin_readcyclecounter()
#else
#define G2_STAMP(i)
#define G2_KSTAMP(i)
#endif
    // ---- block -> (problem, slab, tile) ----
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GEMM2_MAXP; ++i)
        if (i < gg.n && (int)blockIdx.x >= gg.p[i].block0) pi = i;
    const Gemm2Prob& g = gg.p[pi];
    const int tiles = g.tiles_m * g.tiles_n, nblk = tiles * g.splits;
    int local = (int)blockIdx.x - g.block0;
    if (local >= nblk) return;                                              // padding block behind a problem (launcher)
    local = g2_xcd_local(local, nblk);                                      // T1: an XCD (blocks = x mod 8) owns contiguous tiles
    const int z = local / tiles, t = local - z * tiles;
    const int tm = t / g.tiles_n, tn = t - tm * g.tiles_n;
    const int m0 = tm * TM, n0 = tn * TN;
    const int kbeg = z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const int nk = (kend - kbeg + G2_TK - 1) / G2_TK;
    const int wmo = (wave >> 1) * 32 * WM, wno = (wave & 1) * 32 * WN;     // this wave's first tile row / column

    f32x16 acc[WM][WN];
#pragma unroll
    for (int bm = 0; bm < WM; ++bm)
#pragma unroll
        for (int bn = 0; bn < WN; ++bn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[bm][bn][r] = 0.f;

    const bool do_bias = A_KM && g.epi == G2_EPI_PLAIN && g.dbias != nullptr && tn == 0;
    constexpr int BQ = 256 / TM, BK = G2_TK / BQ;                          // bias sums: k-groups per step, k-values per thread
    float bsum = 0.f;

    float* As = lds;
    float* Bs = lds + 2 * ABUF;
    G2Stage<TM> sa0, sa1;                                                  // staging ring, two tiles in flight: a load has two
    G2Stage<TN> sb0, sb1;                                                  // K-steps (~2000 cycles) to land before its LDS write
    G2Frag<WM> xa, ya;
    G2Frag<WN> xb, yb;
    // tile index -> first k of the global load (clamped to the last tile: a redundant reload is cheaper than a conditional
    // load) and of the zero fill (NOT clamped: a tile behind the last one is written as zeros, so the spare iteration of
    // the two-fold unrolled loop multiplies by zero)
    auto kload = [&](int tile) { return kbeg + min(tile, nk - 1) * G2_TK; };
    auto kfill = [&](int tile) { return kbeg + tile * G2_TK; };

    g2_stage_load<A_KM, TM, VEC>(g.A, g.lda, m0, A_KM ? g.Mld : g.M, kbeg, kend, g.K, sa0);
    g2_stage_load<B_KM, TN, VEC>(g.B, g.ldb, n0, B_KM ? g.Nld : g.N, kbeg, kend, g.K, sb0);
    g2_stage_store<A_KM, TM, VEC>(As, sa0, kbeg, kend);
    g2_stage_store<B_KM, TN, VEC>(Bs, sb0, kbeg, kend);
    g2_stage_load<A_KM, TM, VEC>(g.A, g.lda, m0, A_KM ? g.Mld : g.M, kload(1), kend, g.K, sa0);
    g2_stage_load<B_KM, TN, VEC>(g.B, g.ldb, n0, B_KM ? g.Nld : g.N, kload(1), kend, g.K, sb0);
    g2_stage_load<A_KM, TM, VEC>(g.A, g.lda, m0, A_KM ? g.Mld : g.M, kload(2), kend, g.K, sa1);
    g2_stage_load<B_KM, TN, VEC>(g.B, g.ldb, n0, B_KM ? g.Nld : g.N, kload(2), kend, g.K, sb1);
    __syncthreads();
    g2_read_frag<A_KM, WM, TM>(As, wmo, li, lh, 0, xa);
    g2_read_frag<B_KM, WN, TN>(Bs, wno, li, lh, 0, xb);

    G2_STAMP(1);
    // one K-step: tile k sits in LDS buffer c, X holds its half 0; (sa, sb) hold tile k+1 (loaded two K-steps ago)
    auto kstep = [&](int k, int c, G2Stage<TM>& sa, G2Stage<TN>& sb) __attribute__((always_inline)) {
        const float* Ac = As + c * ABUF;
        const float* Bc = Bs + c * BBUF;
        g2_read_frag<A_KM, WM, TM>(Ac, wmo, li, lh, 1, ya);                // half 1 -> Y while half 0 (X) multiplies
        g2_read_frag<B_KM, WN, TN>(Bc, wno, li, lh, 1, yb);
        if (do_bias) {
            const float* col = Ac + (tid / TM) * BK * TM + (tid % TM);
#pragma unroll
            for (int s = 0; s < BK; ++s) bsum += col[s * TM];
        }
        __builtin_amdgcn_sched_barrier(0);
        G2_KSTAMP(4);
        g2_mfma_half<WM, WN>(xa, xb, acc);
        G2_KSTAMP(5);
        // tile k+1: staging registers -> the other LDS buffer (its last readers finished before the previous barrier),
        // then the global loads of tile k+3 into the same registers
#if !(G2_ABLATE & 2)
        g2_stage_store<A_KM, TM, VEC>(As + (c ^ 1) * ABUF, sa, kfill(k + 1), kend);
        g2_stage_store<B_KM, TN, VEC>(Bs + (c ^ 1) * BBUF, sb, kfill(k + 1), kend);
#endif
#if !(G2_ABLATE & 1)
        g2_stage_load<A_KM, TM, VEC>(g.A, g.lda, m0, A_KM ? g.Mld : g.M, kload(k + 3), kend, g.K, sa);
        g2_stage_load<B_KM, TN, VEC>(g.B, g.ldb, n0, B_KM ? g.Nld : g.N, kload(k + 3), kend, g.K, sb);
#endif
        G2_KSTAMP(6);
        __syncthreads();
        G2_KSTAMP(7);
        g2_read_frag<A_KM, WM, TM>(As + (c ^ 1) * ABUF, wmo, li, lh, 0, xa);    // half 0 of tile k+1 -> X while Y multiplies
        g2_read_frag<B_KM, WN, TN>(Bs + (c ^ 1) * BBUF, wno, li, lh, 0, xb);
        __builtin_amdgcn_sched_barrier(0);
        g2_mfma_half<WM, WN>(ya, yb, acc);
        G2_KSTAMP(8);
    };
#pragma unroll 1
    for (int k = 0; k < nk; k += 2) {
        kstep(k, 0, sa0, sb0);
        kstep(k + 1, 1, sa1, sb1);
    }

    G2_STAMP(2);
    if (do_bias) {                                                         // fixed-order sum of the BQ k-group partials
        __syncthreads();
        lds[tid] = bsum;
        __syncthreads();
        if (tid < TM && m0 + tid < g.M) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < BQ; ++q) s += lds[q * TM + tid];
            g.dbias[(long)z * g.bslab + m0 + tid] = s;
        }
    }

    g2_epilogue<A_KM, B_KM, WM, WN, 256>(g, acc, lds, m0, n0, z, wmo, wno, li, lh, true);
    G2_STAMP(3);
}

// =====================================================================================================================
// LDS-DMA variant (16-byte-loadable operands, K a multiple of 32).  Profiling the register-staged kernel above at
// 2048 x 512 x 512 (tools/g2_profile.py: s_memtime stamps inside one K-step) showed where its time goes: MFMA 2 x 540
// cycles, barrier 100, and ~420 cycles in which the wave sits in its four ds_write_b128 (the LDS store path is shared by
// the SIMD pairs and every wave of the CU stores at the same moment) -- without the LDS writes the K loop drops from
// 10.9 to 7.5 us.  `global_load_lds_dwordx4` (one wave instruction = 1 KiB, LDS image lane-linear) has no store pass
// and no staging registers:
//   * k-contiguous operand: LDS rows of 32 floats (128 B, 8 chunks of 16 B), chunk c of row r holds global chunk
//     c ^ ((r >> 1) & 7) -- the swizzle is applied to the per-lane SOURCE address; a ds_read_b128 lane group (16 rows,
//     one logical chunk) then covers all 16 four-bank slots;
//   * k-major operand: LDS [k][T], lane-linear as it is, read with ds_read_b32 / b64 of consecutive lanes;
//   * NBUF LDS buffers, tile k+NBUF-1 is issued at the top of K-step k, `s_waitcnt vmcnt((NBUF-2) * IPT)` + a raw
//     s_barrier before tile k+1 is first read (counted wait: the youngest tile stays in flight across the barrier --
//     `__syncthreads()` would drain it).
template <bool KM, int W, int T>
__device__ __forceinline__ void g2_read_frag_dma(const float* __restrict__ S, int woff, int li, int lh, int h, G2Frag<W>& f) {
    if (KM) {
        g2_read_frag<true, W, T>(S, woff, li, lh, h, f);                   // [k][T] image: same as the register-staged layout
    } else {
#pragma unroll
        for (int b = 0; b < W; ++b) {
            const int row = woff + b * 32 + li, sw = (row >> 1) & 7, q = h * 4 + lh * 2;
            const float4 v0 = *(const float4*)(S + row * 32 + ((q ^ sw) << 2));
            const float4 v1 = *(const float4*)(S + row * 32 + (((q + 1) ^ sw) << 2));
            f.v[b][0] = v0.x; f.v[b][1] = v0.y; f.v[b][2] = v0.z; f.v[b][3] = v0.w;
            f.v[b][4] = v1.x; f.v[b][5] = v1.y; f.v[b][6] = v1.z; f.v[b][7] = v1.w;
        }
    }
}
typedef __attribute__((address_space(3))) void* g2_lds_ptr;
// one operand tile (T rows x 32 k) global -> LDS: T / 32 wave instructions per wave
// AUX: cache policy of the loads (0 = default; 16 = sc1: an operand another work-group of the SAME launch has just written
// with sc1 stores -- gemm2_chain_kernel -- must not come out of this CU's L1)
template <bool KM, int T, int AUX = 0>
__device__ __forceinline__ void g2_dma_tile(const float* __restrict__ P, long ld, int r0, int nrows, int k0, float* __restrict__ S,
                                            int wave, int lane, int kmax = 0x7fffffff) {
#pragma unroll
    for (int j = 0; j < T / 32; ++j) {
        const int piece = j * 4 + wave;                                    // 1 KiB piece of the tile image
        const float* src;
        if (KM) {
            const int k = piece * (256 / T) + lane / (T / 4), c = lane % (T / 4);
            src = P + (long)min(k0 + k, kmax) * ld + min(r0 + c * 4, nrows - 4);      // kmax: ragged reduction tail (gathered B is 0 there)
        } else {
            const int row = piece * 8 + (lane >> 3), c = lane & 7;
            src = P + (long)min(r0 + row, nrows - 1) * ld + k0 + ((c ^ ((row >> 1) & 7)) << 2);
        }
        __builtin_amdgcn_global_load_lds(src, (g2_lds_ptr)(S + piece * 256), 16, 0, AUX);
    }
}

// (forward orientation only: there the column moves with the K-step; in the weight-gradient orientation it is loop-invariant,
// the compiler hoists the divisions, and the shift form measured 7 % SLOWER -- 5.60 -> 6.01 ms at 2.7 M rows x 1728 x 64)
__device__ __forceinline__ int g2_tap(const Gemm2Prob& g, int q) { return g.gsh >= 0 ? q >> g.gsh : q / g.gC; }
__device__ __forceinline__ int g2_chan(const Gemm2Prob& g, int q) { return g.gsh >= 0 ? q & (g.gC - 1) : q % g.gC; }
// The same for a GATHERED operand (see Gemm2Prob::gidx).  The neighbour indices of a tile are ordinary loads; VMEM returns in
// order, so waiting for them would also drain every LDS-DMA issued before -- they are therefore fetched one K-step ahead
// (`gi`: this tile's indices, loaded while the previous tile was issued; refilled here for the next tile `k0n`).
// A neighbour index fetched by an ordinary load the COMPILER DOES NOT SEE (inline asm): hipcc cannot count a loop-carried VMEM
// result and guarded the index registers with `s_waitcnt vmcnt(0)` at the top of the loader loop -- the ring drained to one tile
// there.  The hardware still counts the load (in order with the DMAs), so the ring's counted waits cover it: an index requested by
// issue(t) has landed once issue(t + 1)'s hand-over wait has passed, and issue(t + 2) is its first reader.  The ISA must not copy
// the register between the load and that wait: the wait itself carries the registers as in/out operands (g2_wait_idx below).
__device__ __forceinline__ int g2_idx_load(const int32_t* p) {
    int v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// The counted wait that covers an index set also DEFINES its registers ("+v"): every later use of gi[] depends on this
// statement's output, so hipcc cannot hoist the address arithmetic of the next issue() -- or a copy / spill of the register --
// above the wait and read an index that is still in flight (before this the only safeguard was an ISA inspection).
template <int CNT, int N>
__device__ __forceinline__ void g2_wait_idx(int (&gi)[N]) {
    static_assert(N == 1 || N == 2 || N == 4, "index registers per loader wave");
    if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(gi[0]) : "n"(CNT) : "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(gi[0]), "+v"(gi[1]) : "n"(CNT) : "memory");
    else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(gi[0]), "+v"(gi[1]), "+v"(gi[2]), "+v"(gi[3]) : "n"(CNT) : "memory");
}
template <bool KM, int T>
__device__ __forceinline__ void g2_dma_tile_gather(const Gemm2Prob& g, const float* __restrict__ P, long ld, int r0, int nrows, int k0,
                                                   int k0n, float* __restrict__ S, int wave, int lane, int (&gi)[T / 32]) {
#pragma unroll
    for (int j = 0; j < T / 32; ++j) {
        const int piece = j * 4 + wave;
        int q, qn, rr;                                                     // column inside the virtual operand, its row
        if (KM) {                                                          // B of a weight gradient: rows = reduction index
            const int k = piece * (256 / T) + lane / (T / 4), c = lane % (T / 4);
            q = qn = min(r0 + c * 4, nrows - 4);
            rr = k0 + k;
            // rows behind the reduction's end read zeros -- decided HERE, from the row number, not by skipping the index load:
            // a conditional load is compiled as a branch around it, and a wave whose pieces all lie behind the end then has
            // FEWER VMEM operations in flight than the counted `s_waitcnt vmcnt(IPT + NGI)` of the ring assumes -- the wait
            // let the last DMAs of the tile before pass unfinished, and the multiplying waves read a stale buffer (round 4's
            // "down1 weight gradient once 5.5e-5 off": profiles/round5_sparse_unet_race.md).  Every issue() now performs
            // exactly IPT + NGI operations, whatever the tile.
            const float* src = (gi[j] >= 0 && rr < g.K) ? P + (long)gi[j] * ld + (q % g.gC) : g.gzero + (q % g.gC);
            __builtin_amdgcn_global_load_lds(src, (g2_lds_ptr)(S + piece * 256), 16, 0, 0);
            gi[j] = g2_idx_load(g.gidx + (long)min(k0n + k, g.K - 1) * g.gJ + qn / g.gC);
        } else {                                                           // A of a forward: rows = output rows
            const int row = piece * 8 + (lane >> 3), c = lane & 7;
            rr = min(r0 + row, nrows - 1);
            q = k0 + ((c ^ ((row >> 1) & 7)) << 2);
            qn = k0n + ((c ^ ((row >> 1) & 7)) << 2);
            const float* src = gi[j] >= 0 ? P + (long)gi[j] * ld + g2_chan(g, q) : g.gzero + g2_chan(g, q);
            __builtin_amdgcn_global_load_lds(src, (g2_lds_ptr)(S + piece * 256), 16, 0, 0);
            gi[j] = g2_idx_load(g.gidx + (long)rr * g.gJ + g2_tap(g, qn));
        }
    }
}
template <bool KM, int T>
__device__ __forceinline__ void g2_gather_first(const Gemm2Prob& g, int r0, int nrows, int k0, int wave, int lane, int (&gi)[T / 32]) {
#pragma unroll
    for (int j = 0; j < T / 32; ++j) {
        const int piece = j * 4 + wave;
        if (KM) {
            const int k = piece * (256 / T) + lane / (T / 4), c = lane % (T / 4);
            gi[j] = g2_idx_load(g.gidx + (long)min(k0 + k, g.K - 1) * g.gJ + min(r0 + c * 4, nrows - 4) / g.gC);      // (unconditional: see g2_dma_tile_gather)
        } else {
            const int row = piece * 8 + (lane >> 3), c = lane & 7;
            gi[j] = g2_idx_load(g.gidx + (long)min(r0 + row, nrows - 1) * g.gJ + g2_tap(g, k0 + ((c ^ ((row >> 1) & 7)) << 2)));
        }
    }
}

// LAY: how the four multiplying waves cover the work-group tile -- 0: 2 x 2 (tile 64 WM x 64 WN); 2: 4 x 1 (tile 128 WM x 32 WN)
// for forward problems with at most 32 OUTPUT columns (the 32-channel level of the sparse U-Net: 8.4 M rows -- on a 64-wide
// tile half of every MFMA multiplies columns that do not exist); 1: 1 x 4 (tile 32 WM x 128 WN)
// for problems with at most 32 rows (the weight gradient of a 32-channel convolution over millions of rows: on the 2 x 2 layout
// half of every MFMA multiplies rows that do not exist -- 6.5 ms per launch at 142 TFLOP/s of ISSUED work, 71 of useful).
// NB: LDS stages of the operand ring.  3 keeps two tiles in flight; 2 (one in flight, requested a whole K-step = 4096 MFMA
// cycles ahead) shrinks the 128 x 128 tile from 98 KB to its 68 KB epilogue image: TWO work-groups per CU, i.e. a second
// multiplying wave per SIMD that runs while the first sits at a barrier or waits for a fragment.
template <int WM, int WN, int LAY, int NB = 3>
struct G2DmaShape {
    static constexpr int TM = LAY == 2 ? 128 * WM : (LAY ? 32 * WM : 64 * WM), TN = LAY == 2 ? 32 * WN : (LAY ? 128 * WN : 64 * WN);
    static constexpr int NBUF = NB;
    static constexpr int ABUF = TM * 32, BBUF = TN * 32;
    static constexpr int EPI = TM * (TN + 4);                              // the epilogue's staging image
    static constexpr int LDSF = NBUF * (ABUF + BBUF) > EPI ? NBUF * (ABUF + BBUF) : EPI;
};

// One work-group tile (tm, tn, slab z) of problem g: the whole K loop + epilogue.  AUXA: cache policy of the A-operand loads,
// SC1C: write-through result stores (both for tiles that hand data to / take data from other work-groups of the same launch).
template <bool A_KM, bool B_KM, int WM, int WN, bool GATHER, int LAY, int AUXA, bool SC1C, int NB = 3>
__device__ __forceinline__ void g2_dma_body(const Gemm2Group& gg, const Gemm2Prob& g, int tm, int tn, int z, float* __restrict__ lds) {
    // 8 waves: waves 0-3 multiply (one per SIMD), waves 4-7 only feed the LDS ring.  Issuing one
    // 1 KiB LDS-DMA costs the issuing wave ~90 cycles (measured: 4 of them in front of a half-step's MFMAs stretched it from
    // 512 to 885 cycles) and an in-order wave cannot issue MFMAs meanwhile -- a second wave on the SIMD can.
    using SH = G2DmaShape<WM, WN, LAY, NB>;
    constexpr int TM = SH::TM, TN = SH::TN, NBUF = SH::NBUF, ABUF = SH::ABUF, BBUF = SH::BBUF;
    static_assert(NB == 3 || NB == 2, "two or three LDS stages");
    constexpr int IPT = (TM + TN) / 32;                                    // DMA instructions per tile per loader wave
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3;
    const bool loader = tid >= 256;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = tm * TM, n0 = tn * TN;
    const int kbeg = z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const int nk = (kend - kbeg + G2_TK - 1) / G2_TK;                      // whole steps, except the ragged row tail of a gathered weight gradient
    const int wmo = LAY == 2 ? wave * 32 * WM : (LAY ? 0 : (wave >> 1) * 32 * WM);
    const int wno = LAY == 2 ? 0 : (LAY ? wave * 32 * WN : (wave & 1) * 32 * WN);

    f32x16 acc[WM][WN];
#pragma unroll
    for (int bm = 0; bm < WM; ++bm)
#pragma unroll
        for (int bn = 0; bn < WN; ++bn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[bm][bn][r] = 0.f;
    const bool do_bias = A_KM && g.epi == G2_EPI_PLAIN && g.dbias != nullptr && tn == 0;
    constexpr int BQ = 256 / TM > 0 ? 256 / TM : 1, BK = G2_TK / BQ;
    float bsum = 0.f;
    float* As = lds;
    float* Bs = lds + NBUF * ABUF;

    if (loader) {
        // tile index clamped: the redundant reloads of the last tile keep the outstanding-load count uniform (they land in
        // buffers nobody reads any more)
        constexpr int NGI = GATHER ? (A_KM ? TN : TM) / 32 : 1;
        // GATHER: the neighbour indices of a tile are ordinary loads and VMEM returns in order, so the set a tile's issue()
        // consumes must be OLDER than the previous tile's DMAs -- otherwise waiting for it also waits for those DMAs and only
        // one tile is ever in flight (the first version: indices one tile ahead; the big sparse convolutions ran at 0.58 of
        // the MFMA peak where the plain kernel reaches 0.65).  Two register sets, each refilled TWO tiles ahead by the issue()
        // that has just consumed it: set A serves the even tiles, set B the odd ones.
        int gia[NGI], gib[NGI];
        auto issue = [&](int tile, int (&gi)[NGI]) __attribute__((always_inline)) {
            const int tt = min(tile, nk - 1), buf = tile % NBUF;
            const int kt = kbeg + tt * G2_TK, ktn = kbeg + min(tile + (NB == 2 ? 1 : 2), nk - 1) * G2_TK;   // (two stages: ONE index set, refilled a tile ahead)
            if constexpr (GATHER && !A_KM) g2_dma_tile_gather<false, TM>(g, g.A, g.lda, m0, g.M, kt, ktn, As + buf * ABUF, wave, lane, gi);
            else g2_dma_tile<A_KM, TM, AUXA>(g.A, g.lda, m0, A_KM ? g.Mld : g.M, kt, As + buf * ABUF, wave, lane, g.K - 1);
            if constexpr (GATHER && A_KM) g2_dma_tile_gather<true, TN>(g, g.B, g.ldb, n0, g.N, kt, ktn, Bs + buf * BBUF, wave, lane, gi);
            else g2_dma_tile<B_KM, TN>(g.B, g.ldb, n0, B_KM ? g.Nld : g.N, kt, Bs + buf * BBUF, wave, lane);
        };
        if constexpr (GATHER) {
            const int k1 = kbeg + min(1, nk - 1) * G2_TK;
            if constexpr (A_KM) {
                g2_gather_first<true, TN>(g, n0, g.N, kbeg, wave, lane, gia);
                g2_gather_first<true, TN>(g, n0, g.N, k1, wave, lane, gib);
            } else {
                g2_gather_first<false, TM>(g, m0, g.M, kbeg, wave, lane, gia);
                g2_gather_first<false, TM>(g, m0, g.M, k1, wave, lane, gib);
            }
            g2_wait_idx<0>(gia);                                           // the compiler does not know these loads: the first two tiles' indices are in
            g2_wait_idx<0>(gib);
        }
        // the counted waits allow for the NGI index loads an issue() ends with
        constexpr int NW1 = IPT + (GATHER ? NGI : 0);
        if constexpr (NB == 2) {
            // two stages: tile k + 1 is requested at the top of K-step k (its buffer's readers passed the last barrier) and must have
            // landed at the step's end -- nothing else stays in flight
            issue(0, gia);
            if constexpr (GATHER) g2_wait_idx<0>(gia);
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll 1
            for (int k = 0; k < nk; ++k) {
                issue(k + 1, gia);
                if constexpr (GATHER) g2_wait_idx<0>(gia);
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        } else {
        // (GATHER: a wait that leaves only the LAST issue()'s operations in flight has seen the index loads of the issue() before
        // it land -- that set is what the wait defines: gia after issue(.., gib) and the other way round)
        issue(0, gia);
        issue(1, gib);
        if constexpr (GATHER) g2_wait_idx<NW1>(gia);                       // tile 0 has landed (this wave's pieces), and issue(0)'s indices
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW1) : "memory");
        __builtin_amdgcn_s_barrier();                                      // ... and every other loader's
#pragma unroll 1
        for (int k = 0; k < nk; k += 2) {
            issue(k + 2, gia);                                             // into the buffer whose readers passed the last barrier
            if constexpr (GATHER) g2_wait_idx<NW1>(gib);                   // tile k+1 landed (with the indices its issue() requested), tile k+2 stays in flight
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW1) : "memory");
            __builtin_amdgcn_s_barrier();
            if (k + 1 < nk) {
                issue(k + 3, gib);
                if constexpr (GATHER) g2_wait_idx<NW1>(gia);
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW1) : "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the tail loads must not land in the epilogue's image
        }
    } else {
        G2Frag<WM> xa, ya;
        G2Frag<WN> xb, yb;
        __builtin_amdgcn_s_barrier();
        g2_read_frag_dma<A_KM, WM, TM>(As, wmo, li, lh, 0, xa);
        g2_read_frag_dma<B_KM, WN, TN>(Bs, wno, li, lh, 0, xb);
        G2_STAMP(1);
#pragma unroll 1
        for (int k = 0; k < nk; ++k) {
            const int c = k % NBUF, c1 = (k + 1) % NBUF;
            const float* Ac = As + c * ABUF;
            const float* Bc = Bs + c * BBUF;
            G2_KSTAMP(4);
            g2_mfma_part<WM, WN, 0, 4>(xa, xb, acc);                       // X was read a half-step ago: nothing to wait for
            __builtin_amdgcn_sched_barrier(0);
            g2_read_frag_dma<A_KM, WM, TM>(Ac, wmo, li, lh, 1, ya);        // Y lands under the second half of the X MFMAs
            g2_read_frag_dma<B_KM, WN, TN>(Bc, wno, li, lh, 1, yb);
            if (do_bias) {
                const float* col = Ac + (tid / TM) * BK * TM + (tid % TM);
                const int kk = kbeg + k * G2_TK + (tid / TM) * BK;         // (the clamped rows of a ragged reduction tail do not count)
#pragma unroll
                for (int s = 0; s < BK; ++s) bsum += kk + s < kend ? col[s * TM] : 0.f;
            }
            __builtin_amdgcn_sched_barrier(0);
            g2_mfma_part<WM, WN, 4, 8>(xa, xb, acc);
            __builtin_amdgcn_sched_barrier(0);
            G2_KSTAMP(5);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // my reads of tile k are retired (its buffer is refilled next)
            G2_KSTAMP(6);
            __builtin_amdgcn_s_barrier();
            G2_KSTAMP(7);
            __builtin_amdgcn_sched_barrier(0);
            g2_mfma_part<WM, WN, 0, 4>(ya, yb, acc);
            __builtin_amdgcn_sched_barrier(0);
            g2_read_frag_dma<A_KM, WM, TM>(As + c1 * ABUF, wmo, li, lh, 0, xa);
            g2_read_frag_dma<B_KM, WN, TN>(Bs + c1 * BBUF, wno, li, lh, 0, xb);
            __builtin_amdgcn_sched_barrier(0);
            g2_mfma_part<WM, WN, 4, 8>(ya, yb, acc);
            G2_KSTAMP(8);
        }
        G2_STAMP(2);
    }
    if (do_bias) {
        __syncthreads();
        if (!loader) lds[tid] = bsum;
        __syncthreads();
        if (tid < TM && m0 + tid < g.M) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < BQ; ++q) s += lds[q * TM + tid];
            g.dbias[(long)z * g.bslab + m0 + tid] = s;
        }
    }
    g2_epilogue<A_KM, B_KM, WM, WN, 512, TM, TN, SC1C>(g, acc, lds, m0, n0, z, wmo, wno, li, lh, !loader);
    G2_STAMP(3);
}

template <bool A_KM, bool B_KM, int WM, int WN, bool GATHER = false, int LAY = 0, int NB = 3>
__global__ __launch_bounds__(512) void gemm2_dma_kernel(Gemm2Group gg) {
    __shared__ __attribute__((aligned(1024))) float lds[G2DmaShape<WM, WN, LAY, NB>::LDSF];
#ifdef G2_PROFILE
    if (gg.prof && threadIdx.x == 0) gg.prof[(size_t)blockIdx.x * 4] = __builtin_readcyclecounter();
#endif
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GEMM2_MAXP; ++i)
        if (i < gg.n && (int)blockIdx.x >= gg.p[i].block0) pi = i;
    const Gemm2Prob& g = gg.p[pi];
    const int tiles = g.tiles_m * g.tiles_n, nblk = tiles * g.splits;
    int local = (int)blockIdx.x - g.block0;
    if (local >= nblk) return;
    local = g2_xcd_local(local, nblk);
    const int z = local / tiles, t = local - z * tiles;
    const int tm = t / g.tiles_n, tn = t - tm * g.tiles_n;
    g2_dma_body<A_KM, B_KM, WM, WN, GATHER, LAY, 0, false, NB>(gg, g, tm, tn, z, lds);
}

// =====================================================================================================================
// Chain of dependent layers in ONE launch (Gemm2Chain, gemm2.h).  The small-step regime's network step is ~9 dependent launches
// of 5-20 us; a kernel boundary costs ~5.6 us there while the other network's chain competes for the chip.
// A layer-to-layer dependency only ties together the N / 64 work-groups of one 64-row stripe, so it is carried by a stripe-local
// arrival counter instead: tiles leave as write-through (sc1) 16-byte stores, every storing wave drains them (s_waitcnt
// vmcnt(0)), ONE lane adds to the stripe's counter and polls it (relaxed, agent scope, s_sleep, bounded), and the next phase
// loads its A operand with sc1 LDS-DMA loads -- nothing depends on dispatch order or on where a work-group runs; blocks of a
// stripe are numbered so that they land on one XCD when the dispatcher round-robins (speed only).
// MEASURED (round 3, cfg 2 shapes): bit-identical to the per-layer launches; on an idle chip a stripe hand-off costs what a
// kernel boundary costs (3 forward layers 37.1 us either way; with a memset node zeroing the counters per launch it was 41.2,
// hence the monotonic counters); in the two-stream PPO step it is 14 % SLOWER than the boundaries it replaces (1.98 vs 2.29 M
// env-steps/s) -- MI355X_MICROARCH.md's "cut at every seam" verdict, reproduced.  The learner therefore uses it only on
// request (PARTMANIP_CHAIN=1, algo_utils/network.py).
#define G2_SPIN_LIMIT (1u << 18)
typedef unsigned long long g2_u64;
__device__ __forceinline__ void g2_stripe_barrier(g2_u64* __restrict__ ctr, g2_u64 target, g2_u64* __restrict__ err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // my write-through stores have left
    __syncthreads();                                                       // ... and those of every wave of this work-group
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > G2_SPIN_LIMIT) {                                 // a work-group of the stripe is not resident / has died:
                __hip_atomic_store(err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // give up loudly instead of hanging the GPU
                break;
            }
        }
    }
    __syncthreads();
}

// Phase 0 of a forward chain when the input layer's rows are not 16-byte loadable (K <= 64, e.g. 53 observations): both
// operand tiles go whole through registers into the LDS-DMA image (two 32-wide K-tiles, zero-filled past K), then the four
// multiplying waves run the same fragment reads / MFMA order as the pipelined body.
template <bool SC1C>
__device__ __forceinline__ void g2_smallk_body(const Gemm2Prob& g, int tm, int tn, float* __restrict__ lds) {
    using SH = G2DmaShape<1, 1, 0>;
    constexpr int TM = 64, TN = 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3;
    const bool loader = tid >= 256;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = tm * TM, n0 = tn * TN;
    const int wmo = (wave >> 1) * 32, wno = (wave & 1) * 32;
    float* As = lds;                                                       // two K-tiles of A, then two of B
    float* Bs = lds + 2 * SH::ABUF;
#pragma unroll
    for (int j = 0; j < TM * 64 / 512; ++j) {
        const int e = tid + 512 * j, row = e >> 6, k = e & 63;
        const int kt = k >> 5, c = (k & 31) >> 2, w = k & 3;
        const int dst = kt * SH::ABUF + row * 32 + ((c ^ ((row >> 1) & 7)) << 2) + w;
        const int ga = m0 + row, gb = n0 + row;
        As[dst] = (k < g.K && ga < g.M) ? g.A[(long)ga * g.lda + k] : 0.f;
        Bs[dst] = (k < g.K && gb < g.N) ? g.B[(long)gb * g.ldb + k] : 0.f;
    }
    __syncthreads();
    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    if (!loader) {
        const int nkt = g.K > 32 ? 2 : 1;
        for (int kt = 0; kt < nkt; ++kt) {
            G2Frag<1> xa, xb, ya, yb;
            g2_read_frag_dma<false, 1, TM>(As + kt * SH::ABUF, wmo, li, lh, 0, xa);
            g2_read_frag_dma<false, 1, TN>(Bs + kt * SH::BBUF, wno, li, lh, 0, xb);
            g2_read_frag_dma<false, 1, TM>(As + kt * SH::ABUF, wmo, li, lh, 1, ya);
            g2_read_frag_dma<false, 1, TN>(Bs + kt * SH::BBUF, wno, li, lh, 1, yb);
            g2_mfma_half<1, 1>(xa, xb, acc);
            g2_mfma_half<1, 1>(ya, yb, acc);
        }
    }
    g2_epilogue<false, false, 1, 1, 512, TM, TN, SC1C>(g, acc, lds, m0, n0, 0, wmo, wno, li, lh, !loader);
}

template <bool B_KM>
__global__ __launch_bounds__(512) void gemm2_chain_kernel(Gemm2Chain ch) {
    __shared__ __attribute__((aligned(1024))) float lds[G2DmaShape<1, 1, 0>::LDSF];
    // block -> (stripe, column tile): blocks b = x (mod 8) land on XCD x when the dispatcher round-robins; a stripe's G
    // work-groups are consecutive WITHIN an XCD's block list (their tiles then meet in that XCD's L2 -- speed only)
    const int total = ch.stripes * ch.G;
    int local = (int)blockIdx.x;
    local = g2_xcd_local(local, total);
    const int stripe = local / ch.G, tn = local - stripe * ch.G;
    g2_u64* ctr = ch.bar + stripe;
    g2_u64* err = ch.bar + ch.stripes;
    // The counters are MONOTONIC and never reset (a memset node in front of every launch costs as much as the boundary the
    // chain removes; a per-launch epoch argument would be frozen by graph replay): a launch adds G arrivals per barrier, i.e.
    // per_launch = G * (n - 1) in total, and every earlier launch has completed, so what this work-group reads at its start is
    // (launches so far) * per_launch + (arrivals of THIS launch's first barrier, < G <= per_launch): the launch's base is that
    // value rounded down to a multiple of per_launch.  (64-bit: no wrap in any run length.)
    __shared__ g2_u64 s_base;
    if (threadIdx.x == 0) {
        const g2_u64 per_launch = (g2_u64)ch.G * (g2_u64)(ch.n - 1);
        const g2_u64 v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_base = v - v % per_launch;
        // a launch that was aborted half-way (device error, killed process sharing the workspace) leaves the counter off a
        // launch boundary: what this work-group reads must be boundary + (< G arrivals of this launch's first barrier).  If not,
        // say so in the error word (the host re-zeroes the counters when it sees it) instead of passing barriers early.
        if (v % per_launch >= (g2_u64)ch.G) __hip_atomic_store(err, 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const g2_u64 base = s_base;
    Gemm2Group gg{};                                                       // (profiling hooks of the body: unused here)
    gg.prof = nullptr;
    int ph = 0;
    if (ch.small0) {
        if constexpr (!B_KM) g2_smallk_body<true>(ch.p[0], stripe, tn, lds);
        g2_stripe_barrier(ctr, base + (g2_u64)ch.G, err);
        ph = 1;
    }
#pragma unroll 1
    for (; ph < ch.n; ++ph) {
        // one instantiation for every phase: sc1 loads of A (exact for data of an earlier launch too) and write-through stores
        g2_dma_body<false, B_KM, 1, 1, false, 0, 16, true>(gg, ch.p[ph], stripe, tn, 0, lds);
        if (ph + 1 < ch.n) g2_stripe_barrier(ctr, base + (g2_u64)ch.G * (g2_u64)(ph + 1), err);
    }
}

int gemm2_chain_launch(Gemm2Chain& ch, bool b_kmajor, void* stream) {
    if (ch.n < 2 || ch.n > GEMM2_CHAIN_MAX || !ch.bar) return PM_EINVAL;
    const int M = ch.p[0].M, N = ch.p[0].N;
    if (N % 64 != 0) return PM_EUNSUPPORTED;
    for (int i = 0; i < ch.n; ++i) {
        Gemm2Prob& p = ch.p[i];
        if (p.M != M || p.N != N) return PM_EUNSUPPORTED;                  // equal widths: every phase has the same stripes x G grid
        if (i > 0 && (p.A != ch.p[i - 1].C || p.lda != ch.p[i - 1].ldc || p.K != N)) return PM_EINVAL;
        p.splits = 1;
        p.kchunk = ((p.K + G2_TK - 1) / G2_TK) * G2_TK;
        p.Mld = p.M; p.Nld = p.N;
        p.tiles_m = (M + 63) / 64; p.tiles_n = N / 64; p.block0 = 0;
        p.vecC = (p.ldc % 4 == 0) && (((uintptr_t)p.C & 15) == 0) &&
                 (p.epi != G2_EPI_BIAS_ACT || !p.bias || ((uintptr_t)p.bias & 15) == 0) &&
                 (p.epi != G2_EPI_MUL_DACT || p.act == PM_ACT_NONE || (p.ldh % 4 == 0 && ((uintptr_t)p.H & 15) == 0));
        if (!p.vecC || (long)M * p.ldc * 4 >= 0x7fffffffL) return PM_EUNSUPPORTED;      // 16-byte write-through stores, 31-bit buffer offsets
        const bool small = i == 0 && !b_kmajor && !(p.vecA && p.vecB && p.K % G2_TK == 0);
        if (small) {
            if (p.K > 64) return PM_EUNSUPPORTED;
            ch.small0 = 1;
        } else if (!(p.vecA && p.vecB && p.K % G2_TK == 0)) {
            return PM_EUNSUPPORTED;
        }
    }
    ch.stripes = (M + 63) / 64;
    ch.G = N / 64;
    const int total = ch.stripes * ch.G;
    if (total > GEMM2_CHAIN_MAX_WG) return PM_EUNSUPPORTED;               // every work-group must be resident: one per CU
    if (b_kmajor) hipLaunchKernelGGL((gemm2_chain_kernel<true>), dim3(total), dim3(512), 0, pm_stream(stream), ch);
    else hipLaunchKernelGGL((gemm2_chain_kernel<false>), dim3(total), dim3(512), 0, pm_stream(stream), ch);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// (WM, WN) of a launch: 64 x 64 work-group tiles for small grids; for big ones 128 rows / columns only along a dimension some
// problem actually extends past 64 -- a 128 x 128 tile on an N = 64 problem (the 64-channel sparse convolutions: 2.7 M rows,
// K = 1728) spends half of its MFMAs on columns that do not exist.
#ifndef G2_NBUF_DEFAULT
#define G2_NBUF_DEFAULT 3
#endif
template <bool A_KM, bool B_KM, bool GATHER>
static void g2_launch_dma(const Gemm2Group& g, int wm, int wn, int blocks, void* stream, bool lay1 = false, bool lay2 = false) {
    const dim3 grid(blocks), blk(512);
    if constexpr (A_KM && B_KM) {
        if (lay1) {
            hipLaunchKernelGGL((gemm2_dma_kernel<A_KM, B_KM, 1, 1, GATHER, 1>), grid, blk, 0, pm_stream(stream), g);
            return;
        }
    }
    if constexpr (!A_KM && !B_KM) {
        if (lay2) {
            hipLaunchKernelGGL((gemm2_dma_kernel<A_KM, B_KM, 1, 1, GATHER, 2>), grid, blk, 0, pm_stream(stream), g);
            return;
        }
    }
    if constexpr (!GATHER) {
        constexpr int nb2 = G2_NBUF_DEFAULT;                                // -DG2_NBUF_DEFAULT=2: the two-stage ring for 128 x 128 tiles (A/B builds)
        if (wm == 2 && wn == 2 && nb2 == 2) {
            hipLaunchKernelGGL((gemm2_dma_kernel<A_KM, B_KM, 2, 2, false, 0, 2>), grid, blk, 0, pm_stream(stream), g);
            return;
        }
    }
#ifdef G2_TILE_ENV                                         // A/B builds only: PM_G2_NBUF=2 for every tile shape, gathered operands included
    if (const char* e = getenv("PM_G2_NBUF"); e && atoi(e) == 2) {
        if (wm == 2 && wn == 2) hipLaunchKernelGGL((gemm2_dma_kernel<A_KM, B_KM, 2, 2, GATHER, 0, 2>), grid, blk, 0, pm_stream(stream), g);
        else if (wm == 2) hipLaunchKernelGGL((gemm2_dma_kernel<A_KM, B_KM, 2, 1, GATHER, 0, 2>), grid, blk, 0, pm_stream(stream), g);
        else if (wn == 2) hipLaunchKernelGGL((gemm2_dma_kernel<A_KM, B_KM, 1, 2, GATHER, 0, 2>), grid, blk, 0, pm_stream(stream), g);
        else hipLaunchKernelGGL((gemm2_dma_kernel<A_KM, B_KM, 1, 1, GATHER, 0, 2>), grid, blk, 0, pm_stream(stream), g);
        return;
    }
#endif
    if (wm == 2 && wn == 2) hipLaunchKernelGGL((gemm2_dma_kernel<A_KM, B_KM, 2, 2, GATHER>), grid, blk, 0, pm_stream(stream), g);
    else if (wm == 2) hipLaunchKernelGGL((gemm2_dma_kernel<A_KM, B_KM, 2, 1, GATHER>), grid, blk, 0, pm_stream(stream), g);
    else if (wn == 2) hipLaunchKernelGGL((gemm2_dma_kernel<A_KM, B_KM, 1, 2, GATHER>), grid, blk, 0, pm_stream(stream), g);
    else hipLaunchKernelGGL((gemm2_dma_kernel<A_KM, B_KM, 1, 1, GATHER>), grid, blk, 0, pm_stream(stream), g);
}

template <bool A_KM, bool B_KM>
static int g2_launch_o(Gemm2Group& g, bool big, void* stream) {
    bool vec = true;                                                       // 16-byte loads only if every operand of every problem allows them
    for (int i = 0; i < g.n; ++i) vec = vec && g.p[i].vecA && g.p[i].vecB;
    bool dma = vec, gather = false;
    int maxM = 0, maxN = 0;
    for (int i = 0; i < g.n; ++i) {
        dma = dma && (g.p[i].K % G2_TK == 0) && (g.p[i].kchunk % G2_TK == 0);
        gather = gather || g.p[i].gidx != nullptr;
        maxM = g.p[i].M > maxM ? g.p[i].M : maxM;
        maxN = g.p[i].N > maxN ? g.p[i].N : maxN;
    }
    int wm = big ? 2 : 1, wn = big ? 2 : 1;
    if (big && (dma || gather)) {                                          // the register-staged kernel keeps its two square shapes
        if (maxM <= 64) wm = 1;
        if (maxN <= 64) wn = 1;
    }
    // A GATHERED forward operand (sparse convolutions): 64-row tiles.  Measured at cfg 5's shapes (tools/time_sparse_conv.py,
    // PM_G2_TILE A/B): 2.7 M x 1728 x 64 -- 64 x 64: 5.76 ms, 128 x 64: 6.07; 0.62 M x 3456 x 128 -- 64 x 128: 4.98 ms, 128 x 128:
    // 5.80, 64 x 64: 5.15, 128 x 64: 5.57.  The 128 x 128 tile's three LDS stages (98 KB) leave ONE work-group per CU -- one
    // multiplying wave per SIMD, whose every stall idles the matrix pipe -- and a gathered row costs its loader more issue slots
    // than a plain one; 64 x 128 (74 KB) keeps two work-groups resident.
    if (big && gather && !A_KM) { wm = 1; wn = maxN > 64 ? 2 : 1; }
    // PLAIN forward / data-gradient operands: 128 x 64, for the same reason (74 KB: two work-groups per CU).  tools/time_gemm.py under
    // PM_G2_TILE (profiles/round4_m_gemm_tiles.txt), 128 x 128 -> 128 x 64: 524 288 x 128 x 128 forward 253 -> 232 us, data gradient
    // 290 -> 227; 131 072 x 288 x 256: 207 -> 189 / 307 -> 209; 131 072 x 256 x 512: 377 -> 343 / 338 -> 309 (64 x 128 the same
    // forward, slower data gradients; two stages instead of three: slower on every shape but the first).
    if (big && dma && !gather && !(A_KM && B_KM) && wm == 2 && wn == 2) wn = 1;
#ifdef G2_TILE_ENV                                         // A/B builds only: PM_G2_TILE=21 / 12 / 22 forces the (wm, wn) of forward / data-gradient launches
    if (const char* e = getenv("PM_G2_TILE"); e && dma && !(A_KM && B_KM)) { wm = e[0] - '0'; wn = e[1] - '0'; }
    if (const char* e = getenv("PM_G2_WTILE"); e && (dma || gather) && A_KM && B_KM) { wm = e[0] - '0'; wn = e[1] - '0'; }   // weight gradients
#endif
    // weight gradients of <= 32 output rows: the 32 x 128 tile (1 x 4 waves) of the LDS-DMA kernels
    const bool lay1 = A_KM && B_KM && (dma || gather) && vec && maxM <= 32 && maxN > 64;
    // forward problems of <= 32 output columns over many rows: the 128 x 32 tile (4 x 1 waves)
    const bool lay2 = !A_KM && !B_KM && big && (dma || gather) && vec && maxN <= 32;
    if (lay1 || lay2) wm = wn = 1;
    const int TM = lay1 ? 32 : (lay2 ? 128 : 64 * wm), TN = lay1 ? 128 : (lay2 ? 32 : 64 * wn);
    int blocks = 0;
    for (int i = 0; i < g.n; ++i) {
        Gemm2Prob& p = g.p[i];
        p.vecC = (p.N % 4 == 0) && (p.ldc % 4 == 0) && (p.slab % 4 == 0) && (((uintptr_t)p.C & 15) == 0) &&
                 (p.epi != G2_EPI_BIAS_ACT || !p.bias || ((uintptr_t)p.bias & 15) == 0) &&
                 (p.epi != G2_EPI_MUL_DACT || p.act == PM_ACT_NONE || (p.ldh % 4 == 0 && ((uintptr_t)p.H & 15) == 0));
        if (p.sidx && (!p.vecC && !(p.N % 4 == 0 && p.sC % 4 == 0 && ((uintptr_t)p.C & 15) == 0))) return PM_EINVAL;
        if (p.sidx) p.vecC = 1;                                            // scattered rows: the 16-byte path only (sC % 4 == 0)
        p.tiles_m = (p.M + TM - 1) / TM;
        p.tiles_n = (p.N + TN - 1) / TN;
        p.block0 = blocks;
        blocks += p.tiles_m * p.tiles_n * p.splits;
        blocks = (blocks + 7) & ~7;                                        // problems start on an XCD-0 block
    }
    if (gather) {                                                          // virtual (gathered) operand: LDS-DMA kernels only
        for (int i = 0; i < g.n; ++i) {
            Gemm2Prob& q = g.p[i];
            q.gsh = -1;
            for (int sh = 0; sh < 31; ++sh)
                if (q.gC == (1 << sh)) q.gsh = sh;
            const bool kok = A_KM ? (q.kchunk % G2_TK == 0) : (q.K % G2_TK == 0 && q.kchunk % G2_TK == 0);
            if (!q.gidx || !q.gzero || !vec || !kok || q.gC % 4 != 0 || q.gJ < 1 || (long)q.gJ * q.gC != (A_KM ? q.N : q.K)) return PM_EINVAL;
        }
        if constexpr (A_KM == B_KM) g2_launch_dma<A_KM, B_KM, true>(g, wm, wn, blocks, stream, lay1, lay2);
        else return PM_EUNSUPPORTED;
        PM_CHECK_LAUNCH();
        return PM_OK;
    }
    const dim3 grid(blocks), blk(256);
    if (dma) g2_launch_dma<A_KM, B_KM, false>(g, wm, wn, blocks, stream, lay1, lay2);
    else if (big && vec) hipLaunchKernelGGL((gemm2_kernel<A_KM, B_KM, 2, 2, true>), grid, blk, 0, pm_stream(stream), g);
    else if (big) hipLaunchKernelGGL((gemm2_kernel<A_KM, B_KM, 2, 2, false>), grid, blk, 0, pm_stream(stream), g);
    else if (vec) hipLaunchKernelGGL((gemm2_kernel<A_KM, B_KM, 1, 1, true>), grid, blk, 0, pm_stream(stream), g);
    else hipLaunchKernelGGL((gemm2_kernel<A_KM, B_KM, 1, 1, false>), grid, blk, 0, pm_stream(stream), g);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

#ifdef G2_PROFILE
static unsigned long long* g2_prof_buf = nullptr;                          // A/B builds only (tools/g2_profile.py)
extern "C" void pm_debug_set_gemm_prof(void* p) { g2_prof_buf = (unsigned long long*)p; }
#endif

int gemm2_launch(Gemm2Group& g, bool a_kmajor, bool b_kmajor, void* stream) {
    if (g.n < 1 || g.n > GEMM2_MAXP) return PM_EINVAL;
#ifdef G2_PROFILE
    g.prof = g2_prof_buf;
#else
    g.prof = nullptr;
#endif
    long big_tiles = 0;
    for (int i = 0; i < g.n; ++i) {
        Gemm2Prob& p = g.p[i];
        if (p.splits < 1) p.splits = 1;
        if (p.Mld < p.M) p.Mld = p.M;
        if (p.Nld < p.N) p.Nld = p.N;
        if (p.splits == 1) p.kchunk = ((p.K + G2_TK - 1) / G2_TK) * G2_TK;
        if (p.kchunk <= 0 || p.kchunk % G2_TK != 0 || (long)(p.splits - 1) * p.kchunk >= p.K) return PM_EINVAL;
        // 16-byte loads: k-contiguous needs K % 4 == 0 (and >= 4), k-major needs rows % 4 == 0 (and >= 4); the callers
        // have already checked pointer / leading-dimension alignment
        big_tiles += (long)((p.M + 127) / 128) * ((p.N + 127) / 128) * p.splits;
    }
    const bool big = big_tiles >= 512;                                     // >= 2 work-groups of 128 x 128 per CU
    if (!a_kmajor && !b_kmajor) return g2_launch_o<false, false>(g, big, stream);
    if (!a_kmajor && b_kmajor) return g2_launch_o<false, true>(g, big, stream);
    if (a_kmajor && b_kmajor) return g2_launch_o<true, true>(g, big, stream);
    return PM_EUNSUPPORTED;
}
