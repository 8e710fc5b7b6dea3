// Grouped fp32-MFMA GEMM (gemm2_f32.hip): up to GEMM2_MAXP independent problems of one operand orientation in ONE
// launch -- the Linear layers of the small-step regime (state PPO: 2560 dependent optimiser steps of ~12 launches)
// are launch- and ramp-bound, so the actor's and the critic's layer, or all weight gradients of a network, share a
// grid.  Internal to libpartmanip_hip.so; the C ABI on top is in gemm_f32.hip.
#pragma once
#include "common.h"

#define GEMM2_MAXP 8

enum { G2_EPI_BIAS_ACT = 0, G2_EPI_MUL_DACT = 1, G2_EPI_PLAIN = 2 };

struct Gemm2Prob {
    const float* A; long lda;     // k-contiguous: A[row*lda + k]; k-major: A[k*lda + row]
    const float* B; long ldb;     // k-contiguous: B[col*ldb + k]; k-major: B[k*ldb + col]
    float* C; long ldc;           // C[row*ldc + col] (+ z*slab for split-K slab z)
    const float* bias;            // G2_EPI_BIAS_ACT: per column (may be null)
    const float* H; long ldh;     // G2_EPI_MUL_DACT: C *= 1 - H[row][col]^2
    float* dbias; long bslab;     // G2_EPI_PLAIN with a k-major A: column sums of A (= bias gradient) (+ z*bslab), or null
    long slab;                    // split-K: C offset per slab (elements)
    int M, N, K;                  // rows, cols, reduction
    int Mld, Nld;                 // k-major operands: rows of A / B that may be LOADED (>= M / N, 0 = M / N): a weight gradient whose
                                  // dY or X has 10 or 53 columns inside 16- / 56-float rows takes the 16-byte loaders with Mld = 16 /
                                  // Nld = 56; what the extra columns hold only reaches result rows / columns >= M / N, which are not stored
    int epi, act;
    int vecA, vecB;               // 16-byte global loads allowed for that operand
    int splits, kchunk;           // split-K: number of slabs, reduction range per slab (multiple of 32)
    // gathered operand (sparse convolutions, sparse_voxel.hip): the k-contiguous A of a forward problem / the k-major B of a
    // weight-gradient problem is VIRTUAL -- element (row r, column q) is gsrc-row gidx[r*gJ + q/gC], column q%gC of the matrix
    // at A (resp. B) with its leading dimension, or 0 where the index is negative (read from gzero, >= gC zeros).  The
    // (rows x gJ*gC) operand never exists in HBM.  LDS-DMA path only (gC % 4 == 0, K resp. N = gJ*gC).
    const int32_t* gidx; const float* gzero; int gJ, gC;
    int gsh;                      // log2(gC) when gC is a power of two, else -1 (filled by the launcher): the loader splits a virtual
                                  // column into (tap, channel) per 16-byte piece and K-step -- as a shift / mask instead of two integer
                                  // divisions (~10 VALU instructions each, issued from the SIMDs the MFMA waves run on)
    // scattered result (the input gradient of a convolution whose patches do not overlap): element (row r, column q) goes to
    // row sidx[r*sJ + q/sC], column q%sC of the matrix at C (row stride sC), and is dropped where the index is negative; H
    // (G2_EPI_MUL_DACT) is read at the same place.  sC % 4 == 0.  Rows no element maps to keep their contents.
    const int32_t* sidx; int sJ, sC;
    int vecC;                     // 16-byte stores of C (and loads of bias / H) allowed (filled by the launcher)
    int tiles_m, tiles_n;         // filled by the launcher
    int block0;                   // first work-group of this problem in the grid (filled by the launcher)
};

struct Gemm2Group {
    int n;
    unsigned long long* prof;     // -DG2_PROFILE builds: 4 s_memtime stamps per work-group (tools/g2_profile.py)
    Gemm2Prob p[GEMM2_MAXP];
};

// ---- a CHAIN of dependent problems in one launch (the hidden layers of an MLP forward, or of its data gradient) --------------
// Phase p + 1 reads, as its A operand, the C of phase p.  The dependency is row-local: a 64-row stripe of C(p+1) needs the same
// 64 rows of C(p), all columns -- so the N / 64 work-groups that own a stripe hand their tiles to each other INSIDE the launch
// (write-through stores, one arrival counter per stripe, sc1 loads: MI355X_MICROARCH.md "inter-workgroup visibility", form R1)
// and no kernel boundary separates the layers.  Every work-group of the grid must be resident (<= 256, one per CU).
#define GEMM2_CHAIN_MAX 4
#define GEMM2_CHAIN_MAX_WG 256
struct Gemm2Chain {
    int n;                        // phases
    int stripes, G;               // 64-row stripes, work-groups (64-column tiles) per stripe -- filled by the launcher
    int small0;                   // phase 0 has K <= 64 and rows that are not 16-byte loadable (the input layer): staged through registers
    unsigned long long* bar;      // `stripes` MONOTONIC arrival counters + one error word behind them: zeroed ONCE by the caller, then
                                  // only ever handed to chains of this (stripes, G, n) -- a launch derives its base from what it finds
    Gemm2Prob p[GEMM2_CHAIN_MAX];
};
// b_kmajor: false = forward chain (A, B k-contiguous), true = data-gradient chain (B = W k-major).  PM_EUNSUPPORTED when the
// shapes do not fit the scheme (the caller then issues the layers one by one).
int gemm2_chain_launch(Gemm2Chain& ch, bool b_kmajor, void* stream);

// Launches the group (all problems share the orientation); big = 128x128 work-group tiles (for M*N >> chip), else 64x64.
// Returns PM_OK / PM_E*.  `tag` only names the launch for profiling purposes.
int gemm2_launch(Gemm2Group& g, bool a_kmajor, bool b_kmajor, void* stream);
