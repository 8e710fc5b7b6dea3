// K4/K5: Linear forward / backward as LDS-tiled fp32 MFMA GEMMs (v_mfma_f32_32x32x2_f32:
// exact fp32 products, fp32 accumulate -- bitwise an fmaf chain per output, so results are
// fp32-roundoff-class against the reference's CPU sgemm).
//
// One work-group = 8 waves on a 64x64 tile, K-step 64: waves 0-3 (2x2 blocks of 32x32) take
// k in [0,32) of every K-step and waves 4-7 the same blocks with k in [32,64) (intra-work-group
// split-K, combined through LDS at the end).  The output of these layers (2048 x 512) is only 1024
// MFMA blocks -- one per SIMD -- so without the split every SIMD holds a single wave and nothing
// hides the LDS staging; with it two waves per SIMD alternate staging and MFMA.  Operand tiles are staged in LDS in
// their natural global orientation:
//   "k-contiguous" operand (row r, k fastest):   LDS [64][68]  -> fragments by ds_read_b128;
//        row stride 68 floats: (68*r) mod 64 hits 16 distinct 4-bank slots -> conflict-free.
//   "k-major" operand (k slowest, r fastest):    LDS [64][64]  -> fragments by ds_read_b32,
//        lanes read consecutive floats -> conflict-free.
// Lane l of a wave feeds MFMA row/col (l&31); lanes 0-31 own k in [0,32) of the K-step and
// lanes 32-63 own k in [32,64), so a k-contiguous lane reads its 32 k-values as 8 x 16 B.
#include "common.h"
#include "gemm2.h"

#define GB_M 64
#define GB_N 64
#define GB_K 64
#define LDK (GB_K + 4)   // k-contiguous LDS row stride (floats)
#define LDR 64           // k-major LDS row stride (floats)
#define GB_T 512                      // threads per work-group
#define NV4 (GB_M * GB_K / 4 / GB_T)  // float4 loads per thread per operand tile
#define NSC (GB_M * GB_K / GB_T)      // scalar loads per thread per operand tile
#define KH (GB_K / 4)                 // k-values per lane half per wave k-group (16)

enum { EPI_BIAS_ACT = 0, EPI_MUL_DACT = 1, EPI_PLAIN = 2 };

struct GemmArgs {
    const float* A; long lda;     // k-contig: A[row*lda + k] ; k-major: A[k*lda + row]
    const float* B; long ldb;     // k-contig: B[col*ldb + k] ; k-major: B[k*ldb + col]
    float* C; long ldc;           // C[row*ldc + col]
    const float* bias;            // EPI_BIAS_ACT: per col (may be null)
    const float* H; long ldh;     // EPI_MUL_DACT: C *= 1 - H[row][col]^2
    int M, N, K;                  // GEMM dims (rows, cols, reduction)
    int act;
    int kchunk;                   // split-K: reduction range per blockIdx.z
    long slab;                    // split-K: C offset per blockIdx.z (elements)
    int vecA, vecB;               // 16 B global loads allowed
    float* dbias;                 // EPI_PLAIN (weight gradient): column sums of the k-major A operand
                                  // (= bias gradient), written by the blockIdx.x == 0 tiles per z-slab
};

// ---- global -> register tile loads ------------------------------------------------------
// k-contiguous tile: 64 rows x 32 k.  vec: 2 float4 per thread; scalar: 8 floats per thread.
__device__ __forceinline__ void load_kcontig(const float* __restrict__ P, long ld, int row0, int nrows, int k0,
                                             int kend, int vec, float (&r)[NSC]) {
    const int tid = threadIdx.x;
    if (vec) {
#pragma unroll
        for (int j = 0; j < NV4; ++j) {
            const int idx = tid + GB_T * j, row = idx / (GB_K / 4), k = k0 + ((idx % (GB_K / 4)) << 2);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row0 + row < nrows && k < kend) v = *(const float4*)(P + (long)(row0 + row) * ld + k);
            r[4 * j] = v.x; r[4 * j + 1] = v.y; r[4 * j + 2] = v.z; r[4 * j + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < NSC; ++j) {
            const int idx = tid + GB_T * j, row = idx / GB_K, k = k0 + (idx % GB_K);
            r[j] = (row0 + row < nrows && k < kend) ? P[(long)(row0 + row) * ld + k] : 0.f;
        }
    }
}
__device__ __forceinline__ void store_kcontig(float* __restrict__ S, int vec, const float (&r)[NSC]) {
    const int tid = threadIdx.x;
    if (vec) {
#pragma unroll
        for (int j = 0; j < NV4; ++j) {
            const int idx = tid + GB_T * j, row = idx / (GB_K / 4), k = (idx % (GB_K / 4)) << 2;
            *(float4*)(S + row * LDK + k) = make_float4(r[4 * j], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < NSC; ++j) {
            const int idx = tid + GB_T * j;
            S[(idx / GB_K) * LDK + (idx % GB_K)] = r[j];
        }
    }
}
// k-major tile: GB_K k x 64 rows(cols).
__device__ __forceinline__ void load_kmajor(const float* __restrict__ P, long ld, int row0, int nrows, int k0,
                                            int kend, int vec, float (&r)[NSC]) {
    const int tid = threadIdx.x;
    if (vec) {
#pragma unroll
        for (int j = 0; j < NV4; ++j) {
            const int idx = tid + GB_T * j, k = k0 + (idx >> 4), row = row0 + ((idx & 15) << 2);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < kend && row < nrows) v = *(const float4*)(P + (long)k * ld + row);
            r[4 * j] = v.x; r[4 * j + 1] = v.y; r[4 * j + 2] = v.z; r[4 * j + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < NSC; ++j) {
            const int idx = tid + GB_T * j, k = k0 + (idx >> 6), row = row0 + (idx & 63);
            r[j] = (k < kend && row < nrows) ? P[(long)k * ld + row] : 0.f;
        }
    }
}
__device__ __forceinline__ void store_kmajor(float* __restrict__ S, int vec, const float (&r)[NSC]) {
    const int tid = threadIdx.x;
    if (vec) {
#pragma unroll
        for (int j = 0; j < NV4; ++j) {
            const int idx = tid + GB_T * j;
            *(float4*)(S + (idx >> 4) * LDR + ((idx & 15) << 2)) =
                make_float4(r[4 * j], r[4 * j + 1], r[4 * j + 2], r[4 * j + 3]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < NSC; ++j) {
            const int idx = tid + GB_T * j;
            S[(idx >> 6) * LDR + (idx & 63)] = r[j];
        }
    }
}

template <bool A_KMAJOR, bool B_KMAJOR, int EPI>
__global__ __launch_bounds__(GB_T) void gemm_f32_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[GB_M * LDK];   // 64*68 >= 64*64
    __shared__ __attribute__((aligned(16))) float Bs[GB_N * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int kg = wave >> 2, wq = wave & 3;                  // k-group (intra-WG split-K), output quadrant
    const int wm = (wq >> 1) * 32, wn = (wq & 1) * 32;
    const int kofs = kg * (GB_K / 2) + lh * KH;               // this lane's first k inside a K-step
    const int m0 = blockIdx.y * GB_M, n0 = blockIdx.x * GB_N;
    const int kbeg = blockIdx.z * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool do_bias = (EPI == EPI_PLAIN) && g.dbias != nullptr && blockIdx.x == 0;
    float bsum = 0.f;

    float ra[NSC], rb[NSC];
    if (kbeg < kend) {
        if (A_KMAJOR) load_kmajor(g.A, g.lda, m0, g.M, kbeg, kend, g.vecA, ra);
        else load_kcontig(g.A, g.lda, m0, g.M, kbeg, kend, g.vecA, ra);
        if (B_KMAJOR) load_kmajor(g.B, g.ldb, n0, g.N, kbeg, kend, g.vecB, rb);
        else load_kcontig(g.B, g.ldb, n0, g.N, kbeg, kend, g.vecB, rb);
    }
    for (int k0 = kbeg; k0 < kend; k0 += GB_K) {
        __syncthreads();                       // previous tile's fragment reads done
        if (A_KMAJOR) store_kmajor(As, g.vecA, ra); else store_kcontig(As, g.vecA, ra);
        if (B_KMAJOR) store_kmajor(Bs, g.vecB, rb); else store_kcontig(Bs, g.vecB, rb);
        __syncthreads();
        if (EPI == EPI_PLAIN && A_KMAJOR && do_bias && tid < GB_M) {   // db += column sums of this dY tile
#pragma unroll 8
            for (int k = 0; k < GB_K; ++k) bsum += As[k * LDR + tid];
        }
        if (k0 + GB_K < kend) {                // prefetch next tile while this one is multiplied
            if (A_KMAJOR) load_kmajor(g.A, g.lda, m0, g.M, k0 + GB_K, kend, g.vecA, ra);
            else load_kcontig(g.A, g.lda, m0, g.M, k0 + GB_K, kend, g.vecA, ra);
            if (B_KMAJOR) load_kmajor(g.B, g.ldb, n0, g.N, k0 + GB_K, kend, g.vecB, rb);
            else load_kcontig(g.B, g.ldb, n0, g.N, k0 + GB_K, kend, g.vecB, rb);
        }
        float fa[KH], fb[KH];
        if (A_KMAJOR) {
#pragma unroll
            for (int s = 0; s < KH; ++s) fa[s] = As[(kofs + s) * LDR + wm + li];
        } else {
#pragma unroll
            for (int q = 0; q < KH / 4; ++q) {
                const float4 v = *(const float4*)(As + (wm + li) * LDK + kofs + 4 * q);
                fa[4 * q] = v.x; fa[4 * q + 1] = v.y; fa[4 * q + 2] = v.z; fa[4 * q + 3] = v.w;
            }
        }
        if (B_KMAJOR) {
#pragma unroll
            for (int s = 0; s < KH; ++s) fb[s] = Bs[(kofs + s) * LDR + wn + li];
        } else {
#pragma unroll
            for (int q = 0; q < KH / 4; ++q) {
                const float4 v = *(const float4*)(Bs + (wn + li) * LDK + kofs + 4 * q);
                fb[4 * q] = v.x; fb[4 * q + 1] = v.y; fb[4 * q + 2] = v.z; fb[4 * q + 3] = v.w;
            }
        }
#pragma unroll
        for (int s = 0; s < KH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], fb[s], acc, 0, 0, 0);
    }

    if (EPI == EPI_PLAIN && A_KMAJOR && do_bias && tid < GB_M && m0 + tid < g.M)
        g.dbias[(long)blockIdx.z * g.M + m0 + tid] = bsum;
    // combine the two k-groups: waves 4-7 park their accumulators in LDS, waves 0-3 add them
    __syncthreads();
    float* red = As;                                           // 4 quadrants x 64 lanes x 16 floats = 16 KB
    if (kg == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wq * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += red[(wq * 16 + r) * 64 + lane];
    // epilogue: acc[r] is C[row = wm + (r&3) + 8*(r>>2) + 4*lh][col = wn + li]
    float* C = g.C + (long)blockIdx.z * g.slab;
    const int col = n0 + wn + li;
    if (col >= g.N) return;
    float bias = 0.f;
    if (EPI == EPI_BIAS_ACT && g.bias) bias = g.bias[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < g.M) {
            float v = acc[r];
            if (EPI == EPI_BIAS_ACT) {
                v += bias;
                if (g.act == PM_ACT_TANH) v = pm_tanh(v);
            } else if (EPI == EPI_MUL_DACT) {
                if (g.act == PM_ACT_TANH) {
                    const float h = g.H[(long)row * g.ldh + col];
                    v *= (1.0f - h * h);
                }
            }
            C[(long)row * g.ldc + col] = v;
        }
    }
}

static inline int aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int pm_linear_fwd_f32(const float* X, long ldx, const float* W, long ldw, const float* b, float* Y,
                                 long ldy, int M, int N, int K, int act, void* stream) {
    PM_REQUIRE(X && W && Y && M > 0 && N > 0 && K > 0 && ldx >= K && ldw >= K && ldy >= N);
    PM_REQUIRE(act == PM_ACT_NONE || act == PM_ACT_TANH);
    Gemm2Group gg{};
    gg.n = 1;
    Gemm2Prob& g = gg.p[0];
    g.A = X; g.lda = ldx; g.B = W; g.ldb = ldw; g.C = Y; g.ldc = ldy; g.bias = b;
    g.M = M; g.N = N; g.K = K; g.act = act; g.epi = G2_EPI_BIAS_ACT; g.splits = 1;
    g.vecA = (K % 4 == 0) && (ldx % 4 == 0) && aligned16(X);
    g.vecB = (K % 4 == 0) && (ldw % 4 == 0) && aligned16(W);
    return gemm2_launch(gg, false, false, stream);
}

extern "C" int pm_linear_bwd_data_f32(const float* dY, long lddy, const float* W, long ldw, const float* H,
                                      long ldh, float* dX, long lddx, int M, int N, int K, int act,
                                      void* stream) {
    PM_REQUIRE(dY && W && dX && M > 0 && N > 0 && K > 0 && lddy >= N && ldw >= K && lddx >= K);
    PM_REQUIRE(act == PM_ACT_NONE || (act == PM_ACT_TANH && H && ldh >= K));
    Gemm2Group gg{};
    gg.n = 1;
    Gemm2Prob& g = gg.p[0];
    // C[M x K] = dY[M x N] * W[N x K]: reduction over N; A = dY (k-contiguous), B = W (k-major)
    g.A = dY; g.lda = lddy; g.B = W; g.ldb = ldw; g.C = dX; g.ldc = lddx; g.H = H; g.ldh = ldh;
    g.M = M; g.N = K; g.K = N; g.act = act; g.epi = G2_EPI_MUL_DACT; g.splits = 1;
    g.vecA = (N % 4 == 0) && (lddy % 4 == 0) && aligned16(dY);
    g.vecB = (K % 4 == 0) && (ldw % 4 == 0) && aligned16(W);
    return gemm2_launch(gg, false, true, stream);
}

// ---- weight gradient: split-K over the batch, slab reduction, bias column sums -----------
static inline int bww_splits(int M, int N, int K) {
    const long tiles = (long)((N + GB_M - 1) / GB_M) * ((K + GB_N - 1) / GB_N);
    // target work-group count of the split-K weight-gradient GEMM: one per CU for the mini-batch-sized reductions
    // (M <= 4096: 512 measured -2 %, 128 -12 % on state PPO), two per CU for the long reductions of the point-cloud glue
    long s = (M <= 4096 ? 256 : 512) / tiles;
    const long maxs = (M + GB_K - 1) / GB_K;
    if (s > maxs) s = maxs;
    // reductions over >= 256 K rows with one or two output tiles (conv1 of Conv3DNet: dW (16 x 125) over 7.9 M patch rows;
    // the PointNet++ glue: 64-128 wide over 0.5-2 M rows) get up to 256 slabs: with 32 the 7.9 M-row one ran on 64
    // work-groups at 0.66 TB/s (6.8 ms)
    if (s > (M >= 262144 ? 256 : 32)) s = (M >= 262144 ? 256 : 32);
    if (s < 1) s = 1;
    // every slab must own at least one row of the reduction once the chunk is rounded up to the K-step
    while (s > 1) {
        long kc = (M + s - 1) / s;
        kc = ((kc + GB_K - 1) / GB_K) * GB_K;
        if ((s - 1) * kc < M) break;
        --s;
    }
    return (int)s;
}

extern "C" size_t pm_linear_bwd_weight_workspace_bytes(int M, int N, int K) {
    const int s = bww_splits(M, N, K);
    return s > 1 ? (size_t)s * ((size_t)N * K + N) * sizeof(float) : 16;     // dW slabs + db slabs
}

// dW = sum_z slab_z (fixed order); the trailing N "elements" reduce the bias-gradient slabs.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int S, long slab,
                                                           float* __restrict__ dW, long lddw, int N, int K,
                                                           const float* __restrict__ bslabs, float* __restrict__ db) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    const long nk = (long)N * K;
    if (e < nk) {
        float s = 0.f;
#pragma unroll 4
        for (int z = 0; z < S; ++z) s += slabs[z * slab + e];
        dW[(e / K) * lddw + (e % K)] = s;
    } else if (db && e < nk + N) {
        const int n = (int)(e - nk);
        float s = 0.f;
        for (int z = 0; z < S; ++z) s += bslabs[(long)z * N + n];
        db[n] = s;
    }
}

extern "C" int pm_linear_bwd_weight_f32(const float* dY, long lddy, const float* X, long ldx, float* dW,
                                        long lddw, float* db, int M, int N, int K, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    PM_REQUIRE(dY && X && dW && M > 0 && N > 0 && K > 0 && lddy >= N && ldx >= K && lddw >= K);
    const int S = bww_splits(M, N, K);
    if (S > 1 && (!workspace || workspace_bytes < pm_linear_bwd_weight_workspace_bytes(M, N, K))) return PM_EWORKSPACE;
    Gemm2Group gg{};
    gg.n = 1;
    Gemm2Prob& g = gg.p[0];
    // C[N x K] = dY^T[N x M] * X[M x K]: reduction over M; both operands k-major
    g.A = dY; g.lda = lddy; g.B = X; g.ldb = ldx;
    g.M = N; g.N = K; g.K = M; g.act = 0; g.epi = G2_EPI_PLAIN;
    g.vecA = (N % 4 == 0) && (lddy % 4 == 0) && aligned16(dY);
    g.vecB = (K % 4 == 0) && (ldx % 4 == 0) && aligned16(X);
    float* bslabs = (float*)workspace + (size_t)S * N * K;
    if (S > 1) {
        int kchunk = (M + S - 1) / S;
        kchunk = ((kchunk + GB_K - 1) / GB_K) * GB_K;
        g.C = (float*)workspace; g.ldc = K; g.kchunk = kchunk; g.slab = (long)N * K; g.splits = S;
        g.dbias = db ? bslabs : nullptr; g.bslab = N;
        if ((long)(S - 1) * kchunk >= M) return PM_EINVAL;
    } else {
        g.C = dW; g.ldc = lddw; g.splits = 1; g.slab = 0;
        g.dbias = db; g.bslab = 0;                     // single slab: the column sums ARE the bias gradient
    }
    {
        const int rc = gemm2_launch(gg, true, true, stream);
        if (rc != PM_OK) return rc;
    }
    if (S > 1) {
        const long ne = (long)N * K + (db ? N : 0);
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, pm_stream(stream),
                           (const float*)workspace, S, (long)N * K, dW, lddw, N, K, (const float*)bslabs, db);
    }
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ---- grouped forms: several problems of one kind in ONE launch (include/partmanip_hip.h) -----------------------------
extern "C" int pm_linear_fwd_group_f32(int n, const pm_linear_fwd_desc* d, void* stream) {
    PM_REQUIRE(d && n >= 1 && n <= PM_LINEAR_GROUP_MAX);
    Gemm2Group gg{};
    gg.n = n;
    for (int i = 0; i < n; ++i) {
        const pm_linear_fwd_desc& q = d[i];
        PM_REQUIRE(q.X && q.W && q.Y && q.M > 0 && q.N > 0 && q.K > 0 && q.ldx >= q.K && q.ldw >= q.K && q.ldy >= q.N);
        PM_REQUIRE(q.act == PM_ACT_NONE || q.act == PM_ACT_TANH);
        Gemm2Prob& g = gg.p[i];
        g.A = q.X; g.lda = q.ldx; g.B = q.W; g.ldb = q.ldw; g.C = q.Y; g.ldc = q.ldy; g.bias = q.b;
        g.M = q.M; g.N = q.N; g.K = q.K; g.act = q.act; g.epi = G2_EPI_BIAS_ACT; g.splits = 1;
        g.vecA = (q.K % 4 == 0) && (q.ldx % 4 == 0) && aligned16(q.X);
        g.vecB = (q.K % 4 == 0) && (q.ldw % 4 == 0) && aligned16(q.W);
    }
    return gemm2_launch(gg, false, false, stream);
}

extern "C" int pm_linear_bwd_data_group_f32(int n, const pm_linear_bwd_data_desc* d, void* stream) {
    PM_REQUIRE(d && n >= 1 && n <= PM_LINEAR_GROUP_MAX);
    Gemm2Group gg{};
    gg.n = n;
    for (int i = 0; i < n; ++i) {
        const pm_linear_bwd_data_desc& q = d[i];
        PM_REQUIRE(q.dY && q.W && q.dX && q.M > 0 && q.N > 0 && q.K > 0 && q.lddy >= q.N && q.ldw >= q.K && q.lddx >= q.K);
        PM_REQUIRE(q.act == PM_ACT_NONE || (q.act == PM_ACT_TANH && q.H && q.ldh >= q.K));
        Gemm2Prob& g = gg.p[i];
        g.A = q.dY; g.lda = q.lddy; g.B = q.W; g.ldb = q.ldw; g.C = q.dX; g.ldc = q.lddx; g.H = q.H; g.ldh = q.ldh;
        g.M = q.M; g.N = q.K; g.K = q.N; g.act = q.act; g.epi = G2_EPI_MUL_DACT; g.splits = 1;
        g.vecA = (q.N % 4 == 0) && (q.lddy % 4 == 0) && aligned16(q.dY);
        g.vecB = (q.K % 4 == 0) && (q.ldw % 4 == 0) && aligned16(q.W);
    }
    return gemm2_launch(gg, false, true, stream);
}

extern "C" int pm_linear_bwd_weight_group_f32(int n, const pm_linear_bwd_weight_desc* d, int splits, void* stream) {
    PM_REQUIRE(d && n >= 1 && n <= PM_LINEAR_GROUP_MAX && splits >= 1);
    Gemm2Group gg{};
    gg.n = n;
    for (int i = 0; i < n; ++i) {
        const pm_linear_bwd_weight_desc& q = d[i];
        PM_REQUIRE(q.dY && q.X && q.dW && q.M > 0 && q.N > 0 && q.K > 0 && q.lddy >= q.N && q.ldx >= q.K && q.lddw >= q.K);
        Gemm2Prob& g = gg.p[i];
        g.A = q.dY; g.lda = q.lddy; g.B = q.X; g.ldb = q.ldx; g.C = q.dW; g.ldc = q.lddw;
        g.M = q.N; g.N = q.K; g.K = q.M; g.act = 0; g.epi = G2_EPI_PLAIN;
        g.vecA = (q.N % 4 == 0) && (q.lddy % 4 == 0) && aligned16(q.dY);
        g.vecB = (q.K % 4 == 0) && (q.ldx % 4 == 0) && aligned16(q.X);
        g.dbias = q.db;
        g.splits = splits;
        if (splits > 1) {
            int kchunk = (q.M + splits - 1) / splits;
            kchunk = ((kchunk + 31) / 32) * 32;
            PM_REQUIRE((long)(splits - 1) * kchunk < q.M && q.slab_stride > 0);
            g.kchunk = kchunk; g.slab = q.slab_stride; g.bslab = q.slab_stride;
        }
    }
    return gemm2_launch(gg, true, true, stream);
}
