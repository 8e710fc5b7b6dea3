// K4/K5: Linear forward / backward entry points (C ABI).  The GEMMs run on the grouped fp32-MFMA kernels of gemm2_f32.hip
// (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate -- bitwise an fmaf chain per output, so results are
// fp32-roundoff-class against the reference's CPU sgemm):
//   fwd        Y  = act(X W^T + b)          A = X (k-contiguous), B = W (k-contiguous), bias + activation epilogue
//   bwd_data   dX = (dY W) .* act'(H)       A = dY (k-contiguous), B = W (k-major), derivative-of-activation epilogue
//   bwd_weight dW = dY^T X, db = colsum dY  A = dY, B = X (both k-major), split-K over the batch into slabs, bias
//              gradient from the staged dY tiles; slabs are summed here in fixed order (single-problem call) or by the
//              optimiser launch (grouped call, adam.hip).
#include "common.h"
#include <cstdlib>
#include "gemm2.h"
#include "skinny.h"

#define GB_M 64                       // tile / K-step granularity the split-K heuristic below counts in
#define GB_N 64
#define GB_K 64

static inline int aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }


// ---- skinny layers (N <= 16 outputs: the policy / value heads, 512 -> 10 and 512 -> 1) ----------------------------------
// As GEMMs these are a 64-wide tile with 10 (or 1) live columns on 32 work-groups walking a 16-step K loop: 17 / 8.4 us per
// forward / data-gradient launch in the state-PPO step, where the whole network step should take ~45.  They are row-wise dot
// products (forward: 8.3 us) and a rank-N update per row (data gradient: 5.8 us): memory-bound VALU kernels over all CUs,
// fp32 FMA chains with a fixed (deterministic) summation order.  (The weight gradient, an N x K reduction over the batch,
// stays on the split-K GEMM: a per-work-group-partials VALU version measured 25 us against its 10 + 5.)
// Y[i][n] = act(sum_k X[i][k] W[n][k] + b[n]); one wave per row, lanes over k (float4 per lane per 256-wide chunk)
__global__ __launch_bounds__(SK_T) void skinny_fwd_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ W, long ldw,
                                                          const float* __restrict__ b, float* __restrict__ Y, long ldy, int M, int N,
                                                          int K, int act) {
    extern __shared__ __attribute__((aligned(16))) float sW[];                  // [N][K]
    sk_fill_w(sW, W, ldw, N, K);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = blockIdx.x * 4 + wave; i < M; i += gridDim.x * 4) {
        float acc[SK_MAXN];
        sk_row_dot(X + (long)i * ldx, sW, N, K, lane, acc);
#pragma unroll
        for (int n = 0; n < SK_MAXN; ++n)
            if (n < N) {
                const float s = wave_sum(acc[n]);
                if (lane == n) Y[(long)i * ldy + n] = pm_act(s + (b ? b[n] : 0.f), act);
            }
    }
}
// dX[i][k] = (sum_n dY[i][n] W[n][k]) * dact(H[i][k])
__global__ __launch_bounds__(SK_T) void skinny_dgrad_kernel(const float* __restrict__ dY, long lddy, const float* __restrict__ W, long ldw,
                                                            const float* __restrict__ H, long ldh, float* __restrict__ dX, long lddx,
                                                            int M, int N, int K, int act) {
    extern __shared__ __attribute__((aligned(16))) float sW[];                  // [N][K]
    sk_fill_w(sW, W, ldw, N, K);
    __syncthreads();
    const int k4 = K >> 2;
    const long total = (long)M * k4;
    for (long e = (long)blockIdx.x * SK_T + threadIdx.x; e < total; e += (long)gridDim.x * SK_T) {
        const long i = e / k4;
        const int k = (int)(e - i * k4) * 4;
        const float* dyrow = dY + i * lddy;
        *(float4*)(dX + i * lddx + k) = sk_dgrad4([&](int n) { return dyrow[n]; }, sW, N, K, k, H + i * ldh, act);
    }
}


extern "C" int pm_linear_fwd_f32(const float* X, long ldx, const float* W, long ldw, const float* b, float* Y,
                                 long ldy, int M, int N, int K, int act, void* stream) {
    PM_REQUIRE(X && W && Y && M > 0 && N > 0 && K > 0 && ldx >= K && ldw >= K && ldy >= N);
    PM_REQUIRE(act >= PM_ACT_NONE && act <= PM_ACT_MAX);
    if (M >= 256 && skinny_ok(N, K, X, ldx, W, ldw)) {
        int nb = (M + 3) / 4;                                        // one row per wave (8 rows per work-group measured slower)
        if (nb > 1024) nb = 1024;
        hipLaunchKernelGGL(skinny_fwd_kernel, dim3(nb), dim3(SK_T), (size_t)N * K * 4, pm_stream(stream), X, ldx, W, ldw, b, Y, ldy, M,
                           N, K, act);
        PM_CHECK_LAUNCH();
        return PM_OK;
    }
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
    PM_REQUIRE(act == PM_ACT_NONE || (act > PM_ACT_NONE && act <= PM_ACT_MAX && H && ldh >= K));
    if (M >= 256 && skinny_ok(N, K, W, ldw, dX, lddx) && (act == PM_ACT_NONE || (ldh % 4 == 0 && aligned16(H)))) {
        long nb = ((long)M * (K / 4) + SK_T - 1) / SK_T;
        if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(skinny_dgrad_kernel, dim3((unsigned)nb), dim3(SK_T), (size_t)N * K * 4, pm_stream(stream), dY, lddy, W, ldw,
                           H, ldh, dX, lddx, M, N, K, act);
        PM_CHECK_LAUNCH();
        return PM_OK;
    }
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
    // (<= 32 output rows: the LDS-DMA kernels' 32 x 128 tile, gemm2_f32.hip LAY = 1)
    const long tiles = (N <= 32 && K > 64) ? (K + 127) / 128 : (long)((N + GB_M - 1) / GB_M) * ((K + GB_N - 1) / GB_N);
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
    // long reductions are MFMA-bound and two or three of their work-groups share a CU: the launch takes as long as the fullest
    // CU, i.e. ceil(work-groups / 256) rounds -- pick the slab count near the target that fills its last round best (108 tiles:
    // 4 slabs = 432 work-groups = 2 rounds at 84 %; 7 slabs = 756 = 3 rounds at 98 %)
    if (M >= 262144 && tiles < 256) {
        const long s0 = s;
        double best = 0.0;
        for (long c = (s0 + 1) / 2; c <= 2 * s0 && c <= 256 && c <= maxs; ++c) {
            const long w = tiles * c;
            const double eff = (double)w / (256.0 * ((w + 255) / 256));
            if (eff > best + 0.02 || (eff > best - 1e-9 && labs(c - s0) < labs(s - s0))) {
                if (eff > best) best = eff;
                s = c;
            }
        }
    }
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

// dW = sum_z slab_z; the trailing N "elements" reduce the bias-gradient slabs.  A work-group owns 64 consecutive elements; its
// SR_G waves take the slabs z = w, w + SR_G, ... (chains of S / SR_G loads instead of S: 256 slabs of a 128 x 128 gradient took
// 75 us on 64 single-chain work-groups -- 3.6 % of a PointNet++ step), the partial sums are added in wave order through LDS:
// a fixed order, whatever the launch.
#define SR_G 8
__global__ __launch_bounds__(64 * SR_G) void slab_reduce_kernel(const float* __restrict__ slabs, int S, long slab,
                                                                float* __restrict__ dW, long lddw, int N, int K,
                                                                const float* __restrict__ bslabs, float* __restrict__ db) {
    __shared__ float part[SR_G][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long e = (long)blockIdx.x * 64 + lane;
    const long nk = (long)N * K;
    float s = 0.f;
    if (e < nk) {
#pragma unroll 4
        for (int z = w; z < S; z += SR_G) s += slabs[z * slab + e];
    } else if (db && e < nk + N) {
        const int n = (int)(e - nk);
        for (int z = w; z < S; z += SR_G) s += bslabs[(long)z * N + n];
    }
    part[w][lane] = s;
    __syncthreads();
    if (w == 0) {
        float t = part[0][lane];
#pragma unroll
        for (int q = 1; q < SR_G; ++q) t += part[q][lane];
        if (e < nk) dW[(e / K) * lddw + (e % K)] = t;
        else if (db && e < nk + N) db[e - nk] = t;
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
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((ne + 63) / 64)), dim3(64 * SR_G), 0, pm_stream(stream),
                           (const float*)workspace, S, (long)N * K, dW, lddw, N, K, (const float*)bslabs, db);
    }
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ---- grouped forms: several problems of one kind in ONE launch (include/partmanip_hip.h) -----------------------------
// ---- whole-MLP forms (SURVEY.md 8(b) names pm_mlp_fwd_f32 / pm_mlp_bwd_f32) ------------------------------------------------
// network.py:27-54 `MLP`: n_layers Linear layers, the activation after every layer but the last.  dims[0..n_layers] are the
// layer widths (dims[0] = input features); W / b / H / dW / db are HOST arrays of n_layers DEVICE pointers; H[l] is the
// (M, dims[l+1]) output of layer l (H[n_layers-1] = the network output).  These are the per-layer entry points above, issued
// in order on one stream: same kernels, same bits -- the learner itself uses the grouped / fused-head forms.
extern "C" int pm_mlp_fwd_f32(const float* X, long ldx, int M, int n_layers, const int* dims, const float* const* W,
                              const float* const* b, int act, float* const* H, void* stream) {
    PM_REQUIRE(X && dims && W && H && M > 0 && n_layers > 0 && n_layers <= 64);
    const float* in = X;
    long ldin = ldx;
    for (int l = 0; l < n_layers; ++l) {
        PM_REQUIRE(W[l] && H[l] && dims[l] > 0 && dims[l + 1] > 0);
        const int rc = pm_linear_fwd_f32(in, ldin, W[l], dims[l], b ? b[l] : nullptr, H[l], dims[l + 1], M, dims[l + 1], dims[l],
                                         l + 1 < n_layers ? act : PM_ACT_NONE, stream);
        if (rc != PM_OK) return rc;
        in = H[l];
        ldin = dims[l + 1];
    }
    return PM_OK;
}

static size_t mlp_bwd_scratch_elems(int M, int n_layers, const int* dims) {
    int mx = 0;
    for (int l = 1; l < n_layers; ++l) mx = dims[l] > mx ? dims[l] : mx;      // hidden widths: dz ping-pong buffers
    return ((size_t)M * mx + 3) / 4 * 4;
}

extern "C" size_t pm_mlp_bwd_workspace_bytes(int M, int n_layers, const int* dims) {
    if (!dims || M <= 0 || n_layers <= 0) return 0;
    size_t ww = 16;
    for (int l = 0; l < n_layers; ++l) {
        const size_t w = pm_linear_bwd_weight_workspace_bytes(M, dims[l + 1], dims[l]);
        ww = w > ww ? w : ww;
    }
    return 2 * mlp_bwd_scratch_elems(M, n_layers, dims) * sizeof(float) + ((ww + 15) & ~(size_t)15);
}

// dY (M, dims[n_layers]) = d loss / d output -> dW[l] (dims[l+1], dims[l]), db[l] (may be NULL per layer or as a whole) and,
// if dX != NULL, d loss / d input (M, dims[0]).  X / H as passed to (returned by) pm_mlp_fwd_f32.
extern "C" int pm_mlp_bwd_f32(const float* X, long ldx, int M, int n_layers, const int* dims, const float* const* W,
                              const float* const* H, int act, const float* dY, float* const* dW, float* const* db, float* dX,
                              void* workspace, size_t workspace_bytes, void* stream) {
    PM_REQUIRE(X && dims && W && H && dY && dW && M > 0 && n_layers > 0 && n_layers <= 64);
    if (!workspace || workspace_bytes < pm_mlp_bwd_workspace_bytes(M, n_layers, dims)) return PM_EWORKSPACE;
    if ((uintptr_t)workspace & 15) return PM_EALIGN;
    const size_t se = mlp_bwd_scratch_elems(M, n_layers, dims);
    float* dz[2] = {(float*)workspace, (float*)workspace + se};
    void* wws = (float*)workspace + 2 * se;
    const size_t wwb = workspace_bytes - 2 * se * sizeof(float);
    const float* cur = dY;
    for (int l = n_layers - 1; l >= 0; --l) {
        const float* in = l ? H[l - 1] : X;
        const long ldin = l ? dims[l] : ldx;
        PM_REQUIRE(W[l] && dW[l]);
        int rc = pm_linear_bwd_weight_f32(cur, dims[l + 1], in, ldin, dW[l], dims[l], db ? db[l] : nullptr, M, dims[l + 1], dims[l],
                                          wws, wwb, stream);
        if (rc != PM_OK) return rc;
        if (l > 0) {                                       // through the layer and the activation that produced its input
            float* nxt = dz[l & 1];
            rc = pm_linear_bwd_data_f32(cur, dims[l + 1], W[l], dims[l], H[l - 1], dims[l], nxt, dims[l], M, dims[l + 1], dims[l], act,
                                        stream);
            if (rc != PM_OK) return rc;
            cur = nxt;
        } else if (dX) {
            rc = pm_linear_bwd_data_f32(cur, dims[1], W[0], dims[0], nullptr, 0, dX, dims[0], M, dims[1], dims[0], PM_ACT_NONE, stream);
            if (rc != PM_OK) return rc;
        }
    }
    return PM_OK;
}

extern "C" int pm_linear_fwd_group_f32(int n, const pm_linear_fwd_desc* d, void* stream) {
    PM_REQUIRE(d && n >= 1 && n <= PM_LINEAR_GROUP_MAX);
    Gemm2Group gg{};
    gg.n = n;
    for (int i = 0; i < n; ++i) {
        const pm_linear_fwd_desc& q = d[i];
        PM_REQUIRE(q.X && q.W && q.Y && q.M > 0 && q.N > 0 && q.K > 0 && q.ldx >= q.K && q.ldw >= q.K && q.ldy >= q.N);
        PM_REQUIRE(q.act >= PM_ACT_NONE && q.act <= PM_ACT_MAX);
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
        PM_REQUIRE(q.act == PM_ACT_NONE || (q.act > PM_ACT_NONE && q.act <= PM_ACT_MAX && q.H && q.ldh >= q.K));
        Gemm2Prob& g = gg.p[i];
        g.A = q.dY; g.lda = q.lddy; g.B = q.W; g.ldb = q.ldw; g.C = q.dX; g.ldc = q.lddx; g.H = q.H; g.ldh = q.ldh;
        g.M = q.M; g.N = q.K; g.K = q.N; g.act = q.act; g.epi = G2_EPI_MUL_DACT; g.splits = 1;
        g.vecA = (q.N % 4 == 0) && (q.lddy % 4 == 0) && aligned16(q.dY);
        g.vecB = (q.K % 4 == 0) && (q.ldw % 4 == 0) && aligned16(q.W);
    }
    return gemm2_launch(gg, false, true, stream);
}

// ---- chains: consecutive layers of ONE network in one launch (gemm2.h: Gemm2Chain) ----------------------------------------
extern "C" size_t pm_linear_chain_workspace_bytes(int M) { return (size_t)((M + 63) / 64 + 1) * sizeof(unsigned long long); }

extern "C" int pm_linear_fwd_chain_f32(int n, const pm_linear_fwd_desc* d, void* workspace, size_t workspace_bytes, void* stream) {
    PM_REQUIRE(d && n >= 2 && n <= GEMM2_CHAIN_MAX && workspace);
    if (workspace_bytes < pm_linear_chain_workspace_bytes(d[0].M)) return PM_EWORKSPACE;
    if (((uintptr_t)workspace & 7) != 0) return PM_EALIGN;
    Gemm2Chain ch{};
    ch.n = n;
    ch.bar = (unsigned long long*)workspace;
    for (int i = 0; i < n; ++i) {
        const pm_linear_fwd_desc& q = d[i];
        PM_REQUIRE(q.X && q.W && q.Y && q.M > 0 && q.N > 0 && q.K > 0 && q.ldx >= q.K && q.ldw >= q.K && q.ldy >= q.N);
        PM_REQUIRE(q.act >= PM_ACT_NONE && q.act <= PM_ACT_MAX);
        Gemm2Prob& g = ch.p[i];
        g.A = q.X; g.lda = q.ldx; g.B = q.W; g.ldb = q.ldw; g.C = q.Y; g.ldc = q.ldy; g.bias = q.b;
        g.M = q.M; g.N = q.N; g.K = q.K; g.act = q.act; g.epi = G2_EPI_BIAS_ACT;
        g.vecA = (q.K % 4 == 0) && (q.ldx % 4 == 0) && aligned16(q.X);
        g.vecB = (q.K % 4 == 0) && (q.ldw % 4 == 0) && aligned16(q.W);
    }
    return gemm2_chain_launch(ch, false, stream);
}

extern "C" int pm_linear_bwd_data_chain_f32(int n, const pm_linear_bwd_data_desc* d, void* workspace, size_t workspace_bytes,
                                            void* stream) {
    PM_REQUIRE(d && n >= 2 && n <= GEMM2_CHAIN_MAX && workspace);
    if (workspace_bytes < pm_linear_chain_workspace_bytes(d[0].M)) return PM_EWORKSPACE;
    if (((uintptr_t)workspace & 7) != 0) return PM_EALIGN;
    Gemm2Chain ch{};
    ch.n = n;
    ch.bar = (unsigned long long*)workspace;
    for (int i = 0; i < n; ++i) {
        const pm_linear_bwd_data_desc& q = d[i];
        PM_REQUIRE(q.dY && q.W && q.dX && q.M > 0 && q.N > 0 && q.K > 0 && q.lddy >= q.N && q.ldw >= q.K && q.lddx >= q.K);
        PM_REQUIRE(q.act == PM_ACT_NONE || (q.act > PM_ACT_NONE && q.act <= PM_ACT_MAX && q.H && q.ldh >= q.K));
        Gemm2Prob& g = ch.p[i];
        g.A = q.dY; g.lda = q.lddy; g.B = q.W; g.ldb = q.ldw; g.C = q.dX; g.ldc = q.lddx; g.H = q.H; g.ldh = q.ldh;
        g.M = q.M; g.N = q.K; g.K = q.N; g.act = q.act; g.epi = G2_EPI_MUL_DACT;
        g.vecA = (q.N % 4 == 0) && (q.lddy % 4 == 0) && aligned16(q.dY);
        g.vecB = (q.K % 4 == 0) && (q.ldw % 4 == 0) && aligned16(q.W);
    }
    return gemm2_chain_launch(ch, true, stream);
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
        // loadable columns of the dY / X rows (0 = N / K): see Gemm2Prob::Mld
        const int ncols = q.dy_cols ? q.dy_cols : q.N, kcols = q.x_cols ? q.x_cols : q.K;
        PM_REQUIRE(ncols >= q.N && q.lddy >= ncols && kcols >= q.K && q.ldx >= kcols);
        g.Mld = ncols; g.Nld = kcols;
        g.vecA = (ncols % 4 == 0) && (q.lddy % 4 == 0) && aligned16(q.dY);
        g.vecB = (kcols % 4 == 0) && (q.ldx % 4 == 0) && aligned16(q.X);
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

// ---- sparse convolution = gathered-operand GEMM (sparse_voxel.hip builds the tables) -----------------------------------
// forward:  Y[r][n] = act(sum_{j,c} src[idx[r][j]][c] * W[n][j*C + c] + b[n])        (absent neighbours, idx < 0, contribute 0)
// weight:   dW[n][j*C + c] = sum_r dY[r][n] * src[idx[r][j]][c],  db[n] = sum_r dY[r][n]
// The (rows x J*C) operand is gathered by the GEMM's LDS-DMA loader and never exists in HBM.
extern "C" int pm_sparse_conv_fwd_f32(const float* src, long lds, const int32_t* idx, long rows, int J, int C, const float* W,
                                      long ldw, const float* b, float* Y, long ldy, int N, int act, const float* zero, void* stream) {
    PM_REQUIRE(src && idx && W && Y && zero && rows > 0 && rows < 0x7fffffffL && J > 0 && C > 0 && N > 0 && lds >= C && ldw >= (long)J * C &&
               ldy >= N && act >= PM_ACT_NONE && act <= PM_ACT_MAX);
    PM_REQUIRE(C % 4 == 0 && (J * C) % 32 == 0 && lds % 4 == 0 && ldw % 4 == 0);
    if (!aligned16(src) || !aligned16(W) || !aligned16(zero)) return PM_EALIGN;
    Gemm2Group gg{};
    gg.n = 1;
    Gemm2Prob& g = gg.p[0];
    g.A = src; g.lda = lds; g.B = W; g.ldb = ldw; g.C = Y; g.ldc = ldy; g.bias = b;
    g.M = (int)rows; g.N = N; g.K = J * C; g.act = act; g.epi = G2_EPI_BIAS_ACT; g.splits = 1;
    g.vecA = g.vecB = 1;
    g.gidx = idx; g.gzero = zero; g.gJ = J; g.gC = C;
    return gemm2_launch(gg, false, false, stream);
}

// data gradient of a submanifold convolution as a gathered GEMM of its own (no (rows x J*C_in) column matrix in HBM):
//   dX[s][ci] = (sum_{j,co} dY[idxT[s][j]][co] * Wt[ci][j*Cout + co]) * act'(H[s][ci])
// with the MIRRORED neighbour table idxT[s][j] = idx[s][J-1-j] (row r reads s as its neighbour j  <=>  s reads r as its
// neighbour J-1-j; rows that are nobody's neighbour -- duplicates -- carry -1 everywhere) and the per-tap transposed weight
// view Wt[ci][j*Cout + co] = W[co][j*Cin + ci] (row idxT[s][j] read s through ITS tap j), both built by the caller.  H = the layer input's activation (NULL / act
// NONE: no factor).
extern "C" int pm_sparse_conv_bwd_data_f32(const float* dY, long lddy, const int32_t* idxT, long rows, int J, int Cout,
                                           const float* Wt, long ldwt, const float* H, long ldh, float* dX, long lddx, int Cin,
                                           int act, const float* zero, void* stream) {
    PM_REQUIRE(dY && idxT && Wt && dX && zero && rows > 0 && rows < 0x7fffffffL && J > 0 && Cout > 0 && Cin > 0 && lddy >= Cout &&
               ldwt >= (long)J * Cout && lddx >= Cin && act >= PM_ACT_NONE && act <= PM_ACT_MAX);
    PM_REQUIRE(Cout % 4 == 0 && (J * Cout) % 32 == 0 && lddy % 4 == 0 && ldwt % 4 == 0);
    PM_REQUIRE(!H || ldh >= Cin);
    if (!aligned16(dY) || !aligned16(Wt) || !aligned16(zero)) return PM_EALIGN;
    Gemm2Group gg{};
    gg.n = 1;
    Gemm2Prob& g = gg.p[0];
    g.A = dY; g.lda = lddy; g.B = Wt; g.ldb = ldwt; g.C = dX; g.ldc = lddx;
    g.M = (int)rows; g.N = Cin; g.K = J * Cout; g.splits = 1;
    g.epi = (H && act != PM_ACT_NONE) ? G2_EPI_MUL_DACT : G2_EPI_PLAIN;
    g.act = act; g.H = H; g.ldh = ldh;
    g.vecA = g.vecB = 1;
    g.gidx = idxT; g.gzero = zero; g.gJ = J; g.gC = Cout;
    return gemm2_launch(gg, false, false, stream);
}

// data gradient of a convolution whose patches do not overlap (stride == kernel size), by SCATTER: output row r wrote to /
// read from input row idx[r][j] through tap j and nobody else did, so
//   dX[idx[r][j]][c] = (sum_co dY[r][co] * W[co][j*C + c]) * act'(H[idx[r][j]][c])         (idx < 0: dropped)
// is one plain GEMM (K = Cout) whose epilogue stores each 16-byte piece at its destination row -- no column-gradient matrix,
// no col2im pass.  W is the layer's tap-major weight (Cout x J*C), the forward's B operand as it is.  Destination rows that
// no (r, j) maps to keep their contents (the caller zero-fills when the geometry leaves any).
extern "C" int pm_sparse_conv_bwd_data_scatter_f32(const float* dY, long lddy, const float* W, long ldw, const int32_t* idx,
                                                   long rows, int J, int C, int Cout, const float* H, float* dX, int act,
                                                   void* stream) {
    PM_REQUIRE(dY && W && idx && dX && rows > 0 && rows < 0x7fffffffL && J > 0 && C > 0 && Cout > 0 && lddy >= Cout &&
               ldw >= (long)J * C && act >= PM_ACT_NONE && act <= PM_ACT_MAX);
    PM_REQUIRE(C % 4 == 0 && Cout % 4 == 0 && lddy % 4 == 0 && ldw % 4 == 0);
    if (!aligned16(dY) || !aligned16(W) || !aligned16(dX) || (H && !aligned16(H))) return PM_EALIGN;
    Gemm2Group gg{};
    gg.n = 1;
    Gemm2Prob& g = gg.p[0];
    g.A = dY; g.lda = lddy; g.B = W; g.ldb = ldw; g.C = dX; g.ldc = C;
    g.M = (int)rows; g.N = J * C; g.K = Cout; g.splits = 1;
    g.epi = (H && act != PM_ACT_NONE) ? G2_EPI_MUL_DACT : G2_EPI_PLAIN;
    g.act = act; g.H = H; g.ldh = C;
    g.vecA = g.vecB = 1;
    g.sidx = idx; g.sJ = J; g.sC = C;
    return gemm2_launch(gg, false, true, stream);
}

extern "C" size_t pm_sparse_conv_bwd_weight_workspace_bytes(long rows, int N, int J, int C) {
    return pm_linear_bwd_weight_workspace_bytes((int)rows, N, J * C);
}

extern "C" int pm_sparse_conv_bwd_weight_f32(const float* dY, long lddy, const float* src, long lds, const int32_t* idx, long rows,
                                             int J, int C, float* dW, long lddw, float* db, int N, const float* zero,
                                             void* workspace, size_t workspace_bytes, void* stream) {
    PM_REQUIRE(dY && src && idx && dW && zero && rows > 0 && rows < 0x7fffffffL && J > 0 && C > 0 && N > 0 && lds >= C && lddy >= N &&
               lddw >= (long)J * C);
    PM_REQUIRE(C % 4 == 0 && N % 4 == 0 && lds % 4 == 0 && lddy % 4 == 0);
    if (!aligned16(src) || !aligned16(dY) || !aligned16(zero)) return PM_EALIGN;
    const int M = (int)rows, K = J * C;
    const int S = bww_splits(M, N, K);
    if (S > 1 && (!workspace || workspace_bytes < pm_linear_bwd_weight_workspace_bytes(M, N, K))) return PM_EWORKSPACE;
    Gemm2Group gg{};
    gg.n = 1;
    Gemm2Prob& g = gg.p[0];
    g.A = dY; g.lda = lddy; g.B = src; g.ldb = lds;
    g.M = N; g.N = K; g.K = M; g.act = 0; g.epi = G2_EPI_PLAIN;
    g.vecA = g.vecB = 1;
    g.gidx = idx; g.gzero = zero; g.gJ = J; g.gC = C;
    float* bslabs = (float*)workspace + (size_t)S * N * K;
    if (S > 1) {
        int kchunk = (M + S - 1) / S;
        kchunk = ((kchunk + GB_K - 1) / GB_K) * GB_K;
        g.C = (float*)workspace; g.ldc = K; g.kchunk = kchunk; g.slab = (long)N * K; g.splits = S;
        g.dbias = db ? bslabs : nullptr; g.bslab = N;
    } else {
        g.C = dW; g.ldc = lddw; g.splits = 1; g.kchunk = ((M + 31) / 32) * 32; g.slab = 0;
        g.dbias = db; g.bslab = 0;
    }
    const int rc = gemm2_launch(gg, true, true, stream);
    if (rc != PM_OK) return rc;
    if (S > 1) {
        const long ne = (long)N * K + (db ? N : 0);
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((ne + 63) / 64)), dim3(64 * SR_G), 0, pm_stream(stream),
                           (const float*)workspace, S, (long)N * K, dW, lddw, N, K, (const float*)bslabs, db);
        PM_CHECK_LAUNCH();
    }
    return PM_OK;
}
