// K1 GAE return scan, K2 advantage normalisation, K3 row gather.  HBM / latency bound
// scans: one lane per env, N contiguous => every wave load is a coalesced 256 B row.
#include "common.h"

// ---------------------------------------------------------------------------------- K1
// storage.py:96-112.  Thread n walks t = T-1 .. 0 keeping the running advantage in a
// register; loads of the next UNROLL steps are independent of the recurrence and are
// issued ahead of it.  All arithmetic uses explicit round-to-nearest ops (no FMA
// contraction) so the result is bit-identical to the reference's op-by-op tensors.
template <int UNROLL>
__global__ __launch_bounds__(64) void gae_scan_kernel(const float* __restrict__ rewards,
                                                       const float* __restrict__ values,
                                                       const uint8_t* __restrict__ dones,
                                                       const uint8_t* __restrict__ succs,
                                                       const float* __restrict__ last_values,
                                                       float* __restrict__ returns, float* __restrict__ advantages,
                                                       int T, int N, float gamma, float gamma_lam, int use_succ,
                                                       float succ_value) {
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n >= N) return;
    float adv = 0.0f;
    float nv = last_values[n];
    int t = T - 1;
    while (t >= 0) {
        const int cnt = (t + 1 < UNROLL) ? (t + 1) : UNROLL;
        float r[UNROLL], v[UNROLL];
        uint8_t d[UNROLL], s[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (u < cnt) {
                const long o = (long)(t - u) * N + n;
                r[u] = rewards[o];
                v[u] = values[o];
                d[u] = dones[o];
                s[u] = use_succ ? succs[o] : (uint8_t)0;
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (u < cnt) {
                const long o = (long)(t - u) * N + n;
                const float notdone = d[u] ? 0.0f : 1.0f;
                const float delta = sub_rn(add_rn(r[u], mul_rn(gamma, nv)), v[u]);
                adv = mul_rn(notdone, add_rn(delta, mul_rn(gamma_lam, adv)));
                float ret;
                if (use_succ) {
                    const float sf = s[u] ? 1.0f : 0.0f, nsf = s[u] ? 0.0f : 1.0f;
                    ret = add_rn(mul_rn(nsf, add_rn(adv, v[u])), mul_rn(sf, succ_value));
                } else {
                    ret = add_rn(adv, v[u]);
                }
                returns[o] = ret;
                advantages[o] = sub_rn(ret, v[u]);
                nv = v[u];
            }
        }
        t -= cnt;
    }
}

extern "C" int pm_gae_scan_f32(const float* rewards, const float* values, const uint8_t* dones,
                               const uint8_t* succs, const float* last_values, float* returns,
                               float* advantages, int T, int N, float gamma, float gamma_lam, int use_succ,
                               float succ_value, void* stream) {
    PM_REQUIRE(rewards && values && dones && last_values && returns && advantages);
    PM_REQUIRE(T > 0 && N > 0);
    PM_REQUIRE(!use_succ || succs);
    hipLaunchKernelGGL(gae_scan_kernel<8>, dim3((N + 63) / 64), dim3(64), 0, pm_stream(stream), rewards, values,
                       dones, succs, last_values, returns, advantages, T, N, gamma, gamma_lam, use_succ,
                       succ_value);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ---------------------------------------------------------------------------------- K2
// Two-stage deterministic fp64 reduction of {sum x, sum x^2}; fp64 because the CPU
// reference accumulates mean/std of fp32 tensors in double (acc_type<float,false>).
static constexpr int MOM_THREADS = 256;
static constexpr int MOM_MAX_BLOCKS = 1024;

static inline int moments_blocks(long n) {
    long b = (n + (long)MOM_THREADS * 8 - 1) / ((long)MOM_THREADS * 8);
    if (b < 1) b = 1;
    if (b > MOM_MAX_BLOCKS) b = MOM_MAX_BLOCKS;
    return (int)b;
}

__global__ __launch_bounds__(MOM_THREADS) void moments_partial_kernel(const float* __restrict__ x, long n,
                                                                       double* __restrict__ part) {
    __shared__ double red[MOM_THREADS / 64];
    double s = 0.0, q = 0.0;
    for (long i = (long)blockIdx.x * MOM_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * MOM_THREADS) {
        const double v = (double)x[i];
        s += v;
        q += v * v;
    }
    s = block_sum<double, MOM_THREADS>(s, red);
    q = block_sum<double, MOM_THREADS>(q, red);
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = s;
        part[2 * blockIdx.x + 1] = q;
    }
}

__global__ __launch_bounds__(MOM_THREADS) void moments_final_kernel(const double* __restrict__ part, int nb,
                                                                     double* __restrict__ out) {
    __shared__ double red[MOM_THREADS / 64];
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < nb; i += MOM_THREADS) {
        s += part[2 * i];
        q += part[2 * i + 1];
    }
    s = block_sum<double, MOM_THREADS>(s, red);
    q = block_sum<double, MOM_THREADS>(q, red);
    if (threadIdx.x == 0) {
        out[0] = s;
        out[1] = q;
    }
}

extern "C" size_t pm_moments_workspace_bytes(long n) { return (size_t)moments_blocks(n) * 2 * sizeof(double); }

extern "C" int pm_moments_f64(const float* x, long n, double* moments, void* workspace, size_t workspace_bytes,
                              void* stream) {
    PM_REQUIRE(x && moments && n > 0);
    const int nb = moments_blocks(n);
    if (!workspace || workspace_bytes < (size_t)nb * 2 * sizeof(double)) return PM_EWORKSPACE;
    hipLaunchKernelGGL(moments_partial_kernel, dim3(nb), dim3(MOM_THREADS), 0, pm_stream(stream), x, n,
                       (double*)workspace);
    hipLaunchKernelGGL(moments_final_kernel, dim3(1), dim3(MOM_THREADS), 0, pm_stream(stream),
                       (const double*)workspace, nb, moments);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// x <- (x - mean) / (std_unbiased + eps); mean/std are rounded to fp32 first, as the
// reference's `.mean()` / `.std()` return fp32 tensors (storage.py:114, ppo.py:329).
__global__ __launch_bounds__(256) void normalize_apply_kernel(float* __restrict__ x, long n,
                                                               const double* __restrict__ mom, double count,
                                                               float eps) {
    const double mean_d = mom[0] / count;
    double var = (mom[1] - mom[0] * mean_d) / (count - 1.0);
    if (var < 0.0) var = 0.0;
    const float mean = (float)mean_d;
    const float denom = add_rn((float)sqrt(var), eps);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        x[i] = __fdiv_rn(sub_rn(x[i], mean), denom);
}

extern "C" int pm_normalize_apply_f32(float* x, long n, const double* moments, double count, float eps,
                                      void* stream) {
    PM_REQUIRE(x && moments && n > 0 && count > 1.0);
    long b = (n + 255) / 256;
    if (b > 2048) b = 2048;
    hipLaunchKernelGGL(normalize_apply_kernel, dim3((int)b), dim3(256), 0, pm_stream(stream), x, n, moments, count,
                       eps);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ---------------------------------------------------------------------------------- K3
// dst[i,:] = src[idx[i],:].  One wave per row chunk; float4 when rows are 16 B aligned.
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src,
                                                           const int64_t* __restrict__ idx,
                                                           float* __restrict__ dst, long n_rows, long row_elems,
                                                           long src_ld, long dst_ld, int vec4) {
    const long row = blockIdx.x;
    if (row >= n_rows) return;
    const float* s = src + idx[row] * src_ld;
    float* d = dst + row * dst_ld;
    if (vec4) {
        const float4* s4 = (const float4*)s;
        float4* d4 = (float4*)d;
        for (long i = (long)blockIdx.y * 256 + threadIdx.x; i < row_elems / 4; i += (long)gridDim.y * 256)
            d4[i] = s4[i];
    } else {
        for (long i = (long)blockIdx.y * 256 + threadIdx.x; i < row_elems; i += (long)gridDim.y * 256) d[i] = s[i];
    }
}

extern "C" int pm_gather_rows_f32(const float* src, const int64_t* idx, float* dst, long n_rows, long row_elems,
                                  long src_ld, long dst_ld, void* stream) {
    PM_REQUIRE(src && idx && dst && n_rows > 0 && row_elems > 0 && src_ld >= row_elems && dst_ld >= row_elems);
    const int vec4 = (row_elems % 4 == 0) && (src_ld % 4 == 0) && (dst_ld % 4 == 0) &&
                     (((uintptr_t)src & 15) == 0) && (((uintptr_t)dst & 15) == 0);
    long per = vec4 ? row_elems / 4 : row_elems;
    int gy = (int)((per + 1023) / 1024);
    if (gy < 1) gy = 1;
    if (gy > 64) gy = 64;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)n_rows, gy), dim3(256), 0, pm_stream(stream), src, idx,
                       dst, n_rows, row_elems, src_ld, dst_ld, vec4);
    PM_CHECK_LAUNCH();
    return PM_OK;
}
