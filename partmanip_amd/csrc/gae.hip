// K1 GAE return scan, K2 advantage normalisation, K3 row gather.  HBM / latency bound.
#include "common.h"
#include <stdlib.h>

// ---------------------------------------------------------------------------------- K1
// storage.py:96-112.  The recurrence  adv_t = notdone_t * (delta_t + gamma*lam * adv_{t+1})  is the only serial part of
// the scan: delta_t = r_t + gamma * V_{t+1} - V_t and the two outputs are element-wise.  A work-group owns GAE_E
// consecutive envs and walks T in chunks of GAE_TC steps from the end:
//   phase 1 (256 threads): coalesced loads of r, V, V_next, done -> delta, notdone, V into LDS;
//   phase 2 (GAE_E lanes): the dependent chain over the chunk, three dependent fp32 ops per step, operands from LDS;
//   phase 3 (256 threads): returns / advantages from the chain's values, coalesced stores.
// Round 1 ran the whole scan as one lane per env (loads, stores and twelve ops inside the chain, 64 single-wave
// work-groups on 64 CUs): 40-55 us for 4096 x 128.  With 16 envs per work-group every CU takes part in the
// element-wise phases.  All arithmetic is the reference's op-by-op sequence in explicit round-to-nearest ops (no FMA
// contraction), evaluated per element exactly as before: the result stays bit-identical to the reference's tensors.
#define GAE_TC 128
template <int GAE_E>
__global__ __launch_bounds__(256) void gae_scan_kernel(const float* __restrict__ rewards,
                                                        const float* __restrict__ values,
                                                        const uint8_t* __restrict__ dones,
                                                        const uint8_t* __restrict__ succs,
                                                        const float* __restrict__ last_values,
                                                        float* __restrict__ returns, float* __restrict__ advantages,
                                                        int T, int N, float gamma, float gamma_lam, int use_succ,
                                                        float succ_value) {
    __shared__ float D[GAE_TC * GAE_E];      // delta, then adv
    __shared__ float ND[GAE_TC * GAE_E];     // 1 - done
    __shared__ float V[GAE_TC * GAE_E];
    const int tid = threadIdx.x, n0 = blockIdx.x * GAE_E;
    const int e = tid % GAE_E, r0 = tid / GAE_E;           // phases 1 / 3: rows r0, r0 + 16, ... of the chunk
    const int n = n0 + e;
    const bool live = n < N;
    float adv = 0.0f;                                       // chain state of lane e (tid < GAE_E), carried over chunks
    for (int t_hi = T; t_hi > 0; t_hi -= GAE_TC) {
        const int t_lo = t_hi > GAE_TC ? t_hi - GAE_TC : 0, nt = t_hi - t_lo;
        if (live) {
#pragma unroll 8
            for (int tl = r0; tl < nt; tl += 256 / GAE_E) {
                const int t = t_lo + tl;
                const long o = (long)t * N + n;
                const float r = rewards[o], v = values[o];
                const float nv = (t + 1 < T) ? values[o + N] : last_values[n];
                const float notdone = dones[o] ? 0.0f : 1.0f;
                D[tl * GAE_E + e] = sub_rn(add_rn(r, mul_rn(gamma, nv)), v);
                ND[tl * GAE_E + e] = notdone;
                V[tl * GAE_E + e] = v;
            }
        }
        __syncthreads();
        if (tid < GAE_E && live) {
            int tl = nt - 1;
            for (; tl >= 7; tl -= 8) {                      // operands of eight steps requested together, then the chain
                float d8[8], n8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    d8[u] = D[(tl - u) * GAE_E + tid];
                    n8[u] = ND[(tl - u) * GAE_E + tid];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    adv = mul_rn(n8[u], add_rn(d8[u], mul_rn(gamma_lam, adv)));
                    D[(tl - u) * GAE_E + tid] = adv;
                }
            }
            for (; tl >= 0; --tl) {
                adv = mul_rn(ND[tl * GAE_E + tid], add_rn(D[tl * GAE_E + tid], mul_rn(gamma_lam, adv)));
                D[tl * GAE_E + tid] = adv;
            }
        }
        __syncthreads();
        if (live) {
#pragma unroll 8
            for (int tl = r0; tl < nt; tl += 256 / GAE_E) {
                const long o = (long)(t_lo + tl) * N + n;
                const float a = D[tl * GAE_E + e], v = V[tl * GAE_E + e];
                float ret;
                if (use_succ) {
                    const uint8_t sb = succs[o];
                    const float sf = sb ? 1.0f : 0.0f, nsf = sb ? 0.0f : 1.0f;
                    ret = add_rn(mul_rn(nsf, add_rn(a, v)), mul_rn(sf, succ_value));
                } else {
                    ret = add_rn(a, v);
                }
                returns[o] = ret;
                advantages[o] = sub_rn(ret, v);
            }
        }
        __syncthreads();                                    // the next chunk overwrites the tiles
    }
}

extern "C" int pm_gae_scan_f32(const float* rewards, const float* values, const uint8_t* dones,
                               const uint8_t* succs, const float* last_values, float* returns,
                               float* advantages, int T, int N, float gamma, float gamma_lam, int use_succ,
                               float succ_value, void* stream) {
    PM_REQUIRE(rewards && values && dones && last_values && returns && advantages);
    PM_REQUIRE(T > 0 && N > 0);
    PM_REQUIRE(!use_succ || succs);
    // Envs per work-group (measured, tools/time_gae.py, 128 steps): 4096 envs -- 16 / 32 / 64 all 9.8-11 us (launch-bound: 256 /
    // 128 / 64 work-groups); 32768 envs -- 16: 26.1 us (2.9 TB/s), 32: 18.5 us (4.1 TB/s of the 18 algorithmic B per env-step),
    // 64: 29.7 us (one 98 KB work-group per CU: nothing hides the serial phase).  32 envs = 128-byte row pieces of every time
    // step once that still leaves a work-group per CU; -DPM_GAE_E overrides (A/B builds).
#ifdef PM_GAE_E                                            // A/B builds only (tools/build_ab.py ... -DPM_GAE_E=64)
    const int E = PM_GAE_E;
#else
    const int E = N >= 32 * 256 ? 32 : 16;
#endif
#define GAE_LAUNCH(E_)                                                                                                     \
    hipLaunchKernelGGL(gae_scan_kernel<E_>, dim3((N + (E_) - 1) / (E_)), dim3(256), 0, pm_stream(stream), rewards, values, \
                       dones, succs, last_values, returns, advantages, T, N, gamma, gamma_lam, use_succ, succ_value)
    if (E == 64) GAE_LAUNCH(64);
    else if (E == 32) GAE_LAUNCH(32);
    else GAE_LAUNCH(16);
#undef GAE_LAUNCH
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

// SURVEY.md 8(b) names this entry point: the single-process form of storage.py:114 / ppo.py:329 in one call --
// x <- (x - mean(x)) / (std_unbiased(x) + eps) -- i.e. pm_moments_f64 + pm_normalize_apply_f32 with count = n (data-parallel
// callers use the two halves and all-reduce the moments between them).  The two doubles live at the end of the workspace.
extern "C" size_t pm_adv_normalize_workspace_bytes(long n) { return pm_moments_workspace_bytes(n) + 2 * sizeof(double); }

extern "C" int pm_adv_normalize_f32(float* x, long n, float eps, void* workspace, size_t workspace_bytes, void* stream) {
    PM_REQUIRE(x && n > 1);
    if (!workspace || workspace_bytes < pm_adv_normalize_workspace_bytes(n)) return PM_EWORKSPACE;
    if ((uintptr_t)workspace & 7) return PM_EALIGN;
    double* mom = (double*)((char*)workspace + pm_moments_workspace_bytes(n));
    const int rc = pm_moments_f64(x, n, mom, workspace, pm_moments_workspace_bytes(n), stream);
    if (rc != PM_OK) return rc;
    return pm_normalize_apply_f32(x, n, mom, (double)n, eps, stream);
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
