// K10: clip_grad_norm_ + Adam over one flat fp32 buffer (ppo.py:351-353, 381-382; dagger.py:319).
// HBM bound: reads p,g,m,v (16 B) and writes p,m,v (12 B) per parameter = 28 B/param, in two
// launches (norm partials; update) with the device-side skip predicate and step counter so
// the reference's per-mini-batch host branch (ppo.py:337-338) needs no synchronisation.
#include "common.h"

static constexpr int ADAM_THREADS = 256;
static constexpr int ADAM_VEC = 4;
static constexpr int ADAM_MAX_BLOCKS = 1024;

static inline int adam_blocks(long n) {
    long b = (n + (long)ADAM_THREADS * ADAM_VEC - 1) / ((long)ADAM_THREADS * ADAM_VEC);
    if (b < 1) b = 1;
    if (b > ADAM_MAX_BLOCKS) b = ADAM_MAX_BLOCKS;
    return (int)b;
}

// Stage 1: per-block partial sum of squares over g[0..n_clip) (fp64 partials, deterministic),
// and the step-counter increment (block 0, thread 0), stream-ordered before stage 2 reads it.
__global__ __launch_bounds__(ADAM_THREADS) void grad_sumsq_kernel(const float* __restrict__ g, long n_clip,
                                                                   double* __restrict__ part,
                                                                   int32_t* __restrict__ state,
                                                                   const float* __restrict__ skip_flag) {
    __shared__ double red[ADAM_THREADS / 64];
    double s = 0.0;
    for (long i = (long)blockIdx.x * ADAM_THREADS + threadIdx.x; i < n_clip; i += (long)gridDim.x * ADAM_THREADS) {
        const double v = (double)g[i];
        s += v * v;
    }
    s = block_sum<double, ADAM_THREADS>(s, red);
    if (threadIdx.x == 0) {
        part[blockIdx.x] = s;
        if (blockIdx.x == 0) {
            const bool skip = skip_flag && skip_flag[0] != 0.0f;
            if (!skip) state[0] += 1;
        }
    }
}

// Stage 2: every block re-reduces the (<=1024) partials, derives the clip coefficient and the
// bias corrections (fp64, once per block) and applies the update to its slice.
__global__ __launch_bounds__(ADAM_THREADS) void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                                  float* __restrict__ m, float* __restrict__ v,
                                                                  long n, long n_clip, float max_norm, double lr,
                                                                  double b1, double b2, double eps,
                                                                  const int32_t* __restrict__ state,
                                                                  const float* __restrict__ skip_flag,
                                                                  const double* __restrict__ part, int nparts,
                                                                  float* __restrict__ gnorm_out) {
    __shared__ double red[ADAM_THREADS / 64];
    __shared__ float s_coef, s_step_size, s_bc2_sqrt;
    const bool skip = skip_flag && skip_flag[0] != 0.0f;
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += ADAM_THREADS) s += part[i];
    s = block_sum<double, ADAM_THREADS>(s, red);
    if (threadIdx.x == 0) {
        const float total = (float)sqrt(s);                         // torch computes the norm in fp32
        float coef = 1.0f;
        if (max_norm > 0.0f) {
            coef = max_norm / (total + 1e-6f);                        // clip_grad_norm_: clamp(max_norm/(norm+1e-6), max=1)
            if (coef > 1.0f) coef = 1.0f;
        }
        s_coef = coef;
        const int t = state[0];
        const double bc1 = 1.0 - pow(b1, (double)t);
        const double bc2 = 1.0 - pow(b2, (double)t);
        s_step_size = (float)(lr / bc1);
        s_bc2_sqrt = (float)sqrt(bc2);
        if (blockIdx.x == 0 && gnorm_out) gnorm_out[0] = total;
    }
    __syncthreads();
    if (skip) return;
    const float coef = s_coef, step_size = s_step_size, bc2s = s_bc2_sqrt;
    const float fb1 = (float)b1, fb2 = (float)b2, feps = (float)eps;
    const float omb1 = (float)(1.0 - b1), omb2 = (float)(1.0 - b2);
    for (long i = (long)blockIdx.x * ADAM_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * ADAM_THREADS) {
        float gi = g[i];
        if (i < n_clip) gi = gi * coef;
        // torch: exp_avg.lerp_(grad, 1-b1); exp_avg_sq.mul_(b2).addcmul_(grad, grad, value=1-b2)
        const float mi = m[i] + omb1 * (gi - m[i]);
        const float vi = v[i] * fb2 + omb2 * gi * gi;
        (void)fb1;
        const float denom = sqrtf(vi) / bc2s + feps;
        p[i] = p[i] - step_size * (mi / denom);
        m[i] = mi;
        v[i] = vi;
    }
}

extern "C" size_t pm_clip_adam_workspace_bytes(long n) { return (size_t)adam_blocks(n) * sizeof(double); }

extern "C" int pm_clip_adam_step_f32(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n,
                                     long n_clip, float max_norm, double lr, double b1, double b2, double eps,
                                     int32_t* state, const float* skip_flag, float* gnorm_out, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    PM_REQUIRE(params && grads && exp_avg && exp_avg_sq && state && n > 0 && n_clip >= 0 && n_clip <= n);
    const int nb = adam_blocks(n);
    if (!workspace || workspace_bytes < (size_t)nb * sizeof(double)) return PM_EWORKSPACE;
    const long nc = (max_norm > 0.0f) ? n_clip : 0;
    hipLaunchKernelGGL(grad_sumsq_kernel, dim3(nb), dim3(ADAM_THREADS), 0, pm_stream(stream), grads, nc,
                       (double*)workspace, state, skip_flag);
    hipLaunchKernelGGL(clip_adam_kernel, dim3(nb), dim3(ADAM_THREADS), 0, pm_stream(stream), params, grads, exp_avg,
                       exp_avg_sq, n, n_clip, max_norm, lr, b1, b2, eps, state, skip_flag,
                       (const double*)workspace, nb, gnorm_out);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ---- several optimisers per launch pair, with the split-K slab sum folded into the norm pass -------------------------
#define ADAM_GROUP_MAX 4
struct AdamGroup {
    int n;
    int block0[ADAM_GROUP_MAX + 1];
    pm_clip_adam_desc d[ADAM_GROUP_MAX];
};

// block 0, thread 0 of the norm pass: the step counter (unless the step is skipped) and the step's running sums
__device__ __forceinline__ void adam_block0_tail(const pm_clip_adam_desc& d) {
    // (volatile: in the data-parallel form the flag was just rewritten through d.dp_scal, which may alias it)
    const bool skip = d.skip_flag && ((const volatile float*)d.skip_flag)[0] != 0.0f;
    if (!skip) d.state[0] += 1;
    if (d.stats_acc) {                                     // pm_ppo_accumulate_stats_f32 (losses.hip), same arithmetic
        float* acc = d.stats_acc;
        const volatile float* scal = d.stats_scal;
        if (d.stats_which == 0) {
            const float loss = scal[0], kl = scal[1], sk = scal[2];
            if (kl > acc[2]) acc[2] = kl;
            if (sk == 0.0f) {
                acc[0] += loss;
                acc[1] += kl;
                acc[3] += 1.0f;
            }
        } else {
            acc[4] += scal[0];
            acc[5] += 1.0f;
        }
    }
}

__global__ __launch_bounds__(ADAM_THREADS) void grad_sumsq_group_kernel(AdamGroup G) {
    __shared__ double red[ADAM_THREADS / 64];
    int k = 0;
#pragma unroll
    for (int i = 1; i < ADAM_GROUP_MAX; ++i)
        if (i < G.n && (int)blockIdx.x >= G.block0[i]) k = i;
    const pm_clip_adam_desc& d = G.d[k];
    const int nb = G.block0[k + 1] - G.block0[k], b = (int)blockIdx.x - G.block0[k];
    const long n_clip = d.max_norm > 0.0f ? d.n_clip : 0;
    const long n_sum = d.n_extra > 0 ? d.n_sum : 0;
    const long hi = n_clip > n_sum ? n_clip : n_sum;
    float* __restrict__ g = d.grads;
    double s = 0.0;
    if (d.dp_scal != nullptr || (d.grad_scale != 0.0f && d.grad_scale != 1.0f)) {
        // data-parallel step: grads hold the all-reduce SUM over the ranks -> the mean, in place, then the plain norm pass
        // (n_extra == 0 by contract: the slabs were folded before the reduce, pm_grad_slab_sum_f32)
        const float sc = d.grad_scale != 0.0f ? d.grad_scale : 1.0f;       // (a one-rank group: x 1.0f, exact)
        const long n = d.n;
        const bool al4 = ((uintptr_t)g & 15) == 0;
        const long nv4 = al4 ? (n & ~3L) : 0;
        for (long i = ((long)b * ADAM_THREADS + threadIdx.x) * 4; i < nv4; i += (long)nb * ADAM_THREADS * 4) {
            float4 gv = *(const float4*)(g + i);
            gv.x *= sc; gv.y *= sc; gv.z *= sc; gv.w *= sc;
            *(float4*)(g + i) = gv;
            if (i < n_clip) {
                s += (i + 0 < n_clip ? (double)gv.x * (double)gv.x : 0.0) + (i + 1 < n_clip ? (double)gv.y * (double)gv.y : 0.0) +
                     (i + 2 < n_clip ? (double)gv.z * (double)gv.z : 0.0) + (i + 3 < n_clip ? (double)gv.w * (double)gv.w : 0.0);
            }
        }
        for (long i = nv4 + (long)b * ADAM_THREADS + threadIdx.x; i < n; i += (long)nb * ADAM_THREADS) {
            const float gi = g[i] * sc;
            g[i] = gi;
            if (i < n_clip) s += (double)gi * (double)gi;
        }
        s = block_sum<double, ADAM_THREADS>(s, red);
        if (threadIdx.x == 0) {
            ((double*)d.workspace)[b] = s;
            if (b == 0) {
                if (d.dp_scal) {                               // the reduced scalars -> means; the KL predicate from the REDUCED kl
                    d.dp_scal[0] *= sc;
                    d.dp_scal[1] *= sc;
                    if (d.dp_kl_desired > 0.0f) d.dp_scal[2] = d.dp_scal[1] > d.dp_kl_desired ? 1.0f : 0.0f;
                }
                adam_block0_tail(d);
            }
        }
        return;
    }
    // 16-byte body (4 elements per thread and trip, the common prefix of the two ranges rounded down), scalar tail
    const bool al = (((uintptr_t)g | (uintptr_t)d.extra) & 15) == 0 && (d.extra_stride & 3) == 0;
    const long lo = n_clip < n_sum ? n_clip : n_sum;
    const long nv = al ? ((n_sum > 0 && n_clip > 0 ? lo : hi) & ~3L) : 0;
    const bool vs = n_sum > 0, vc = n_clip > 0;             // inside [0, nv) an element is in every non-empty range
    for (long i = ((long)b * ADAM_THREADS + threadIdx.x) * 4; i < nv; i += (long)nb * ADAM_THREADS * 4) {
        float4 gv = *(const float4*)(g + i);
        if (vs) {                                          // slab 0 is `grads` itself: add slabs 1.. in fixed order
            for (int z = 0; z < d.n_extra; ++z) {
                const float4 e = *(const float4*)(d.extra + (long)z * d.extra_stride + i);
                gv.x += e.x; gv.y += e.y; gv.z += e.z; gv.w += e.w;
            }
            *(float4*)(g + i) = gv;
        }
        if (vc) s += (double)gv.x * (double)gv.x + (double)gv.y * (double)gv.y + (double)gv.z * (double)gv.z + (double)gv.w * (double)gv.w;
    }
    for (long i = nv + (long)b * ADAM_THREADS + threadIdx.x; i < hi; i += (long)nb * ADAM_THREADS) {
        float gi = g[i];
        if (i < n_sum) {
            for (int z = 0; z < d.n_extra; ++z) gi += d.extra[(long)z * d.extra_stride + i];
            g[i] = gi;
        }
        if (i < n_clip) s += (double)gi * (double)gi;
    }
    s = block_sum<double, ADAM_THREADS>(s, red);
    if (threadIdx.x == 0) {
        ((double*)d.workspace)[b] = s;
        if (b == 0) adam_block0_tail(d);
    }
}

__global__ __launch_bounds__(ADAM_THREADS) void clip_adam_group_kernel(AdamGroup G) {
    __shared__ double red[ADAM_THREADS / 64];
    __shared__ float s_coef, s_step_size, s_bc2_sqrt;
    int k = 0;
#pragma unroll
    for (int i = 1; i < ADAM_GROUP_MAX; ++i)
        if (i < G.n && (int)blockIdx.x >= G.block0[i]) k = i;
    const pm_clip_adam_desc& d = G.d[k];
    const int nb = G.block0[k + 1] - G.block0[k], b = (int)blockIdx.x - G.block0[k];
    const bool skip = d.skip_flag && d.skip_flag[0] != 0.0f;
    const double* part = (const double*)d.workspace;
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += ADAM_THREADS) s += part[i];
    s = block_sum<double, ADAM_THREADS>(s, red);
    if (threadIdx.x == 0) {                                // identical arithmetic to clip_adam_kernel above
        const float total = (float)sqrt(s);
        float coef = 1.0f;
        if (d.max_norm > 0.0f) {
            coef = d.max_norm / (total + 1e-6f);
            if (coef > 1.0f) coef = 1.0f;
        }
        s_coef = coef;
        const int t = d.state[0];
        const double bc1 = 1.0 - pow(d.b1, (double)t);
        const double bc2 = 1.0 - pow(d.b2, (double)t);
        s_step_size = (float)(d.lr / bc1);
        s_bc2_sqrt = (float)sqrt(bc2);
        if (b == 0 && d.gnorm_out) d.gnorm_out[0] = total;
    }
    __syncthreads();
    if (skip) return;
    const float coef = s_coef, step_size = s_step_size, bc2s = s_bc2_sqrt;
    const float fb2 = (float)d.b2, feps = (float)d.eps;
    const float omb1 = (float)(1.0 - d.b1), omb2 = (float)(1.0 - d.b2);
    float* __restrict__ p = d.params;
    const float* __restrict__ g = d.grads;
    float* __restrict__ m = d.exp_avg;
    float* __restrict__ v = d.exp_avg_sq;
    const long n = d.n, n_clip = d.n_clip;
    auto upd = [&](float gi, bool clipped, float& pi, float& mi, float& vi) __attribute__((always_inline)) {
        if (clipped) gi = gi * coef;
        mi = mi + omb1 * (gi - mi);
        vi = vi * fb2 + omb2 * gi * gi;
        const float denom = sqrtf(vi) / bc2s + feps;
        pi = pi - step_size * (mi / denom);
    };
    // 16-byte body, scalar tail (element-wise the same arithmetic)
    const bool al = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
    const long nv = al ? (n & ~3L) : 0;
    for (long i = ((long)b * ADAM_THREADS + threadIdx.x) * 4; i < nv; i += (long)nb * ADAM_THREADS * 4) {
        const float4 gv = *(const float4*)(g + i);
        float4 pv = *(const float4*)(p + i), mv = *(const float4*)(m + i), vv = *(const float4*)(v + i);
        upd(gv.x, i < n_clip, pv.x, mv.x, vv.x);
        upd(gv.y, i + 1 < n_clip, pv.y, mv.y, vv.y);
        upd(gv.z, i + 2 < n_clip, pv.z, mv.z, vv.z);
        upd(gv.w, i + 3 < n_clip, pv.w, mv.w, vv.w);
        *(float4*)(p + i) = pv;
        *(float4*)(m + i) = mv;
        *(float4*)(v + i) = vv;
    }
    for (long i = nv + (long)b * ADAM_THREADS + threadIdx.x; i < n; i += (long)nb * ADAM_THREADS) {
        float pi = p[i], mi = m[i], vi = v[i];
        upd(g[i], i < n_clip, pi, mi, vi);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
}

__global__ __launch_bounds__(ADAM_THREADS) void grad_slab_sum_kernel(float* __restrict__ g, const float* __restrict__ extra,
                                                                      long extra_stride, long n_sum, int n_extra) {
    const bool al = (((uintptr_t)g | (uintptr_t)extra) & 15) == 0 && (extra_stride & 3) == 0;
    const long nv = al ? (n_sum & ~3L) : 0;
    for (long i = ((long)blockIdx.x * ADAM_THREADS + threadIdx.x) * 4; i < nv; i += (long)gridDim.x * ADAM_THREADS * 4) {
        float4 gv = *(const float4*)(g + i);
        for (int z = 0; z < n_extra; ++z) {                // the order of grad_sumsq_group_kernel: slab 0, then 1, 2, ...
            const float4 e = *(const float4*)(extra + (long)z * extra_stride + i);
            gv.x += e.x; gv.y += e.y; gv.z += e.z; gv.w += e.w;
        }
        *(float4*)(g + i) = gv;
    }
    for (long i = nv + (long)blockIdx.x * ADAM_THREADS + threadIdx.x; i < n_sum; i += (long)gridDim.x * ADAM_THREADS) {
        float gi = g[i];
        for (int z = 0; z < n_extra; ++z) gi += extra[(long)z * extra_stride + i];
        g[i] = gi;
    }
}

extern "C" int pm_grad_slab_sum_f32(float* grads, const float* extra, long extra_stride, long n_sum, int n_extra, void* stream) {
    PM_REQUIRE(grads && n_sum >= 0 && n_extra >= 0 && (n_extra == 0 || extra));
    if (n_sum == 0 || n_extra == 0) return PM_OK;
    hipLaunchKernelGGL(grad_slab_sum_kernel, dim3(adam_blocks(n_sum)), dim3(ADAM_THREADS), 0, pm_stream(stream), grads, extra,
                       extra_stride, n_sum, n_extra);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

extern "C" int pm_clip_adam_group_f32(int n, const pm_clip_adam_desc* d, void* stream) {
    PM_REQUIRE(d && n >= 1 && n <= ADAM_GROUP_MAX);
    AdamGroup G{};
    G.n = n;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const pm_clip_adam_desc& q = d[i];
        PM_REQUIRE(q.params && q.grads && q.exp_avg && q.exp_avg_sq && q.state && q.workspace && q.n > 0 && q.n_clip >= 0 &&
                   q.n_clip <= q.n && q.n_extra >= 0 && (q.n_extra == 0 || (q.extra && q.n_sum >= 0 && q.n_sum <= q.n)));
        const bool dp = q.dp_scal != nullptr || (q.grad_scale != 0.0f && q.grad_scale != 1.0f);
        PM_REQUIRE(!dp || q.n_extra == 0);                 // data-parallel form: slabs are folded before the all-reduce
        if (((uintptr_t)q.workspace & 7) != 0) return PM_EALIGN;
        G.d[i] = q;
        G.block0[i] = blocks;
        blocks += adam_blocks(q.n);
    }
    G.block0[n] = blocks;
    for (int i = n + 1; i <= ADAM_GROUP_MAX; ++i) G.block0[i] = blocks;
    hipLaunchKernelGGL(grad_sumsq_group_kernel, dim3(blocks), dim3(ADAM_THREADS), 0, pm_stream(stream), G);
    hipLaunchKernelGGL(clip_adam_group_kernel, dim3(blocks), dim3(ADAM_THREADS), 0, pm_stream(stream), G);
    PM_CHECK_LAUNCH();
    return PM_OK;
}
