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
