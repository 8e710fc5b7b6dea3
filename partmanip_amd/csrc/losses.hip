// K8 PPO actor loss, K9 value loss, K11 DAgger loss: one pass over a (B, A<=64) mini-batch,
// forward value + backward gradient in the same launch.  These are a few hundred KB of
// traffic per call; the point of fusing is to remove ~40 tiny ATen launches and two host
// syncs per mini-batch (ppo.py:326-357), not bandwidth.
#include "common.h"
#include "skinny.h"

#define LOSS_THREADS 1024
#define LOSS_WAVES (LOSS_THREADS / 64)
#define MAX_A 64

__device__ __forceinline__ float deactivate(float a, float max_action, int act_tanh) {
    // actor_critic.py:93-100
    if (!act_tanh) return a;
    float u = a / max_action;
    u = fminf(fmaxf(u, -1.0f + 1e-5f), 1.0f - 1e-5f);
    return atanhf(u);
}

// ---------------------------------------------------------------------------------- K8
// Stage 1: up to 128 work-groups of 256 threads each write dmu and a partial {loss, kl, dlog_std[A]};
// stage 2 (one wave) sums the partials in fixed order.  (A single 1024-thread work-group took
// 63 us for 2048 rows -- 12 % of a state-PPO optimiser step; the row math is atanh/exp/log heavy.)
#define AL_THREADS 256
#define AL_MAXG 128
struct ActorLossPart {
    double loss, kl;
    float dls[MAX_A];
};

// One lane per (row, action) element, AP = 16 or 64 lanes per row (A <= AP): the first version walked a row's A actions in
// one thread -- A dependent rounds of four loads, an atanh, an exp -- and took 19 us for 2048 x 10, a sixth of a state-PPO
// actor step's dependent chain.  Row sums (M, kl) are butterfly sums inside the AP-lane group; the per-action dlog_std terms
// stay in their lane across the work-group's rows and meet in LDS at the end (fixed order).
template <int AP>
__global__ __launch_bounds__(AL_THREADS) void ppo_actor_loss_part_kernel(
    const float* __restrict__ mu, long ldmu, const float* __restrict__ log_std, const float* __restrict__ actions,
    long lda, const float* __restrict__ old_logp, const float* __restrict__ adv, const float* __restrict__ old_mu,
    long ldom, const float* __restrict__ old_sigma, long ldos, int B, int A, float max_action, int act_tanh,
    float eps_clip, const double* __restrict__ adv_moments, double adv_count, float* __restrict__ dmu, long lddmu,
    ActorLossPart* __restrict__ parts) {
    constexpr int RPB = AL_THREADS / AP;                   // rows per work-group pass
    __shared__ float s_ls[MAX_A], s_s[MAX_A];
    __shared__ double red[AL_THREADS / 64];
    __shared__ float red_a[RPB][AP];
    const int tid = threadIdx.x, r = tid / AP, a = tid % AP;
    if (tid < A) {
        const float ls = log_std[tid];
        const float e = expf(ls);
        s_ls[tid] = ls;
        s_s[tid] = e * e;                 // effective std (actor_critic.py:74) == square(sigma.exp()) of ppo.py:333
    }
    __syncthreads();
    float sum_logs = 0.f;
    for (int k = 0; k < A; ++k) sum_logs += logf(s_s[k]);
    const float k_log2pi = (float)A * 1.8378770664093453f;

    float a_mean = 0.f, a_den = 1.f;
    if (adv_moments) {                    // ppo.py:329 mini_adv_norm
        const double m = adv_moments[0] / adv_count;
        double var = (adv_moments[1] - adv_moments[0] * m) / (adv_count - 1.0);
        if (var < 0.0) var = 0.0;
        a_mean = (float)m;
        a_den = (float)sqrt(var) + 1e-8f;
    }
    const float invB = 1.0f / (float)B;
    const bool on_a = a < A;
    const float sa = on_a ? s_s[a] : 1.f, lsa = on_a ? s_ls[a] : 0.f;
    double loss_acc = 0.0, kl_acc = 0.0;
    float cacc = 0.f;                                       // this lane's action: sum over its rows of d loss / d log_std[a]
    for (int base = blockIdx.x * RPB; base < B; base += gridDim.x * RPB) {   // uniform trip count per block
        const int i = base + r;
        const bool row = i < B, on = row && on_a;
        float z = 0.f, klt = 0.f;
        if (on) {
            const float m = mu[i * ldmu + a];
            const float x = deactivate(actions[i * lda + a], max_action, act_tanh);
            const float os = old_sigma[i * ldos + a];
            z = (x - m) / sa;
            const float eo = expf(os);
            const float dm = old_mu[i * ldom + a] - m;
            klt = lsa - os + (eo * eo + dm * dm) / (2.0f * sa) - 0.5f;
        }
        float M = z * z, kl = klt;
#pragma unroll
        for (int o = AP / 2; o > 0; o >>= 1) {
            M += __shfl_xor(M, o, 64);
            kl += __shfl_xor(kl, o, 64);
        }
        float g = 0.f;
        if (row) {
            const float logp = -0.5f * (k_log2pi + M) - sum_logs;
            const float ratio = expf(logp - old_logp[i]);
            float ad = adv[i];
            if (adv_moments) ad = (ad - a_mean) / a_den;
            const float s1 = -ad * ratio;
            const float s2 = -ad * fminf(fmaxf(ratio, 1.0f - eps_clip), 1.0f + eps_clip);
            if (a == 0) {
                loss_acc += (double)fmaxf(s1, s2);
                kl_acc += (double)kl;
            }
            // d max(s1,s2)/d ratio: -ad through the unclipped branch (incl. the tie inside the clip
            // range, where torch splits the gradient in two halves that re-add); 0 when clipped.
            g = (s1 >= s2 ? -ad : 0.0f) * ratio * invB;       // d loss / d logp_i
        }
        if (on) {
            dmu[i * lddmu + a] = g * z / sa;                  // d logp / d mu = z / s
            cacc += g * (2.0f * z * z - 2.0f);                // d logp / d log_std (s = exp(2 ls))
        }
    }
    const double loss_sum = block_sum<double, AL_THREADS>(loss_acc, red);
    const double kl_sum = block_sum<double, AL_THREADS>(kl_acc, red);
    red_a[r][a] = cacc;
    __syncthreads();
    ActorLossPart* part = parts + blockIdx.x;
    if (tid < A) {
        float t = 0.f;
        for (int k = 0; k < RPB; ++k) t += red_a[k][tid];
        part->dls[tid] = t;
    }
    if (tid == 0) {
        part->loss = loss_sum;
        part->kl = kl_sum;
    }
}

// final stage (one wave): butterfly sums over the work-groups' partials instead of G dependent loads in one lane
template <bool COHERENT = false>                           // true: the partials were written by other work-groups of THIS launch
__device__ __forceinline__ void actor_loss_final(const ActorLossPart* parts, int G, int B, int A,
                                                 const float* __restrict__ log_std, float desired_kl, float* __restrict__ scal_out,
                                                 float* __restrict__ dlog_std, int tid, const float* sum_logs_known = nullptr) {
    const bool has = tid < G, has2 = tid + 64 < G;         // G <= 128: lane g holds partials g and g + 64
    auto ldd = [&](const double* p) { return COHERENT ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; };
    auto ldf = [&](const float* p) { return COHERENT ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; };
    double ls = has ? ldd(&parts[tid].loss) : 0.0, ks = has ? ldd(&parts[tid].kl) : 0.0;
    if (has2) {
        ls += ldd(&parts[tid + 64].loss);
        ks += ldd(&parts[tid + 64].kl);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ls += __shfl_xor(ls, o, 64);
        ks += __shfl_xor(ks, o, 64);
    }
    for (int a0 = 0; a0 < A; a0 += 16) {                   // sixteen independent loads in flight, then their butterflies
        float t[16], u[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) t[j] = (has && a0 + j < A) ? ldf(&parts[tid].dls[a0 + j]) : 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) u[j] = (has2 && a0 + j < A) ? ldf(&parts[tid + 64].dls[a0 + j]) : 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t[j] += u[j];
#pragma unroll
        for (int j = 0; j < 16; ++j) t[j] = wave_sum(t[j]);
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (tid == j && a0 + j < A) dlog_std[a0 + j] = t[j];
    }
    if (tid == 0) {
        float sum_logs = 0.f;
        if (sum_logs_known) {                              // (the caller has this very sum already)
            sum_logs = *sum_logs_known;
        } else {
            for (int a = 0; a < A; ++a) {
                const float e = expf(log_std[a]);
                sum_logs += logf(e * e);
            }
        }
        const float klm = (float)(ks / (double)B);
        scal_out[0] = (float)(ls / (double)B);
        scal_out[1] = klm;
        scal_out[2] = (klm > desired_kl) ? 1.0f : 0.0f;
        scal_out[3] = 0.5f * (float)A * (1.0f + 1.8378770664093453f) + sum_logs;   // entropy (same every row)
    }
}
__global__ __launch_bounds__(64) void ppo_actor_loss_final_kernel(const ActorLossPart* __restrict__ parts, int G, int B,
                                                                   int A, const float* __restrict__ log_std,
                                                                   float desired_kl, float* __restrict__ scal_out,
                                                                   float* __restrict__ dlog_std) {
    actor_loss_final(parts, G, B, A, log_std, desired_kl, scal_out, dlog_std, threadIdx.x);
}

// ---- the policy head and its loss as ONE launch (small-step regime) ------------------------------------------------------
// mu = H W^T + b (row-wise, skinny.h) -> the loss rows above -> dmu -> dH = (dmu W) .* act'(H) -> the last work-group to
// finish sums the partials.  Replaces four dependent launches of a state-PPO actor step (head forward 9 us, loss 9.7, final
// 8, head data gradient 5.7 of a ~200 us chain) by one; every stage keeps the arithmetic and the summation order of the
// separate kernels (same work-group partition as ppo_actor_loss_part_kernel<16>), so the results are bit-identical.
#define AH_ROWS 16
#define AH_THREADS 1024
// dynamic LDS (the head's A x K weights) the fused head kernels accept.  Their static arrays add < 8 KB; a gfx950 work-group
// may allocate up to 160 KB (MI355X_MICROARCH.md, LDS: "a single workgroup may declare all 160 KiB"), and this library is built
// for gfx950 only -- 64 KB + 8 KB can therefore never fail at launch (tests run A = 16, K = 1024 = exactly 64 KB).
#define AH_MAX_DYN_LDS (64 * 1024)
static_assert(AH_MAX_DYN_LDS + 8 * 1024 <= 160 * 1024, "fused head kernels: weights + static arrays must fit a gfx950 work-group's LDS");
__global__ __launch_bounds__(AH_THREADS) void ppo_actor_head_kernel(
    const float* __restrict__ H, long ldh, const float* __restrict__ W, long ldw, const float* __restrict__ bias, int K, int hact,
    const float* __restrict__ log_std, const float* __restrict__ actions, long lda, const float* __restrict__ old_logp,
    const float* __restrict__ adv, const float* __restrict__ old_mu, long ldom, const float* __restrict__ old_sigma, long ldos,
    int B, int A, float max_action, int act_tanh, float eps_clip, float desired_kl, const double* __restrict__ adv_moments,
    double adv_count, float* __restrict__ mu_out, long ldmu, float* __restrict__ dmu, long lddmu, float* __restrict__ dH, long lddh,
    ActorLossPart* __restrict__ parts, unsigned int* __restrict__ counter, float* __restrict__ scal_out,
    float* __restrict__ dlog_std) {
    constexpr int AP = 16, RPB = AH_ROWS;
    extern __shared__ __attribute__((aligned(16))) float sW[];                  // [A][K]
    __shared__ float s_ls[MAX_A], s_s[MAX_A];
    __shared__ double red[AH_THREADS / 64];
    __shared__ float red_a[RPB][AP];
    __shared__ float s_mu[RPB][AP], s_dmu[RPB][AP];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool lt = tid < AL_THREADS;                      // the loss rows run on the first 256 threads (16 lanes per row)
    const int r = lt ? tid / AP : 0, a = tid % AP;
    sk_fill_w<AH_THREADS>(sW, W, ldw, A, K);
    if (tid < A) {
        const float ls = log_std[tid];
        const float e = expf(ls);
        s_ls[tid] = ls;
        s_s[tid] = e * e;
    }
    __syncthreads();
    float sum_logs = 0.f;
    for (int k = 0; k < A; ++k) sum_logs += logf(s_s[k]);
    const float k_log2pi = (float)A * 1.8378770664093453f;
    float a_mean = 0.f, a_den = 1.f;
    if (adv_moments) {
        const double m = adv_moments[0] / adv_count;
        double var = (adv_moments[1] - adv_moments[0] * m) / (adv_count - 1.0);
        if (var < 0.0) var = 0.0;
        a_mean = (float)m;
        a_den = (float)sqrt(var) + 1e-8f;
    }
    const float invB = 1.0f / (float)B;
    const bool on_a = a < A;
    const float sa = on_a ? s_s[a] : 1.f, lsa = on_a ? s_ls[a] : 0.f;
    double loss_acc = 0.0, kl_acc = 0.0;
    float cacc = 0.f;
    const int k4 = K >> 2;
    for (int base = blockIdx.x * RPB; base < B; base += gridDim.x * RPB) {
        // the loss rows' inputs do not depend on mu: request them now, they land under the head forward
        const int i = base + r;
        const bool row = lt && i < B, on = row && on_a;
        float in_act = 0.f, in_os = 0.f, in_om = 0.f, in_olp = 0.f, in_adv = 0.f;
        if (on) {
            in_act = actions[i * lda + a];
            in_os = old_sigma[i * ldos + a];
            in_om = old_mu[i * ldom + a];
        }
        if (row) {
            in_olp = old_logp[i];
            in_adv = adv[i];
        }
        // ---- head forward: one row of the pass per wave
        {
            const int rr = wave, i = base + rr;
            if (i < B) {                                    // wave-uniform
                float acc[SK_MAXN];
                sk_row_dot(H + (long)i * ldh, sW, A, K, lane, acc);
#pragma unroll
                for (int n = 0; n < SK_MAXN; ++n)
                    if (n < A) {
                        const float s = wave_sum(acc[n]);
                        if (lane == n) {
                            const float m = s + (bias ? bias[n] : 0.f);
                            s_mu[rr][n] = m;
                            if (mu_out) mu_out[(long)i * ldmu + n] = m;
                        }
                    }
            }
        }
        __syncthreads();
        // ---- loss rows (ppo_actor_loss_part_kernel<16>, mu from LDS)
        float z = 0.f, klt = 0.f;
        if (on) {
            const float m = s_mu[r][a];
            const float x = deactivate(in_act, max_action, act_tanh);
            const float os = in_os;
            z = (x - m) / sa;
            const float eo = expf(os);
            const float dm = in_om - m;
            klt = lsa - os + (eo * eo + dm * dm) / (2.0f * sa) - 0.5f;
        }
        float M = z * z, kl = klt;
#pragma unroll
        for (int o = AP / 2; o > 0; o >>= 1) {
            M += __shfl_xor(M, o, 64);
            kl += __shfl_xor(kl, o, 64);
        }
        float g = 0.f;
        if (row) {
            const float logp = -0.5f * (k_log2pi + M) - sum_logs;
            const float ratio = expf(logp - in_olp);
            float ad = in_adv;
            if (adv_moments) ad = (ad - a_mean) / a_den;
            const float s1 = -ad * ratio;
            const float s2 = -ad * fminf(fmaxf(ratio, 1.0f - eps_clip), 1.0f + eps_clip);
            if (a == 0) {
                loss_acc += (double)fmaxf(s1, s2);
                kl_acc += (double)kl;
            }
            g = (s1 >= s2 ? -ad : 0.0f) * ratio * invB;
        }
        float dm_out = 0.f;
        if (on) {
            dm_out = g * z / sa;
            dmu[i * lddmu + a] = dm_out;
            cacc += g * (2.0f * z * z - 2.0f);
        }
        if (lt) s_dmu[r][a] = dm_out;
        __syncthreads();
        // ---- head data gradient: (row, 4 columns) items of the pass over the work-group
        for (int e = tid; e < RPB * k4; e += AH_THREADS) {
            const int rr = e / k4, k = (e - rr * k4) * 4;
            const long ii = base + rr;
            if (ii < B)
                *(float4*)(dH + ii * lddh + k) = sk_dgrad4([&](int n) { return s_dmu[rr][n]; }, sW, A, K, k, H + ii * ldh, hact);
        }
        __syncthreads();                                   // s_mu / s_dmu are rewritten by the next pass
    }
    // (threads 256.. carry zeros: the same sums, in the same order, as the 256-thread loss kernel's)
    const double loss_sum = block_sum<double, AH_THREADS>(loss_acc, red);
    const double kl_sum = block_sum<double, AH_THREADS>(kl_acc, red);
    if (lt) red_a[r][a] = cacc;
    __syncthreads();
    // The partials cross work-groups (and XCDs, whose L2s are not coherent with each other) INSIDE the launch.  A device-scope
    // fence would do it -- and writes back every dirty L2 line first, the 4 MB of dH included: the kernel took 49 us with
    // __threadfence() against 13 without.  Instead the partials themselves travel as device-scope atomic stores / loads
    // (write-through, sc1) and the ordinary stores are left to the end-of-kernel release.
    ActorLossPart* part = parts + blockIdx.x;
    if (lt && tid < A) {
        float t = 0.f;
        for (int k = 0; k < RPB; ++k) t += red_a[k][tid];
        __hip_atomic_store(&part->dls[tid], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) {
        __hip_atomic_store(&part->loss, loss_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&part->kl, kl_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // my partials have reached memory ...
    __syncthreads();                                       // ... and so have this work-group's
    // ---- the last work-group to get here sums the partials (fixed order: the result does not depend on which one it is)
#ifdef AH_NO_FINAL                                          // A/B builds only (tools/build_ab.sh): what the counter round trip + the final stage cost --
    if (tid == 0) s_last = 0;                              // 19.9 -> 14.7 us per launch at 2048 x 512 x 10; a separate final launch costs more
#else
    if (tid == 0) s_last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
#endif
    __syncthreads();
    if (s_last && tid < 64) {
        actor_loss_final<true>(parts, (int)gridDim.x, B, A, log_std, desired_kl, scal_out, dlog_std, tid, &sum_logs);
        if (tid == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
}

static inline int actor_loss_blocks(int B) {           // sized for the 16-lanes-per-row form (16 rows per work-group pass)
    int g = (B + AL_THREADS / 16 - 1) / (AL_THREADS / 16);
    return g < 1 ? 1 : (g > AL_MAXG ? AL_MAXG : g);
}

extern "C" size_t pm_ppo_actor_loss_workspace_bytes(int B) { return (size_t)actor_loss_blocks(B) * sizeof(ActorLossPart); }

extern "C" int pm_ppo_actor_loss_fwd_bwd_f32(const float* mu, long ldmu, const float* log_std,
                                             const float* actions, long lda, const float* old_logp,
                                             const float* adv, const float* old_mu, long ldom,
                                             const float* old_sigma, long ldos, int B, int A, float max_action,
                                             int act_tanh, float eps_clip, float desired_kl,
                                             const double* adv_moments, double adv_count, float* scal_out,
                                             float* dmu, long lddmu, float* dlog_std, void* workspace,
                                             size_t workspace_bytes, void* stream) {
    PM_REQUIRE(mu && log_std && actions && old_logp && adv && old_mu && old_sigma && scal_out && dmu && dlog_std);
    PM_REQUIRE(B > 0 && A > 0 && A <= MAX_A && max_action > 0.f);
    PM_REQUIRE(!adv_moments || adv_count > 1.0);
    const int G = actor_loss_blocks(B);
    if (!workspace || workspace_bytes < (size_t)G * sizeof(ActorLossPart) || ((uintptr_t)workspace & 7)) return PM_EWORKSPACE;
    ActorLossPart* parts = (ActorLossPart*)workspace;
    if (A <= 16)
        hipLaunchKernelGGL(ppo_actor_loss_part_kernel<16>, dim3(G), dim3(AL_THREADS), 0, pm_stream(stream), mu, ldmu, log_std,
                           actions, lda, old_logp, adv, old_mu, ldom, old_sigma, ldos, B, A, max_action, act_tanh, eps_clip,
                           adv_moments, adv_count, dmu, lddmu, parts);
    else
        hipLaunchKernelGGL(ppo_actor_loss_part_kernel<64>, dim3(G), dim3(AL_THREADS), 0, pm_stream(stream), mu, ldmu, log_std,
                           actions, lda, old_logp, adv, old_mu, ldom, old_sigma, ldos, B, A, max_action, act_tanh, eps_clip,
                           adv_moments, adv_count, dmu, lddmu, parts);
    hipLaunchKernelGGL(ppo_actor_loss_final_kernel, dim3(1), dim3(64), 0, pm_stream(stream), parts, G, B, A, log_std,
                       desired_kl, scal_out, dlog_std);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

extern "C" int pm_ppo_actor_head_supported(const float* H, long ldh, const float* W, long ldw, int A, int K, const float* dH,
                                           long lddh) {
    // the kernel adds ~4.5 KB of static LDS (red, red_a, s_mu, s_dmu, ...) to the A*K*4 bytes of weights (AH_MAX_DYN_LDS above)
    return A <= 16 && skinny_ok(A, K, H, ldh, W, ldw) && (long)A * K * 4 <= AH_MAX_DYN_LDS && lddh % 4 == 0 && sk_aligned16(dH);
}

extern "C" int pm_ppo_actor_head_f32(const float* H, long ldh, const float* W, long ldw, const float* bias, int K, int hidden_act,
                                     const float* log_std, const float* actions, long lda, const float* old_logp,
                                     const float* adv, const float* old_mu, long ldom, const float* old_sigma, long ldos, int B,
                                     int A, float max_action, int act_tanh, float eps_clip, float desired_kl,
                                     const double* adv_moments, double adv_count, float* scal_out, float* mu_out, long ldmu,
                                     float* dmu, long lddmu, float* dH, long lddh, float* dlog_std, void* workspace,
                                     size_t workspace_bytes, unsigned int* counter, void* stream) {
    PM_REQUIRE(H && W && log_std && actions && old_logp && adv && old_mu && old_sigma && scal_out && dmu && dH && dlog_std && counter);
    PM_REQUIRE(B > 0 && A > 0 && K > 0 && max_action > 0.f && ldh >= K && ldw >= K && lddh >= K && lddmu >= A);
    PM_REQUIRE(hidden_act >= PM_ACT_NONE && hidden_act <= PM_ACT_MAX);
    PM_REQUIRE(!adv_moments || adv_count > 1.0);
    PM_REQUIRE(!mu_out || ldmu >= A);
    if (!pm_ppo_actor_head_supported(H, ldh, W, ldw, A, K, dH, lddh)) return PM_EUNSUPPORTED;
    const int G = actor_loss_blocks(B);
    if (!workspace || workspace_bytes < (size_t)G * sizeof(ActorLossPart) || ((uintptr_t)workspace & 7)) return PM_EWORKSPACE;
    hipLaunchKernelGGL(ppo_actor_head_kernel, dim3(G), dim3(AH_THREADS), (size_t)A * K * 4, pm_stream(stream), H, ldh, W, ldw, bias, K,
                       hidden_act, log_std, actions, lda, old_logp, adv, old_mu, ldom, old_sigma, ldos, B, A, max_action, act_tanh,
                       eps_clip, desired_kl, adv_moments, adv_count, mu_out, ldmu, dmu, lddmu, dH, lddh, (ActorLossPart*)workspace,
                       counter, scal_out, dlog_std);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// log-prob / entropy rows only (actor_critic.py:71-82, 36-47)
__global__ __launch_bounds__(256) void gaussian_logp_kernel(const float* __restrict__ mu, long ldmu,
                                                             const float* __restrict__ log_std,
                                                             const float* __restrict__ actions, long lda, int B,
                                                             int A, float max_action, int act_tanh,
                                                             float* __restrict__ logp, float* __restrict__ entropy) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    float M = 0.f, sum_logs = 0.f;
    for (int a = 0; a < A; ++a) {
        const float e = expf(log_std[a]);
        const float s = e * e;
        sum_logs += logf(s);
        const float x = deactivate(actions[i * lda + a], max_action, act_tanh);
        const float z = (x - mu[i * ldmu + a]) / s;
        M += z * z;
    }
    if (logp) logp[i] = -0.5f * ((float)A * 1.8378770664093453f + M) - sum_logs;
    if (entropy) entropy[i] = 0.5f * (float)A * (1.0f + 1.8378770664093453f) + sum_logs;
}

extern "C" int pm_gaussian_logp_f32(const float* mu, long ldmu, const float* log_std, const float* actions,
                                    long lda, int B, int A, float max_action, int act_tanh, float* logp,
                                    float* entropy, void* stream) {
    PM_REQUIRE(mu && log_std && actions && B > 0 && A > 0 && (logp || entropy) && max_action > 0.f);
    hipLaunchKernelGGL(gaussian_logp_kernel, dim3((B + 255) / 256), dim3(256), 0, pm_stream(stream), mu, ldmu,
                       log_std, actions, lda, B, A, max_action, act_tanh, logp, entropy);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// Backward of the two rows above for the autograd bridge (partmanip_amd/autograd.py: the reference's own `update()`
// differentiates through ActorCritic.update_act_cri with torch autograd, ppo.py:326,347-348):
//   dmu[i][a]  = dlogp[i] * (x - mu) / s^2,  s = exp(log_std)^2 (the covariance factor of actor_critic.py:74-75)
//   dlog_std[a] = sum_i dlogp[i] * (2 z^2 - 2) + dent[i] * 2,  z = (x - mu) / s
// dlog_std: one work-group, fixed-order fp64 sums (deterministic).
__global__ __launch_bounds__(LOSS_THREADS) void gaussian_logp_bwd_kernel(const float* __restrict__ mu, long ldmu,
                                                                          const float* __restrict__ log_std,
                                                                          const float* __restrict__ actions, long lda,
                                                                          int B, int A, float max_action, int act_tanh,
                                                                          const float* __restrict__ dlogp,
                                                                          const float* __restrict__ dent,
                                                                          float* __restrict__ dmu, long lddm,
                                                                          float* __restrict__ dlog_std) {
    __shared__ double red[LOSS_WAVES];
    for (int a = 0; a < A; ++a) {
        const float e = expf(log_std[a]);
        const float s = e * e;
        double acc = 0.0;
        for (int i = threadIdx.x; i < B; i += LOSS_THREADS) {
            const float gl = dlogp ? dlogp[i] : 0.f;
            const float x = deactivate(actions[i * lda + a], max_action, act_tanh);
            const float z = (x - mu[i * ldmu + a]) / s;
            if (dmu) dmu[i * lddm + a] = gl * z / s;
            acc += (double)(gl * (2.0f * z * z - 2.0f)) + (dent ? (double)(2.0f * dent[i]) : 0.0);
        }
        acc = block_sum<double, LOSS_THREADS>(acc, red);
        if (threadIdx.x == 0 && dlog_std) dlog_std[a] = (float)acc;
        __syncthreads();
    }
}

extern "C" int pm_gaussian_logp_bwd_f32(const float* mu, long ldmu, const float* log_std, const float* actions, long lda,
                                        int B, int A, float max_action, int act_tanh, const float* dlogp,
                                        const float* dent, float* dmu, long lddm, float* dlog_std, void* stream) {
    PM_REQUIRE(mu && log_std && actions && B > 0 && A > 0 && (dmu || dlog_std) && max_action > 0.f);
    hipLaunchKernelGGL(gaussian_logp_bwd_kernel, dim3(1), dim3(LOSS_THREADS), 0, pm_stream(stream), mu, ldmu, log_std,
                       actions, lda, B, A, max_action, act_tanh, dlogp, dent, dmu, lddm, dlog_std);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// d/d(mu) of actor_critic.py:84-91: out = tanh(mu) * max_action  ->  dmu = dout * max_action * (1 - (out / max_action)^2)
__global__ __launch_bounds__(256) void action_activation_bwd_kernel(const float* __restrict__ out, const float* __restrict__ dout,
                                                                     float* __restrict__ dmu, long n, float max_action,
                                                                     int act_tanh) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float t = out[i] / max_action;
        dmu[i] = act_tanh ? dout[i] * max_action * (1.0f - t * t) : dout[i];
    }
}

extern "C" int pm_action_activation_bwd_f32(const float* out, const float* dout, float* dmu, long n, float max_action,
                                            int act_tanh, void* stream) {
    PM_REQUIRE(out && dout && dmu && n > 0 && max_action > 0.f);
    long b = (n + 255) / 256;
    if (b > 1024) b = 1024;
    hipLaunchKernelGGL(action_activation_bwd_kernel, dim3((int)b), dim3(256), 0, pm_stream(stream), out, dout, dmu, n,
                       max_action, act_tanh);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ---------------------------------------------------------------------------------- K9
__global__ __launch_bounds__(LOSS_THREADS) void value_loss_kernel(const float* __restrict__ V,
                                                                   const float* __restrict__ returns,
                                                                   const float* __restrict__ old_values, int B,
                                                                   int clipped, float eps_clip,
                                                                   const float* __restrict__ clip_mean_extern,
                                                                   float grad_scale, float* __restrict__ scal_out,
                                                                   float* __restrict__ dV, long lddv) {
    __shared__ double red[LOSS_WAVES];
    const int tid = threadIdx.x;
    float d = 0.f;
    if (clipped) {                       // ppo.py:370: delta = mean(|eps * V_old|) over the mini-batch
        if (clip_mean_extern) {
            d = clip_mean_extern[0];
        } else {
            double acc = 0.0;
            for (int i = tid; i < B; i += LOSS_THREADS) acc += (double)fabsf(eps_clip * old_values[i]);
            acc = block_sum<double, LOSS_THREADS>(acc, red);
            d = (float)(acc / (double)B);
        }
    }
    double acc = 0.0;
    const float k = 2.0f / (float)B * grad_scale;
    for (int i = tid; i < B; i += LOSS_THREADS) {
        float tgt = returns[i];
        if (clipped) {
            const float ov = old_values[i];
            tgt = ov + fminf(fmaxf(returns[i] - ov, -d), d);
        }
        const float e = V[i] - tgt;
        acc += (double)(e * e);
        dV[i * lddv] = k * e;
    }
    acc = block_sum<double, LOSS_THREADS>(acc, red);
    if (tid == 0) {
        scal_out[0] = (float)(acc / (double)B);
        scal_out[1] = d;
    }
}

extern "C" int pm_value_loss_fwd_bwd_f32(const float* V, const float* returns, const float* old_values, int B,
                                         int clipped, float eps_clip, const float* clip_mean_extern,
                                         float grad_scale, float* scal_out, float* dV, long lddv, void* stream) {
    PM_REQUIRE(V && returns && scal_out && dV && B > 0 && lddv >= 1);
    PM_REQUIRE(!clipped || old_values);
    hipLaunchKernelGGL(value_loss_kernel, dim3(1), dim3(LOSS_THREADS), 0, pm_stream(stream), V, returns, old_values,
                       B, clipped, eps_clip, clip_mean_extern, grad_scale, scal_out, dV, lddv);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ---- the value head and its loss as ONE launch (small-step regime), the critic's counterpart of ppo_actor_head_kernel:
// V = H w + b (row-wise, skinny.h) -> value_loss_kernel's rows -> dV -> dH = (dV w) .* act'(H) -> the last work-group sums the
// loss partials.  V, dV and dH carry the bits of the three separate launches (the clip width d is summed in value_loss_kernel's
// order by every work-group); the loss SCALAR is the same sum of squares in a different association (double precision).
#define VH_MAXG 128
__global__ __launch_bounds__(AH_THREADS) void value_head_kernel(
    const float* __restrict__ H, long ldh, const float* __restrict__ W, const float* __restrict__ bias, int K, int hact,
    const float* __restrict__ returns, const float* __restrict__ old_values, int B, int clipped, float eps_clip,
    const float* __restrict__ clip_mean_extern, float grad_scale, float* __restrict__ V_out, float* __restrict__ dV, long lddv,
    float* __restrict__ dH, long lddh, double* __restrict__ parts, unsigned int* __restrict__ counter, float* __restrict__ scal_out) {
    extern __shared__ __attribute__((aligned(16))) float sW[];                  // [K]
    __shared__ double red[AH_THREADS / 64];
    __shared__ float s_dv[AH_ROWS];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    sk_fill_w<AH_THREADS>(sW, W, K, 1, K);
    float d = 0.f;
    if (clipped) {                       // ppo.py:370: delta = mean(|eps * V_old|) over the mini-batch (value_loss_kernel's order)
        if (clip_mean_extern) {
            d = clip_mean_extern[0];
        } else {
            double acc = 0.0;
            for (int i = tid; i < B; i += AH_THREADS) acc += (double)fabsf(eps_clip * old_values[i]);
            acc = block_sum<double, AH_THREADS>(acc, red);
            d = (float)(acc / (double)B);
        }
    }
    __syncthreads();
    const float kq = 2.0f / (float)B * grad_scale;
    const int k4 = K >> 2;
    double acc = 0.0;
    for (int base = blockIdx.x * AH_ROWS; base < B; base += gridDim.x * AH_ROWS) {
        const int i = base + wave;                             // one row per wave
        if (i < B) {
            float a[SK_MAXN];
            sk_row_dot(H + (long)i * ldh, sW, 1, K, lane, a);
            const float s = wave_sum(a[0]);
            if (lane == 0) {
                const float v = s + (bias ? bias[0] : 0.f);
                float tgt = returns[i];
                if (clipped) {
                    const float ov = old_values[i];
                    tgt = ov + fminf(fmaxf(returns[i] - ov, -d), d);
                }
                const float e = v - tgt;
                acc += (double)(e * e);
                const float g = kq * e;
                dV[(long)i * lddv] = g;
                s_dv[wave] = g;
                if (V_out) V_out[i] = v;
            }
        }
        __syncthreads();
        for (int e = tid; e < AH_ROWS * k4; e += AH_THREADS) {
            const int rr = e / k4, k = (e - rr * k4) * 4;
            const long ii = base + rr;
            if (ii < B)
                *(float4*)(dH + ii * lddh + k) = sk_dgrad4([&](int) { return s_dv[rr]; }, sW, 1, K, k, H + ii * ldh, hact);
        }
        __syncthreads();
    }
    acc = block_sum<double, AH_THREADS>(acc, red);
    if (tid == 0) {
        __hip_atomic_store(parts + blockIdx.x, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // (see ppo_actor_head_kernel)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (s_last && tid < 64) {
        const int G = (int)gridDim.x;
        double s = tid < G ? __hip_atomic_load(parts + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        if (tid + 64 < G) s += __hip_atomic_load(parts + tid + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s = wave_sum(s);
        if (tid == 0) {
            scal_out[0] = (float)(s / (double)B);
            scal_out[1] = d;
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

extern "C" size_t pm_value_head_workspace_bytes(void) { return VH_MAXG * sizeof(double); }
extern "C" int pm_value_head_supported(const float* H, long ldh, const float* W, int K, const float* dH, long lddh) {
    return skinny_ok(1, K, H, ldh, W, K) && (long)K * 4 <= AH_MAX_DYN_LDS && lddh % 4 == 0 && sk_aligned16(dH);
}
extern "C" int pm_value_head_f32(const float* H, long ldh, const float* W, const float* bias, int K, int hidden_act,
                                 const float* returns, const float* old_values, int B, int clipped, float eps_clip,
                                 const float* clip_mean_extern, float grad_scale, float* scal_out, float* V_out, float* dV, long lddv,
                                 float* dH, long lddh, void* workspace, size_t workspace_bytes, unsigned int* counter, void* stream) {
    PM_REQUIRE(H && W && returns && scal_out && dV && dH && counter && B > 0 && K > 0 && ldh >= K && lddh >= K && lddv >= 1);
    PM_REQUIRE(!clipped || old_values);
    PM_REQUIRE(hidden_act >= PM_ACT_NONE && hidden_act <= PM_ACT_MAX);
    if (!pm_value_head_supported(H, ldh, W, K, dH, lddh)) return PM_EUNSUPPORTED;
    if (!workspace || workspace_bytes < pm_value_head_workspace_bytes() || ((uintptr_t)workspace & 7)) return PM_EWORKSPACE;
    int G = (B + AH_ROWS - 1) / AH_ROWS;
    if (G > VH_MAXG) G = VH_MAXG;
    hipLaunchKernelGGL(value_head_kernel, dim3(G), dim3(AH_THREADS), (size_t)K * 4, pm_stream(stream), H, ldh, W, bias, K, hidden_act,
                       returns, old_values, B, clipped, eps_clip, clip_mean_extern, grad_scale, V_out, dV, lddv, dH, lddh,
                       (double*)workspace, counter, scal_out);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ---------------------------------------------------------------------------------- K11
__global__ __launch_bounds__(LOSS_THREADS) void mse_tanh_loss_kernel(const float* __restrict__ stu_mu, long lds,
                                                                      const float* __restrict__ tea_mu, long ldt,
                                                                      int B, int A, float max_action, int act_tanh,
                                                                      float grad_scale, float* __restrict__ scal_out,
                                                                      float* __restrict__ dstu, long ldd) {
    __shared__ double red[LOSS_WAVES];
    const long n = (long)B * A;
    const float k = 2.0f / (float)n * grad_scale;
    double acc = 0.0;
    for (long e = threadIdx.x; e < n; e += LOSS_THREADS) {
        const long i = e / A, a = e % A;
        const float sm = stu_mu[i * lds + a], tm = tea_mu[i * ldt + a];
        float sa, ta, ds;
        if (act_tanh & 1) {
            const float th = pm_tanh(sm);
            sa = th * max_action;
            ta = (act_tanh & 2) ? tm : pm_tanh(tm) * max_action;      // bit 1: the target is a recorded ACTION (bc.py:139)
            ds = max_action * (1.0f - th * th);
        } else {
            sa = sm;
            ta = tm;
            ds = 1.0f;
        }
        const float diff = ta - sa;
        acc += (double)(diff * diff);
        dstu[i * ldd + a] = -k * diff * ds;
    }
    acc = block_sum<double, LOSS_THREADS>(acc, red);
    if (threadIdx.x == 0) scal_out[0] = (float)(acc / (double)n);
}

extern "C" int pm_mse_tanh_loss_fwd_bwd_f32(const float* stu_mu, long lds, const float* tea_mu, long ldt, int B,
                                            int A, float max_action, int act_tanh, float grad_scale,
                                            float* scal_out, float* dstu_mu, long ldd, void* stream) {
    PM_REQUIRE(stu_mu && tea_mu && scal_out && dstu_mu && B > 0 && A > 0 && max_action > 0.f);
    hipLaunchKernelGGL(mse_tanh_loss_kernel, dim3(1), dim3(LOSS_THREADS), 0, pm_stream(stream), stu_mu, lds, tea_mu,
                       ldt, B, A, max_action, act_tanh, grad_scale, scal_out, dstu_mu, ldd);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

__global__ __launch_bounds__(256) void action_activation_kernel(const float* __restrict__ mu, float* __restrict__ out,
                                                                 long n, float max_action, int act_tanh) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        out[i] = act_tanh ? pm_tanh(mu[i]) * max_action : mu[i];
}

extern "C" int pm_action_activation_f32(const float* mu, float* out, long n, float max_action, int act_tanh,
                                        void* stream) {
    PM_REQUIRE(mu && out && n > 0);
    long b = (n + 255) / 256;
    if (b > 1024) b = 1024;
    hipLaunchKernelGGL(action_activation_kernel, dim3((int)b), dim3(256), 0, pm_stream(stream), mu, out, n,
                       max_action, act_tanh);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ---------------------------------------------------------------------------------- stats
// ppo.py:335-336 (kl_max), :355-357 (sums over non-skipped actor steps), :384 (value loss).
__global__ void ppo_accumulate_stats_kernel(float* __restrict__ acc, const float* __restrict__ scal, int which) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (which == 0) {
        const float loss = scal[0], kl = scal[1], skip = scal[2];
        if (kl > acc[2]) acc[2] = kl;
        if (skip == 0.0f) {
            acc[0] += loss;
            acc[1] += kl;
            acc[3] += 1.0f;
        }
    } else {
        acc[4] += scal[0];
        acc[5] += 1.0f;
    }
}

extern "C" int pm_ppo_accumulate_stats_f32(float* acc, const float* scal, int which, void* stream) {
    PM_REQUIRE(acc && scal && (which == 0 || which == 1));
    hipLaunchKernelGGL(ppo_accumulate_stats_kernel, dim3(1), dim3(64), 0, pm_stream(stream), acc, scal, which);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ---------------------------------------------------------------------------------- rollout side
// actor_critic.py:36-47 `random_act_cri` after the two network forwards, in one launch: the sample of
// MultivariateNormal(mu, scale_tril = diag(sigma^2)) from caller-provided standard-normal noise (so the global torch
// RNG stream stays the reference's), its log-prob, and the squashed action that goes to the simulator.
__global__ __launch_bounds__(256) void gaussian_sample_kernel(const float* __restrict__ mu, long ldmu,
                                                               const float* __restrict__ log_std,
                                                               const float* __restrict__ eps, int B, int A,
                                                               float max_action, int act_tanh,
                                                               float* __restrict__ actions, float* __restrict__ logp,
                                                               float* __restrict__ log_std_rows) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    float M = 0.f, sum_logs = 0.f;
    for (int a = 0; a < A; ++a) {
        const float e = expf(log_std[a]);
        const float s = e * e;
        sum_logs += logf(s);
        const float m = mu[i * ldmu + a];
        const float x = add_rn(m, mul_rn(s, eps[(long)i * A + a]));     // loc + scale_tril @ eps, op by op
        const float z = (x - m) / s;
        M += z * z;
        actions[(long)i * A + a] = act_tanh ? pm_tanh(x) * max_action : x;
        if (log_std_rows) log_std_rows[(long)i * A + a] = log_std[a];
    }
    logp[i] = -0.5f * ((float)A * 1.8378770664093453f + M) - sum_logs;
}

extern "C" int pm_gaussian_sample_f32(const float* mu, long ldmu, const float* log_std, const float* eps, int B, int A,
                                      float max_action, int act_tanh, float* actions, float* logp, float* log_std_rows,
                                      void* stream) {
    PM_REQUIRE(mu && log_std && eps && actions && logp && B > 0 && A > 0 && ldmu >= A && max_action > 0.f);
    hipLaunchKernelGGL(gaussian_sample_kernel, dim3((B + 255) / 256), dim3(256), 0, pm_stream(stream), mu, ldmu, log_std,
                       eps, B, A, max_action, act_tanh, actions, logp, log_std_rows);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// RMS.py:10-18 `RunningMeanStd.update` + RMS.py:40-45 `Normalization.__call__`: one read of the (N, D) observation
// batch for the column moments (fp64 sums over row slabs, fixed-order finish), one read + one write for the
// normalisation -- HBM-bound; the torch expression makes five passes and three (N, D) temporaries.
#define RMS_ROWSPLIT 64
__global__ __launch_bounds__(256) void rms_partial_kernel(const float* __restrict__ x, long ldx, int N, int D,
                                                           double* __restrict__ part) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= D) return;
    const int per = (N + RMS_ROWSPLIT - 1) / RMS_ROWSPLIT;
    const int r0 = blockIdx.y * per, r1 = min(N, r0 + per);
    double s = 0.0, q = 0.0;
#pragma unroll 4
    for (int r = r0; r < r1; ++r) {
        const double v = (double)x[(long)r * ldx + col];
        s += v;
        q += v * v;
    }
    part[((size_t)blockIdx.y * D + col) * 2] = s;
    part[((size_t)blockIdx.y * D + col) * 2 + 1] = q;
}
__global__ __launch_bounds__(256) void rms_finish_kernel(const double* __restrict__ part, int nsplit, double N, int D,
                                                          int n_new, float* __restrict__ mean, float* __restrict__ S,
                                                          float* __restrict__ stdv, double* __restrict__ mom_out) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= D) return;
    double s = 0.0, q = 0.0;
    for (int y = 0; y < nsplit; ++y) {
        s += part[((size_t)y * D + col) * 2];
        q += part[((size_t)y * D + col) * 2 + 1];
    }
    if (mom_out) {                                      // moments only (data-parallel: the caller all-reduces them)
        mom_out[(size_t)col * 2] = s;
        mom_out[(size_t)col * 2 + 1] = q;
        return;
    }
    const double m64 = s / N;
    double w64 = q / N - m64 * m64;                     // mean((x - cur)^2), exact to fp64 round-off
    if (w64 < 0.0) w64 = 0.0;
    const float cur = (float)m64, within = (float)w64, prev = mean[col], n = (float)n_new;
    const float d = prev - cur;
    const float between = d * d * (float)(n_new - 1) / n;               // RMS.py:16, left to right
    mean[col] = prev + (cur - prev) / n;                                // RMS.py:15
    const float Sn = S[col] + within + between;
    S[col] = Sn;
    stdv[col] = sqrtf(Sn / n);                                          // RMS.py:17
}
__global__ __launch_bounds__(256) void rms_normalize_kernel(const float* __restrict__ x, long ldx, int N, int D,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ stdv, float* __restrict__ out,
                                                             long ldo) {
    const long total = (long)N * D;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / D;
        const int c = (int)(e - r * D);
        out[r * ldo + c] = (x[r * ldx + c] - mean[c]) / stdv[c];        // RMS.py:44 (a true division, as the reference)
    }
}

extern "C" size_t pm_rms_update_workspace_bytes(int D) { return D > 0 ? (size_t)RMS_ROWSPLIT * D * 2 * sizeof(double) : 0; }

extern "C" int pm_rms_update_f32(const float* x, long ldx, int N, int D, int n_new, float* mean, float* S, float* stdv,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    PM_REQUIRE(x && mean && S && stdv && N > 0 && D > 0 && ldx >= D && n_new >= 1);
    if (!workspace || workspace_bytes < pm_rms_update_workspace_bytes(D)) return PM_EWORKSPACE;
    if (((uintptr_t)workspace & 7) != 0) return PM_EALIGN;
    double* part = (double*)workspace;
    hipLaunchKernelGGL(rms_partial_kernel, dim3((D + 255) / 256, RMS_ROWSPLIT), dim3(256), 0, pm_stream(stream), x, ldx, N,
                       D, part);
    hipLaunchKernelGGL(rms_finish_kernel, dim3((D + 255) / 256), dim3(256), 0, pm_stream(stream), part, RMS_ROWSPLIT,
                       (double)N, D, n_new, mean, S, stdv, (double*)nullptr);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// The same update in two halves for the data-parallel learner (each rank holds an env shard of the batch): column
// moments {sum x, sum x^2} (fp64, interleaved per column) -> [the caller sums them over ranks] -> the RMS.py:10-18
// update from moments over `n_rows` rows.  One process calling both halves = pm_rms_update_f32.
extern "C" int pm_rms_moments_f64(const float* x, long ldx, int N, int D, double* mom, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    PM_REQUIRE(x && mom && N > 0 && D > 0 && ldx >= D);
    if (!workspace || workspace_bytes < pm_rms_update_workspace_bytes(D)) return PM_EWORKSPACE;
    if (((uintptr_t)workspace & 7) != 0 || ((uintptr_t)mom & 7) != 0) return PM_EALIGN;
    double* part = (double*)workspace;
    hipLaunchKernelGGL(rms_partial_kernel, dim3((D + 255) / 256, RMS_ROWSPLIT), dim3(256), 0, pm_stream(stream), x, ldx, N,
                       D, part);
    hipLaunchKernelGGL(rms_finish_kernel, dim3((D + 255) / 256), dim3(256), 0, pm_stream(stream), part, RMS_ROWSPLIT,
                       (double)N, D, 1, (float*)nullptr, (float*)nullptr, (float*)nullptr, mom);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

extern "C" int pm_rms_apply_moments_f32(const double* mom, long n_rows, int D, int n_new, float* mean, float* S,
                                        float* stdv, void* stream) {
    PM_REQUIRE(mom && mean && S && stdv && n_rows > 0 && D > 0 && n_new >= 1);
    if (((uintptr_t)mom & 7) != 0) return PM_EALIGN;
    hipLaunchKernelGGL(rms_finish_kernel, dim3((D + 255) / 256), dim3(256), 0, pm_stream(stream), mom, 1, (double)n_rows,
                       D, n_new, mean, S, stdv, (double*)nullptr);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

extern "C" int pm_rms_normalize_f32(const float* x, long ldx, int N, int D, const float* mean, const float* stdv,
                                    float* out, long ldo, void* stream) {
    PM_REQUIRE(x && mean && stdv && out && N > 0 && D > 0 && ldx >= D && ldo >= D);
    const long total = (long)N * D;
    long nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(rms_normalize_kernel, dim3((unsigned)nb), dim3(256), 0, pm_stream(stream), x, ldx, N, D, mean, stdv,
                       out, ldo);
    PM_CHECK_LAUNCH();
    return PM_OK;
}
