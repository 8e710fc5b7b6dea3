/*
 * partmanip_hip.h -- C ABI of libpartmanip_hip.so (MI355X / gfx950 only).
 *
 * The reference (PKU-EPIC/PartManip) is 100 % Python on PyTorch and has no FFI
 * of its own; the boundary it offers is the Python API of
 * algorithms/ppo.py, algorithms/dagger.py and the algorithms/algo_utils package.
 * This header is the NEW native boundary underneath that API: every entry
 * point replaces the ATen work one reference call site dispatches (cited per
 * function as file:line under /root/reference).  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers + sizes only; all pointers are DEVICE pointers unless
 *     the name ends in _host; no torch types; no hidden allocations.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*,
 *     NULL = the null stream), never synchronises, keeps no global state and
 *     is re-entrant.  Scratch memory is supplied by the caller
 *     (pm_*_workspace_bytes tells how much).
 *   - return value: PM_OK (0) or a negative PM_E* code; a HIP launch error is
 *     returned as -(1000 + hipError_t).
 *   - matrices are row-major fp32 with an explicit row stride in ELEMENTS
 *     (ld*), weights use torch.nn.Linear layout (out_features, in_features).
 */
#ifndef PARTMANIP_HIP_H
#define PARTMANIP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PM_OK 0
#define PM_EINVAL (-1)     /* bad argument (null pointer, non-positive size, unsupported shape) */
#define PM_EWORKSPACE (-2) /* workspace too small                                                */
#define PM_EALIGN (-3)     /* pointer / stride not aligned as required                           */
#define PM_EUNSUPPORTED (-4) /* shape has no fused instantiation: use the generic kernels instead  */

#define PM_ACT_NONE 0
#define PM_ACT_TANH 1
/* network.py:7-24 `get_activation`: the other activations a cfg may name (Linear epilogues only: the fused point-cloud /
 * voxel encoders are tanh kernels).  Each derivative is a function of the activation OUTPUT h, which is what the
 * backward kernels are handed: relu 1[h>0]; lrelu (slope 0.01) h>0 ? 1 : 0.01; elu (alpha 1) h>0 ? 1 : h+1;
 * selu h>0 ? lambda : h + lambda*alpha; sigmoid h(1-h). */
#define PM_ACT_RELU 2
#define PM_ACT_LRELU 3
#define PM_ACT_ELU 4
#define PM_ACT_SELU 5
#define PM_ACT_SIGMOID 6
#define PM_ACT_MAX 6

/* ABI version: major*10000 + minor*100 + patch */
#define PM_ABI_VERSION 152 /* bumped whenever an entry point is added or a signature changes */
int pm_version(void);      /* returns PM_ABI_VERSION of the built library: loaders compare it with their header */

/* ------------------------------------------------------------------ K1  GAE return scan
 * Replaces RolloutStorage.compute_returns, algorithms/algo_utils/storage.py:96-112.
 * rewards/values/returns/advantages: (T,N) fp32, N contiguous; dones/succs: (T,N) bytes
 * (torch.bool storage); last_values: (N).  gamma_lam = (float)(gamma*lam) computed in
 * double by the caller exactly as Python does.  use_succ==0 reproduces
 * `default_succ_value is None`.  Bit-exact with the reference (no FMA contraction).
 * Algorithmic HBM traffic: 18 B per env-step (8 B r,V + 2 B masks in, 8 B ret,adv out). */
int pm_gae_scan_f32(const float* rewards, const float* values, const uint8_t* dones, const uint8_t* succs,
                    const float* last_values, float* returns, float* advantages, int T, int N, float gamma,
                    float gamma_lam, int use_succ, float succ_value, void* stream);

/* ------------------------------------------------------------------ K2  advantage normalisation
 * storage.py:113-114 (whole batch) and ppo.py:329 (mini-batch): (x-mean)/(std_unbiased+eps).
 * pm_moments_f64 writes {sum, sum of squares} (fp64, deterministic two-stage reduction) to
 * moments[0..1]; data-parallel callers all-reduce those two doubles (and the count) before
 * pm_normalize_apply_f32.  Workspace: pm_moments_workspace_bytes(n). */
size_t pm_moments_workspace_bytes(long n);
int pm_moments_f64(const float* x, long n, double* moments, void* workspace, size_t workspace_bytes, void* stream);
int pm_normalize_apply_f32(float* x, long n, const double* moments, double count, float eps, void* stream);
/* The single-process form in ONE call (the name SURVEY.md 8b lists): x <- (x - mean) / (std_unbiased + eps) over x's n
 * elements = pm_moments_f64 + pm_normalize_apply_f32 with count = n.  Workspace (8-byte aligned): pm_adv_normalize_workspace_bytes. */
size_t pm_adv_normalize_workspace_bytes(long n);
int pm_adv_normalize_f32(float* x, long n, float eps, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ K3  mini-batch gather
 * ppo.py:317-324,361-364 / dagger.py:307-308: dst[i,:] = src[idx[i],:] for `sampler: random`
 * (sequential mini-batches are contiguous slices and need no copy). idx: int64 device. */
int pm_gather_rows_f32(const float* src, const int64_t* idx, float* dst, long n_rows, long row_elems,
                       long src_ld, long dst_ld, void* stream);

/* ------------------------------------------------------------------ K4/K5  Linear layers (fp32 MFMA)
 * network.py:53-54 (MLP) and network.py:154-160,196 (PointNet.final_mlp) forward, and the
 * autograd backward ppo.py:348,378 / dagger.py:318 triggers for them.
 *   fwd : Y[M,N]  = act(X[M,K] * W[N,K]^T + b[N])
 *   bwd_data  : dX[M,K] = (dY[M,N] * W[N,K]) .* act'(H[M,K])   (H = the layer input, i.e. the previous
 *               layer's activation OUTPUT; act' for tanh is 1-H^2, the others as listed at PM_ACT_*; H may be NULL with
 *               PM_ACT_NONE)
 *   bwd_weight: dW[N,K] = dY^T * X ; db[N] = column sums of dY (db may be NULL)
 * v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate. */
int pm_linear_fwd_f32(const float* X, long ldx, const float* W, long ldw, const float* b, float* Y, long ldy,
                      int M, int N, int K, int act, void* stream);
int pm_linear_bwd_data_f32(const float* dY, long lddy, const float* W, long ldw, const float* H, long ldh,
                           float* dX, long lddx, int M, int N, int K, int act, void* stream);
size_t pm_linear_bwd_weight_workspace_bytes(int M, int N, int K);
int pm_linear_bwd_weight_f32(const float* dY, long lddy, const float* X, long ldx, float* dW, long lddw,
                             float* db, int M, int N, int K, void* workspace, size_t workspace_bytes,
                             void* stream);

/* Whole-MLP forms (the names SURVEY.md 8b lists; network.py:27-54 `MLP`: activation after every layer but the last).  dims[0..
 * n_layers] = layer widths; W, b, H, dW, db = HOST arrays of n_layers DEVICE pointers; H[l] (M, dims[l+1]) = output of layer l.
 * They issue the per-layer entry points above in order on `stream` (same kernels, same bits); pm_mlp_bwd_f32 returns dW[l],
 * db[l] (db or db[l] may be NULL) and, when dX != NULL, the input gradient (M, dims[0]).  Workspace 16-byte aligned. */
int pm_mlp_fwd_f32(const float* X, long ldx, int M, int n_layers, const int* dims, const float* const* W, const float* const* b,
                   int act, float* const* H, void* stream);
size_t pm_mlp_bwd_workspace_bytes(int M, int n_layers, const int* dims);
int pm_mlp_bwd_f32(const float* X, long ldx, int M, int n_layers, const int* dims, const float* const* W, const float* const* H,
                   int act, const float* dY, float* const* dW, float* const* db, float* dX, void* workspace,
                   size_t workspace_bytes, void* stream);

/* Grouped forms (new; the reference runs one ATen GEMM per layer per network): up to PM_LINEAR_GROUP_MAX independent
 * problems of the SAME kind in ONE launch.  The small-step regime (state PPO, ppo.py:315-384: 2560 dependent optimiser
 * steps per iteration on 2048 x 512 matrices = one MFMA block per SIMD) is bound by per-launch ramp / drain and by
 * single-wave stalls; the actor's and the critic's layer l are independent (ppo.py:73-74: disjoint parameters), and so
 * are all weight gradients of a backward pass, so they share a grid.  Each desc has exactly the meaning of the
 * corresponding single-problem call.
 * pm_linear_bwd_weight_group_f32 with splits > 1 does NOT reduce: slab z (z = 0..splits-1) of problem i's dW / db is
 * written at dW + z * slab_stride / db + z * slab_stride (the reduction over M is cut into `splits` equal ranges of
 * whole 32-row steps); the caller sums the slabs in fixed order -- pm_clip_adam_group_f32 does it while it computes the
 * gradient norm.  With splits == 1 dW / db are final. */
#define PM_LINEAR_GROUP_MAX 8
typedef struct pm_linear_fwd_desc {
    const float* X; long ldx; const float* W; long ldw; const float* b; float* Y; long ldy; int M, N, K, act;
} pm_linear_fwd_desc;
typedef struct pm_linear_bwd_data_desc {
    const float* dY; long lddy; const float* W; long ldw; const float* H; long ldh; float* dX; long lddx; int M, N, K, act;
} pm_linear_bwd_data_desc;
typedef struct pm_linear_bwd_weight_desc {
    const float* dY; long lddy; const float* X; long ldx; float* dW; long lddw; float* db; long slab_stride; int M, N, K;
    int dy_cols, x_cols; /* 0, or the number of columns of a dY / X row that may be READ (>= N / K, <= the row stride): rows of 10 or 53
                          * floats inside 16- / 56-float strides take the 16-byte loaders with dy_cols = 16 / x_cols = 56; the extra
                          * columns' contents do not reach dW / db */
    int pad_;
} pm_linear_bwd_weight_desc;
int pm_linear_fwd_group_f32(int n, const pm_linear_fwd_desc* d, void* stream);
int pm_linear_bwd_data_group_f32(int n, const pm_linear_bwd_data_desc* d, void* stream);
int pm_linear_bwd_weight_group_f32(int n, const pm_linear_bwd_weight_desc* d, int splits, void* stream);
/* Chains: n (2..4) CONSECUTIVE layers of one network in one launch -- d[i + 1].X must be d[i].Y (forward: network.py:53-54
 * `self.model(x)`), resp. d[i + 1].dY must be d[i].dX (data gradient, top layer first) -- for the small-step regime, where a
 * kernel boundary between two 15 us layers costs a third of a layer.  A 64-row stripe of layer l + 1 depends on the same 64
 * rows of layer l only, so the work-groups of a stripe hand their tiles over inside the launch (write-through stores, one
 * arrival counter per stripe in `workspace`, bounded spins).  Same arithmetic per layer as the single-problem entry points.
 * Needs equal layer widths N (a multiple of 64), 16-byte-loadable operands with K % 32 == 0 (the FIRST forward layer may
 * instead have K <= 64 and unaligned rows: the input layer), and ceil(M / 64) * N / 64 <= 256 work-groups; otherwise returns
 * PM_EUNSUPPORTED (-4) and the caller issues the layers one by one.  workspace: pm_linear_chain_workspace_bytes(M) bytes,
 * 8-byte aligned, ZEROED ONCE by the caller and then handed only to chains of the same (M, N, n): the arrival counters in it
 * are monotonic (no memset per launch, graph-replay safe); its LAST 8-byte word is non-zero once a spin has given up
 * (results of that launch are invalid). */
size_t pm_linear_chain_workspace_bytes(int M);
int pm_linear_fwd_chain_f32(int n, const pm_linear_fwd_desc* d, void* workspace, size_t workspace_bytes, void* stream);
int pm_linear_bwd_data_chain_f32(int n, const pm_linear_bwd_data_desc* d, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ K6/K7  PointNet encoder
 * network.py:147-153,172-181: per-point shared MLP C->128->256->512 (act,act,none) over
 * (B, P, C) clouds fused with the symmetric max(/mean) pooling; the (B,P,512) activation
 * is never materialised.  feat (B, ldf) receives [max(512) | mean(512) if max_mean];
 * argmax (B,512) int32 is the pooling index (lowest index on ties, as torch.max).
 * x points are read at x + b*ldx + p*C (the reference's flat (B, P*C [+proprio]) rows).
 * Weights in torch layout: W1 (128,C), W2 (256,128), W3 (512,256).
 * pm_pointnet_pack_weights_f32 re-lays W2/W3 into the MFMA-operand order the kernels
 * stream (call after every optimiser step); packed size = pm_pointnet_packed_elems(). */
size_t pm_pointnet_packed_elems(void);
int pm_pointnet_pack_weights_f32(const float* W2, const float* W3, float* packed, void* stream);
int pm_pointnet_enc_fwd_f32(const float* x, long ldx, int B, int P, int C, int sub_mean, const float* W1,
                            const float* b1, const float* b2, const float* b3, const float* packed,
                            int max_mean, float* feat, long ldf, int32_t* argmax,
                            float* h2_save /* NULL, or (B, P, 256): the layer-2 activations for the backward */,
                            int act /* PM_ACT_* of the two hidden layers (network.py:147-153 takes any of get_activation's
                                       seven; PM_ACT_TANH -- every shipped cfg -- runs the tuned packed-tanh kernels) */,
                            void* stream);
/* OPT-IN split-bf16 forward: same contract as pm_pointnet_enc_fwd_f32, but the two big per-point GEMMs run as
 * a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on bf16 MFMAs with fp32 accumulation (~1e-5 relative instead of ~1e-7;
 * 5.3x less matrix-pipe time).  P must be a multiple of 128.  `packed` = pm_pointnet_packed_bf3_bytes() bytes
 * written by pm_pointnet_pack_weights_bf3 (hi/lo bf16 planes of W2, W3 in MFMA B-operand order). */
size_t pm_pointnet_packed_bf3_bytes(void);
int pm_pointnet_pack_weights_bf3(const float* W2, const float* W3, void* packed, void* stream);
int pm_pointnet_enc_fwd_bf3(const float* x, long ldx, int B, int P, int C, int sub_mean, const float* W1,
                            const float* b1, const float* b2, const float* b3, const void* packed, int max_mean,
                            float* feat, long ldf, int32_t* argmax, float* h2_save /* as pm_pointnet_enc_fwd_f32 */,
                            void* stream);
/* OPT-IN fp32-class split-bf16 forward: same contract again, with every operand split into THREE bf16 planes and the
 * six products a0b0, a0b1, a1b0, a0b2, a1b1, a2b0 accumulated in fp32 (dropped terms <= 2^-24 relative: the error
 * against fp64 is that of the fp32 MFMA kernel, at 2.7x less matrix-pipe time).  P must be a multiple of 64.
 * `packed` = pm_pointnet_packed_bf6_bytes() bytes written by pm_pointnet_pack_weights_bf6. */
size_t pm_pointnet_packed_bf6_bytes(void);
int pm_pointnet_pack_weights_bf6(const float* W2, const float* W3, void* packed, void* stream);
int pm_pointnet_enc_fwd_bf6(const float* x, long ldx, int B, int P, int C, int sub_mean, const float* W1,
                            const float* b1, const float* b2, const float* b3, const void* packed, int max_mean,
                            float* feat, long ldf, int32_t* argmax, float* h2_save /* as pm_pointnet_enc_fwd_f32 */,
                            void* stream);
/* Backward of the above w.r.t. the six encoder parameters given dfeat (B, ldf) =
 * [d max(512) | d mean(512)].  Uses the pooling structure: the gradient of the 512-wide
 * layer-3 output is (d mean)/P on every point plus (d max) on the argmax point only, so
 * layers 1-2 are recomputed per tile and only the 256->128 GEMMs run dense.
 * Workspace: pm_pointnet_enc_bwd_workspace_bytes(B). Gradients are WRITTEN (not accumulated). */
size_t pm_pointnet_enc_bwd_workspace_bytes(int B, int P, int C);
int pm_pointnet_enc_bwd_f32(const float* x, long ldx, int B, int P, int C, int sub_mean, const float* W1,
                            const float* b1, const float* b2, const float* W3, const float* packed,
                            int max_mean, const float* dfeat, long ldf, const int32_t* argmax, float* dW1,
                            float* db1, float* dW2, float* db2, float* dW3, float* db3,
                            const float* h2_saved /* NULL = recompute layer 2; else what the forward saved */,
                            int act /* as the forward's */, void* workspace, size_t workspace_bytes, void* stream);
/* OPT-IN fp32-class split-bf16 form of the same call (csrc/pointnet_enc_bwd_bf6.h): the two dense GEMMs of the backward
 * (dW2 = dz2^T h1, dh1 = dz2 W2) run on bf16 MFMAs with every operand split into three bf16 planes and six products summed in
 * the fp32 accumulator -- the fp32 kernel's error level at 2.7x less matrix-pipe time; everything else is the fp32 code.  tanh
 * encoders with the forward's saved layer 2 only (h2_saved != NULL).  packed_w2_bf6 = pm_pointnet_packed_bwd_bf6_bytes()
 * bytes written by pm_pointnet_pack_weights_bwd_bf6 from the current W2 (16-byte aligned); same workspace as the fp32 call. */
size_t pm_pointnet_packed_bwd_bf6_bytes(void);
int pm_pointnet_pack_weights_bwd_bf6(const float* W2, void* packed, void* stream);
int pm_pointnet_enc_bwd_bf6(const float* x, long ldx, int B, int P, int C, int sub_mean, const float* W1,
                            const float* b1, const float* b2, const float* W3, const float* packed,
                            const void* packed_w2_bf6, int max_mean, const float* dfeat, long ldf, const int32_t* argmax,
                            float* dW1, float* db1, float* dW2, float* db2, float* dW3, float* db3, const float* h2_saved,
                            void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ K8  PPO actor loss
 * actor_critic.py:74-78,93-100 + ppo.py:327-344 in one pass over a mini-batch of B rows:
 * atanh(clamp(a/max_a)), Gaussian log-prob with std = exp(log_std)^2, analytic KL to the
 * stored (mu_old, log_std_old), ratio, clipped surrogate; and its backward.
 * scal_out[0]=surrogate loss, [1]=kl mean, [2]=skip flag (kl_mean > desired_kl, as 0/1 float),
 * [3]=entropy (identical for every row; unused by the loss, ppo.py:326).  dmu (B,A) and
 * dlog_std (A) are d(loss)/d(.) and are ALWAYS written: the reference's `continue`
 * (ppo.py:337-338) is realised by handing &scal_out[2] to pm_clip_adam_step_f32 as its
 * skip flag, so no host sync sits between loss and optimiser.  adv_moments: NULL, or
 * {sum,sumsq} of the mini-batch advantages (count adv_count) for ppo.py:329's
 * mini_adv_norm (data-parallel callers all-reduce them first).  Two launches: <= 64 work-groups
 * of 256 rows, then a one-wave fixed-order reduction of their partials (deterministic). */
int pm_ppo_actor_loss_fwd_bwd_f32(const float* mu, long ldmu, const float* log_std, const float* actions,
                                  long lda, const float* old_logp, const float* adv, const float* old_mu,
                                  long ldom, const float* old_sigma, long ldos, int B, int A, float max_action,
                                  int act_tanh, float eps_clip, float desired_kl, const double* adv_moments,
                                  double adv_count, float* scal_out, float* dmu, long lddmu, float* dlog_std,
                                  void* workspace, size_t workspace_bytes, void* stream);
size_t pm_ppo_actor_loss_workspace_bytes(int B);
/* The policy head and its loss as ONE launch (small-step regime; ppo.py:322-347 from the last hidden activation on):
 * mu = H W^T + bias (H: B x K last hidden activations, W: A x K, the head of network.py:40) -> the loss rows of
 * pm_ppo_actor_loss_fwd_bwd_f32 -> dmu -> dH = (dmu W) .* act'(H) (hidden_act: PM_ACT_* of the hidden layers) -> the last
 * work-group to finish reduces the partials.  Same arithmetic and summation order as pm_linear_fwd_f32 (row-wise form) +
 * pm_ppo_actor_loss_fwd_bwd_f32 + pm_linear_bwd_data_f32: bit-identical outputs.  mu_out: NULL or B x A.  counter: one
 * device uint32, zero before the first launch (the kernel leaves it zero).  Shapes: A <= 16, K % 4 == 0, A * K * 4 <= 64 KiB,
 * 16-byte aligned rows (pm_ppo_actor_head_supported); otherwise PM_EUNSUPPORTED: use the three entry points separately. */
int pm_ppo_actor_head_supported(const float* H, long ldh, const float* W, long ldw, int A, int K, const float* dH, long lddh);
int pm_ppo_actor_head_f32(const float* H, long ldh, const float* W, long ldw, const float* bias, int K, int hidden_act,
                          const float* log_std, const float* actions, long lda, const float* old_logp, const float* adv,
                          const float* old_mu, long ldom, const float* old_sigma, long ldos, int B, int A, float max_action,
                          int act_tanh, float eps_clip, float desired_kl, const double* adv_moments, double adv_count,
                          float* scal_out, float* mu_out, long ldmu, float* dmu, long lddmu, float* dH, long lddh,
                          float* dlog_std, void* workspace, size_t workspace_bytes, unsigned int* counter, void* stream);
/* forward-only: log-prob/entropy rows, actor_critic.py:71-82 (used by rollout + tests). */
int pm_gaussian_logp_f32(const float* mu, long ldmu, const float* log_std, const float* actions, long lda,
                         int B, int A, float max_action, int act_tanh, float* logp, float* entropy,
                         void* stream);

/* ------------------------------------------------------------------ K9  value loss
 * ppo.py:368-374: mean((ret-V)^2), or the clipped variant with the batch-mean clip width.
 * clip_mean_extern: NULL or device float = all-reduced mean(|eps*V_old|) (data parallel).
 * scal_out[0] = loss.  dV (B rows of stride lddv >= 1) = d loss / dV * grad_scale. */
int pm_value_loss_fwd_bwd_f32(const float* V, const float* returns, const float* old_values, int B,
                              int clipped, float eps_clip, const float* clip_mean_extern, float grad_scale,
                              float* scal_out, float* dV, long lddv, void* stream);

/* The value head and its loss as ONE launch (small-step regime; ppo.py:362-377 from the last hidden activation on):
 * V = H w + bias (H: B x K, w: the (1 x K) head of network.py:40) -> the loss above -> dV (rows of stride lddv) ->
 * dH = (dV w) .* act'(H) -> the last work-group to finish sums the loss partials.  V, dV, dH: the bits of pm_linear_fwd_f32
 * (row-wise form) + pm_value_loss_fwd_bwd_f32 + pm_linear_bwd_data_f32; scal_out[0]: the same sum in another association
 * (double precision).  V_out: NULL or B.  counter: one device uint32, zero before the first launch (left zero).  K % 4 == 0,
 * K * 4 <= 64 KiB, 16-byte aligned rows (pm_value_head_supported), otherwise PM_EUNSUPPORTED. */
size_t pm_value_head_workspace_bytes(void);
int pm_value_head_supported(const float* H, long ldh, const float* W, int K, const float* dH, long lddh);
int pm_value_head_f32(const float* H, long ldh, const float* W, const float* bias, int K, int hidden_act, const float* returns,
                      const float* old_values, int B, int clipped, float eps_clip, const float* clip_mean_extern,
                      float grad_scale, float* scal_out, float* V_out, float* dV, long lddv, float* dH, long lddh,
                      void* workspace, size_t workspace_bytes, unsigned int* counter, void* stream);

/* ------------------------------------------------------------------ K11 DAgger loss
 * dagger.py:310-314: mean((tanh(tea_mu)*max_a - tanh(stu_mu)*max_a)^2) over B*A and
 * d/d stu_mu.  act_tanh==0 -> identity squashing (actor_critic.py:87-88).  act_tanh==3: the student is
 * squashed but `tea_mu` already holds recorded ACTIONS -- bc.py:139 `(actions - stu_act).pow(2).mean()`. */
int pm_mse_tanh_loss_fwd_bwd_f32(const float* stu_mu, long lds, const float* tea_mu, long ldt, int B, int A,
                                 float max_action, int act_tanh, float grad_scale, float* scal_out,
                                 float* dstu_mu, long ldd, void* stream);
/* actor_critic.py:84-91 */
int pm_action_activation_f32(const float* mu, float* out, long n, float max_action, int act_tanh, void* stream);
/* Backward halves of pm_gaussian_logp_f32 / pm_action_activation_f32 for callers that differentiate through
 * ActorCritic.update_act_cri / update_act with torch autograd, the way the reference's own update() does
 * (ppo.py:326,347-348; dagger.py:312-318; partmanip_amd/autograd.py wraps them in torch.autograd.Function):
 * dmu (B, A) = dlogp[i] * (x - mu) / s^2 and dlog_std (A) = sum_i dlogp[i] (2 z^2 - 2) + 2 dent[i], with s = exp(log_std)^2,
 * z = (x - mu) / s, x = atanh(clamp(a / max_action)) -- dlogp / dent (B) are the incoming gradients (either may be NULL). */
int pm_gaussian_logp_bwd_f32(const float* mu, long ldmu, const float* log_std, const float* actions, long lda, int B, int A,
                             float max_action, int act_tanh, const float* dlogp, const float* dent, float* dmu, long lddm,
                             float* dlog_std, void* stream);
int pm_action_activation_bwd_f32(const float* out, const float* dout, float* dmu, long n, float max_action, int act_tanh,
                                 void* stream);

/* ------------------------------------------------------------------ K10 clip + Adam
 * nn.utils.clip_grad_norm_ (ppo.py:351,381) fused with torch.optim.Adam.step (ppo.py:353,382,
 * dagger.py:319) over ONE flat fp32 parameter buffer: the L2 norm / clip coefficient
 * min(1, max_norm/(norm+1e-6)) covers the first n_clip elements only (the actor optimiser
 * also owns log_std, which the reference does not clip); max_norm<=0 disables clipping.
 * state (device, 4 x int32): [0]=step count t (incremented here unless skipped).
 * skip_flag: NULL or device float; non-zero -> the whole step is a no-op (ppo.py:337-338).
 * Adam: betas (b1,b2), eps, bias-corrected exactly as torch (denom = sqrt(v)/sqrt(1-b2^t)+eps,
 * step = lr/(1-b1^t)).  Workspace: pm_clip_adam_workspace_bytes(n). */
size_t pm_clip_adam_workspace_bytes(long n);
/* The same step for several optimisers in two launches (ppo.py:353 and :382 are independent: optimizer_actor and
 * optimizer_critic), each optionally summing split-K gradient slabs first: grads[i] += sum_{s=1..n_extra} extra[(s-1) *
 * extra_stride + i] for i < n_sum (fixed order; grads is slab 0 of pm_linear_bwd_weight_group_f32), then exactly
 * pm_clip_adam_step_f32.  workspace: pm_clip_adam_workspace_bytes(n) doubles per desc, 8-byte aligned. */
typedef struct pm_clip_adam_desc {
    float* params; float* grads; float* exp_avg; float* exp_avg_sq; long n, n_clip; const float* extra; long extra_stride;
    long n_sum; int n_extra; float max_norm; double lr, b1, b2, eps; int32_t* state; const float* skip_flag;
    float* gnorm_out; void* workspace;
    /* optional (stats_acc != NULL): pm_ppo_accumulate_stats_f32(stats_acc, stats_scal, stats_which) folded into the norm pass */
    float* stats_acc; const float* stats_scal; int stats_which;
    /* data-parallel step (new: the reference has no collective): `grads` and the step's scalars arrive as the
     * all-reduce SUM over W ranks.  The form is on when dp_scal != NULL or grad_scale is neither 0 nor 1; it requires n_extra == 0
     * (fold the slabs with pm_grad_slab_sum_f32 BEFORE the reduce).  grad_scale = 1/W: the norm pass first rewrites grads[0..n) *= grad_scale
     * (mean gradient; clip + Adam then see exactly what a single process with the W-fold mini-batch computes).  dp_scal
     * (NULL = off): the step's scalar record {loss, kl, skip}: [0], [1] *= grad_scale, and when dp_kl_desired > 0 the KL
     * early-stop predicate of ppo.py:337-338 is re-taken from the REDUCED kl ([2] = kl > dp_kl_desired) before skip_flag /
     * stats are read -- every rank takes the same branch without a host round trip or an extra launch. */
    float grad_scale; float dp_kl_desired; float* dp_scal;
} pm_clip_adam_desc;
int pm_clip_adam_group_f32(int n, const pm_clip_adam_desc* d, void* stream);
/* grads[i] += sum_{s=1..n_extra} extra[(s-1) * extra_stride + i] for i < n_sum, in the order pm_clip_adam_group_f32's norm
 * pass uses (bit-identical): the split-K slabs of pm_linear_bwd_weight_group_f32 folded BEFORE a gradient all-reduce, so
 * that the message is one slab and not n_extra + 1. */
int pm_grad_slab_sum_f32(float* grads, const float* extra, long extra_stride, long n_sum, int n_extra, void* stream);
int pm_clip_adam_step_f32(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n,
                          long n_clip, float max_norm, double lr, double b1, double b2, double eps,
                          int32_t* state, const float* skip_flag, float* gnorm_out, void* workspace,
                          size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ scalar bookkeeping of ppo.update
 * ppo.py:335-336,355-357,384: device-side running sums so the host never syncs per
 * mini-batch. acc (device, 8 floats): [0]=sum surrogate,[1]=sum kl,[2]=kl max,[3]=count,
 * [4]=sum value loss,[5]=n value steps.  which: 0 = actor step (reads scal[0..2]), 1 = critic. */
int pm_ppo_accumulate_stats_f32(float* acc, const float* scal, int which, void* stream);

/* ------------------------------------------------------------------ K12-K14 point-set operators
 * K12: utils/depth2tsdf.py:113,160 call pytorch3d.ops.sample_farthest_points(points, K)
 * (un-vendored; defaults: start index 0, no lengths).  idx_out (B,K) int32; squared-L2
 * running-min distance; next = argmax, lowest index on ties.  PARITY UNPINNED (no reference
 * implementation in tree) -- checked against the restatement `oracle/ref_cpu.py::fps` / `ball_query` / `group_points`
 * (tests/test_gpu_kernels.py, tests/test_gpu_fuzz.py: indices bit-exact).
 * K13/K14: PointNet++ ball query / grouping -- absent from the reference (README.md:23,30),
 * mandated by BASELINE.json.north_star; first `nsample` in-radius indices in ascending
 * order, padded with the first hit, all-zero row when the ball is empty. */
size_t pm_fps_workspace_bytes(int B, int P);   /* 0 when the cloud fits in registers (P <= 8192) */
int pm_fps_f32(const float* xyz, int B, int P, int D, int K, int32_t* idx_out, void* workspace,
               size_t workspace_bytes, void* stream);
int pm_ball_query_f32(const float* xyz, const float* centers, int B, int P, int S, float radius, int nsample,
                      int32_t* idx_out, void* stream);
int pm_group_points_f32(const float* feat, const int32_t* idx, int B, int P, int C, int S, int nsample,
                        float* out, void* stream);
/* (the gradient of the gather: per source point the sum of its rows of dout in ASCENDING row order -- one work-group per cloud inverts
 * the index table in LDS, no floating-point atomics, every element of dfeat written -- when 2 P + S nsample <= ~15.7 k; larger
 * tables: fp32 atomics into the zero-filled dfeat, summation order not fixed.  pm_group_concat_bwd_f32 alike.) */
int pm_group_points_bwd_f32(const float* dout, const int32_t* idx, int B, int P, int C, int S, int nsample,
                            float* dfeat, void* stream);

/* ------------------------------------------------------------------ K15 PointNet++ set-abstraction glue
 * (absent from the reference; north-star mandated; parity unpinned, own oracle).
 * group_concat: out[b,s,j,:] = [xyz[b,idx]-centers[b,s] (3) | feat[b,idx,:] (Cf) | zero pad to ldo]
 * (the rows the shared per-point MLP consumes); its backward scatter-adds the feature columns
 * (fixed-order sums like pm_group_points_bwd_f32; fp32 atomics into the zero-filled dfeat only beyond its size limit).  maxpool_rows: max over the nsample axis of
 * (G, nsample, C) with the lowest arg-max index; its backward writes every element of dx. */
int pm_group_concat_f32(const float* xyz, const float* feat, const float* centers, const int32_t* idx, int B, int P,
                        int Cf, int S, int nsample, int ldo, float* out, void* stream);
/* Column-block copy: dst[r][d_b .. d_b + e_b - s_b) = src[r][s_b .. e_b) for two blocks b (an empty block: s == e); the other
 * columns of dst in [col0, dst_cols) are zeroed when zero_other != 0; columns below col0 are left alone.  A non-empty block must land
 * inside [col0, dst_cols), the two blocks must not overlap, src may be NULL when both blocks are empty (zero-only), and dst == src is
 * allowed only with blocks that read below col0 (PM_EINVAL otherwise -- never a silently skipped block).  The PointNet++ plug-in's
 * glue around its GEMMs in one launch each (weights in operand column order padded to the K-step, gradients back, [xyz | 0] behind the
 * group-all rows' features). */
int pm_col_blocks_f32(float* dst, long ldd, const float* src, long lds, long rows, int dst_cols, int col0, int s0, int e0, int d0,
                      int s1, int e1, int d1, int zero_other, void* stream);
/* dst[q] = table[q] >= 0 ? src[table[q]] : 0 for q < n.  Every weight-derived operand copy of a network (the pack entry points'
 * outputs, aligned / padded column blocks) is such a gather of the flat parameter buffer with a table fixed by the layouts: the host
 * records the table once (by running the pack entry points on index-valued weights) and refreshes all copies with this ONE launch per
 * forward instead of one tiny launch per copy. */
int pm_gather_copy_f32(float* dst, const float* src, const int32_t* table, long n, void* stream);
int pm_group_concat_bwd_f32(const float* dout, const int32_t* idx, int B, int P, int Cf, int S, int nsample, int ldo,
                            float* dfeat, void* stream);
int pm_maxpool_rows_f32(const float* x, long G, int nsample, int C, float* out, long ldo, int32_t* arg, void* stream);
int pm_maxpool_rows_bwd_f32(const float* dout, long lddo, const int32_t* arg, long G, int nsample, int C,
                            const float* y_tanh /* NULL, or the pooled tanh outputs: dx *= 1-y^2 */, float* dx,
                            void* stream);

/* ------------------------------------------------------------------ depth images -> world-frame cloud
 * utils/depth2tsdf.py:142-157 (TSDFVolume.depth2pc before the sampling at :160): pinhole back-projection of
 * depth (B, M, H, W), rigid transform with cam_pose (M, 4, 4 row-major, camera -> world), points outside the
 * OPEN box (lo, hi) zeroed.  out (B, M*H*W, 3).  lo / hi are HOST pointers to 3 floats.  K12 then samples it. */
int pm_depth_backproject_f32(const float* depth, int B, int M, int H, int W, const float* cam_pose, float fx,
                             float fy, float cx, float cy, const float* lo, const float* hi, float* out,
                             void* stream);

/* Crop compaction + variable-length FPS for the step above.  pm_depth_compact_f32: out (B, P, 3) receives, per
 * env and in the original order, every non-zero point plus the FIRST zero point; lengths[b] = how many.  FPS over
 * that prefix selects the same sequence of POINTS as FPS over the full cropped cloud (the zeroed points are one
 * candidate).  pm_fps_varlen_f32: clouds are the first lengths[b] rows of a (B, ld, D) buffer; K rounds always
 * (no -1 padding: once every point is taken the lowest index repeats, as on the full cloud); workspace
 * B*ld floats when ld > 8192. */
int pm_depth_compact_f32(const float* xyz, int B, int P, float* out, int32_t* lengths, void* stream);
/* Workspace: 0 when every cloud fits in registers (ld <= 8192).  With the full reservation (8-byte aligned) camera-sized xyz
 * clouds run on pm_fps_varlen_groups(B, ld, D) >= 2 work-groups per cloud, each keeping its chunk on chip; the partners of a
 * cloud wait for each other INSIDE the launch, with ONE bounded poll budget per work-group and launch (~1 s; pm_fps_config.spin_limit).
 * A work-group that exhausts it gives up ONCE: it latches the flag, sets the reservation's last 8 bytes (which every other
 * work-group of the launch watches and follows) and returns; the call then DEGRADES on the device: a launch queued behind it
 * re-samples the batch's big clouds on one work-group each when -- and only when -- that word is set (no host round trip, the
 * indices are valid either way; the word stays set for diagnostics until the next multi-work-group call clears it).  Other
 * kernels on the device only delay the hand-offs (their work-groups drain); two such launches running concurrently on two
 * streams can hold each other's CUs until the budget trips -- issue them from one stream, or cap the group count (pm_fps_config.max_groups,
 * 0 / 1 = one work-group per cloud) when CUs are masked or shared. */
/* Launch policy of the multi-work-group sampler, PASSED by the caller (the library reads no environment variable and keeps no
 * state between calls); NULL = the defaults.  max_groups: cap on the work-groups per cloud (< 0: none; 0 / 1: one work-group per
 * cloud -- when CUs are masked or shared); resident_cus: the CUs the caller knows the launch can occupy (0: all the device
 * reports -- wrong under a CU mask or beside a long-running kernel, which is why it is the caller's to say); spin_limit (when
 * spin_limit_set): the poll budget of a hand-off; legacy_shape: the 1024-thread x 16-point launch shape (A/B). */
typedef struct pm_fps_config {
    int max_groups;
    int resident_cus;
    unsigned spin_limit;
    int spin_limit_set;
    int legacy_shape;
} pm_fps_config;
size_t pm_fps_varlen_workspace_bytes(int B, int ld);
int pm_fps_varlen_groups(int B, int ld, int D);                                  /* = _cfg(..., NULL) */
int pm_fps_varlen_groups_cfg(int B, int ld, int D, const pm_fps_config* cfg);
int pm_fps_varlen_cfg_f32(const float* xyz, int B, int ld, int D, int K, const int32_t* lengths, int pad, int32_t* idx_out,
                          const pm_fps_config* cfg, void* workspace, size_t workspace_bytes, void* stream);
int pm_fps_varlen_f32(const float* xyz, int B, int ld, int D, int K, const int32_t* lengths,
                      int pad /* 1: pytorch3d semantics, -1 once a cloud is exhausted; 0: keep sampling (see above) */,
                      int32_t* idx_out, void* workspace, size_t workspace_bytes, void* stream);
/* utils/depth2tsdf.py:103-119 (`TSDFVolume.sparse_voxel` after pm_tsdf_integrate_f32): pm_tsdf_select_f32 writes, per
 * env and in row-major voxel order (= torch.where), the integer coordinates (as floats) of the voxels with
 * lo < tsdf < hi into coords (B, res^3, 3) and their count into lengths; pm_fps_varlen_f32 (pad = 1) samples them;
 * pm_tsdf_sparse_gather_f32 emits out (B, K, 4) = (x, y, z, tsdf), padding indices reading voxel (0,0,0). */
int pm_tsdf_select_f32(const float* vol, int B, int res, float lo, float hi, float* coords, int32_t* lengths,
                       void* stream);
int pm_tsdf_sparse_gather_f32(const float* coords, const int32_t* idx, const float* vol, int B, int res, int K,
                              float* out, void* stream);

/* algorithms/algo_utils/network.py:72,119 -- Conv3DNet's INPUT layer, Conv3d(1, Cout, k, stride, padding k/2), as a direct
 * stencil over the single-channel volume (no patch matrix): x (B, D, H, W) through element strides (sb, sd, sh, sw);
 * wt = conv.weight viewed (Cout, k^3) TRANSPOSED to (k^3, Cout); y (B*Do*Ho*Wo, ldy) rows of Cout = act(conv + bias)
 * (channels-last, what the next layer's pm_im2col3d_f32 reads through strides).  pm_conv3d_c1_wgrad_f32: dW (Cout, lddw
 * >= k^3) and db (Cout, may be NULL) from dz (rows, lddz) = the gradient at this layer's pre-activation.  Instantiated
 * for k = 5, Cout = 16 (pm_conv3d_c1_supported); other shapes return PM_EUNSUPPORTED: use im2col + the Linear kernels.
 * batch_index (may be NULL): volume b of the batch is row batch_index[b] (stride sb) of a larger store -- DAgger's
 * mini-batches are random rows of the ring (dagger.py:305-306): the 0.8 GB gathered copy of 1600 volumes is not made. */
int pm_conv3d_c1_supported(int k, int Cout);
size_t pm_conv3d_c1_wgrad_workspace_bytes(int Cout);
int pm_conv3d_c1_fwd_f32(const float* x, int B, int D, int H, int W, int k, int stride, int pad, long sb, long sd, long sh,
                         long sw, const float* wt, const float* bias, int Cout, int act, float* y, long ldy,
                         const int64_t* batch_index, void* stream);
int pm_conv3d_c1_wgrad_f32(const float* dz, long lddz, const float* x, int B, int D, int H, int W, int k, int stride,
                           int pad, long sb, long sd, long sh, long sw, int Cout, float* dW, long lddw, float* db,
                           const int64_t* batch_index, void* workspace, size_t workspace_bytes, void* stream);
/* ------------------------------------------------------------------ Conv3D students: patch gather / scatter
 * network.py:56-94 (`Conv3DNet` / `Encoder`: nn.Conv3d(k, stride, padding = k/2)).  A convolution runs as
 * cols = im2col(x) -> pm_linear_fwd_f32 with conv.weight viewed (Cout, Cin*k^3) -> rows (b, od, oh, ow) x Cout;
 * its backward as pm_linear_bwd_weight_f32 / pm_linear_bwd_data_f32 on the same cols and dx = col2im(dcols).
 * x / dx are addressed by ELEMENT strides (sb, sc, sd, sh, sw): channels-first inputs and channels-last layer
 * outputs both work.  cols / dcols: (B*Do*Ho*Wo, ldc), column (c, kd, kh, kw) with c slowest; im2col zero-fills
 * columns [C*k^3, ldc).  Do = (D + 2*pad - k)/stride + 1 etc.  col2im overwrites every dx element. */
int pm_im2col3d_f32(const float* x, int B, int C, int D, int H, int W, int k, int stride, int pad, long sb, long sc,
                    long sd, long sh, long sw, float* cols, int ldc, void* stream);
int pm_col2im3d_f32(const float* dcols, int B, int C, int D, int H, int W, int k, int stride, int pad, long sb,
                    long sc, long sd, long sh, long sw, const float* y_tanh /* NULL, or x itself when it is an activation
                    output laid out like dx: dx *= act'(x), 1 - x^2 for tanh */, int act /* PM_ACT_* of that layer */,
                    float* dx, int ldc, void* stream);

/* ------------------------------------------------------------------ TSDF integration (observation side)
 * utils/depth2tsdf.py:68-86 `TSDFVolume.integrate`: depth (B, M, H*W) -> out (B, V) truncated signed distances
 * averaged over the views that see each voxel.  pix_idx (M, V) int32 = row*W + col of the voxel's pixel in view m,
 * or -1 outside the frustum; pix_z (M, V) = its camera-frame depth (both built at register_camera time,
 * depth2tsdf.py:41-60).  trunc = 4 voxels; default_tsdf for voxels no view sees. */
int pm_tsdf_integrate_f32(const float* depth, const int32_t* pix_idx, const float* pix_z, int B, int M, long HW,
                          long V, float trunc, float default_tsdf, float* out, void* stream);

/* ------------------------------------------------------------------ K15 fused set-abstraction level
 * One PointNet++ SA level (north_star; not in the reference snapshot, README.md:23,30) as one forward and
 * one backward kernel: ball-query groups of `nsample` = 32 rows [xyz[idx]-centre | feat[idx]] -> Linear C1,
 * tanh -> Linear C2, tanh -> Linear C3, tanh -> max over the group.  Layer 1's feature part is passed in
 * pre-multiplied per SOURCE point: Y (B*P, C1) = feat * W1[:, 3:3+Cf]^T (pm_linear_fwd_f32, no bias, no
 * activation), NULL when the level has no input features.  The grouped rows and per-row activations never
 * reach HBM.  pooled (B*S, ldp) = tanh(max_j z3 + b3), arg (B*S, C3) = lowest row j attaining the max of the
 * pre-activation.  Instantiated for (C1,C2,C3) in {(64,64,128), (128,128,256)} and nsample = 32
 * (pm_sa_supported); other shapes return PM_EUNSUPPORTED and the caller uses K4/K13-K15 separately.
 * Backward: dpooled (B*S, lddp) -> dW1[:, 0:3] (leading dimension lddw1; other columns untouched), db1, dW2,
 * db2, dW3, db3 (all overwritten) and dY (B*P, C1), which the caller zero-fills (fp32 atomics), or NULL. */
int pm_sa_supported(int C1, int C2, int C3, int nsample);
size_t pm_sa_packed_elems(int C1, int C2, int C3);
int pm_sa_pack_weights_f32(const float* W2, const float* W3, int C1, int C2, int C3, float* packed, void* stream);
int pm_sa_fwd_f32(const float* xyz, const float* centers, const int32_t* idx, const float* Y, int B, int P, int S,
                  int nsample, const float* W1, long ldw1, const float* b1, const float* b2, const float* b3,
                  const float* packed, int C1, int C2, int C3, float* pooled, long ldp, int32_t* arg,
                  float* h2_save /* NULL, or (B*S*32, C2): layer-2 activations kept for the backward */, void* stream);
size_t pm_sa_bwd_workspace_bytes(int C1, int C2, int C3);
int pm_sa_bwd_f32(const float* xyz, const float* centers, const int32_t* idx, const float* Y, int B, int P, int S,
                  int nsample, const float* W1, long ldw1, const float* b1, const float* b2, const float* W3,
                  const float* packed, int C1, int C2, int C3, const float* pooled, long ldp, const int32_t* arg,
                  const float* dpooled, long lddp, float* dW1, long lddw1, float* db1, float* dW2, float* db2,
                  float* dW3, float* db3, float* dY, const float* h2_saved /* NULL = recompute layer 2 */,
                  void* workspace, size_t workspace_bytes, void* stream);

/* Duplicate-free ("packed") form of the fused level.  Ball query pads a short group with copies of its first hit; a
 * copy computes the same activations as row 0 of its group and the max-pool keeps the LOWEST row attaining the maximum,
 * so copies never win and never receive a gradient: the level is exactly the level over each group's DISTINCT rows.
 * pm_sa_plan_i32 (coordinates only: once per neighbourhood table) lists those rows back to back and cuts them into tiles
 * of whole groups; pm_sa_fwd_packed_f32 / pm_sa_bwd_packed_f32 run the same per-row arithmetic as pm_sa_fwd_f32 /
 * pm_sa_bwd_f32 on the tiles.  pooled, arg and the saved layer 2 are bit-identical to the dense entry points' (arg is
 * the row inside the group in both); weight gradients agree to fp32 summation order.
 *   grow    (B*S + 1)         first packed row of each group, grow[B*S] = R
 *   rowmap  (2 * B*S*nsample) capacity: {flat source point b*P + idx, (group - first group of its tile) << 8 | row inside the
 *                             group} per packed row; R pairs are written
 *   relxyz  (4 * B*S*nsample) capacity: xyz[source point] - centre[group] per packed row (x, y, z, 0): the staging of the level
 *                             kernels is two coalesced loads per row (coordinates only: weights never enter the plan)
 *   tiles   (4 * B*S)         capacity: {first row, first group, groups, rows} per tile; T quadruples are written
 *   totals  (4)               [0] = R packed rows, [1] = T tiles -- read by the kernels on the device (persistent
 *                             work-groups), never by the host: nothing here synchronises
 *   h2_save / h2_saved        (R, C2) -- capacity (B*S*nsample, C2) when R is not read back
 * tile_rows / tile_groups come from pm_sa_packed_tile (they are the kernels' tile shapes for that level shape). */
int pm_sa_packed_tile(int C1, int C2, int C3, int* tile_rows, int* tile_groups);
size_t pm_sa_plan_workspace_bytes(int B, int S);
int pm_sa_plan_i32(const int32_t* idx, const float* xyz, const float* centers, int B, int P, int S, int nsample, int tile_rows,
                   int tile_groups, int32_t* grow, int32_t* rowmap, float* relxyz, int32_t* tiles, int32_t* totals, void* workspace,
                   size_t workspace_bytes, void* stream);
int pm_sa_fwd_packed_f32(const float* Y, int B, int P, int S, const int32_t* grow, const int32_t* rowmap, const float* relxyz,
                         const int32_t* tiles, const int32_t* totals, const float* W1, long ldw1, const float* b1, const float* b2,
                         const float* b3, const float* packed, int C1, int C2, int C3, float* pooled, long ldp, int32_t* arg,
                         float* h2_save,
                         const float* tail_xyz /* (B*S, 3) or NULL: columns [C3, C3 + tail_cols) of every pooled row = (x, y, z, 0 ...) -- the
                                                  rows then ARE a group-all level's input rows [features | xyz | 0], no tail copy */,
                         int tail_cols, void* stream);
int pm_sa_bwd_packed_f32(const float* Y, int B, int P, int S, const int32_t* grow, const int32_t* rowmap, const float* relxyz,
                         const int32_t* tiles, const int32_t* totals, const float* W1, long ldw1, const float* b1, const float* b2,
                         const float* W3, const float* packed, int C1, int C2, int C3, const float* pooled, long ldp,
                         const int32_t* arg, const float* dpooled, long lddp, float* dW1, long lddw1, float* db1, float* dW2,
                         float* db2, float* dW3, float* db3, float* dY /* fp32 atomics: A/B only */,
                         float* dz1_rows /* (R, C1): the layer-1 gradient per packed row, plain stores (deterministic path) */,
                         int dw1_zero_end /* columns [3, dw1_zero_end) of dW1 are zeroed (a level without input features: padding) */,
                         const float* h2_saved, void* workspace, size_t workspace_bytes, void* stream);
/* The gradient of a level's per-source-point layer-1 rows (Y) WITHOUT floating-point atomics: a source point sits in several
 * groups, so dY[point] = sum of dz1 over its packed rows.  pm_sa_plan_inverse_i32 (coordinates only, once per plan) lists every
 * point's packed rows in ASCENDING order (CSR: inv_start (B*P + 1), inv_rows (R; capacity B*S*nsample)); pm_sa_dy_segsum_f32 adds
 * them in that order -- bit-reproducible run to run, no zero-fill of dY (points in no group get zeros).  P <= 7000. */
int pm_sa_plan_inverse_i32(const int32_t* rowmap, const int32_t* grow, int B, int P, int S, int32_t* inv_start, int32_t* inv_rows,
                           void* stream);
int pm_sa_dy_segsum_f32(const float* dz1, const int32_t* inv_start, const int32_t* inv_rows, long npoints, int C1, float* dY, long lddy,
                        void* stream);
/* The same sums CONSUMED in place (C1 = CF = 128): dY never reaches HBM.  Per tile of 64 source points the rows are summed into LDS
 * (same order, bit-identical to pm_sa_dy_segsum_f32), then  dfeat = dY * W1f  (the gradient the level below receives) and
 * dW1[:, 3 : 3 + CF] = dY^T * feat  run on fp32 MFMA; columns [3 + CF, dw1_cols) of dW1 are zeroed (pad columns never receive
 * data).  packedW = W1[:, 3 : 3 + CF] in operand order (pm_sa_dy_consume_pack_f32, after every update).  dfeat / dY may be NULL
 * (dY: an optional copy of the sums).  Replaces zero-fill + scatter + two Linear launches + slab reduction + column copy. */
int pm_sa_dy_consume_supported(int C1, int CF);
size_t pm_sa_dy_consume_packed_elems(int C1, int CF);
size_t pm_sa_dy_consume_workspace_bytes(int C1, int CF);
int pm_sa_dy_consume_pack_f32(const float* W1, long ldw1, int C1, int CF, float* packed, void* stream);
int pm_sa_dy_consume_f32(const float* dz1, const int32_t* inv_start, const int32_t* inv_rows, long npoints, int C1, int CF,
                         const float* feat, long ldf, const float* packedW, float* dfeat, long lddf, float* dW1, long lddw1,
                         int dw1_cols, float* dY, long lddy, void* workspace, size_t workspace_bytes, void* stream);

/* ---- PointNet++ group-all level: its LAST layer fused with the max over the cloud (csrc/sa_groupall.hip) ----------------
 * (BASELINE.json cfg 3's backbone; the reference ships no PointNet++ source -- `PointNet2` in algo_utils/network.py.)
 *   feat[b][c] = max_r tanh(H[b*R + r][:] . W[c][:] + bias[c]),  argmax[b][c] = the lowest row r attaining it
 * H: (B*R) x CK activations of the layer before (row-major, ld = CK), W: CO x CK.  Supported: CK = 256, CO = 512, R = 64
 * (pm_sa_groupall_supported; callers fall back to pm_linear_fwd_f32 + pm_maxpool_rows_f32 otherwise).
 * pack: W -> the MFMA operand order the forward streams (pm_sa_groupall_packed_elems floats); re-pack after every update.
 * bwd: dz[b][c] = dfeat[b][c] * (1 - feat[b][c]^2);  dH[b*R + r][:] = (sum over c with argmax[b][c] == r of dz[b][c] * W[c][:]) * (1 - H^2)
 * for EVERY row (rows without a winner: 0);  dW[c][:] = sum_b dz[b][c] * H[b*R + argmax[b][c]][:];  dbias[c] = sum_b dz[b][c].
 * Fixed summation orders (run-to-run identical). */
int pm_sa_groupall_supported(int CK, int CO, int R);
size_t pm_sa_groupall_packed_elems(int CK, int CO);
int pm_sa_groupall_pack_f32(const float* W, int CK, int CO, float* packed, void* stream);
int pm_sa_groupall_fwd_f32(const float* H, int B, int R, int CK, int CO, const float* bias, const float* packed, float* feat, long ldf,
                           int32_t* argmax, void* stream);
size_t pm_sa_groupall_bwd_workspace_bytes(int B, int CK, int CO);
int pm_sa_groupall_bwd_f32(const float* dfeat, long lddf, const float* feat, long ldf, const int32_t* argmax, const float* W,
                           const float* H, int B, int R, int CK, int CO, float* dH, float* dW, float* dbias, void* workspace,
                           size_t workspace_bytes, void* stream);

/* ---- rollout side ------------------------------------------------------------------------
 * algorithms/algo_utils/actor_critic.py:36-47 `random_act_cri` after the two network forwards: x = mu + sigma^2 * eps
 * (MultivariateNormal(mu, scale_tril = diag(sigma^2)).sample() with the caller's standard-normal eps (B, A), so the
 * torch RNG stream stays the reference's), logp (B) = its log_prob, actions (B, A) = tanh(x) * max_action (or x),
 * log_std_rows (B, A) = log_std repeated (may be NULL). */
int pm_gaussian_sample_f32(const float* mu, long ldmu, const float* log_std, const float* eps, int B, int A,
                           float max_action, int act_tanh, float* actions, float* logp, float* log_std_rows,
                           void* stream);
/* algorithms/algo_utils/RMS.py:10-18 `RunningMeanStd.update(x)` for one (N, D) batch: n_new = the reference's n AFTER
 * its increment; mean / S / std (D floats each) are updated in place.  RMS.py:40-45 `Normalization.__call__`:
 * out = (x - mean) / std. */
size_t pm_rms_update_workspace_bytes(int D);
int pm_rms_update_f32(const float* x, long ldx, int N, int D, int n_new, float* mean, float* S, float* std,
                      void* workspace, size_t workspace_bytes, void* stream);
int pm_rms_normalize_f32(const float* x, long ldx, int N, int D, const float* mean, const float* std, float* out,
                         long ldo, void* stream);
/* The same update in two halves for the data-parallel learner (new functionality: the reference has no distributed
 * path; each rank holds an env shard of the batch RMS.py:10-18 sees): mom[2*c] = sum_r x[r][c], mom[2*c+1] =
 * sum_r x[r][c]^2 (fp64; workspace as pm_rms_update_workspace_bytes(D)) -> the caller all-reduces `mom` ->
 * pm_rms_apply_moments_f32 applies RMS.py:10-18 for a batch of n_rows rows with those moments. */
int pm_rms_moments_f64(const float* x, long ldx, int N, int D, double* mom, void* workspace, size_t workspace_bytes,
                       void* stream);
int pm_rms_apply_moments_f32(const double* mom, long n_rows, int D, int n_new, float* mean, float* S, float* std,
                             void* stream);

/* ------------------------------------------------------------------ sparse-voxel U-Net building blocks
 * `network.name: SparseUNet` -- the "3D Sparse-UNet" backbone README.md:30 names; its code is NOT in the reference
 * snapshot (README.md:23): PARITY UNPINNED, restated in oracle/ref_cpu.py.  Input rows are the reference's 'depth_sparse'
 * observation (tasks/hand_base.py:335-336; utils/depth2tsdf.py:88-120): (x, y, z, f) with integer voxel coordinates.
 * Geometry: dense per-cloud index grids (R^3 int32; empty = 0x7fffffff), duplicates resolve to the lowest row.
 *   pm_voxel_grid0_f32: x (B, ldx) rows of P points x C floats -> grid (B * R^3) = lowest row b*P+i at that coordinate,
 *     coords (B*P, 4) int32 (b, x, y, z) clamped to [0, R), feat (B*P, 4) = (f, x/R, y/R, z/R) (may be NULL).
 *   pm_voxel_nbr27_i32: nbr (rows, 27) = row at coords + (dx,dy,dz), o = (dx+1)*9 + (dy+1)*3 + (dz+1), or -1.
 *   pm_voxel_down_count_i32 / _build_i32: the 2x strided level.  count: grid_c (B * Rc^3, Rc = ceil(Rf/2)) marks the parent
 *     cells, counts[b] = occupied cells of cloud b.  The caller prefix-sums counts into base (and sizes the level:
 *     rows_c = sum), then build numbers the cells in cell order from base[b] (grid_c <- global row; coords_c (rows_c, 4)),
 *     child (rows_c, 8) = fine row in slot (x&1)*4 + (y&1)*2 + (z&1) or -1, and per fine row parent / parent_canon (-1 for a
 *     duplicate coordinate: only the lowest row is its parent's child) / slot.
 * Arithmetic: a sparse convolution = pm_rows_gather_f32 + the Linear kernels on (rows x J*C).
 *   pm_rows_gather_f32: dst[r][j*C + c] = idx[r][j] >= 0 ? src[idx[r][j]][c] : 0      (C % 4 == 0, 16-byte rows)
 *   pm_rows_gather_bwd_f32: dsrc[r][c] (=, or += if accumulate) (sum_j dcols[t][blk*C + c]) * (y_tanh ? 1 - y_tanh[r][c]^2 : 1)
 *     with t = tidx[r][reverse ? J-1-j : j] (skipped when < 0) and blk = j (mode 0), tslot[r][j] (mode 1) or 0 (mode 2);
 *     self_col >= 0: rows with tidx[r][self_col] != r get 0 (duplicates are nobody's neighbour).  Fixed summation order. */
int pm_voxel_grid0_f32(const float* x, long ldx, int B, int P, int C, int R, int32_t* grid, int32_t* coords, float* feat,
                       void* stream);
int pm_voxel_nbr27_i32(const int32_t* coords, long rows, const int32_t* grid, int R, int32_t* nbr,
                       int ld /* row stride of nbr, >= 27; columns 27 .. ld-1 are written as -1 (padding taps) */, void* stream);
/* out[r][o] = nbr[r][26 - o] where row r is the canonical row of its cell (nbr[r][13] == r), else -1: the table the data
 * gradient of a 3^3 submanifold convolution gathers through (pm_sparse_conv_bwd_data_f32). */
int pm_voxel_mirror27_i32(const int32_t* nbr, long rows, int32_t* out, void* stream);
int pm_voxel_down_count_i32(const int32_t* coords_f, long rows_f, int B, int Rc, int32_t* grid_c, int32_t* counts, void* stream);
int pm_voxel_down_build_i32(const int32_t* coords_f, long rows_f, const int32_t* grid_f, int Rf, int B, int Rc,
                            const int32_t* base, int32_t* grid_c, long rows_c, int32_t* coords_c, int32_t* child,
                            int32_t* parent, int32_t* parent_canon, int32_t* slot, void* stream);
int pm_rows_gather_f32(const float* src, long lds, const int32_t* idx, long rows, int J, int C, float* dst, long ldd,
                       void* stream);
/* Fused forms: the gather happens inside the GEMM's LDS-DMA loader, the (rows x J*C) operand never exists in HBM
 * (J*C % 32 == 0; `zero`: >= C zeros, 16-byte aligned, read for absent neighbours).
 *   fwd:        Y[r][n] = act(sum_{j,c} src[idx[r][j]][c] * W[n][j*C + c] + b[n])
 *   bwd_weight: dW[n][j*C + c] = sum_r dY[r][n] * src[idx[r][j]][c],  db[n] = sum_r dY[r][n]   (db may be NULL) */
int pm_sparse_conv_fwd_f32(const float* src, long lds, const int32_t* idx, long rows, int J, int C, const float* W, long ldw,
                           const float* b, float* Y, long ldy, int N, int act, const float* zero, void* stream);
/* Data gradient of a submanifold convolution as a gathered GEMM (no column matrix in HBM):
 *   dX[s][ci] = (sum_{j,co} dY[idxT[s][j]][co] * Wt[ci][j*Cout + co]) * act'(H[s][ci])
 * idxT = the mirrored neighbour table (idxT[s][j] = idx[s][J-1-j], all -1 for rows that are nobody's neighbour),
 * Wt[ci][j*Cout + co] = W[co][j*Cin + ci] (row idxT[s][j] read s through ITS tap j); H = the activation the layer read (NULL: no factor).  (J*Cout) % 32 == 0. */
int pm_sparse_conv_bwd_data_f32(const float* dY, long lddy, const int32_t* idxT, long rows, int J, int Cout, const float* Wt,
                                long ldwt, const float* H, long ldh, float* dX, long lddx, int Cin, int act, const float* zero,
                                void* stream);
/* Data gradient by scatter, for convolutions whose patches do not overlap (stride == kernel size): every input row is
 * idx[r][j] for at most one (r, j), so  dX[idx[r][j]][c] = (sum_co dY[r][co] * W[co][j*C + c]) * act'(H[idx[r][j]][c])
 * is one GEMM whose epilogue stores each 16-byte piece at its destination row (idx < 0: dropped; rows nobody maps to keep
 * their contents).  W = the tap-major weight (Cout x J*C) of the forward.  accumulate: 0 = overwrite dsrc; 1 = add the (already multiplied) result to what dsrc holds; 2 = dsrc holds a RAW
 * contribution that is added to the gathered sum BEFORE the activation derivative ((skip + strided) * act'). */
int pm_sparse_conv_bwd_data_scatter_f32(const float* dY, long lddy, const float* W, long ldw, const int32_t* idx, long rows,
                                        int J, int C, int Cout, const float* H, float* dX, int act, void* stream);
size_t pm_sparse_conv_bwd_weight_workspace_bytes(long rows, int N, int J, int C);
int pm_sparse_conv_bwd_weight_f32(const float* dY, long lddy, const float* src, long lds, const int32_t* idx, long rows, int J,
                                  int C, float* dW, long lddw, float* db, int N, const float* zero, void* workspace,
                                  size_t workspace_bytes, void* stream);
int pm_rows_gather_bwd_f32(const float* dcols, long ldc, const int32_t* tidx, const int32_t* tslot, int mode, int reverse,
                           int self_col, long rows, int J, int C, const float* y_tanh, long ldy, int accumulate, float* dsrc,
                           long lds, void* stream);
/* The same with a COMPACTED column gradient: the table holds rows of a level, dcols only a subset of them; rowmap[row] = that
 * row's row in dcols, or -1 (its contribution is zero).  rowmap NULL: pm_rows_gather_bwd_f32. */
int pm_rows_gather_bwd_mapped_f32(const float* dcols, long ldc, const int32_t* tidx, const int32_t* tslot, int mode, int reverse,
                                  int self_col, long rows, int J, int C, const float* y_tanh, long ldy, int accumulate,
                                  float* dsrc, long lds, const int32_t* rowmap, void* stream);
/* ... and with a SPARSE raw skip contribution added before the activation derivative (accumulate == 2 without the dense
 * zero-filled tensor): row r additionally receives skip[skipmap[r]][:] when skipmap[r] >= 0; dsrc is overwritten. */
int pm_rows_gather_bwd_skip_f32(const float* dcols, long ldc, const int32_t* tidx, const int32_t* tslot, int mode, int reverse,
                                int self_col, long rows, int J, int C, const float* y_tanh, long ldy, float* dsrc, long lds,
                                const float* skip, long ldskip, const int32_t* skipmap, void* stream);

/* ------------------------------------------------------------------ row bookkeeping of the compact decoder backward
 * (partmanip_amd/algo_utils/network.py::SparseUNet._decoder_backward_compact: after the cloud-wide max-pool only the winners'
 * rows and their ancestors carry a gradient through the decoder -- at most S = c0 rows per cloud and level).  Integer work, fixed
 * order, no sort library.  A Level-2 binder reproduces the SparseUNet backward from these + the entry points above.
 *   pm_rows_uniq_i32: per cloud b the ids v[i] = map ? (src[b][i] == pad_in ? pad_out : map[src[b][i]]) : src[b][i] + b*row_base,
 *     i < S <= 64 -> u (B, S) = the cloud's DISTINCT ids ascending, padded with pad_out (a dummy row id >= every real id);
 *     um = u with -1 in the padding slots; rank (B, S) = slot of id i in u[b].
 *   pm_child_sum_f32: y[b*S + j][:] = sum_{i ascending, rank[b][i] == j} x[b*S + i][:]  (children summed into their parent's slot).
 *   pm_rowmap_scatter_i32: map[0..n) = -1; map[ids[k]] = k for ids[k] != pad  (row -> compact slot).
 *   pm_table_rows_i32: out[k][0..J) = sel[k] >= 0 ? table[sel[k]][0..J) : -1  (rows of an index table; -1 rows = absent taps).
 *   pm_voxel_vcat_table_i32: out[r] = {parent[r]*m .. parent[r]*m + m-1, r + m*rows_hi}: the (m + 1)-tap gather table of a
 *     virtual [unpool(coarse) | skip] operand whose halves share one buffer of skip-width rows.
 *   pm_exclusive_scan_i32: base[i] = counts[0] + .. + counts[i-1], total[0] = sum (pm_voxel_down_*'s `base`). */
int pm_rows_uniq_i32(const int32_t* src, long lds, int B, int S, long row_base, const int32_t* map, int pad_in, int pad_out,
                     int32_t* u, int32_t* um, int32_t* rank, void* stream);
int pm_child_sum_f32(const float* x, long ldx, const int32_t* rank, int B, int S, int C, float* y, long ldy, void* stream);
int pm_rowmap_scatter_i32(int32_t* map, long n, const int32_t* ids, long N, int pad, void* stream);
int pm_table_rows_i32(const int32_t* table, long ldt, int J, const int32_t* sel, long N, int32_t* out, void* stream);
int pm_voxel_vcat_table_i32(const int32_t* parent, long rows, int m, long rows_hi, int32_t* out, void* stream);
int pm_exclusive_scan_i32(const int32_t* counts, int n, int32_t* base, int32_t* total, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PARTMANIP_HIP_H */
