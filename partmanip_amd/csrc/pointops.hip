// K12 farthest point sampling, K13 ball query, K14 grouping (+ backward).
// Integer-index outputs: bit-exact against the CPU restatement used by the tests (same fp32 distance
// expression ((dx*dx)+dy*dy)+dz*dz with no FMA contraction, same tie-breaks).
// The reference tree holds no implementation of these: parity unpinned.
#include <type_traits>
#include "common.h"
#include <stdlib.h>

#define FPS_NT 1024
#define FPS_MAXD 4
#define FPS_RPT 8            // register-resident points per thread (P <= 8192)

__device__ __forceinline__ float dist2_rn(const float* a, const float* b, int D) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < FPS_MAXD; ++d)
        if (d < D) {
            const float t = sub_rn(a[d], b[d]);
            s = (d == 0) ? mul_rn(t, t) : add_rn(s, mul_rn(t, t));
        }
    return s;
}

// (value, index) arg-max with lowest index on ties, across the work-group.
__device__ __forceinline__ int block_argmax(float v, int i, float* sv, int* si) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(i, o, 64);
        if (ov > v || (ov == v && oi < i)) {
            v = ov;
            i = oi;
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) {
        sv[w] = v;
        si[w] = i;
    }
    __syncthreads();
    float bv = sv[0];
    int bi = si[0];
    for (int k = 1; k < FPS_NT / 64; ++k) {
        const float ov = sv[k];
        const int oi = si[k];
        if (ov > bv || (ov == bv && oi < bi)) {
            bv = ov;
            bi = oi;
        }
    }
    return bi;
}

// One work-group per cloud.  REG variant keeps the cloud and the running min-distance in
// registers (P <= 8192); the streaming variant re-reads points from L2/HBM and keeps the
// min-distance in a caller-supplied fp32 workspace (depth2pc-sized clouds: P = 442 368).
// PAD: pytorch3d's fixed-length semantics (K > P -> -1 indices); !PAD: keep sampling (all min-distances are 0 by
// then, so the lowest index wins) -- what the full, uncompacted cloud would have produced (see depth_compact_kernel).
template <bool REG, bool PAD>
__device__ __forceinline__ void fps_body(const float* __restrict__ pts, int P, int D, int K,
                                         int32_t* __restrict__ idx_b, float* __restrict__ mind_g, float* sv, int* si,
                                         float* sel) {
    const int tid = threadIdx.x;
    float px[FPS_RPT][FPS_MAXD];
    float mind[FPS_RPT];
    if (REG) {
#pragma unroll
        for (int r = 0; r < FPS_RPT; ++r) {
            const int p = tid + r * FPS_NT;
            mind[r] = INFINITY;
#pragma unroll
            for (int d = 0; d < FPS_MAXD; ++d) px[r][d] = (p < P && d < D) ? pts[(long)p * D + d] : 0.f;
        }
    } else {
        for (int p = tid; p < P; p += FPS_NT) mind_g[p] = INFINITY;
    }
    int cur = 0;
    for (int j = 0; j < K; ++j) {
        if (PAD && j >= P) {                // more samples than points: pytorch3d pads with -1
            if (tid == 0) idx_b[j] = -1;
            continue;
        }
        if (tid == 0) idx_b[j] = cur;
        if (tid < D) sel[tid] = pts[(long)cur * D + tid];
        __syncthreads();
        float s[FPS_MAXD];
#pragma unroll
        for (int d = 0; d < FPS_MAXD; ++d) s[d] = (d < D) ? sel[d] : 0.f;
        float bv = -1.0f;
        int bi = 0x7fffffff;
        if (REG) {
#pragma unroll
            for (int r = 0; r < FPS_RPT; ++r) {
                const int p = tid + r * FPS_NT;
                if (p < P) {
                    const float d2 = dist2_rn(px[r], s, D);
                    mind[r] = fminf(mind[r], d2);
                    if (mind[r] > bv) {     // strict: keeps the lowest index within the thread
                        bv = mind[r];
                        bi = p;
                    }
                }
            }
        } else {
            for (int p = tid; p < P; p += FPS_NT) {
                float q[FPS_MAXD];
#pragma unroll
                for (int d = 0; d < FPS_MAXD; ++d) q[d] = (d < D) ? pts[(long)p * D + d] : 0.f;
                const float m = fminf(mind_g[p], dist2_rn(q, s, D));
                mind_g[p] = m;
                if (m > bv) {
                    bv = m;
                    bi = p;
                }
            }
        }
        cur = block_argmax(bv, bi, sv, si);
    }
}

template <bool REG>
__global__ __launch_bounds__(FPS_NT) void fps_kernel(const float* __restrict__ xyz, int P, int D, int K,
                                                      int32_t* __restrict__ idx_out, float* __restrict__ mind_ws) {
    __shared__ float sv[FPS_NT / 64];
    __shared__ int si[FPS_NT / 64];
    __shared__ float sel[FPS_MAXD];
    const int b = blockIdx.x;
    fps_body<REG, true>(xyz + (long)b * P * D, P, D, K, idx_out + (long)b * K, REG ? nullptr : mind_ws + (long)b * P, sv,
                        si, sel);
}

// Variable-length clouds (rows of a (B, ld, D) buffer, the first lengths[b] of each are points): the path is
// chosen per cloud on the device (no host sync on the lengths) -- registers if it fits, streaming otherwise.
__global__ __launch_bounds__(FPS_NT) void fps_varlen_kernel(const float* __restrict__ xyz, int ld, int D, int K,
                                                             const int32_t* __restrict__ lengths, int pad,
                                                             int32_t* __restrict__ idx_out,
                                                             float* __restrict__ mind_ws, int mode,
                                                             const unsigned long long* __restrict__ gave_up) {
    __shared__ float sv[FPS_NT / 64];
    __shared__ int si[FPS_NT / 64];
    __shared__ float sel[FPS_MAXD];
    const int b = blockIdx.x, n = lengths[b];
    // mode 0: every cloud.  mode 1: the clouds that fit one work-group's registers (fps_multi_kernel samples the big ones of
    // this batch).  mode 2: the big clouds, and only if the multi-work-group launch in front of this one GAVE UP (its error
    // word is set): the batch degrades to the one-work-group streaming sampler instead of returning invalid indices -- no host
    // round trip; in the normal case this launch returns at once.
    if (mode == 1 && n > FPS_NT * FPS_RPT) return;
    if (mode == 2 && (n <= FPS_NT * FPS_RPT || __hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull)) return;
    const float* pts = xyz + (long)b * ld * D;
    int32_t* idx_b = idx_out + (long)b * K;
    if (n <= 0) {                                        // empty cloud: pytorch3d yields -1 everywhere
        for (int j = threadIdx.x; j < K; j += FPS_NT) idx_b[j] = -1;
    } else if (n <= FPS_NT * FPS_RPT) {
        if (pad) fps_body<true, true>(pts, n, D, K, idx_b, nullptr, sv, si, sel);
        else fps_body<true, false>(pts, n, D, K, idx_b, nullptr, sv, si, sel);
    } else {
        if (pad) fps_body<false, true>(pts, n, D, K, idx_b, mind_ws + (long)b * ld, sv, si, sel);
        else fps_body<false, false>(pts, n, D, K, idx_b, mind_ws + (long)b * ld, sv, si, sel);
    }
}

// Wave-per-cloud variant for xyz clouds of up to 2048 points (the learner's sizes: 1024 -> 256 -> 64): the cloud
// and its running min-distances live in ONE wave's registers (PPL points per lane, interleaved so the loads
// coalesce), each round is PPL distance updates + a 6-step (value, index, x, y, z) butterfly arg-max over the 64
// lanes -- no LDS, no barrier, no reload of the selected point.  Four clouds per 256-thread work-group.
// Same fp32 expression and tie-breaks as fps_kernel (bit-identical indices).
template <int PPL>
__global__ __launch_bounds__(256) void fps_wave_kernel(const float* __restrict__ xyz, int B, int P, int K,
                                                        int32_t* __restrict__ idx_out) {
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* pts = xyz + (long)b * P * 3;
    float px[PPL], py[PPL], pz[PPL], mind[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        const int p = lane + 64 * j;
        const bool ok = p < P;
        px[j] = ok ? pts[p * 3] : 0.f;
        py[j] = ok ? pts[p * 3 + 1] : 0.f;
        pz[j] = ok ? pts[p * 3 + 2] : 0.f;
        mind[j] = INFINITY;
    }
    int cur = 0;
    float cx = __shfl(px[0], 0, 64), cy = __shfl(py[0], 0, 64), cz = __shfl(pz[0], 0, 64);
    for (int k = 0; k < K; ++k) {
        if (k >= P) {                           // more samples than points: pytorch3d pads with -1
            if (lane == 0) idx_out[(long)b * K + k] = -1;
            continue;
        }
        if (lane == 0) idx_out[(long)b * K + k] = cur;
        float bv = -1.0f, bx = 0.f, by = 0.f, bz = 0.f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const int p = lane + 64 * j;
            if (p < P) {
                const float tx = sub_rn(px[j], cx), ty = sub_rn(py[j], cy), tz = sub_rn(pz[j], cz);
                const float d2 = add_rn(add_rn(mul_rn(tx, tx), mul_rn(ty, ty)), mul_rn(tz, tz));
                const float m = fminf(mind[j], d2);
                mind[j] = m;
                if (m > bv) {                   // strict: keeps the lowest index within the lane
                    bv = m; bi = p; bx = px[j]; by = py[j]; bz = pz[j];
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            const float ox = __shfl_xor(bx, o, 64), oy = __shfl_xor(by, o, 64), oz = __shfl_xor(bz, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov; bi = oi; bx = ox; by = oy; bz = oz;
            }
        }
        cur = bi; cx = bx; cy = by; cz = bz;
    }
}

extern "C" size_t pm_fps_workspace_bytes(int B, int P) {
    return (P > FPS_NT * FPS_RPT) ? (size_t)B * P * sizeof(float) : 0;
}

extern "C" int pm_fps_f32(const float* xyz, int B, int P, int D, int K, int32_t* idx_out, void* workspace,
                          size_t workspace_bytes, void* stream) {
    PM_REQUIRE(xyz && idx_out && B > 0 && P > 0 && D >= 1 && D <= FPS_MAXD && K > 0);
    if (D == 3 && P <= 2048) {
#define FPS_WAVE_LAUNCH(PPL)                                                                                  \
    hipLaunchKernelGGL(fps_wave_kernel<PPL>, dim3((B + 3) / 4), dim3(256), 0, pm_stream(stream), xyz, B, P, K, idx_out)
        if (P <= 64) FPS_WAVE_LAUNCH(1);
        else if (P <= 128) FPS_WAVE_LAUNCH(2);
        else if (P <= 256) FPS_WAVE_LAUNCH(4);
        else if (P <= 512) FPS_WAVE_LAUNCH(8);
        else if (P <= 1024) FPS_WAVE_LAUNCH(16);
        else FPS_WAVE_LAUNCH(32);
#undef FPS_WAVE_LAUNCH
    } else if (P <= FPS_NT * FPS_RPT) {
        hipLaunchKernelGGL(fps_kernel<true>, dim3(B), dim3(FPS_NT), 0, pm_stream(stream), xyz, P, D, K, idx_out,
                           (float*)nullptr);
    } else {
        if (!workspace || workspace_bytes < (size_t)B * P * sizeof(float)) return PM_EWORKSPACE;
        hipLaunchKernelGGL(fps_kernel<false>, dim3(B), dim3(FPS_NT), 0, pm_stream(stream), xyz, P, D, K, idx_out,
                           (float*)workspace);
    }
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ---------------------------------------------------------------------------------- K13
// One wave per centre: 64 points tested per step, __ballot gives the in-radius mask and
// the popcount of the lower lanes is each hit's output slot (ascending index order).
__global__ __launch_bounds__(256) void ball_query_kernel(const float* __restrict__ xyz,
                                                          const float* __restrict__ centers, int B, int P, int S,
                                                          float r2, int nsample, int32_t* __restrict__ idx_out) {
    const int lane = threadIdx.x & 63;
    const long q = (long)blockIdx.x * 4 + (threadIdx.x >> 6);      // centre id in [0, B*S)
    if (q >= (long)B * S) return;
    const int b = (int)(q / S);
    const float* pts = xyz + (long)b * P * 3;
    const float c[3] = {centers[q * 3], centers[q * 3 + 1], centers[q * 3 + 2]};
    int32_t* out = idx_out + q * nsample;
    int cnt = 0, first = 0;
    for (int base = 0; base < P && cnt < nsample; base += 64) {
        const int p = base + lane;
        bool hit = false;
        if (p < P) {
            const float a[3] = {pts[(long)p * 3], pts[(long)p * 3 + 1], pts[(long)p * 3 + 2]};
            hit = dist2_rn(a, c, 3) < r2;
        }
        const unsigned long long mask = __ballot(hit);
        if (mask) {
            if (cnt == 0) first = base + __builtin_ctzll(mask);
            const int slot = cnt + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
            if (hit && slot < nsample) out[slot] = p;
            cnt += __builtin_popcountll(mask);
        }
    }
    if (cnt > nsample) cnt = nsample;
    for (int j = cnt + lane; j < nsample; j += 64) out[j] = first;   // pad with the first hit (0 if none)
}

extern "C" int pm_ball_query_f32(const float* xyz, const float* centers, int B, int P, int S, float radius,
                                 int nsample, int32_t* idx_out, void* stream) {
    PM_REQUIRE(xyz && centers && idx_out && B > 0 && P > 0 && S > 0 && nsample > 0 && radius > 0.f);
    const long nq = (long)B * S;
    const float r2 = radius * radius;
    hipLaunchKernelGGL(ball_query_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, pm_stream(stream), xyz, centers,
                       B, P, S, r2, nsample, idx_out);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ---------------------------------------------------------------------------------- K14
// out[b,s,j,:] = feat[b, idx[b,s,j], :]
__global__ __launch_bounds__(256) void group_points_kernel(const float* __restrict__ feat,
                                                            const int32_t* __restrict__ idx, int B, int P, int C,
                                                            long n_rows, int rows_per_b, float* __restrict__ out) {
    const long total = n_rows * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long row = e / C;
        const int c = (int)(e - row * C);
        const int b = (int)(row / rows_per_b);
        out[e] = feat[((long)b * P + idx[row]) * C + c];
    }
}

// ---- the gradient of a row gather WITHOUT floating-point atomics ----------------------------------------------------------
// dfeat[b, p, :] = sum over the cloud's rows r with idx[r] == p, in ASCENDING r, of dout[r, col0 : col0 + C].  One work-group per
// cloud builds the inverse of its index table in LDS -- integer counts (LDS atomics: order-free), prefix, cursor fill, insertion sort
// of every point's short list -- and then every (point, channel) is ONE sequential chain: bit-reproducible, every element of dfeat
// written (no zero-fill needed).  Used when 2 P + rows-per-cloud ints fit 64 KB of LDS; beyond that the atomic kernels below run.
#define SCAT_LDS_INTS (16000)
__global__ __launch_bounds__(256) void scatter_rows_det_kernel(const float* __restrict__ dout, long ldo, int col0,
                                                                const int32_t* __restrict__ idx, int P, int C, int rows_per_b,
                                                                float* __restrict__ dfeat) {
    extern __shared__ int sh[];                          // cnt[P + 1] | cur[P] | part[256] | list[rows_per_b]
    int* cnt = sh;
    int* cur = sh + P + 1;
    int* part = cur + P;
    int* list = part + 256;
    const long b = blockIdx.x;
    const int32_t* ib = idx + b * rows_per_b;
    for (int p = threadIdx.x; p <= P; p += 256) cnt[p] = 0;
    __syncthreads();
    for (int r = threadIdx.x; r < rows_per_b; r += 256) atomicAdd(&cnt[ib[r]], 1);
    __syncthreads();
    const int per = (P + 255) / 256, lo = threadIdx.x * per, hi = lo + per < P ? lo + per : P;
    int sum = 0;
    for (int p = lo; p < hi; ++p) sum += cnt[p];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;
    for (int p = lo; p < hi; ++p) {
        cur[p] = run;
        run += cnt[p];
    }
    __syncthreads();
    for (int r = threadIdx.x; r < rows_per_b; r += 256) list[atomicAdd(&cur[ib[r]], 1)] = r;
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += 256) {           // cur[p] is now the END of p's list
        const int e = cur[p], s0 = e - cnt[p];
        for (int i = s0 + 1; i < e; ++i) {
            const int v = list[i];
            int j = i - 1;
            while (j >= s0 && list[j] > v) {
                list[j + 1] = list[j];
                --j;
            }
            list[j + 1] = v;
        }
    }
    __syncthreads();
    const float* db = dout + b * rows_per_b * ldo + col0;
    float* fb = dfeat + b * (long)P * C;
    for (long e = threadIdx.x; e < (long)P * C; e += 256) {
        const int p = (int)(e / C), c = (int)(e - (long)p * C);
        const int en = cur[p], s0 = en - cnt[p];
        float acc = 0.f;
        for (int j = s0; j < en; ++j) acc += db[(long)list[j] * ldo + c];
        fb[e] = acc;
    }
}
static bool scatter_rows_det(const float* dout, long ldo, int col0, const int32_t* idx, int B, int P, int C, int rows_per_b,
                             float* dfeat, void* stream) {
    const long ints = 2L * P + 1 + 256 + rows_per_b;
    if (ints > SCAT_LDS_INTS) return false;
    hipLaunchKernelGGL(scatter_rows_det_kernel, dim3(B), dim3(256), (size_t)ints * sizeof(int), pm_stream(stream), dout, ldo, col0, idx, P,
                       C, rows_per_b, dfeat);
    return true;
}

__global__ __launch_bounds__(256) void group_points_bwd_kernel(const float* __restrict__ dout,
                                                                const int32_t* __restrict__ idx, int B, int P, int C,
                                                                long n_rows, int rows_per_b,
                                                                float* __restrict__ dfeat) {
    const long total = n_rows * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long row = e / C;
        const int c = (int)(e - row * C);
        const int b = (int)(row / rows_per_b);
        atomicAdd(&dfeat[((long)b * P + idx[row]) * C + c], dout[e]);
    }
}

extern "C" int pm_group_points_f32(const float* feat, const int32_t* idx, int B, int P, int C, int S, int nsample,
                                   float* out, void* stream) {
    PM_REQUIRE(feat && idx && out && B > 0 && P > 0 && C > 0 && S > 0 && nsample > 0);
    const long n_rows = (long)B * S * nsample;
    long nb = (n_rows * C + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(group_points_kernel, dim3((unsigned)nb), dim3(256), 0, pm_stream(stream), feat, idx, B, P, C,
                       n_rows, S * nsample, out);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// Deterministic (fixed ascending-row order per source point, scatter_rows_det_kernel) when 2 P + S nsample <= ~15.7 k; beyond that:
// fp32 atomics into a dfeat the caller zero-filled (summation order not fixed).  Callers zero-fill either way.
extern "C" int pm_group_points_bwd_f32(const float* dout, const int32_t* idx, int B, int P, int C, int S,
                                       int nsample, float* dfeat, void* stream) {
    PM_REQUIRE(dout && idx && dfeat && B > 0 && P > 0 && C > 0 && S > 0 && nsample > 0);
    if (scatter_rows_det(dout, C, 0, idx, B, P, C, S * nsample, dfeat, stream)) {
        PM_CHECK_LAUNCH();
        return PM_OK;
    }
    const long n_rows = (long)B * S * nsample;
    long nb = (n_rows * C + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(group_points_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, pm_stream(stream), dout, idx, B, P, C,
                       n_rows, S * nsample, dfeat);
    PM_CHECK_LAUNCH();
    return PM_OK;
}


// ---------------------------------------------------------------------------------- K15 pieces
// PointNet++ set-abstraction glue (absent from the reference, north-star mandated):
//   group_concat: out[b,s,j,:] = [ xyz[b,idx]-center[b,s] (3) | feat[b,idx,:] (Cf) | 0-pad ]  rows of ldo floats
//   maxpool_rows: out[g,c] = max_j x[g,j,c], arg = lowest j attaining it  (x: (G, ns, C))
__global__ __launch_bounds__(256) void group_concat_kernel(const float* __restrict__ xyz, const float* __restrict__ feat,
                                                            const float* __restrict__ centers,
                                                            const int32_t* __restrict__ idx, int P, int Cf, int S,
                                                            int ns, long n_rows, int ldo, float* __restrict__ out) {
    const long total = n_rows * ldo;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long row = e / ldo;
        const int c = (int)(e - row * ldo);
        const long q = row / ns;                         // (b, s) flat
        const int b = (int)(q / S);
        const long src = (long)b * P + idx[row];
        float v = 0.f;
        if (c < 3) v = sub_rn(xyz[src * 3 + c], centers[q * 3 + c]);
        else if (c < 3 + Cf) v = feat[src * Cf + (c - 3)];
        out[e] = v;
    }
}

extern "C" int pm_group_concat_f32(const float* xyz, const float* feat, const float* centers, const int32_t* idx,
                                   int B, int P, int Cf, int S, int nsample, int ldo, float* out, void* stream) {
    PM_REQUIRE(xyz && centers && idx && out && B > 0 && P > 0 && Cf >= 0 && S > 0 && nsample > 0 && ldo >= 3 + Cf);
    PM_REQUIRE(Cf == 0 || feat);
    const long n_rows = (long)B * S * nsample;
    long nb = (n_rows * ldo + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(group_concat_kernel, dim3((unsigned)nb), dim3(256), 0, pm_stream(stream), xyz, feat, centers, idx,
                       P, Cf, S, nsample, n_rows, ldo, out);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// d feat[b, idx, :] += d out[row, 3:3+Cf]   (dfeat zero-filled by the caller; fp32 atomics -- only for sizes beyond scatter_rows_det)
__global__ __launch_bounds__(256) void group_concat_bwd_kernel(const float* __restrict__ dout,
                                                                const int32_t* __restrict__ idx, int P, int Cf, int S,
                                                                int ns, long n_rows, int ldo, float* __restrict__ dfeat) {
    const long total = n_rows * Cf;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long row = e / Cf;
        const int c = (int)(e - row * Cf);
        const int b = (int)(row / ((long)S * ns));
        atomicAdd(&dfeat[((long)b * P + idx[row]) * Cf + c], dout[row * ldo + 3 + c]);
    }
}

extern "C" int pm_group_concat_bwd_f32(const float* dout, const int32_t* idx, int B, int P, int Cf, int S, int nsample,
                                       int ldo, float* dfeat, void* stream) {
    PM_REQUIRE(dout && idx && dfeat && B > 0 && P > 0 && Cf > 0 && S > 0 && nsample > 0 && ldo >= 3 + Cf);
    if (scatter_rows_det(dout, ldo, 3, idx, B, P, Cf, S * nsample, dfeat, stream)) {     // (as pm_group_points_bwd_f32)
        PM_CHECK_LAUNCH();
        return PM_OK;
    }
    const long n_rows = (long)B * S * nsample;
    long nb = (n_rows * Cf + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(group_concat_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, pm_stream(stream), dout, idx, P, Cf, S,
                       nsample, n_rows, ldo, dfeat);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

__global__ __launch_bounds__(256) void maxpool_rows_kernel(const float* __restrict__ x, long G, int ns, int C,
                                                            float* __restrict__ out, long ldo, int32_t* __restrict__ arg) {
    const long total = G * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long g = e / C;
        const int c = (int)(e - g * C);
        const float* p = x + g * ns * C + c;
        float m = p[0];
        int am = 0;
        for (int j = 1; j < ns; ++j) {
            const float v = p[(long)j * C];
            if (v > m) {
                m = v;
                am = j;
            }
        }
        out[g * ldo + c] = m;
        arg[e] = am;
    }
}

// Long groups (a whole cloud's rows: the SparseUNet's feature pooling, ns = 4096): one work-group per group, thread
// (chunk = t / C, c = t % C) scans rows chunk, chunk + 256/C, ... -- a wave reads whole consecutive rows -- and the 256/C
// candidates of a channel meet in LDS.  Same result as the serial scan: the greatest value, the lowest row among equals.
__global__ __launch_bounds__(256) void maxpool_rows_long_kernel(const float* __restrict__ x, int ns, int C,
                                                                 float* __restrict__ out, long ldo, int32_t* __restrict__ arg) {
    __shared__ float sm[256];
    __shared__ int sa[256];
    const long g = blockIdx.x;
    const int t = threadIdx.x, tpc = 256 / C, chunk = t / C, c = t - chunk * C;
    const float* p = x + g * ns * C + c;
    float m = -INFINITY;
    int am = 0x7fffffff;
    int j = chunk;
    for (; j + 3 * tpc < ns; j += 4 * tpc) {
        const float v0 = p[(long)j * C], v1 = p[(long)(j + tpc) * C], v2 = p[(long)(j + 2 * tpc) * C], v3 = p[(long)(j + 3 * tpc) * C];
        if (v0 > m || am == 0x7fffffff) { m = v0; am = j; }
        if (v1 > m) { m = v1; am = j + tpc; }
        if (v2 > m) { m = v2; am = j + 2 * tpc; }
        if (v3 > m) { m = v3; am = j + 3 * tpc; }
    }
    for (; j < ns; j += tpc) {
        const float v = p[(long)j * C];
        if (v > m || am == 0x7fffffff) { m = v; am = j; }
    }
    sm[t] = m;
    sa[t] = am;
    __syncthreads();
    if (t < C) {
        for (int k = 1; k < tpc; ++k) {
            const float v = sm[k * C + t];
            const int a = sa[k * C + t];
            if (a != 0x7fffffff && (am == 0x7fffffff || v > m || (v == m && a < am))) { m = v; am = a; }
        }
        out[g * ldo + t] = m;
        arg[g * C + t] = am;
    }
}

extern "C" int pm_maxpool_rows_f32(const float* x, long G, int nsample, int C, float* out, long ldo, int32_t* arg,
                                   void* stream) {
    PM_REQUIRE(x && out && arg && G > 0 && nsample > 0 && C > 0 && ldo >= C);
    if (nsample >= 256 && C <= 256 && 256 % C == 0 && G < 0x7fffffffL) {
        hipLaunchKernelGGL(maxpool_rows_long_kernel, dim3((unsigned)G), dim3(256), 0, pm_stream(stream), x, nsample, C, out, ldo, arg);
        PM_CHECK_LAUNCH();
        return PM_OK;
    }
    long nb = (G * C + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(maxpool_rows_kernel, dim3((unsigned)nb), dim3(256), 0, pm_stream(stream), x, G, nsample, C, out,
                       ldo, arg);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// dx[g,j,c] = dout[g,c] * act'(y[g,j,c]) if j == arg[g,c] else 0  (writes every element of dx);
// y_tanh (nullable) = the pooled tensor itself when it is a tanh output: act' = 1 - y^2.
__global__ __launch_bounds__(256) void maxpool_rows_bwd_kernel(const float* __restrict__ dout, long lddo,
                                                                const int32_t* __restrict__ arg, long G, int ns, int C,
                                                                const float* __restrict__ y_tanh,
                                                                float* __restrict__ dx) {
    const long total = G * ns * C;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c = (int)(e % C);
        const long gj = e / C;
        const int j = (int)(gj % ns);
        const long g = gj / ns;
        float v = 0.f;
        if (arg[g * C + c] == j) {
            v = dout[g * lddo + c];
            if (y_tanh) {
                const float y = y_tanh[e];
                v *= (1.0f - y * y);
            }
        }
        dx[e] = v;
    }
}

extern "C" int pm_maxpool_rows_bwd_f32(const float* dout, long lddo, const int32_t* arg, long G, int nsample, int C,
                                       const float* y_tanh, float* dx, void* stream) {
    PM_REQUIRE(dout && arg && dx && G > 0 && nsample > 0 && C > 0 && lddo >= C);
    long nb = (G * nsample * C + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(maxpool_rows_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, pm_stream(stream), dout, lddo, arg, G,
                       nsample, C, y_tanh, dx);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ---------------------------------------------------------------------------------------------- column-block copies
// dst[r][d_b .. d_b + (e_b - s_b)) = src[r][s_b .. e_b) for up to two blocks b; every other column of dst below `dst_cols` is
// written as zero when zero_other is set.  ONE launch for the glue the PointNet++ plug-in needs around the GEMMs -- a layer's
// weight with its columns in the operand's order and padded to the K-step, its gradient back in the parameter's order, the
// aligned copy of W1's feature columns, [xyz | 0] behind the pooled features of the group-all rows -- each of which was two to
// four strided copy_ / zero_ launches of the tensor library (round 4: 6 % of the PointNet++ step's kernel time).
// (dst / src carry no __restrict__: the zero-only form -- no block, src == NULL -- and disjoint column ranges of ONE buffer are legal)
__global__ __launch_bounds__(256) void col_blocks_kernel(float* dst, long ldd, const float* src, long lds,
                                                          long rows, int dst_cols, int col0, int s0, int e0, int d0, int s1, int e1,
                                                          int d1, int zero_other) {
    const int w = dst_cols - col0;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < rows * w; e += (long)gridDim.x * 256) {
        const long r = e / w;
        const int c = col0 + (int)(e - r * w);
        if (c >= d0 && c < d0 + (e0 - s0)) dst[r * ldd + c] = src[r * lds + s0 + (c - d0)];
        else if (c >= d1 && c < d1 + (e1 - s1)) dst[r * ldd + c] = src[r * lds + s1 + (c - d1)];
        else if (zero_other) dst[r * ldd + c] = 0.f;
    }
}

extern "C" int pm_col_blocks_f32(float* dst, long ldd, const float* src, long lds, long rows, int dst_cols, int col0, int s0, int e0,
                                 int d0, int s1, int e1, int d1, int zero_other, void* stream) {
    const int n0 = e0 - s0, n1 = e1 - s1;
    PM_REQUIRE(dst && rows > 0 && dst_cols > 0 && col0 >= 0 && col0 < dst_cols && ldd >= dst_cols);
    PM_REQUIRE(s0 >= 0 && n0 >= 0 && s1 >= 0 && n1 >= 0);
    PM_REQUIRE(src || (n0 == 0 && n1 == 0));                   // src == NULL: zero-only (no block to copy)
    // a block must land inside the written range [col0, dst_cols) -- one that starts below col0 would silently not be copied --
    // read inside a source row, and the two blocks must not overlap (the first would win)
    PM_REQUIRE(n0 == 0 || (d0 >= col0 && d0 + n0 <= dst_cols && e0 <= lds));
    PM_REQUIRE(n1 == 0 || (d1 >= col0 && d1 + n1 <= dst_cols && e1 <= lds));
    PM_REQUIRE(n0 == 0 || n1 == 0 || d0 + n0 <= d1 || d1 + n1 <= d0);
    // the same buffer as source and destination: only with column ranges that do not touch what is written
    if (src && (const float*)dst == src && (n0 || n1)) {
        PM_REQUIRE(lds == ldd);
        PM_REQUIRE(n0 == 0 || e0 <= col0);
        PM_REQUIRE(n1 == 0 || e1 <= col0);
    }
    const long n = rows * (dst_cols - col0);
    long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(col_blocks_kernel, dim3((unsigned)blocks), dim3(256), 0, pm_stream(stream), dst, ldd, src, lds, rows, dst_cols, col0,
                       s0, e0, d0, s1, e1, d1, zero_other);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ---- every weight-derived operand copy of a network in ONE launch ---------------------------------------------------------
// The MFMA kernels stream their weights from operand-order copies (pm_sa_pack_weights_f32, pm_sa_groupall_pack_f32,
// pm_sa_dy_consume_pack_f32), the Linear kernels want 16-byte-aligned / K-step-padded copies of misaligned column blocks
// (pm_col_blocks_f32): six tiny launches per PointNet++ forward + backward, each of which queues behind the other network's
// persistent kernels.  All of them are "dst[q] = parameter[t(q)] or 0" with t fixed by the layouts, so the host records t ONCE
// (it runs the pack entry points on index-valued weights) and every later refresh is this one gather.
__global__ __launch_bounds__(256) void gather_copy_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                           const int32_t* __restrict__ table, long n) {
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < n; q += (long)gridDim.x * 256) {
        const int t = table[q];
        dst[q] = t >= 0 ? src[t] : 0.f;
    }
}

extern "C" int pm_gather_copy_f32(float* dst, const float* src, const int32_t* table, long n, void* stream) {
    PM_REQUIRE(dst && src && table && n > 0 && dst != src);
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gather_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, pm_stream(stream), dst, src, table, n);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

extern "C" int pm_version(void) { return PM_ABI_VERSION; }

// ---- camera-sized clouds on SEVERAL work-groups per cloud ----------------------------------------------------------------
// The streaming rounds of fps_body re-read every surviving point (12 B) and its running min-distance (4 B + 4 B back) from
// L2 / Infinity Cache in each of the K rounds -- 176 GB per depth2pc call at 64 envs x 134 k points, 2.8 TB/s from the 64
// CUs that one work-group per env occupies.  Here G work-groups share a cloud (G * B <= the CU count, so that
// all of them are resident) and each keeps ITS chunk of the cloud ON CHIP for all K rounds: FM_RPT points per thread in
// registers (x, y, z, min-distance: 4 x FM_RPT VGPRs), the next FM_LDS points in LDS, only what is left streams (four
// points per trip).  A round is the distance update of the chunk, a work-group arg-max and ONE hand-off between the G
// work-groups of the cloud: each publishes its candidate -- value, index, coordinates -- as five 8-byte {round, word} granules
// (agent-scope atomic stores: write-through, the data IS the flag -- MI355X_MICROARCH.md form R2), one wave of every work-group
// sweeps the granules of its cloud until all carry this round's tag, and every work-group takes the same lowest-index arg-max.
// Same fp32 distance expression and tie-breaks as fps_body: bit-identical indices.
//
// Shape (per depth2pc call, 64 envs x ~134 k points, K = 1024; profiles/round4_h_fps_multi_ab.txt):
//   1024 threads x 16 points + 8 192 in LDS, (value, index) carried through the sweep      8.9 ms   (rounds 2-3)
//   the same with packed-fp32 distances                                                    9.0      (v_pk_*_f32 is NOT double rate here)
//   512 x 48 + 9 728: no streamed pass                                                     7.7
//   + the winner's coordinates inside the hand-off (no load, one barrier less)             7.7      (no gain alone)
//   + value-only sweep, index recovered afterwards (fm_min / fm_max3 below)                5.7
//   512 x 50 + 10 176 (253 VGPRs, no spill; 35 776 points on chip)                         5.6
#define FM_NT 512
#define FM_RPT 50
#define FM_LDS 10176
#ifndef FM_STR_U
#define FM_STR_U 4            // streamed points per thread and trip (A/B: 8 halves the dependent round trips of the streamed pass)
#endif
#define FM_MAXG 8
#define FM_GRAN 8             // 64-bit words per work-group and set: value, index, x, y, z of its candidate (+3 pad: one 64-byte line)
#define FM_SLOT_WORDS (2 * 256 * FM_GRAN)     // two sets x <= 256 work-groups; the error word sits right behind them
#define FM_SPIN_LIMIT (1u << 20)   // polls a work-group may spend waiting for partners over the WHOLE launch (~1 s); normal: a few thousand
typedef unsigned long long fm_u64;
// NT threads, RPT points per thread in registers, LP points in LDS (16 B each, structure of arrays).  (1024, 16, 8192): four waves
// per SIMD, 24 576 points on chip; (512, 44 | 48, 9728): two waves per SIMD with 256 registers each, 32 256 | 34 304 points on chip
// -- a quarter of a 134 k-point cloud then needs (almost) no streamed pass (`PM_FM_CFG`; bench.py --workload depth2pc).
//
// The sweep carries VALUES only.  Per pair of points: 8 packed-fp32 instructions for the two squared distances (v_pk_add / v_pk_mul:
// IEEE per element, no contraction under -ffp-contract=off, so the sums keep the reference's op-by-op rounding), two v_min, one
// v_max3 -- 5.5 instructions per point against 12 with an (value, index) pair carried through every point (compare, two selects,
// a range predicate, the canonicalising v_max the compiler puts in front of fminf).  The index is recovered AFTERWARDS, by the few
// threads whose own maximum equals the work-group's: their first point (ascending index) holding that value, an LDS atomic min
// over those.  Lowest index among equal values, as fps_body: bit-identical indices.
typedef float fm_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ fm_f32x2 fm_dist2_pk(fm_f32x2 x, fm_f32x2 y, fm_f32x2 z, const float* s) {
    const fm_f32x2 tx = x - (fm_f32x2){s[0], s[0]}, ty = y - (fm_f32x2){s[1], s[1]}, tz = z - (fm_f32x2){s[2], s[2]};
    fm_f32x2 d = tx * tx;
    d = d + ty * ty;
    d = d + tz * tz;
    return d;
}
// min / max of values that are never NaN on the right (a NaN distance leaves the running minimum alone, as fminf does)
__device__ __forceinline__ float fm_min(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float fm_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// maximum over the wave, in every lane: four DPP steps inside each row of 16 (quad swaps, half-row and row mirrors -- VALU
// operand modifiers, no LDS round trip as in __shfl_xor), then the four row results through scalar registers
__device__ __forceinline__ float fm_wave_max(float v) {
#define FM_DPP_MAX(CTRL) v = fm_max3(v, v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false)))
    FM_DPP_MAX(0xB1);     // quad_perm [1, 0, 3, 2]
    FM_DPP_MAX(0x4E);     // quad_perm [2, 3, 0, 1]
    FM_DPP_MAX(0x141);    // row_half_mirror
    FM_DPP_MAX(0x140);    // row_mirror
#undef FM_DPP_MAX
    const int vi = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(vi, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(vi, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(vi, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(vi, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// loops over the register-resident points are unrolled by TEMPLATE (constant indices from the first optimisation pass on): with a
// `#pragma unroll` loop the arrays are still memory when the branch tree of fm_pick is simplified, the tree's reads get folded into
// one dynamically indexed read, and the arrays then stay in scratch for the sweep as well (24 scratch loads per round, measured)
template <int I, int N, int STEP, class F>
__device__ __forceinline__ void fm_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        fm_static_for<I + STEP, N, STEP>(f);
    }
}
// px[r], py[r], pz[r] for a wave-uniform r: a binary search of scalar branches over statically indexed registers
template <int LO, int HI, int RPT>
__device__ __forceinline__ void fm_pick(const float (&px)[RPT], const float (&py)[RPT], const float (&pz)[RPT], int r, float& x, float& y, float& z) {
    if constexpr (HI - LO == 1) {
        x = px[LO]; y = py[LO]; z = pz[LO];
        asm volatile("" : "+v"(x), "+v"(y), "+v"(z));    // keeps the leaves' reads apart: merged, they become ONE dynamically indexed read
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (r < MID) fm_pick<LO, MID, RPT>(px, py, pz, r, x, y, z);
        else fm_pick<MID, HI, RPT>(px, py, pz, r, x, y, z);
    }
}
template <bool PAD, int NT, int RPT, int LP>
__global__ __launch_bounds__(NT) void fps_multi_kernel(const float* __restrict__ xyz, int ld, int K,
                                                       const int32_t* __restrict__ lengths, int32_t* __restrict__ idx_out,
                                                       float* __restrict__ mind_ws, fm_u64* __restrict__ slots, int G,
                                                       unsigned spin_limit) {
    extern __shared__ __attribute__((aligned(16))) float fm_lds[];           // x[LP], y[LP], z[LP], min-distance[LP]
    static_assert(RPT % 2 == 0 && LP % 2 == 0, "points are paired for the packed distance");
    float* const lx = fm_lds;
    float* const ly = lx + LP;
    float* const lz = ly + LP;
    float* const lm = lz + LP;
    __shared__ float sv[NT / 64];
    __shared__ float sel[4];
    __shared__ int s_idx;                                // the work-group's winning index of the round (atomic min of the candidates)
    __shared__ int s_cur;
    __shared__ int s_dead;                               // latched give-up: the whole work-group leaves at the end of the round
    if (threadIdx.x == 0) {
        s_dead = 0;
        s_idx = 0x7fffffff;
    }
    unsigned spent = 0;                                  // polls spent waiting, CUMULATIVE over the launch (one budget, not one per round)
    const int b = blockIdx.x / G, g = blockIdx.x - b * G, tid = threadIdx.x;
    const int n = lengths[b];
    const float* pts = xyz + (long)b * ld * 3;
    int32_t* idx_b = idx_out + (long)b * K;
    fm_u64* err = slots + FM_SLOT_WORDS;                 // fixed place (the last word of the reservation): the host reads it
    if (n <= FPS_NT * FPS_RPT) return;                   // small / empty clouds: fps_varlen_kernel (launched beside this one) samples them
    const int chunk = (n + G - 1) / G, lo = g * chunk, hi = min(n, lo + chunk), cnt = max(hi - lo, 0);
    const int n_reg = min(cnt, NT * RPT), n_lds = min(cnt - n_reg, LP), n_str = cnt - n_reg - n_lds;
    const int n_lds2 = (n_lds + 1) & ~1;                 // the LDS points are swept in pairs
    // a slot beyond the chunk holds the running minimum -1: it stays -1 (every distance is >= 0) and never reaches a maximum
    float px[RPT], py[RPT], pz[RPT], md[RPT];
    fm_static_for<0, RPT, 1>([&](auto R) __attribute__((always_inline)) {
        constexpr int r = decltype(R)::value;
        const int l = tid + r * NT;
        const bool in = l < n_reg;
        const float* q = pts + (long)(lo + (in ? l : 0)) * 3;
        px[r] = in ? q[0] : 0.f; py[r] = in ? q[1] : 0.f; pz[r] = in ? q[2] : 0.f;
        md[r] = in ? INFINITY : -1.0f;
    });
    for (int l = tid; l < n_lds2; l += NT) {
        const bool in = l < n_lds;
        const float* q = pts + (long)(lo + n_reg + (in ? l : 0)) * 3;
        lx[l] = in ? q[0] : 0.f; ly[l] = in ? q[1] : 0.f; lz[l] = in ? q[2] : 0.f;
        lm[l] = in ? INFINITY : -1.0f;
    }
    float* mind_g = mind_ws + (long)b * ld + lo + n_reg + n_lds;             // streamed remainder
    const float* pstr = pts + (long)(lo + n_reg + n_lds) * 3;
    for (int l = tid; l < n_str; l += NT) mind_g[l] = INFINITY;
    // TWO granule sets, used by alternate rounds: a work-group that has passed round j's sweep may publish round j + 1 while a
    // slower partner is still sweeping round j -- into the other set, so the sweep always finds round j's tags (nobody can be
    // two rounds ahead: round j + 1's sweep needs every partner's round-j + 1 candidate).  A candidate is FIVE tagged words --
    // value, index and the point's coordinates -- so the next round starts from the hand-off itself: no load of the winner's
    // coordinates in the K-round dependent chain.
    fm_u64* my0 = slots + (long)blockIdx.x * FM_GRAN;
    const fm_u64* cloud0 = slots + (long)b * G * FM_GRAN;
    const long set_stride = (long)gridDim.x * FM_GRAN;     // <= 256 work-groups (all resident): FM_SLOT_WORDS covers both sets
    int cur = 0;
    if (tid < 3) sel[tid] = pts[tid];
    __syncthreads();
    for (int j = 0; j < K; ++j) {
        if (PAD && j >= n) {
            if (tid == 0 && g == 0) idx_b[j] = -1;
            continue;
        }
        if (tid == 0 && g == 0) idx_b[j] = cur;
        const float s[3] = {sel[0], sel[1], sel[2]};
        float bv = -1.0f;
        fm_static_for<0, RPT, 2>([&](auto R) __attribute__((always_inline)) {
            constexpr int r = decltype(R)::value;
            const fm_f32x2 d = fm_dist2_pk((fm_f32x2){px[r], px[r + 1]}, (fm_f32x2){py[r], py[r + 1]}, (fm_f32x2){pz[r], pz[r + 1]}, s);
            md[r] = fm_min(md[r], d.x);
            md[r + 1] = fm_min(md[r + 1], d.y);
            bv = fm_max3(bv, md[r], md[r + 1]);
        });
        for (int l = 2 * tid; l < n_lds2; l += 2 * NT) {
            const fm_f32x2 d = fm_dist2_pk(*(const fm_f32x2*)(lx + l), *(const fm_f32x2*)(ly + l), *(const fm_f32x2*)(lz + l), s);
            fm_f32x2 m = *(const fm_f32x2*)(lm + l);
            m.x = fm_min(m.x, d.x);
            m.y = fm_min(m.y, d.y);
            *(fm_f32x2*)(lm + l) = m;
            bv = fm_max3(bv, m.x, m.y);
        }
        for (int l0 = tid; l0 < n_str; l0 += FM_STR_U * NT) {   // FM_STR_U points per trip, their loads issued back to back
            float q[FM_STR_U][3], mo[FM_STR_U];
#pragma unroll
            for (int u = 0; u < FM_STR_U; ++u) {
                const int l = l0 + u * NT, lc = l < n_str ? l : l0;
                q[u][0] = pstr[(long)lc * 3]; q[u][1] = pstr[(long)lc * 3 + 1]; q[u][2] = pstr[(long)lc * 3 + 2];
                mo[u] = mind_g[lc];
            }
#pragma unroll
            for (int u = 0; u < FM_STR_U; ++u) {
                const int l = l0 + u * NT;
                if (l < n_str) {
                    const float m = fm_min(mo[u], dist2_rn(q[u], s, 3));
                    mind_g[l] = m;
                    bv = fmaxf(bv, m);
                }
            }
        }
        // ---- the work-group's maximum ...
        const float wv = fm_wave_max(bv);
        if ((tid & 63) == 0) sv[tid >> 6] = wv;
        __syncthreads();
        float wmax = sv[0];
#pragma unroll
        for (int k = 1; k < NT / 64; ++k) wmax = fmaxf(wmax, sv[k]);
        // ... and the lowest index holding it: only threads whose own maximum IS the work-group's look (their points in ascending order)
        int mine = 0x7fffffff, mr = 0;
        if (bv == wmax && wmax >= 0.0f) {
            int rr = -1;
            fm_static_for<0, RPT, 1>([&](auto R) __attribute__((always_inline)) {
                constexpr int r = RPT - 1 - decltype(R)::value;        // descending: the lowest matching r stays
                if (md[r] == wmax) rr = r;
            });
            if (rr >= 0) {
                mine = lo + tid + rr * NT;
                mr = rr;
            } else {
                for (int l = 2 * tid; l < n_lds2 && mine == 0x7fffffff; l += 2 * NT) {
                    if (lm[l] == wmax) mine = lo + n_reg + l;
                    else if (lm[l + 1] == wmax) mine = lo + n_reg + l + 1;
                }
                for (int l = tid; l < n_str && mine == 0x7fffffff; l += NT)
                    if (mind_g[l] == wmax) mine = lo + n_reg + n_lds + l;
            }
            atomicMin(&s_idx, mine);
        }
        __syncthreads();
        const int wbi = s_idx;
        // ---- hand-off between the G work-groups of this cloud (round tag j + 1: never 0, the slots are zeroed per call)
        const fm_u64 tag = (fm_u64)(unsigned)(j + 1) << 32;
        fm_u64* my = my0 + (j & 1) * set_stride;
        // the ONE thread that owns the winning point publishes it; a work-group without a candidate (empty chunk) publishes the
        // losing value -1 from thread 0
        if (mine == wbi && (wbi != 0x7fffffff || tid == 0)) {
            float cx = 0.f, cy = 0.f, cz = 0.f;
            if (wbi != 0x7fffffff) {
                const int l = wbi - lo;
                if (l < n_reg) {
                    fm_pick<0, RPT, RPT>(px, py, pz, __builtin_amdgcn_readfirstlane(mr), cx, cy, cz);
                } else if (l < n_reg + n_lds) {
                    cx = lx[l - n_reg]; cy = ly[l - n_reg]; cz = lz[l - n_reg];
                } else {
                    const float* q = pstr + (long)(l - n_reg - n_lds) * 3;
                    cx = q[0]; cy = q[1]; cz = q[2];
                }
            }
            const float pv = wbi != 0x7fffffff ? wmax : -1.0f;
            __hip_atomic_store(my, tag | (fm_u64)__float_as_uint(pv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(my + 1, tag | (fm_u64)(unsigned)wbi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(my + 2, tag | (fm_u64)__float_as_uint(cx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(my + 3, tag | (fm_u64)__float_as_uint(cy), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(my + 4, tag | (fm_u64)__float_as_uint(cz), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid < 64) {
            const fm_u64* cloud = cloud0 + (j & 1) * set_stride;
            const bool word = tid < FM_GRAN * G && (tid & (FM_GRAN - 1)) < 5;
            fm_u64 x = 0;
            for (;;) {
                x = word ? __hip_atomic_load(cloud + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
                if (__all((x >> 32) == (tag >> 32))) break;
                // A partner is not resident / has given up: give up too, ONCE -- the flag is latched for the work-group and the
                // error word tells every other work-group of the launch (they look at it every 64 polls) and the fallback launch
                // behind this one (fps_varlen_kernel mode 2), which re-samples the batch's big clouds on one work-group each.
                ++spent;
                const bool told = (spent & 63u) == 0u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull;
                if (spent > spin_limit || told) {
                    if (tid == 0) {
                        __hip_atomic_store(err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        s_dead = 1;
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            // lanes 8k .. 8k+4 hold work-group k's value / index / coordinates: the same lowest-index arg-max on every work-group
            const int xl = (int)(unsigned)x;
            int kw = 0;
            float cv = __int_as_float(__builtin_amdgcn_readlane(xl, 0));
            int ci = __builtin_amdgcn_readlane(xl, 1);
            for (int k = 1; k < G; ++k) {
                const float ov = __int_as_float(__builtin_amdgcn_readlane(xl, FM_GRAN * k));
                const int oi = __builtin_amdgcn_readlane(xl, FM_GRAN * k + 1);
                if (ov > cv || (ov == cv && oi < ci)) {
                    cv = ov;
                    ci = oi;
                    kw = k;
                }
            }
            const int wc = __shfl(xl, FM_GRAN * kw + (tid < 5 ? tid : 0), 64);     // lanes 2..4: the winner's x, y, z
            if (tid >= 2 && tid < 5) sel[tid - 2] = __int_as_float(wc);
            if (tid == 0) {
                s_cur = ci;
                s_idx = 0x7fffffff;                      // (every thread has read this round's; the candidates of the next come after its first barrier)
            }
        }
        __syncthreads();
        if (s_dead) return;                              // gave up: no further sweeps, no further spinning (the fallback launch re-samples)
        cur = s_cur;
    }
}

extern "C" size_t pm_fps_varlen_workspace_bytes(int B, int ld) {
    // running min-distances of the streamed part (clouds beyond the register-resident size) + the hand-off granules of the
    // several-work-groups-per-cloud kernel (two sets of 2 per work-group, <= 256 work-groups, + the error word), 8-byte aligned behind them
    const size_t mind = ld > FPS_NT * FPS_RPT ? (((size_t)B * ld * sizeof(float) + 7) & ~(size_t)7) : 0;
    return mind + (mind ? (size_t)(FM_SLOT_WORDS + 1) * sizeof(fm_u64) : 0);
}

// Work-groups per cloud of the multi-work-group sampler for a (B, ld, D) batch; < 2: the one-work-group kernels run.
// Everything that used to come from the environment is the CALLER's to pass (pm_fps_config; NULL = defaults): the group cap
// (0 / 1 switches the multi-work-group path off, e.g. when the caller masks CUs or shares the device with another long-running
// launch: the partners of a cloud must all be resident), the CUs the caller knows to be available to the launch (0 = all the
// device reports), the poll budget, and the launch shape.
static int fps_groups_cfg(int B, int ld, int D, const pm_fps_config* cfg) {
    if (B <= 0 || D != 3 || ld <= FPS_NT * FPS_RPT) return 1;
    int ncu = cfg && cfg->resident_cus > 0 ? cfg->resident_cus : pm_cu_count();
    if (ncu > 256) ncu = 256;
    int G = ncu / B;
    if (G > FM_MAXG) G = FM_MAXG;
    if (cfg && cfg->max_groups >= 0 && cfg->max_groups < G) G = cfg->max_groups;
    return G < 1 ? 1 : G;
}
extern "C" int pm_fps_varlen_groups(int B, int ld, int D) { return fps_groups_cfg(B, ld, D, nullptr); }
extern "C" int pm_fps_varlen_groups_cfg(int B, int ld, int D, const pm_fps_config* cfg) { return fps_groups_cfg(B, ld, D, cfg); }

extern "C" int pm_fps_varlen_cfg_f32(const float* xyz, int B, int ld, int D, int K, const int32_t* lengths, int pad, int32_t* idx_out,
                                     const pm_fps_config* cfg, void* workspace, size_t workspace_bytes, void* stream);
extern "C" int pm_fps_varlen_f32(const float* xyz, int B, int ld, int D, int K, const int32_t* lengths, int pad,
                                 int32_t* idx_out, void* workspace, size_t workspace_bytes, void* stream) {
    return pm_fps_varlen_cfg_f32(xyz, B, ld, D, K, lengths, pad, idx_out, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int pm_fps_varlen_cfg_f32(const float* xyz, int B, int ld, int D, int K, const int32_t* lengths, int pad, int32_t* idx_out,
                                     const pm_fps_config* cfg_in, void* workspace, size_t workspace_bytes, void* stream) {
    PM_REQUIRE(xyz && lengths && idx_out && B > 0 && ld > 0 && D >= 1 && D <= FPS_MAXD && K > 0);
    if (ld > FPS_NT * FPS_RPT && (!workspace || workspace_bytes < (size_t)B * ld * sizeof(float))) return PM_EWORKSPACE;
    // several work-groups per cloud: xyz clouds, few enough clouds that G >= 2 work-groups each are all resident (one
    // 1024-thread work-group of 128 VGPRs per CU), and the caller handed over the larger workspace
    const int G = fps_groups_cfg(B, ld, D, cfg_in);
    const bool full_ws = ld > FPS_NT * FPS_RPT && workspace_bytes >= pm_fps_varlen_workspace_bytes(B, ld) && ((uintptr_t)workspace & 7) == 0;
    if (G >= 2 && full_ws) {
        const size_t mind = (((size_t)B * ld * sizeof(float) + 7) & ~(size_t)7);
        fm_u64* slots = (fm_u64*)((char*)workspace + mind);
        // round tags start at 1 and the give-up word is per call: the granules are cleared in front of every multi-work-group launch
        if (hipMemsetAsync(slots, 0, (size_t)(FM_SLOT_WORDS + 1) * sizeof(fm_u64), pm_stream(stream)) != hipSuccess) return PM_EINVAL;
        const unsigned limit = cfg_in && cfg_in->spin_limit_set ? cfg_in->spin_limit : FM_SPIN_LIMIT;    // (tests force the give-up path with 0)
        const int cfg = cfg_in && cfg_in->legacy_shape ? 0 : 1;     // legacy_shape (A/B): the 1024-thread x 16-point shape of rounds 2-3
#define FM_LAUNCH(PAD_, NT_, RPT_, LP_)                                                                                      \
        hipLaunchKernelGGL((fps_multi_kernel<PAD_, NT_, RPT_, LP_>), dim3(B * G), dim3(NT_), (size_t)(LP_) * 16, pm_stream(stream), \
                           xyz, ld, K, lengths, idx_out, (float*)workspace, slots, G, limit)
        if (cfg == 0) { if (pad) FM_LAUNCH(true, 1024, 16, 8192); else FM_LAUNCH(false, 1024, 16, 8192); }
        else { if (pad) FM_LAUNCH(true, FM_NT, FM_RPT, FM_LDS); else FM_LAUNCH(false, FM_NT, FM_RPT, FM_LDS); }
#undef FM_LAUNCH
        // the clouds of the batch that fit one work-group's registers (decided per cloud on the device, no host sync on the lengths)
        hipLaunchKernelGGL(fps_varlen_kernel, dim3(B), dim3(FPS_NT), 0, pm_stream(stream), xyz, ld, D, K, lengths, pad,
                           idx_out, (float*)workspace, 1, (const unsigned long long*)nullptr);
        // ... and, only if a work-group above gave up, the big clouds once more on one work-group each (returns at once otherwise)
        hipLaunchKernelGGL(fps_varlen_kernel, dim3(B), dim3(FPS_NT), 0, pm_stream(stream), xyz, ld, D, K, lengths, pad,
                           idx_out, (float*)workspace, 2, (const unsigned long long*)(slots + FM_SLOT_WORDS));
        PM_CHECK_LAUNCH();
        return PM_OK;
    }
    hipLaunchKernelGGL(fps_varlen_kernel, dim3(B), dim3(FPS_NT), 0, pm_stream(stream), xyz, ld, D, K, lengths, pad,
                       idx_out, (float*)workspace, 0, (const unsigned long long*)nullptr);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ---------------------------------------------------------------------------------- crop compaction
// The cropped world cloud (depth2tsdf.py:155-159) is mostly points zeroed to the origin.  For farthest point
// sampling they are ONE point: FPS on [all non-zero points + the first zero point, original order kept] selects
// the same sequence of POINTS as FPS on the full cloud (equal candidates have equal distances, ties go to the
// lowest index, and the first point of the cloud is always kept) -- and reads 3-20x less per round.
// One work-group per env: pass 1 finds the first zero point (block min), pass 2 is a stable stream compaction:
// wave ballot + popcount for the slot inside a wave, an LDS prefix over the 16 waves, a running base per chunk.
__global__ __launch_bounds__(1024) void depth_compact_kernel(const float* __restrict__ xyz, int P,
                                                              float* __restrict__ out, int32_t* __restrict__ lengths) {
    __shared__ int wcnt[16];
    __shared__ int s_first;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pts = xyz + (long)b * P * 3;
    float* dst = out + (long)b * P * 3;
    if (tid == 0) s_first = 0x7fffffff;
    __syncthreads();
    int first = 0x7fffffff;
    for (int p = tid; p < P; p += 1024)
        if (pts[p * 3] == 0.f && pts[p * 3 + 1] == 0.f && pts[p * 3 + 2] == 0.f) {
            first = p;                                    // ascending p per thread: the first hit is the thread's minimum
            break;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o, 64));
    if (lane == 0 && first != 0x7fffffff) atomicMin(&s_first, first);
    __syncthreads();
    first = s_first;
    int base = 0;
    for (int p0 = 0; p0 < P; p0 += 1024) {
        const int p = p0 + tid;
        float x = 0.f, y = 0.f, z = 0.f;
        bool keep = false;
        if (p < P) {
            x = pts[p * 3]; y = pts[p * 3 + 1]; z = pts[p * 3 + 2];
            keep = (x != 0.f || y != 0.f || z != 0.f) || p == first;
        }
        const unsigned long long m = __ballot(keep);
        const int within = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[wave] = __popcll(m);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int c = wcnt[w];
            before += (w < wave) ? c : 0;
            total += c;
        }
        if (keep) {
            const long q = base + before + within;
            dst[q * 3] = x; dst[q * 3 + 1] = y; dst[q * 3 + 2] = z;
        }
        base += total;
        __syncthreads();
    }
    if (tid == 0) lengths[b] = base;
}

extern "C" int pm_depth_compact_f32(const float* xyz, int B, int P, float* out, int32_t* lengths, void* stream) {
    PM_REQUIRE(xyz && out && lengths && B > 0 && P > 0 && xyz != out);
    hipLaunchKernelGGL(depth_compact_kernel, dim3(B), dim3(1024), 0, pm_stream(stream), xyz, P, out, lengths);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ---------------------------------------------------------------------------------- sparse TSDF voxels
// utils/depth2tsdf.py:103-119 (`TSDFVolume.sparse_voxel` after the integration): per env the voxels whose TSDF lies
// in the open band (lo, hi) -- `torch.where` order = row-major voxel index -- go to farthest point sampling as
// integer coordinates, and the sampled (x, y, z, tsdf) rows are the 'depth_sparse' observation of the PointNet
// learner.  Same stable stream compaction as depth_compact_kernel (wave ballot + popcount, LDS prefix over waves).
__global__ __launch_bounds__(1024) void tsdf_select_kernel(const float* __restrict__ vol, int V, int res, float lo,
                                                            float hi, float* __restrict__ coords,
                                                            int32_t* __restrict__ lengths) {
    __shared__ int wcnt[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* vb = vol + (long)b * V;
    float* dst = coords + (long)b * V * 3;
    int base = 0;
    for (int v0 = 0; v0 < V; v0 += 1024) {
        const int v = v0 + tid;
        bool keep = false;
        if (v < V) {
            const float t = vb[v];
            keep = t < hi && t > lo;
        }
        const unsigned long long m = __ballot(keep);
        const int within = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[wave] = __popcll(m);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int c = wcnt[w];
            before += (w < wave) ? c : 0;
            total += c;
        }
        if (keep) {
            const long q = base + before + within;
            dst[q * 3] = (float)(v / (res * res));
            dst[q * 3 + 1] = (float)((v / res) % res);
            dst[q * 3 + 2] = (float)(v % res);
        }
        base += total;
        __syncthreads();
    }
    if (tid == 0) lengths[b] = base;
}

// out[b,k,:] = (x, y, z, vol[b,x,y,z]) of the sampled voxel; a padding index (-1: fewer candidates than K) reads as
// voxel (0,0,0), as the reference's zero-filled gather does (depth2tsdf.py:116-119).
__global__ __launch_bounds__(256) void tsdf_sparse_gather_kernel(const float* __restrict__ coords,
                                                                  const int32_t* __restrict__ idx,
                                                                  const float* __restrict__ vol, int V, int res, int K,
                                                                  long total, float* __restrict__ out) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long b = e / K;
        const int i = idx[e];
        float x = 0.f, y = 0.f, z = 0.f;
        if (i >= 0) {
            const float* c = coords + ((long)b * V + i) * 3;
            x = c[0]; y = c[1]; z = c[2];
        }
        const int v = ((int)x * res + (int)y) * res + (int)z;
        out[e * 4] = x; out[e * 4 + 1] = y; out[e * 4 + 2] = z;
        out[e * 4 + 3] = vol[b * V + v];
    }
}

extern "C" int pm_tsdf_select_f32(const float* vol, int B, int res, float lo, float hi, float* coords,
                                  int32_t* lengths, void* stream) {
    PM_REQUIRE(vol && coords && lengths && B > 0 && res > 0 && res <= 1024);
    hipLaunchKernelGGL(tsdf_select_kernel, dim3(B), dim3(1024), 0, pm_stream(stream), vol, res * res * res, res, lo, hi,
                       coords, lengths);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

extern "C" int pm_tsdf_sparse_gather_f32(const float* coords, const int32_t* idx, const float* vol, int B, int res,
                                         int K, float* out, void* stream) {
    PM_REQUIRE(coords && idx && vol && out && B > 0 && res > 0 && K > 0);
    const long total = (long)B * K;
    long nb = (total + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(tsdf_sparse_gather_kernel, dim3((unsigned)nb), dim3(256), 0, pm_stream(stream), coords, idx, vol,
                       res * res * res, res, K, total, out);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ---------------------------------------------------------------------------------- depth -> world cloud
// utils/depth2tsdf.py:142-157 (`TSDFVolume.depth2pc` before the sampling): back-project every pixel of
// every view with the pinhole intrinsics, move it to world coordinates with the view's pose, and zero
// the points outside the open workspace box (lo, hi) so that farthest point sampling (K12) sees them as one
// point at the origin.  Pure streaming: 4 B in, 12 B out per pixel.  Rounding follows the reference's tensor
// expression bit for bit (checked against its own output, tests/golden/depth2pc_small.npz):
//   p0 = ((col - cx) * z) / fx,  p1 = ((row - cy) * z) / fy,  p2 = z   (each op rounded)
//   w[d] = fma(p2, R[d][2], fma(p1, R[d][1], p0 * R[d][0])) + t[d]      (torch.bmm's K=3 inner product)
__global__ __launch_bounds__(256) void depth_backproject_kernel(const float* __restrict__ depth, long n_per_env,
                                                                 int HW, int W, const float* __restrict__ pose,
                                                                 float fx, float fy, float cx, float cy, float lo0,
                                                                 float lo1, float lo2, float hi0, float hi1, float hi2,
                                                                 long total, float* __restrict__ out) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e % n_per_env;                     // pixel inside the env: view * HW + pix
        const int m = (int)(r / HW), pix = (int)(r - (long)m * HW);
        const int row = pix / W, col = pix - row * W;
        const float z = depth[e];
        const float p0 = __fdiv_rn(mul_rn(sub_rn((float)col, cx), z), fx);
        const float p1 = __fdiv_rn(mul_rn(sub_rn((float)row, cy), z), fy);
        const float* T = pose + m * 16;                   // row-major 4x4
        float w[3];
#pragma unroll
        for (int d = 0; d < 3; ++d)
            w[d] = add_rn(__fmaf_rn(z, T[d * 4 + 2], __fmaf_rn(p1, T[d * 4 + 1], mul_rn(p0, T[d * 4]))), T[d * 4 + 3]);
        const bool ok = w[0] < hi0 && w[1] < hi1 && w[2] < hi2 && w[0] > lo0 && w[1] > lo1 && w[2] > lo2;
        out[e * 3] = ok ? w[0] : 0.f;
        out[e * 3 + 1] = ok ? w[1] : 0.f;
        out[e * 3 + 2] = ok ? w[2] : 0.f;
    }
}

extern "C" int pm_depth_backproject_f32(const float* depth, int B, int M, int H, int W, const float* cam_pose,
                                        float fx, float fy, float cx, float cy, const float* lo, const float* hi,
                                        float* out, void* stream) {
    PM_REQUIRE(depth && cam_pose && lo && hi && out && B > 0 && M > 0 && H > 0 && W > 0);
    const long total = (long)B * M * H * W;
    long nb = (total + 255) / 256;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(depth_backproject_kernel, dim3((unsigned)nb), dim3(256), 0, pm_stream(stream), depth,
                       (long)M * H * W, H * W, W, cam_pose, fx, fy, cx, cy, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2],
                       total, out);
    PM_CHECK_LAUNCH();
    return PM_OK;
}
