// K6/K7: PointNet per-point shared MLP (C->128->256->512, tanh,tanh,none) fused with the
// symmetric max(/mean) pooling -- network.py:147-153,172-181 -- forward and backward.
//
// Data layout in HBM
//   x       (B, ldx)      one row per env-step: P points x C floats (+ optional proprio tail)
//   packed  768 KB        W2 / W3 / W2-for-backward re-laid in MFMA B-operand order so that a
//                         wave's 16 B-per-lane load is one contiguous 1 KB (L2-resident)
//   feat    (B, ldf)      [max(512) | mean(512)]   argmax (B,512) int32
//   never materialised: the (B,P,128/256/512) activations (4.3 GB per net at B=2048).
//
// Forward: one work-group (8 waves: PN_FWD_NW) owns one cloud and walks it in tiles of 64 points
// (the backward uses 32-point tiles so that two work-groups fit one CU, see below):
//   layer 1 (K=C<=8)  VALU          -> H1 tile in LDS  [64][132]
//   layer 2 (K=128)   fp32 MFMA     -> H2 tile in LDS  [64][260]   (aliases H1 after a barrier)
//   layer 3 (K=256)   fp32 MFMA     -> running max / argmax / sum per channel in registers
// A operands (activations) come from LDS by ds_read_b128 (row stride = K+4 floats: bank-
// conflict free for the 16-lane b128 groups); B operands (weights) stream from L2 straight
// into registers, one N-slice per wave, so the only LDS traffic is the activation tile.
// MFMA = v_mfma_f32_32x32x2_f32 with the k-split convention: lanes 0-31 own k in [0,K/2),
// lanes 32-63 own k in [K/2,K); points are MFMA rows, channels MFMA columns, so pooling over
// points is an in-lane reduction over the 16 accumulator registers + one lane^32 exchange.
#include "common.h"

#define PN_C1 128
#define PN_C2 256
#define PN_C3 512
#define PN_TM 64
#define PN_LD1 (PN_C1 + 4)
#define PN_LD2 (PN_C2 + 4)
#define PN_MAXC 8
#ifndef PN_ABLATE
#define PN_ABLATE 0          // profiling only (wrong results): forward 1 = no layer-3 MFMAs, 2 = no layer-2 MFMAs, 256 = no layer 1 (VALU), 512 = no tanh in the layer-2
                             // epilogue, 1024 = no pooling, 2048 = no saved-h2 copy, 4096 = no point staging; 4 = no operand traffic in the MFMA loops; 8 .. 128: backward
#endif
#ifndef PN_FWD_NW
#define PN_FWD_NW 8          // waves per forward work-group (4 or 8)
#endif

#include "mfma_f32.h"

#define PN_P2_OFF 0                      // [8][16][64][4]  W2 as fwd B operand
#define PN_P3_OFF 32768                  // [16][32][64][4] W3 as fwd B operand
#define PN_P2T_OFF (32768 + 131072)      // [4][32][64][4]  W2 as bwd (dh1 = dz2 * W2) B operand
#define PN_P2T16_OFF (32768 + 131072 + 32768)   // [8][16][64][4] W2 as B operand of the 16x16x4 MFMA (pn_bwd16_kernel's dh1)
#define PN_PACKED_DATA (32768 + 131072 + 32768 + 32768)
#define PN_PACKED_ELEMS (PN_PACKED_DATA + 1024)   // + 4 KB tail pad: the operand stream prefetches one group past the end

extern "C" size_t pm_pointnet_packed_elems(void) { return PN_PACKED_ELEMS; }

__global__ __launch_bounds__(256) void pn_pack_kernel(const float* __restrict__ W2, const float* __restrict__ W3,
                                                       float* __restrict__ packed) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= PN_PACKED_ELEMS) return;
    if (i >= PN_PACKED_DATA) {
        packed[i] = 0.f;
        return;
    }
    const int e = i & 3, lane = (i >> 2) & 63, li = lane & 31, lh = lane >> 5;
    if (i < PN_P3_OFF) {                                   // W2 fwd: half = 64 k, 16 groups of 4
        const int s4 = (i >> 8) & 15, nb = i >> 12;
        packed[i] = W2[(nb * 32 + li) * PN_C1 + lh * 64 + s4 * 4 + e];
    } else if (i < PN_P2T_OFF) {                           // W3 fwd: half = 128 k, 32 groups
        const int j = i - PN_P3_OFF, s4 = (j >> 8) & 31, nb = j >> 13;
        packed[i] = W3[(nb * 32 + li) * PN_C2 + lh * 128 + s4 * 4 + e];
    } else if (i < PN_P2T16_OFF) {                         // W2 bwd: B[k=out][j=in], half = 128 k
        const int j = i - PN_P2T_OFF, s4 = (j >> 8) & 31, nb = j >> 13;
        packed[i] = W2[(lh * 128 + s4 * 4 + e) * PN_C1 + nb * 32 + li];
    } else {                                               // W2 bwd, 16x16x4: lane quarter q owns k in [64q, 64q+64)
        const int j = i - PN_P2T16_OFF, g = (j >> 8) & 15, nb = j >> 12, l16 = lane & 15, q = lane >> 4;
        packed[i] = W2[(q * 64 + g * 4 + e) * PN_C1 + nb * 16 + l16];
    }
}

extern "C" int pm_pointnet_pack_weights_f32(const float* W2, const float* W3, float* packed, void* stream) {
    PM_REQUIRE(W2 && W3 && packed);
    hipLaunchKernelGGL(pn_pack_kernel, dim3((PN_PACKED_ELEMS + 255) / 256), dim3(256), 0, pm_stream(stream), W2, W3,
                       packed);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ---- shared device pieces -----------------------------------------------------------------

// centroid of the first 3 coordinates of a cloud (network.py:172-173 `sub_mean`)
template <int NT>
__device__ __forceinline__ void cloud_centroid(const float* __restrict__ xb, int P, int C, double* red, float (&cen)[3]) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int p = threadIdx.x; p < P; p += NT) {
        s0 += (double)xb[p * C];
        s1 += (double)xb[p * C + 1];
        s2 += (double)xb[p * C + 2];
    }
    s0 = block_sum<double, NT>(s0, red);
    s1 = block_sum<double, NT>(s1, red);
    s2 = block_sum<double, NT>(s2, red);
    cen[0] = (float)(s0 / P);
    cen[1] = (float)(s1 / P);
    cen[2] = (float)(s2 / P);
}

// ---- activation of the shared per-point MLP (network.py:147-153: `activation` = any of get_activation's seven) -----------
// TANH = true: the tuned path (packed tanh, 1 - h^2) -- every shipped cfg; false: pm_act / pm_dact with the runtime code.
template <bool TANH>
__device__ __forceinline__ f32x2 pn_act2(float a, float b, int act) {
    if constexpr (TANH) return pm_tanh2(a, b);
    else return (f32x2){pm_act(a, act), pm_act(b, act)};
}
template <bool TANH>
__device__ __forceinline__ float pn_act1(float a, int act) {
    if constexpr (TANH) return pm_tanh(a);
    else return pm_act(a, act);
}
template <bool TANH>
__device__ __forceinline__ float pn_dact(float h, int act) {                  // derivative through the OUTPUT h
    if constexpr (TANH) return 1.0f - h * h;
    else return pm_dact(h, act);
}

// stage one tile of TM points (C floats each) into Xs[TM][PN_MAXC], optionally re-centred
template <int TM, int NT>
__device__ __forceinline__ void stage_points(const float* __restrict__ xb, int tile, int C, int sub_mean,
                                             const float (&cen)[3], float* __restrict__ Xs) {
    for (int i = threadIdx.x; i < TM * PN_MAXC; i += NT) {
        const int p = i >> 3, d = i & 7;
        float v = 0.f;                                   // slots d >= C stay zero (layer1_tile reads float4s)
        if (d < C) {
            v = xb[(tile * TM + p) * C + d];
            if (sub_mean && d < 3) v -= cen[d];
        }
        Xs[i] = v;
    }
}

// layer 1: thread (c = tid&127, part = tid>>7) computes tanh(b1[c] + W1[c,:] . x[p,:]) for its PPT points.
// CT = compile-time channel count (3: xyz clouds, 4: depth_sparse); 0 = generic runtime C <= 8.
template <int CT, int TM, int NT, bool TANH = true>
__device__ __forceinline__ void layer1_tile(const float* __restrict__ Xs, const float* __restrict__ W1,
                                            const float* __restrict__ b1, int C, float* __restrict__ H1, int act = PM_ACT_TANH) {
    constexpr int PPT = TM * 128 / NT;           // points per thread
    const int c = threadIdx.x & 127, p0 = (threadIdx.x >> 7) * PPT;
    const float b1c = b1[c];
    if (CT == 3 || CT == 4) {                  // one broadcast ds_read_b128 per point, no branches
        float w[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) w[d] = (d < CT) ? W1[c * CT + d] : 0.f;
        static_assert(PPT % 2 == 0, "layer 1 pairs points for the packed tanh");
#pragma unroll 4
        for (int p = p0; p < p0 + PPT; p += 2) {
            float s2[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4 xv = *(const float4*)(Xs + (p + j) * PN_MAXC);
                float s = fmaf(w[0], xv.x, b1c);
                s = fmaf(w[1], xv.y, s);
                s = fmaf(w[2], xv.z, s);
                if (CT == 4) s = fmaf(w[3], xv.w, s);
                s2[j] = s;
            }
            const f32x2 t = pn_act2<TANH>(s2[0], s2[1], act);
            H1[p * PN_LD1 + c] = t.x;
            H1[(p + 1) * PN_LD1 + c] = t.y;
        }
    } else {
        float w1[PN_MAXC];
#pragma unroll
        for (int d = 0; d < PN_MAXC; ++d) w1[d] = (d < C) ? W1[c * C + d] : 0.f;
        for (int p = p0; p < p0 + PPT; ++p) {
            const float4 x0 = *(const float4*)(Xs + p * PN_MAXC), x1 = *(const float4*)(Xs + p * PN_MAXC + 4);
            float s = b1c;                     // Xs slots d >= C are zero-filled by stage_points
            s = fmaf(w1[0], x0.x, s); s = fmaf(w1[1], x0.y, s); s = fmaf(w1[2], x0.z, s); s = fmaf(w1[3], x0.w, s);
            s = fmaf(w1[4], x1.x, s); s = fmaf(w1[5], x1.y, s); s = fmaf(w1[6], x1.z, s); s = fmaf(w1[7], x1.w, s);
            H1[p * PN_LD1 + c] = pn_act1<TANH>(s, act);
        }
    }
}

// ---- MFMA operand streaming without register copies ----------------------------* W2^T, K = 128
template <int MB, int NB>
__device__ __forceinline__ void layer2_mfma(const float* __restrict__ H1, const float4* __restrict__ P2v, int wave,
                                            int lane, f32x16 (&acc)[MB][NB]) {
    const int li = lane & 31, lh = lane >> 5;
    mfma_stream<MB, NB, 16>(H1 + li * PN_LD1 + lh * 64, PN_LD1, P2v + (size_t)(wave * NB) * 16 * 64 + lane, acc);
}

// layer 2 epilogue: H2[row][ch] = tanh(acc + b2[ch])
template <int MB, int NB, bool TANH = true>
__device__ __forceinline__ void layer2_store(const f32x16 (&acc)[MB][NB], const float* __restrict__ b2, int wave, int lane,
                                             float* __restrict__ H2, float* __restrict__ h2_global = nullptr, int act = PM_ACT_TANH) {
    const int li = lane & 31, lh = lane >> 5;
    float b2v[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) b2v[nb] = b2[(wave * NB + nb) * 32 + li];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 v2 = pn_act2<TANH>(acc[mb][nb][r] + b2v[nb], acc[mb][nb][r + 1] + b2v[nb], act);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = mb * 32 + ((r + j) & 3) + 8 * ((r + j) >> 2) + 4 * lh;
                    const float v = j ? v2.y : v2.x;
                    H2[row * PN_LD2 + (wave * NB + nb) * 32 + li] = v;
                    if (h2_global) h2_global[row * PN_C2 + (wave * NB + nb) * 32 + li] = v;   // 128-B row segments per half-wave
                }
            }
}

// =================================================================================== forward
// NW waves per work-group (template): the 512-channel output is split NW ways.  NW = 8 (512
// threads, 64 channels per wave in layer 3, <=128 VGPRs) puts FOUR waves on every SIMD (two
// work-groups per CU), so VALU / LDS / barrier phases of one wave hide under the MFMA phases of
// three others; NW = 4 (128 channels per wave, 256 VGPRs, two waves per SIMD) is kept for A/B.
template <int CT, int NW, bool TANH = true>
__global__ __launch_bounds__(NW * 64, NW / 2) void pn_fwd_kernel(const float* __restrict__ x, long ldx, int P, int C,
                                                                  int sub_mean, const float* __restrict__ W1,
                                                                  const float* __restrict__ b1,
                                                                  const float* __restrict__ b2,
                                                                  const float* __restrict__ b3,
                                                                  const float* __restrict__ packed, int max_mean,
                                                                  float* __restrict__ feat, long ldf,
                                                                  int32_t* __restrict__ argmax,
                                                                  float* __restrict__ h2_save, int act) {
    constexpr int NT = NW * 64, NB2 = 8 / NW, NB3 = 16 / NW;
    __shared__ __attribute__((aligned(16))) float smem[PN_TM * PN_LD2 + PN_TM * PN_MAXC + 32];
    float* H = smem;                             // H1 [64][132] then H2 [64][260] (aliased)
    float* Xs = smem + PN_TM * PN_LD2;
    double* red = (double*)(Xs + PN_TM * PN_MAXC);

    const int b = blockIdx.x, tid = threadIdx.x, lane0 = tid & 63, wave = tid >> 6;
    const float* xb = x + (long)b * ldx;
    const float4* P2v = (const float4*)(packed + PN_P2_OFF);
    const float4* P3v = (const float4*)(packed + PN_P3_OFF);

    float cen[3] = {0.f, 0.f, 0.f};
    if (sub_mean) cloud_centroid<NT>(xb, P, C, red, cen);

    float vmax[NB3], vsum[NB3];
    int imax[NB3];
#pragma unroll
    for (int nb = 0; nb < NB3; ++nb) {
        vmax[nb] = -INFINITY;
        vsum[nb] = 0.f;
        imax[nb] = 0;
    }

    const int ntiles = P / PN_TM;
    stage_points<PN_TM, NT>(xb, 0, C, sub_mean, cen, Xs);
    for (int tile = 0; tile < ntiles; ++tile) {
        // Launder the lane id once per tile: every LDS / packed-weight address below is then recomputed from
        // it inside the tile (a few VALU ops) instead of being hoisted to kernel entry as ~30 lane-constant
        // VGPRs that do not fit the 128-register budget and get spilled to scratch (134 MB of spill stores
        // per launch in the PMC WRITE_SIZE of the first 8-wave build).
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int li = lane & 31, lh = lane >> 5;
        __syncthreads();                          // Xs staged; previous tile's layer-3 reads of H are done
        if (!(PN_ABLATE & 256)) layer1_tile<CT, PN_TM, NT, TANH>(Xs, W1, b1, C, H, act);
        __syncthreads();
        {
            f32x16 acc2[2][NB2];
            zero_acc<2, NB2>(acc2);
#if !(PN_ABLATE & 2)
            layer2_mfma<2, NB2>(H, P2v, wave, lane, acc2);
#endif
            __syncthreads();                      // every wave has finished reading H1 (and Xs)
            layer2_store<2, NB2, (PN_ABLATE & 512) ? false : TANH>(acc2, b2, wave, lane, H, nullptr, (PN_ABLATE & 512) ? PM_ACT_NONE : act);
            if (!(PN_ABLATE & 4096) && tile + 1 < ntiles) stage_points<PN_TM, NT>(xb, tile + 1, C, sub_mean, cen, Xs);
        }
        __syncthreads();

        // ---- layer 3: 64 points x this wave's NB3*32 channels, K = 256 ------------------------
        f32x16 acc[2][NB3];
#pragma unroll
        for (int nb = 0; nb < NB3; ++nb) {
            const float b3c = b3[(wave * NB3 + nb) * 32 + li];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][nb][r] = acc[1][nb][r] = b3c;
        }
#if !(PN_ABLATE & 1)
        mfma_stream<2, NB3, 32>(H + li * PN_LD2 + lh * 128, PN_LD2, P3v + (size_t)(wave * NB3) * 32 * 64 + lane, acc);
#endif
        // ---- pooling over this tile's 64 points (rows), in increasing point order ------------
#if (PN_ABLATE & 1024) && defined(__HIP_DEVICE_COMPILE__)   // timing probe: the layer-3 results stay live without the pooling arithmetic
#pragma unroll
        for (int nb = 0; nb < NB3; ++nb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) asm volatile("" ::"v"(acc[mb][nb]));
#endif
#pragma unroll
        for (int nb = 0; nb < ((PN_ABLATE & 1024) ? 0 : NB3); ++nb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[mb][nb][r];
                    const int p = tile * PN_TM + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (v > vmax[nb]) {
                        vmax[nb] = v;
                        imax[nb] = p;
                    }
                    if constexpr (!TANH) {               // unbounded activations: track a NaN explicitly (see the epilogue)
                        if (v != v && vmax[nb] == vmax[nb]) {       // the FIRST NaN sticks: no later `>` is true against it
                            vmax[nb] = v;
                            imax[nb] = p;
                        }
                    }
                    vsum[nb] += v;
                }
        // Training forward: the H2 tile (still intact in LDS) also goes to HBM, 1 KB rows, so that the backward
        // LOADS h2 instead of recomputing layer 2 -- 1 KB per point against 65.5 kFLOP of exact-fp32 MFMA work
        // (64 FLOP/B where the machine balance is ~20).  Costs the forward +0.25 ms per 2048 clouds (instruction
        // issue: 16 LDS/VMEM ops per thread and tile; plain, nt and sc1 stores, or storing from the layer-2
        // epilogue registers, all measure the same) and saves the backward 0.69 ms.
        if (h2_save && !(PN_ABLATE & 2048)) {
            float* dst = h2_save + ((long)b * P + (long)tile * PN_TM) * PN_C2;
#pragma unroll 2
            for (int i = 0; i < PN_TM * PN_C2 / 4 / NT; ++i) {
                const int q = tid + NT * i, row = q >> 6, c4 = q & 63;
                const f32x4 v = *(const f32x4*)(H + row * PN_LD2 + 4 * c4);
                *(f32x4*)(dst + row * PN_C2 + 4 * c4) = v;
            }
        }
    }
    // lanes l and l^32 hold the two interleaved row sets of the same channel
    const int li = lane0 & 31, lh = lane0 >> 5;
#pragma unroll
    for (int nb = 0; nb < NB3; ++nb) {
        const float ov = __shfl_xor(vmax[nb], 32, 64);
        const int oi = __shfl_xor(imax[nb], 32, 64);
        const float os = __shfl_xor(vsum[nb], 32, 64);
        float v = vmax[nb];
        int i = imax[nb];
        if (!TANH && (ov != ov || v != v)) {        // a NaN in either half: torch.max returns NaN and the index of the first one
            if (ov != ov && (v == v || oi < i)) {
                v = ov;
                i = oi;
            }
        } else if (ov > v || (ov == v && oi < i)) {
            v = ov;
            i = oi;
        }
        if (lh == 0) {
            const int ch = (wave * NB3 + nb) * 32 + li;
            const float sm = vsum[nb] + os;
            // tanh: |h| <= 1, so the column sum is NaN exactly when the column holds a NaN (no overflow, no inf - inf) and the
            // strict > above skipped it: torch.max returns NaN.  Other activations are unbounded (a column with +inf and -inf
            // sums to NaN without holding one): they track the NaN itself, and its first index, above.
            if (TANH && sm != sm) v = sm;
            feat[(long)b * ldf + ch] = v;
            if (max_mean) feat[(long)b * ldf + PN_C3 + ch] = sm / (float)P;
            argmax[(long)b * PN_C3 + ch] = i;
        }
    }
}

extern "C" int pm_pointnet_enc_fwd_f32(const float* x, long ldx, int B, int P, int C, int sub_mean,
                                       const float* W1, const float* b1, const float* b2, const float* b3,
                                       const float* packed, int max_mean, float* feat, long ldf, int32_t* argmax,
                                       float* h2_save, int act, void* stream) {
    PM_REQUIRE(x && W1 && b1 && b2 && b3 && packed && feat && argmax);
    PM_REQUIRE(act > PM_ACT_NONE && act <= PM_ACT_MAX);
    if (h2_save && ((uintptr_t)h2_save & 15) != 0) return PM_EALIGN;
    PM_REQUIRE(B > 0 && P > 0 && P % PN_TM == 0 && C >= 1 && C <= PN_MAXC && ldx >= (long)P * C);
    PM_REQUIRE(ldf >= PN_C3 * (max_mean ? 2 : 1));
    PM_REQUIRE(!sub_mean || C >= 3);
    if (((uintptr_t)packed & 15) != 0) return PM_EALIGN;
#define PN_FWD_LAUNCH_(CT, TH)                                                                             \
    hipLaunchKernelGGL((pn_fwd_kernel<CT, PN_FWD_NW, TH>), dim3(B), dim3(PN_FWD_NW * 64), 0, pm_stream(stream), x, ldx, P, \
                       C, sub_mean, W1, b1, b2, b3, packed, max_mean, feat, ldf, argmax, h2_save, act)
#define PN_FWD_LAUNCH(CT)                        \
    do {                                         \
        if (act == PM_ACT_TANH) PN_FWD_LAUNCH_(CT, true); \
        else PN_FWD_LAUNCH_(CT, false);          \
    } while (0)
    if (C == 3) PN_FWD_LAUNCH(3);
    else if (C == 4) PN_FWD_LAUNCH(4);
    else PN_FWD_LAUNCH(0);
#undef PN_FWD_LAUNCH
#undef PN_FWD_LAUNCH_
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ================================================================================== backward
// Gradient of the pooled features w.r.t. the layer-3 output is structured:
//   G[b,p,c] = dmean[b,c]/P  (every point)  +  dmax[b,c] * [p == argmax[b,c]]
// so with  u[b,:] = dmean[b,:] * W3  (B x 256, a small GEMM done before this kernel)
//   d h2[b,p,:] = u[b,:]/P + sum_{c: argmax[b,c]=p} dmax[b,c] * W3[c,:]
//   d W3        = (dmean/P)^T * (sum_p h2[b,p,:])  +  sum_{b,c} dmax[b,c] e_c (x) h2[b,argmax[b,c],:]
// and only layers 1-2 need a dense backward.  Per cloud the kernel recomputes h1,h2 tile by
// tile (same code as the forward), turns the H2 tile into dz2 in place, and runs
//   dW2 += dz2^T * h1     (MFMA, both operands from LDS, accumulator lives in registers for
//                          the whole kernel: 256x128 over 4 waves x 128 VGPRs)
//   dh1  = dz2 * W2       (MFMA, B operand streamed from L2)  ->  dz1 = dh1 .* (1-h1^2)
//   dW1/db1 += dz1^T [x 1] (VALU, K = C+1)
// It also emits, per cloud, sum_p h2 (scaled 1/P) and the h2 rows at the argmax points
// (Hg[b,c,:]) from which two small follow-up kernels build dW3/db3.
// Tiles are 32 points: 79.5 KB of LDS per work-group => TWO work-groups per CU (two waves per
// SIMD from independent work-groups), so one group's VALU / LDS / barrier phases hide under the
// other's MFMA phases (the 64-point, one-group-per-CU version ran the MFMA pipe at 38 %).
// Work-groups are persistent over clouds (grid <= 512) so the dW2 partials stay small.

#define PN_BT 32                 // backward tile (points)
#ifndef PN_BWD_MAXG
#define PN_BWD_MAXG 512
#endif

struct PnBwdPart {          // per-work-group partial sums (floats)
    float dW2[PN_C2 * PN_C1];
    float db2[PN_C2];
    float dW1[PN_C1 * PN_MAXC];
    float db1[PN_C1];
};

// in-LDS bitonic sort of 512 int keys by 256 threads (ascending)
__device__ __forceinline__ void bitonic_sort_512(int* keys) {
#pragma unroll 1                                  // rolled: unrolling hoists 45 lane-constant address pairs
    for (int k = 2; k <= 512; k <<= 1) {          // to kernel entry and spills them (rule: recompute, don't hoist)
#pragma unroll 1
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            const int t = threadIdx.x;
            const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // lower index of the pair
            const int ixj = i | j;
            const int a = keys[i], c = keys[ixj];
            const bool up = ((i & k) == 0);
            if ((a > c) == up) {
                keys[i] = c;
                keys[ixj] = a;
            }
        }
    }
    __syncthreads();
}

// dW1 / db1 (K = C) of one finished tile: thread (c, half) over its 16 points of DZ1 / Xs
#define PN_DW1_ACCUM(XS)                                                         \
    {                                                                            \
        const int c_ = tid & 127, p0_ = (tid >> 7) * (BT / 2);                   \
        _Pragma("unroll 4") for (int p = p0_; p < p0_ + BT / 2; ++p) {           \
            const float dz = DZ1[p * PN_LD1 + c_];                               \
            const float4 x0 = *(const float4*)((XS) + p * PN_MAXC);              \
            db1acc += dz;                                                        \
            dW1acc[0] = fmaf(dz, x0.x, dW1acc[0]);                               \
            dW1acc[1] = fmaf(dz, x0.y, dW1acc[1]);                               \
            dW1acc[2] = fmaf(dz, x0.z, dW1acc[2]);                               \
            dW1acc[3] = fmaf(dz, x0.w, dW1acc[3]);                               \
            if (CT != 3 && CT != 4) {                                            \
                const float4 x1 = *(const float4*)((XS) + p * PN_MAXC + 4);      \
                dW1acc[4] = fmaf(dz, x1.x, dW1acc[4]);                           \
                dW1acc[5] = fmaf(dz, x1.y, dW1acc[5]);                           \
                dW1acc[6] = fmaf(dz, x1.z, dW1acc[6]);                           \
                dW1acc[7] = fmaf(dz, x1.w, dW1acc[7]);                           \
            }                                                                    \
        }                                                                        \
    }
// SAVED: h2 comes from HBM (written by the training forward) instead of a layer-2 recompute: one MFMA phase and
// one barrier less per tile, and the row-owner pass reads its rows straight from global memory (1 KB coalesced
// rows, issued before the tile's first barrier) -- it no longer waits for other waves' layer-2 output.
template <int CT, bool SAVED, bool TANH = true>
__global__ __launch_bounds__(256, 2) void pn_bwd_kernel(
    const float* __restrict__ x, long ldx, int B, int P, int C, int sub_mean, const float* __restrict__ W1,
    const float* __restrict__ b1, const float* __restrict__ b2, const float* __restrict__ W3,
    const float* __restrict__ packed, int max_mean, const float* __restrict__ dfeat, long ldf,
    const int32_t* __restrict__ argmax, const float* __restrict__ U, float* __restrict__ H2sum,
    float* __restrict__ Hg, int32_t* __restrict__ slotmap, PnBwdPart* __restrict__ parts,
    const float* __restrict__ h2_saved, int act) {
    constexpr int BT = PN_BT;
    __shared__ __attribute__((aligned(16))) float smem[BT * PN_LD1 * 2 + BT * PN_LD2 + 2 * BT * PN_MAXC + PN_C2 + PN_C3 +
                                                        PN_C3 + 520 + 4 * PN_C2 + 16];
    float* H1 = smem;                                   // [32][132]
    float* DZ1 = H1 + BT * PN_LD1;                      // [32][132]
    float* H2 = DZ1 + BT * PN_LD1;                      // [32][260]  h2, then dz2 in place
    float* Xs0 = H2 + BT * PN_LD2;                      // 2 x [32][8]: tile t and t+1 (double-buffered)
    float* Us = Xs0 + 2 * BT * PN_MAXC;                 // [256]  u[b,:]/P
    float* Gm = Us + PN_C2;                             // [512]  dmax[b,:]
    int* keys = (int*)(Gm + PN_C3);                     // [512]  sorted (point<<9 | channel)
    unsigned short* offs = (unsigned short*)(keys + PN_C3);   // [P/8+1 <= 1025] first key index of each wave's row block
    float* wred = (float*)(keys + PN_C3) + 520;         // [4][256] cross-wave reductions
    double* red = (double*)(wred + 4 * PN_C2);

    const int tid = threadIdx.x, lane0 = tid & 63, wave = tid >> 6;
    const int lane = lane0, li = lane0 & 31, lh = lane0 >> 5;
    const float4* P2v = (const float4*)(packed + PN_P2_OFF);
    const float4* P2Tv = (const float4*)(packed + PN_P2T_OFF);
    const float invP = 1.0f / (float)P;

    // accumulators that live for the whole kernel
    f32x16 accW2[2][4];                                  // dW2[out = wave*64+mb*32+row][in = nb*32+li]
    zero_acc<2, 4>(accW2);
    float4 db2acc = make_float4(0.f, 0.f, 0.f, 0.f);   // columns 4*lane..+3 over this wave's rows
    float dW1acc[PN_MAXC], db1acc = 0.f;                 // channel tid&127, point half tid>>7
#pragma unroll
    for (int d = 0; d < PN_MAXC; ++d) dW1acc[d] = 0.f;

    const int ntiles = P / BT;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const float* xb = x + (long)b * ldx;
        float cen[3] = {0.f, 0.f, 0.f};
        __syncthreads();
        if (sub_mean) cloud_centroid<256>(xb, P, C, red, cen);
        // ---- per-cloud setup: u/P, dmax, CSR of argmax by point ---------------------------
        Us[tid] = max_mean ? U[(long)b * PN_C2 + tid] * invP : 0.f;
        for (int c = tid; c < PN_C3; c += 256) {
            Gm[c] = dfeat[(long)b * ldf + c];
            keys[c] = (argmax[(long)b * PN_C3 + c] << 9) | c;
        }
#if !(PN_ABLATE & 16)
        bitonic_sort_512(keys);                           // by point, then channel: deterministic order
#endif
        for (int p = tid; p <= P / (BT / 4); p += 256) {   // offs[i] = #keys with point < i * (rows per wave)
            int lo = 0, hi = PN_C3;
            const int target = (p * (BT / 4)) << 9;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (keys[mid] < target) lo = mid + 1; else hi = mid;
            }
            offs[p] = (unsigned short)lo;
        }
        // slot[e] = number of distinct arg-max points before sorted entry e (prefix count over the 512 keys):
        // each distinct point's h2 row is stored ONCE (Hg[b, slot, :]) and the channels that share it find
        // it through slotmap[b, c] -- typically 3-5x less HBM traffic than one row per channel.
        {
            const int e0 = 2 * tid, e1 = e0 + 1;
            const int q0 = keys[e0] >> 9, q1 = keys[e1] >> 9, qm = (e0 > 0) ? (keys[e0 - 1] >> 9) : -1;
            const int f0 = (q0 != qm), f1 = (q1 != q0);
            int v = f0 + f1;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(v, o, 64);
                if (lane >= o) v += t;
            }
            int* wtot = (int*)wred;
            __syncthreads();                                   // every neighbour key has been read
            if (lane == 63) wtot[wave] = v;
            __syncthreads();
            int base = 0;
            for (int w2 = 0; w2 < wave; ++w2) base += wtot[w2];
            const int s1 = base + v - 1, s0 = s1 - f1;
            keys[e0] |= s0 << 22;
            keys[e1] |= s1 << 22;
            slotmap[(long)b * PN_C3 + (keys[e0] & 511)] = s0;
            slotmap[(long)b * PN_C3 + (keys[e1] & 511)] = s1;
        }
        float4 h2s = make_float4(0.f, 0.f, 0.f, 0.f);    // sum_p h2[p][4*lane..] over this wave's rows

        // Four barriers per tile: the VALU phases of neighbouring tiles are merged (layer 1 of tile t runs
        // next to dW1/db1 of tile t-1; the points of tile t+1 are staged next to the layer-2 epilogue).
        stage_points<BT, 256>(xb, 0, C, sub_mean, cen, Xs0);
        for (int tile = 0; tile < ntiles; ++tile) {
            int lane = lane0;                              // laundered per tile: see pn_fwd_kernel
            asm volatile("" : "+v"(lane));
            const int li = lane & 31, lh = lane >> 5;
            float* Xs = Xs0 + (tile & 1) * BT * PN_MAXC;
            float* Xo = Xs0 + ((tile & 1) ^ 1) * BT * PN_MAXC;
            // SAVED: the first two of this wave's 8 saved-h2 rows are requested before the barrier and layer 1, so
            // their HBM latency runs under that VALU phase
            const float* hsrc = SAVED ? h2_saved + ((long)b * P + tile * BT + wave * (BT / 4)) * PN_C2 + 4 * lane : nullptr;
            float4 h_nx = make_float4(0.f, 0.f, 0.f, 0.f), h_nx2 = h_nx;
            if (SAVED) {
                h_nx = *(const float4*)hsrc;
                h_nx2 = *(const float4*)(hsrc + PN_C2);
            }
            __syncthreads();                               // (A) Xs staged; DZ1 of tile t-1 complete; H1/H2 free
            layer1_tile<CT, BT, 256, TANH>(Xs, W1, b1, C, H1, act);
            if (tile > 0) PN_DW1_ACCUM(Xo)
            if (!SAVED) {
                __syncthreads();                           // (B)
                f32x16 acc2[1][2];
                zero_acc<1, 2>(acc2);
#if !(PN_ABLATE & 32)
                layer2_mfma<1, 2>(H1, P2v, wave, lane, acc2);
#endif
                layer2_store<1, 2, TANH>(acc2, b2, wave, lane, H2, nullptr, act);
                if (tile + 1 < ntiles) stage_points<BT, 256>(xb, tile + 1, C, sub_mean, cen, Xo);
                __syncthreads();                           // (C)
            }
            // ---- row-owner pass: wave w owns rows w*8..w*8+7; h2 -> dz2 (into the LDS tile) -------------
#if !(PN_ABLATE & 8)
            {
                constexpr int RPW = BT / 4;
                const int p0 = tile * BT + wave * RPW;
                const float4 u4 = *(const float4*)(Us + 4 * lane);
                const int e_end = offs[p0 / RPW + 1];
                int e = offs[p0 / RPW];
                // Row boundaries inside this wave's run of sorted keys (slot<<22 | point<<9 | channel) come from
                // one ballot per row over the run (lane j looks at entry e + j) -- wave-uniform, no per-point
                // table, so clouds up to 4096 points fit the same LDS; runs longer than 64 entries (rare) search.
                const int e0 = e, span = e_end - e0;
                const int pj = (lane < span) ? ((keys[e0 + lane] >> 9) & 0x1FFF) : 0x7FFFFFFF;
                int c_next = 0;
                float4 w_next = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < e_end) {
                    c_next = keys[e] & 511;
                    w_next = *(const float4*)(W3 + (long)c_next * PN_C2 + 4 * lane);
                }
                // SAVED: this wave's rows come from HBM, two rows ahead of their use (h_nx, h_nx2)
                for (int rr = 0; rr < RPW; ++rr) {
                    float* hrow = H2 + (wave * RPW + rr) * PN_LD2 + 4 * lane;
                    float4 h;
                    if (SAVED) {
                        h = h_nx;
                        h_nx = h_nx2;
                        if (rr + 2 < RPW) h_nx2 = *(const float4*)(hsrc + (rr + 2) * PN_C2);
                    } else {
                        h = *(const float4*)hrow;
                    }
                    h2s.x += h.x; h2s.y += h.y; h2s.z += h.z; h2s.w += h.w;
                    float4 S = make_float4(0.f, 0.f, 0.f, 0.f);
                    int row_end;
                    if (span <= 64) {
                        row_end = e0 + __popcll(__ballot(pj <= p0 + rr));
                    } else {
                        int lo = e, hi = e_end;
                        const int target = (p0 + rr + 1) << 9;
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            if ((keys[mid] & 0x3FFFFF) < target) lo = mid + 1; else hi = mid;
                        }
                        row_end = lo;
                    }
                    if (e < row_end)                                  // this point is some channel's arg-max
                        *(float4*)(Hg + ((long)b * PN_C3 + (keys[e] >> 22)) * PN_C2 + 4 * lane) = h;
                    for (; e < row_end; ++e) {
                        const int c = c_next;
                        const float4 w3 = w_next;
                        if (e + 1 < e_end) {
                            c_next = keys[e + 1] & 511;
                            w_next = *(const float4*)(W3 + (long)c_next * PN_C2 + 4 * lane);
                        }
                        const float g = Gm[c];
                        S.x += g * w3.x; S.y += g * w3.y; S.z += g * w3.z; S.w += g * w3.w;
                    }
                    float4 dz;
                    dz.x = (u4.x + S.x) * pn_dact<TANH>(h.x, act);
                    dz.y = (u4.y + S.y) * pn_dact<TANH>(h.y, act);
                    dz.z = (u4.z + S.z) * pn_dact<TANH>(h.z, act);
                    dz.w = (u4.w + S.w) * pn_dact<TANH>(h.w, act);
                    *(float4*)hrow = dz;
                    db2acc.x += dz.x; db2acc.y += dz.y; db2acc.z += dz.z; db2acc.w += dz.w;
                }
            }
#endif
            __syncthreads();                               // (D)
            // SAVED: the next tile's points are staged here (Xo was read by the dW1 accumulation before (D))
            if (SAVED && tile + 1 < ntiles) stage_points<BT, 256>(xb, tile + 1, C, sub_mean, cen, Xo);
            // ---- dW2 += dz2^T * h1 : K = 32 points (lanes<32: point s, lanes>=32: point 16+s) ----
            {
                const float* Ap = H2 + (lh * (BT / 2)) * PN_LD2 + wave * 64 + li;     // A[i=out][k=pt] = dz2[pt][out]
                const float* Bp = H1 + (lh * (BT / 2)) * PN_LD1 + li;                 // B[k=pt][j=in] = h1[pt][in]
                float a0p, a1p, bvp[4], a0q, a1q, bvq[4];
#define DW2_LOAD(a0, a1, bv, s_)                                   \
    a0 = Ap[(s_) * PN_LD2]; a1 = Ap[(s_) * PN_LD2 + 32];           \
    _Pragma("unroll") for (int nb = 0; nb < 4; ++nb) bv[nb] = Bp[(s_) * PN_LD1 + nb * 32];
#define DW2_MMA(a0, a1, bv)                                         \
    _Pragma("unroll") for (int nb = 0; nb < 4; ++nb) {              \
        accW2[0][nb] = MFMA(a0, bv[nb], accW2[0][nb]);              \
        accW2[1][nb] = MFMA(a1, bv[nb], accW2[1][nb]);              \
    }
                DW2_LOAD(a0p, a1p, bvp, 0)
#pragma unroll 1
                for (int s = 0; s < ((PN_ABLATE & 32) ? 0 : BT / 2); s += 2) {
                    DW2_LOAD(a0q, a1q, bvq, s + 1)
                    DW2_MMA(a0p, a1p, bvp)
                    DW2_LOAD(a0p, a1p, bvp, s + 2)      // unconditional (last trip reads rows past the tile: discarded)
                    DW2_MMA(a0q, a1q, bvq)
                }
#undef DW2_LOAD
#undef DW2_MMA
            }
            // ---- dh1 = dz2 * W2 : 32 points x this wave's 32 input channels, K = 256 ----------
            {
                f32x16 accH[1][1];
                zero_acc<1, 1>(accH);
#if !(PN_ABLATE & 32)
                mfma_stream<1, 1, 32>(H2 + li * PN_LD2 + lh * 128, PN_LD2, P2Tv + (size_t)wave * 32 * 64 + lane, accH);
#endif
                // dz1 = dh1 .* (1 - h1^2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh, col = wave * 32 + li;
                    const float h = H1[row * PN_LD1 + col];
                    DZ1[row * PN_LD1 + col] = accH[0][0][r] * pn_dact<TANH>(h, act);
                }
            }
        }
        __syncthreads();
        PN_DW1_ACCUM(Xs0 + ((ntiles - 1) & 1) * BT * PN_MAXC)           // dW1/db1 of the cloud's last tile
        // ---- per-cloud output: sum_p h2 / P (for dW3's mean term) ---------------------------
        __syncthreads();
        *(float4*)(wred + wave * PN_C2 + 4 * lane) = h2s;
        __syncthreads();
        H2sum[(long)b * PN_C2 + tid] = (wred[tid] + wred[PN_C2 + tid] + wred[2 * PN_C2 + tid] + wred[3 * PN_C2 + tid]) * invP;
    }

    // ---- write this work-group's partial sums ------------------------------------------------
    PnBwdPart* part = parts + blockIdx.x;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int out = wave * 64 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                part->dW2[out * PN_C1 + nb * 32 + li] = accW2[mb][nb][r];
            }
    __syncthreads();
    *(float4*)(wred + wave * PN_C2 + 4 * lane) = db2acc;
    __syncthreads();
    part->db2[tid] = wred[tid] + wred[PN_C2 + tid] + wred[2 * PN_C2 + tid] + wred[3 * PN_C2 + tid];
    __syncthreads();
    {   // combine the two point-halves of dW1/db1 through LDS (H1 region: 2*128*9 floats <= 32*132)
        float* t = H1;
        const int c = tid & 127, half = tid >> 7;
#pragma unroll
        for (int d = 0; d < PN_MAXC; ++d) t[(half * 128 + c) * (PN_MAXC + 1) + d] = dW1acc[d];
        t[(half * 128 + c) * (PN_MAXC + 1) + PN_MAXC] = db1acc;
        __syncthreads();
        if (half == 0) {
#pragma unroll
            for (int d = 0; d < PN_MAXC; ++d)
                part->dW1[c * PN_MAXC + d] = t[c * (PN_MAXC + 1) + d] + t[(128 + c) * (PN_MAXC + 1) + d];
            part->db1[c] = t[c * (PN_MAXC + 1) + PN_MAXC] + t[(128 + c) * (PN_MAXC + 1) + PN_MAXC];
        }
    }
}
#undef PN_DW1_ACCUM

// ---- 16-wave variant of the saved-h2 backward: ONE work-group of 1024 threads per CU ------------------------
// The 4-wave kernel above keeps the whole 256x128 dW2 accumulator in 4 waves (128 VGPRs each), which caps a CU at
// two waves per SIMD, and its VALU / memory phases wait for HBM and L2 round trips between barriers.  Here
//   * 16 waves share the dW2 accumulator (two 32x32 blocks = 32 registers each): four waves per SIMD feed the pipe;
//   * dh1 = dz2 * W2 (32 points x 128 channels) runs on v_mfma_f32_16x16x4_f32 over the full K = 256 (no K-split
//     exchange) on EIGHT of the waves, each owning 16 channels of both 16-point blocks so that one W2 fragment from L2
//     feeds eight MFMAs; it is finished in registers: dz1 = dh1 .* (1 - h1^2) feeds dW1 / db1 directly;
//   * the H1 / H2 tiles are double-buffered in LDS and the tile loop is software-pipelined by one tile: the memory
//     requests of tile t+1's VALU stage (saved h2 rows, the finished dh2 rows of its arg-max points) are issued BEFORE the
//     MFMA stage of tile t and consumed after it, so their round trips run under the MFMAs.  One barrier per 32-point tile.
// Round-2 timeline work (s_memtime stamps per wave and tile, -DPN_PROFILE + tools/pn_profile.py; 2048 clouds):
//   3.00 ms  round-1 structure: each wave walked the arg-max keys of its two points itself (one dependent L2 round trip
//            per key; a point that wins 10+ channels held all 16 waves at the tile barrier: slowest wave 6.9 k cycles of
//            VALU stage against a mean of 3.3 k)
//   2.74 ms  the per-point sums S = sum_c dmax[c] W3[c,:] come finished from pn_bwd_prep_kernel (one row load per point)
//   2.62 ms  dh1 on eight waves with shared B fragments: the loop ran at 39 cycles per 32-cycle MFMA because every
//            global_load_dwordx4 costs its SIMD ~27 issue cycles (the same with every wave reading ONE L1-resident block;
//            with the loads removed the loop runs at 32.1) -- the dW2 loop, LDS-fed, runs at 99 % of the matrix rate
// What the stamps also showed: a SIMD serves its waves strictly oldest-first, so waves 0-3 run ahead and 12-15 trail;
// that is harmless here because fp32 MFMA and VALU share the issue port and the pipe has work whenever ANY
// wave has.  Measured without effect or slower on this skeleton, all removed: the VALU stage before the MFMA stage (all
// waves, or only the younger half via a barrier placed mid-iteration), s_setprio by stage / by progress / by wave group,
// three LDS operand sets in rotation for the dW2 loop, two accumulator chains in dh1, the dh1 prologue loads hoisted
// above the dW2 loop, layer-1 weights requested before the MFMA stage.  Round 1 (kept for the record): 64-point tiles
// with the VALU results held in registers across the stage (spills, 5.6 ms).
#ifndef PN_BWD16
#define PN_BWD16 1
#endif
#ifndef PN_DH1_GROUP
#define PN_DH1_GROUP 0           // which half of the 16 waves runs the dh1 GEMM (0: waves 0-7, 1: waves 8-15)
#endif
// ---- per-cloud arg-max bookkeeping of pn_bwd16_kernel, hoisted into its own launch ---------------------------------------
// Inside the 16-wave kernel the 512-key bitonic sort is 45 barrier-separated stages in which 256 of 1024 threads work
// while the matrix pipe idles (0.2 ms of a 2.95 ms launch, tools/time_enc.py ablation).  It depends on the forward's
// arg-max only, so one small work-group per cloud does it up front -- 2048 of them fill the chip -- and the backward kernel
// just loads the results: keys (slot<<22 | point<<9 | channel, ascending), the first key of every 2-point row block
// (`offs`), and slotmap[b][c] = index of channel c's arg-max point among the cloud's distinct arg-max points.
__global__ __launch_bounds__(256) void pn_bwd_prep_kernel(const int32_t* __restrict__ argmax, int P, int32_t* __restrict__ keys_g,
                                                           unsigned short* __restrict__ offs_g, int32_t* __restrict__ slotmap,
                                                           const float* __restrict__ dfeat, long ldf,
                                                           const float* __restrict__ W3, float* __restrict__ Sg) {
    __shared__ int keys[PN_C3];
    __shared__ int wtot[4];
    __shared__ float Gm[PN_C3];                     // dmax[b,:]
    __shared__ unsigned short start[PN_C3 + 1];     // first sorted entry of every distinct arg-max point (slot)
    __shared__ int nd_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane0 = tid & 63, wave = tid >> 6;
    keys[tid] = (argmax[(long)b * PN_C3 + tid] << 9) | tid;
    keys[tid + 256] = (argmax[(long)b * PN_C3 + tid + 256] << 9) | (tid + 256);
#pragma unroll 1
    for (int k = 2; k <= 512; k <<= 1) {
#pragma unroll 1
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            const int i = ((tid & ~(j - 1)) << 1) | (tid & (j - 1));
            const int ixj = i | j;
            const int a = keys[i], c = keys[ixj];
            if ((a > c) == ((i & k) == 0)) {
                keys[i] = c;
                keys[ixj] = a;
            }
        }
    }
    __syncthreads();
    const int nblk = P / 2;                                      // row blocks of 2 points (RPW of the 16-wave kernel)
    for (int p = tid; p <= nblk; p += 256) {                     // offs[i] = #keys with point < 2 i
        int lo = 0, hi = PN_C3;
        const int target = (p * 2) << 9;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (keys[mid] < target) lo = mid + 1; else hi = mid;
        }
        offs_g[(long)b * (nblk + 1) + p] = (unsigned short)lo;
    }
    // slot[e] = number of distinct arg-max points before sorted entry e
    const int e0 = 2 * tid, e1 = e0 + 1;
    const int q0 = keys[e0] >> 9, q1 = keys[e1] >> 9, qm = (e0 > 0) ? (keys[e0 - 1] >> 9) : -1;
    const int f1 = (q1 != q0);
    int v = (q0 != qm) + f1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (lane0 >= o) v += t;
    }
    if (lane0 == 63) wtot[wave] = v;
    __syncthreads();
    int base = 0;
    for (int w2 = 0; w2 < wave; ++w2) base += wtot[w2];
    const int s1 = base + v - 1, s0 = s1 - f1;
    const int k0 = keys[e0] | (s0 << 22), k1 = keys[e1] | (s1 << 22);
    keys_g[(long)b * PN_C3 + e0] = k0;
    keys_g[(long)b * PN_C3 + e1] = k1;
    slotmap[(long)b * PN_C3 + (k0 & 511)] = s0;
    slotmap[(long)b * PN_C3 + (k1 & 511)] = s1;
    // ---- S[slot,:] = sum over the channels c that picked this point of dmax[b,c] * W3[c,:], channels ascending (the
    // order of the sorted keys).  The 16-wave kernel used to walk these runs itself, one dependent L2 round trip per key,
    // and a point that wins many channels (a few extreme points always do) held its whole work-group at the tile
    // barrier; here 2048 small work-groups walk them side by side and the big kernel loads ONE finished row per arg-max
    // point.  (All reads of keys[] above are in front of the wtot barrier.)
    Gm[tid] = dfeat[(long)b * ldf + tid];
    Gm[tid + 256] = dfeat[(long)b * ldf + tid + 256];
    keys[e0] = k0;
    keys[e1] = k1;
    if (q0 != qm) start[s0] = (unsigned short)e0;
    if (f1) start[s1] = (unsigned short)e1;
    if (tid == 255) {
        start[s1 + 1] = (unsigned short)PN_C3;
        nd_s = s1 + 1;
    }
    __syncthreads();
    const int nd = nd_s;
    for (int sl = wave; sl < nd; sl += 4) {
        const int a = start[sl], z = start[sl + 1];
        float4 S = make_float4(0.f, 0.f, 0.f, 0.f);
        int c_next = keys[a] & 511;
        float4 w_next = *(const float4*)(W3 + (long)c_next * PN_C2 + 4 * lane0);
        for (int e = a; e < z; ++e) {
            const int c = c_next;
            const float4 w3 = w_next;
            if (e + 1 < z) {
                c_next = keys[e + 1] & 511;
                w_next = *(const float4*)(W3 + (long)c_next * PN_C2 + 4 * lane0);
            }
            const float g = Gm[c];
            S.x += g * w3.x; S.y += g * w3.y; S.z += g * w3.z; S.w += g * w3.w;
        }
        *(float4*)(Sg + ((long)b * PN_C3 + sl) * PN_C2 + 4 * lane0) = S;
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#ifdef PN_PROFILE   // A/B builds only (tools/pn_profile.py): s_memtime stamps of work-groups 0-3, first cloud, tiles 0-15, every wave
__device__ unsigned long long pn_prof[4 * 16 * 16 * 8];
extern "C" int pm_debug_pn_prof_read(void* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(pn_prof), sizeof(unsigned long long) * 4 * 16 * 16 * 8) == hipSuccess ? 0 : 1;
}
#define PN_STAMP(i)                                                                                         \
    if (prof_tile >= 0 && lane0 == 0) pn_prof[((blockIdx.x * 16 + prof_tile) * 16 + wave) * 8 + (i)] = __builtin_readcyclecounter()
#else
#define PN_STAMP(i)
#endif
template <int CT, bool TANH = true>
__global__ __launch_bounds__(1024, 1) void pn_bwd16_kernel(
    const float* __restrict__ x, long ldx, int B, int P, int C, int sub_mean, const float* __restrict__ W1,
    const float* __restrict__ b1, const float* __restrict__ W3, const float* __restrict__ packed, int max_mean,
    const float* __restrict__ dfeat, long ldf, const int32_t* __restrict__ argmax, const float* __restrict__ U,
    float* __restrict__ H2sum, float* __restrict__ Hg, const int32_t* __restrict__ keys_g,
    const unsigned short* __restrict__ offs_g, PnBwdPart* __restrict__ parts, const float* __restrict__ h2_saved,
    const float* __restrict__ Sg, int act) {
    constexpr int BT = 32, NT = 1024, NW = 16, RPW = BT / NW, PPT = BT * PN_C1 / NT;
    static_assert(RPW == 2, "the row-owner pass finds its two rows' slots at the ends of the wave's key run");
    constexpr int NXC = (CT == 3 || CT == 4) ? 4 : PN_MAXC;      // point coordinates that can be non-zero
    constexpr int XSZ = BT * PN_MAXC, H1SZ = BT * PN_LD1, H2SZ = BT * PN_LD2;
    __shared__ __attribute__((aligned(16))) float smem[2 * H1SZ + 2 * H2SZ + 3 * XSZ + PN_C2 + PN_C3 + PN_C3 + 1028 + 32];
    float* H1b = smem;                                  // 2 x [32][132]  h1 of tiles t, t+1
    float* H2b = H1b + 2 * H1SZ;                        // 2 x [32][260]  dz2 of tiles t, t+1 (h2 only passes through registers);
                                                        //                end of cloud / kernel: [16][256] wave partial sums
    float* Xs0 = H2b + 2 * H2SZ;                        // 3 x [32][8]: points of tiles t, t+1, t+2
    float* Us = Xs0 + 3 * XSZ;                          // [256]  u[b,:]/P
    float* Gm = Us + PN_C2;                             // [512]  dmax[b,:]
    int* keys = (int*)(Gm + PN_C3);                     // [512]  sorted (slot<<22 | point<<9 | channel)
    unsigned short* offs = (unsigned short*)(keys + PN_C3);   // [P/2+1 <= 2049] first key of each wave's row block
    double* red = (double*)((float*)(keys + PN_C3) + 1028);   // [16]
    float* wred = H2b;

    const int tid = threadIdx.x, lane0 = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float4* P2T16 = (const float4*)(packed + PN_P2T16_OFF);
    const float invP = 1.0f / (float)P;

    const int w2_m = wave & 7, w2_n0 = (wave >> 3) * 2;  // dW2[out = w2_m*32 + row][in = (w2_n0 + j)*32 + li]
    const int hnb = wave & 7;                            // dh1: channels hnb*16.. (both 16-point blocks), by one wave group
    const bool dh_on = (wave >> 3) == PN_DH1_GROUP;
    f32x16 accW2[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW2[nb][r] = 0.f;
    float4 db2acc = make_float4(0.f, 0.f, 0.f, 0.f);   // columns 4*lane..+3 over this wave's rows
    float dW1acc[NXC], db1acc = 0.f;                     // channel hnb*16 + (lane&15) over this lane's 4 rows per tile
#pragma unroll
    for (int d = 0; d < NXC; ++d) dW1acc[d] = 0.f;

    const int ntiles = P / BT;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const float* xb = x + (long)b * ldx;
        float cen[3] = {0.f, 0.f, 0.f};
        __syncthreads();
        if (sub_mean) cloud_centroid<NT>(xb, P, C, red, cen);
        // ---- per-cloud setup: u/P, dmax; the sorted arg-max keys and their row-block offsets come from pn_bwd_prep_kernel --
        if (tid < PN_C2) Us[tid] = max_mean ? U[(long)b * PN_C2 + tid] * invP : 0.f;
        if (tid < PN_C3) keys[tid] = keys_g[(long)b * PN_C3 + tid];
        for (int p = tid; p <= P / RPW; p += NT) offs[p] = offs_g[(long)b * (P / RPW + 1) + p];
        __syncthreads();
        float4 h2s = make_float4(0.f, 0.f, 0.f, 0.f);    // sum_p h2[p][4*lane..] over this wave's rows

        // ---- VALU stage of tile tt, in two parts.  valu_issue: everything that waits on HBM / L2 is REQUESTED (this
        // wave's two saved-h2 rows, its run of sorted keys, the W3 row of the first key); valu_finish: layer 1 -> H1n and
        // the row-owner pass (wave w owns rows 2w, 2w+1) h2 -> dz2 -> H2n.  The MFMA stage of the previous tile runs
        // between the two, so the round trips are over when valu_finish starts.
        float4 hrows[RPW], srow[RPW];
        int slot[RPW];                                   // >= 0: this row's point is an arg-max (its Hg / Sg slot)
#ifdef PN_PROFILE
        int prof_tile = -1;
#endif
        auto valu_issue = [&](int tt, int tl) __attribute__((always_inline)) {
            const int lane = tl & 63;
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr)
                hrows[rr] = *(const float4*)(h2_saved + ((long)b * P + tt * BT + wave * RPW + rr) * PN_C2 + 4 * lane);
            const int p0 = tt * BT + wave * RPW;
            // the wave's key run [e, e_end) covers its two points: the first key belongs to row 0 iff its point is p0,
            // the last to row 1 iff its point is p0 + 1 (keys are sorted by point)
            const int e = __builtin_amdgcn_readfirstlane((int)offs[p0 / RPW]);
            const int e_end = __builtin_amdgcn_readfirstlane((int)offs[p0 / RPW + 1]);
            slot[0] = slot[1] = -1;
            srow[0] = srow[1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < e_end) {
                const int kf = __builtin_amdgcn_readfirstlane(keys[e]);
                const int kl = __builtin_amdgcn_readfirstlane(keys[e_end - 1]);
                if (((kf >> 9) & 0x1FFF) == p0) slot[0] = kf >> 22;
                if (((kl >> 9) & 0x1FFF) == p0 + 1) slot[1] = kl >> 22;
#pragma unroll
                for (int rr = 0; rr < RPW; ++rr)
                    if (slot[rr] >= 0) srow[rr] = *(const float4*)(Sg + ((long)b * PN_C3 + slot[rr]) * PN_C2 + 4 * lane);
            }
        };
        auto valu_finish = [&](int tt, const float* Xs, float* H1n, float* H2n, int tl) __attribute__((always_inline)) {
            const int lane = tl & 63;
            if (!(PN_ABLATE & 128)) {
                const int c = tl & 127, p0 = (tl >> 7) * PPT, CC = (CT == 3 || CT == 4) ? CT : C;
                const float b1c = b1[c];
                float w[NXC];
#pragma unroll
                for (int d = 0; d < NXC; ++d) w[d] = d < CC ? W1[c * CC + d] : 0.f;
                float z[PPT];
#pragma unroll
                for (int i = 0; i < PPT; ++i) {
                    const float4 xv = *(const float4*)(Xs + (p0 + i) * PN_MAXC);
                    float sacc = fmaf(w[0], xv.x, b1c);
                    sacc = fmaf(w[1], xv.y, sacc);
                    sacc = fmaf(w[2], xv.z, sacc);
                    if (CT != 3) sacc = fmaf(w[3], xv.w, sacc);
                    if (CT != 3 && CT != 4) {
                        const float4 x1 = *(const float4*)(Xs + (p0 + i) * PN_MAXC + 4);
                        sacc = fmaf(w[4 % NXC], x1.x, sacc); sacc = fmaf(w[5 % NXC], x1.y, sacc);
                        sacc = fmaf(w[6 % NXC], x1.z, sacc); sacc = fmaf(w[7 % NXC], x1.w, sacc);
                    }
                    z[i] = sacc;
                }
#pragma unroll
                for (int i = 0; i < PPT; i += 2) {
                    const f32x2 t = pn_act2<TANH>(z[i], z[i + 1], act);
                    H1n[(p0 + i) * PN_LD1 + c] = t.x;
                    H1n[(p0 + i + 1) * PN_LD1 + c] = t.y;
                }
            }
            if (!(PN_ABLATE & 8)) {
                const float4 u4 = *(const float4*)(Us + 4 * lane);
#pragma unroll
                for (int rr = 0; rr < RPW; ++rr) {
                    const float4 h = hrows[rr], S = srow[rr];
                    h2s.x += h.x; h2s.y += h.y; h2s.z += h.z; h2s.w += h.w;
                    if (slot[rr] >= 0)                                 // this point is some channel's arg-max
                        *(float4*)(Hg + ((long)b * PN_C3 + slot[rr]) * PN_C2 + 4 * lane) = h;
                    float4 dz;
                    dz.x = (u4.x + S.x) * pn_dact<TANH>(h.x, act);
                    dz.y = (u4.y + S.y) * pn_dact<TANH>(h.y, act);
                    dz.z = (u4.z + S.z) * pn_dact<TANH>(h.z, act);
                    dz.w = (u4.w + S.w) * pn_dact<TANH>(h.w, act);
                    *(float4*)(H2n + (wave * RPW + rr) * PN_LD2 + 4 * lane) = dz;
                    db2acc.x += dz.x; db2acc.y += dz.y; db2acc.z += dz.z; db2acc.w += dz.w;
                }
            }
        };
        // ---- MFMA stage of the tile in H1c / H2c (its points in Xs) ------------------------------------------------
        auto mfma_stage = [&](const float* H1c, const float* H2c, const float* Xs, int tl) __attribute__((always_inline)) {
            const int lane = tl & 63;
            {   // dW2 += dz2^T * h1 : K = 32 points (lanes < 32: point s, lanes >= 32: point 16 + s)
                const int li = lane & 31, lh = lane >> 5;
                const float* Ap = H2c + (lh * (BT / 2)) * PN_LD2 + w2_m * 32 + li;            // A[i=out][k=pt] = dz2[pt][out]
                const float* Bp = H1c + (lh * (BT / 2)) * PN_LD1 + w2_n0 * 32 + li;           // B[k=pt][j=in] = h1[pt][in]
                float ap, bvp[2], aq, bvq[2];
#define DW2_LOAD(a_, bv, s_) \
    a_ = Ap[(s_) * PN_LD2];  \
    _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) bv[nb] = Bp[(s_) * PN_LD1 + nb * 32];
#define DW2_MMA(a_, bv) _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) accW2[nb] = MFMA(a_, bv[nb], accW2[nb]);
                DW2_LOAD(ap, bvp, 0)
#pragma unroll 1
                for (int s = 0; s < ((PN_ABLATE & 32) ? 0 : BT / 2); s += 2) {
                    DW2_LOAD(aq, bvq, s + 1)
                    DW2_MMA(ap, bvp)
                    DW2_LOAD(ap, bvp, s + 2)            // unconditional (last trip reads one row past the half: discarded)
                    DW2_MMA(aq, bvq)
                }
#undef DW2_LOAD
#undef DW2_MMA
            }
            PN_STAMP(2);
            if (dh_on) {
                // dh1 = dz2 * W2 on 16x16x4, by HALF of the waves: wave (hnb) owns channels hnb*16.. of BOTH 16-point
                // blocks, so one B fragment from L2 feeds eight MFMAs instead of four.  (With one block per wave every
                // fragment was fetched twice and the loop ran at 39 cycles per 32-cycle MFMA: each global_load_dwordx4
                // costs its SIMD ~27 issue cycles whatever the L1 / L2 hit rate -- tools/pn_profile.py, A/B probes.)  The
                // other eight waves go straight from the dW2 loop to their VALU stage, which then runs beside these MFMAs.
                // Lane (l16 = lane&15, q = lane>>4) holds A[row l16][k = 64q + 4g + e], B[k][col l16] and the results of rows
                // 4q..4q+3 (acc0) / 16+4q..16+4q+3 (acc1), column l16.
                const int l16 = lane & 15, q = lane >> 4;
                const float* Ap = H2c + l16 * PN_LD2 + q * 64;
                const float4* Bp = P2T16 + (size_t)hnb * 16 * 64 + lane;
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                // the B stream comes from L2 (~700 clk): four groups in flight (b0..b3); A from LDS: one group ahead
                float4 a0 = *(const float4*)Ap, c0 = *(const float4*)(Ap + 16 * PN_LD2), a1, c1;
                float4 b0 = Bp[0], b1v = Bp[64], b2v = Bp[128], b3v = Bp[192];
#define DH_MMA(a_, c_, b_)                  \
    acc0 = MFMA16(a_.x, b_.x, acc0);        \
    acc1 = MFMA16(c_.x, b_.x, acc1);        \
    acc0 = MFMA16(a_.y, b_.y, acc0);        \
    acc1 = MFMA16(c_.y, b_.y, acc1);        \
    acc0 = MFMA16(a_.z, b_.z, acc0);        \
    acc1 = MFMA16(c_.z, b_.z, acc1);        \
    acc0 = MFMA16(a_.w, b_.w, acc0);        \
    acc1 = MFMA16(c_.w, b_.w, acc1);
#define DH_STEP(acur, ccur, anext, cnext, bcur, g_)                                                                      \
    anext = *(const float4*)(Ap + ((g_) + 1) * 4);                                                                       \
    cnext = *(const float4*)(Ap + 16 * PN_LD2 + ((g_) + 1) * 4);                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    DH_MMA(acur, ccur, bcur)                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    bcur = Bp[(size_t)((g_) + 4) * 64];        /* unconditional: up to 4 groups past the end (next block / tail pad) */ \
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
                for (int g = 0; g < ((PN_ABLATE & 64) ? 0 : 16); g += 4) {
                    DH_STEP(a0, c0, a1, c1, b0, g)
                    DH_STEP(a1, c1, a0, c0, b1v, g + 1)
                    DH_STEP(a0, c0, a1, c1, b2v, g + 2)
                    DH_STEP(a1, c1, a0, c0, b3v, g + 3)
                }
#undef DH_STEP
#undef DH_MMA
                PN_STAMP(3);
                // dz1 = dh1 .* (1 - h1^2); dW1 / db1 straight from the accumulator registers
                const int col = hnb * 16 + l16;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int row = (r >> 2) * 16 + 4 * q + (r & 3);
                    const float h = H1c[row * PN_LD1 + col];
                    const float dz = (r < 4 ? acc0[r & 3] : acc1[r & 3]) * pn_dact<TANH>(h, act);
                    const float4 x0 = *(const float4*)(Xs + row * PN_MAXC);
                    db1acc += dz;
                    dW1acc[0] = fmaf(dz, x0.x, dW1acc[0]);
                    dW1acc[1] = fmaf(dz, x0.y, dW1acc[1]);
                    dW1acc[2] = fmaf(dz, x0.z, dW1acc[2]);
                    dW1acc[3] = fmaf(dz, x0.w, dW1acc[3]);
                    if (CT != 3 && CT != 4) {
                        const float4 x1 = *(const float4*)(Xs + row * PN_MAXC + 4);
                        dW1acc[4 % NXC] = fmaf(dz, x1.x, dW1acc[4 % NXC]);
                        dW1acc[5 % NXC] = fmaf(dz, x1.y, dW1acc[5 % NXC]);
                        dW1acc[6 % NXC] = fmaf(dz, x1.z, dW1acc[6 % NXC]);
                        dW1acc[7 % NXC] = fmaf(dz, x1.w, dW1acc[7 % NXC]);
                    }
                }
            }
        };

        // ---- pipeline prologue: tile 0 through the VALU stage ----------------------------------------------------
        stage_points<BT, NT>(xb, 0, C, sub_mean, cen, Xs0);
        if (ntiles > 1) stage_points<BT, NT>(xb, 1, C, sub_mean, cen, Xs0 + XSZ);
        __syncthreads();
        {
            int tl = tid;
            asm volatile("" : "+v"(tl));
            valu_issue(0, tl);
            valu_finish(0, Xs0, H1b, H2b, tl);
        }
        __syncthreads();
        int ix = 0;                                        // Xs buffer of tile t (t+1: ix+1, t+2: ix+2, mod 3)
        for (int tile = 0; tile < ntiles; ++tile) {
            int tl = tid;                                  // laundered per tile: recompute addresses, don't hoist
            asm volatile("" : "+v"(tl));                   // (the stages derive lane / channel / point indices from it)
            const int ix1 = ix == 2 ? 0 : ix + 1, ix2 = ix1 == 2 ? 0 : ix1 + 1, cur = tile & 1;
            const bool more = tile + 1 < ntiles;
            // the points of tile t+2 are requested here as well (threads 0-255, one coordinate each) and only stored to
            // LDS at the end of the interval: a load -> ds_write right before the barrier exposed one HBM round trip
            // per tile (0.2 ms per launch)
            float xnext = 0.f;
            const bool stage2 = tile + 2 < ntiles && tl < XSZ;
            if (stage2 && (tl & 7) < C) {
                xnext = xb[((tile + 2) * BT + (tl >> 3)) * C + (tl & 7)];
                if (sub_mean && (tl & 7) < 3) xnext -= cen[tl & 7];
            }
#ifdef PN_PROFILE
            prof_tile = (blockIdx.x < 4 && b == (int)blockIdx.x && tile < 16) ? tile : -1;
#endif
            PN_STAMP(0);
            if (more) valu_issue(tile + 1, tl);
            PN_STAMP(1);
            mfma_stage(H1b + cur * H1SZ, H2b + cur * H2SZ, Xs0 + ix * XSZ, tl);
            PN_STAMP(4);
            if (more) valu_finish(tile + 1, Xs0 + ix1 * XSZ, H1b + (cur ^ 1) * H1SZ, H2b + (cur ^ 1) * H2SZ, tl);
            if (stage2) Xs0[ix2 * XSZ + tl] = xnext;
            PN_STAMP(5);
            __syncthreads();                               // tile t's buffers are free, tile t+1's complete
            PN_STAMP(6);
            ix = ix1;
        }
        *(float4*)(wred + wave * PN_C2 + 4 * lane0) = h2s;
        __syncthreads();
        if (tid < PN_C2) {
            float sum = 0.f;
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) sum += wred[w2 * PN_C2 + tid];
            H2sum[(long)b * PN_C2 + tid] = sum * invP;
        }
    }

    // ---- write this work-group's partial sums ------------------------------------------------
    PnBwdPart* part = parts + blockIdx.x;
    {
        const int li = lane0 & 31, lh = lane0 >> 5;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int out = w2_m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                part->dW2[out * PN_C1 + (w2_n0 + nb) * 32 + li] = accW2[nb][r];
            }
    }
    __syncthreads();
    *(float4*)(wred + wave * PN_C2 + 4 * lane0) = db2acc;
    float* t1 = H2b + NW * PN_C2;                       // dW1/db1: the 8 row subsets (point half, lane quarter) of a channel
    {                                                   // meet in LDS: [8][128][9] floats behind the wave partials
        const int c = hnb * 16 + (lane0 & 15), part_i = (wave >> 3) * 4 + (lane0 >> 4);   // the idle group adds zeros
#pragma unroll
        for (int d = 0; d < PN_MAXC; ++d) t1[(part_i * 128 + c) * (PN_MAXC + 1) + d] = d < NXC ? dW1acc[d % NXC] : 0.f;
        t1[(part_i * 128 + c) * (PN_MAXC + 1) + PN_MAXC] = db1acc;
    }
    __syncthreads();
    if (tid < PN_C2) {
        float sum = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) sum += wred[w2 * PN_C2 + tid];
        part->db2[tid] = sum;
    }
    if (tid < PN_C1) {
#pragma unroll
        for (int d = 0; d <= PN_MAXC; ++d) {
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) sum += t1[(q * 128 + tid) * (PN_MAXC + 1) + d];
            if (d < PN_MAXC) part->dW1[tid * PN_MAXC + d] = sum;
            else part->db1[tid] = sum;
        }
    }
}

// sum the per-work-group partials in fixed order, two stages (68 MB of partials: a single pass with one
// thread per element and 512 dependent-latency loads took 127 us; 16-way split + final takes ~25 us)
#define PN_RED_SPLIT 16
__global__ __launch_bounds__(256) void pn_bwd_reduce1_kernel(const PnBwdPart* __restrict__ parts, int G,
                                                              float* __restrict__ tmp) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = (int)(sizeof(PnBwdPart) / sizeof(float));
    if (i >= n) return;
    const int per = (G + PN_RED_SPLIT - 1) / PN_RED_SPLIT;
    const int g0 = blockIdx.y * per, g1 = min(G, g0 + per);
    const float* base = (const float*)parts;
    float s = 0.f;
#pragma unroll 8
    for (int g = g0; g < g1; ++g) s += base[(size_t)g * n + i];
    tmp[(size_t)blockIdx.y * n + i] = s;
}

__global__ __launch_bounds__(256) void pn_bwd_reduce2_kernel(const float* __restrict__ tmp, int C,
                                                              float* __restrict__ dW1, float* __restrict__ db1,
                                                              float* __restrict__ dW2, float* __restrict__ db2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = (int)(sizeof(PnBwdPart) / sizeof(float));
    if (i >= n) return;
    float s = 0.f;
#pragma unroll
    for (int y = 0; y < PN_RED_SPLIT; ++y) s += tmp[(size_t)y * n + i];
    const int o_db2 = PN_C2 * PN_C1, o_dW1 = o_db2 + PN_C2, o_db1 = o_dW1 + PN_C1 * PN_MAXC;
    if (i < o_db2) dW2[i] = s;
    else if (i < o_dW1) db2[i - o_db2] = s;
    else if (i < o_db1) {
        const int j = i - o_dW1, c = j / PN_MAXC, d = j % PN_MAXC;
        if (d < C) dW1[c * C + d] = s;
    } else db1[i - o_db1] = s;
}

// dW3[c,:] (+)= sum_b dmax[b,c] * Hg[b, slot(b,c), :] ;  db3[c] = sum_b (dmax[b,c] + dmean[b,c])
// One 1 KB row read per (cloud, channel) = 1 GB for 2048 clouds although only ~250 MB of distinct rows exist (several
// channels share an arg-max point), so the kernel is laid out for L2 reuse: the clouds are split into PN_DW3_SPLIT
// ranges with the range index as blockIdx.x -- consecutive work-groups go round-robin to the 8 XCDs, so a range (and
// its rows) stays on ONE XCD's L2 -- and each work-group handles PN_DW3_CPB channels so that the whole grid is
// resident at once and the groups of a range walk its clouds together.  Partials are added in fixed order by
// pn_dw3_finish_kernel.  (One work-group per channel over all clouds: 0.31 ms; this layout: see DESIGN.md 3.2.)
#define PN_DW3_SPLIT 16
#define PN_DW3_CPB 4
__global__ __launch_bounds__(256) void pn_dw3_gather_kernel(const float* __restrict__ dfeat, long ldf, int B,
                                                             const float* __restrict__ Hg,
                                                             const int32_t* __restrict__ slotmap,
                                                             float* __restrict__ tmp) {
    const int c0 = blockIdx.y * PN_DW3_CPB, k = threadIdx.x;
    const int per = (B + PN_DW3_SPLIT - 1) / PN_DW3_SPLIT;
    const int b0 = blockIdx.x * per, b1 = min(B, b0 + per);
    float acc[PN_DW3_CPB];
#pragma unroll
    for (int j = 0; j < PN_DW3_CPB; ++j) acc[j] = 0.f;
#pragma unroll 4
    for (int b = b0; b < b1; ++b) {
        const int4 sl = *(const int4*)(slotmap + (long)b * PN_C3 + c0);
        const float* gp = dfeat + (long)b * ldf + c0;       // (no alignment assumption on the caller's gradient rows)
        const float4 g = make_float4(gp[0], gp[1], gp[2], gp[3]);
        const float* rows = Hg + (long)b * PN_C3 * PN_C2 + k;
        acc[0] += g.x * rows[(long)sl.x * PN_C2];
        acc[1] += g.y * rows[(long)sl.y * PN_C2];
        acc[2] += g.z * rows[(long)sl.z * PN_C2];
        acc[3] += g.w * rows[(long)sl.w * PN_C2];
    }
#pragma unroll
    for (int j = 0; j < PN_DW3_CPB; ++j) tmp[((size_t)blockIdx.x * PN_C3 + c0 + j) * PN_C2 + k] = acc[j];
}
__global__ __launch_bounds__(256) void pn_dw3_finish_kernel(const float* __restrict__ dfeat, long ldf, int B,
                                                             int max_mean, const float* __restrict__ tmp,
                                                             float* __restrict__ dW3, float* __restrict__ db3) {
    __shared__ float red[4];
    const int c = blockIdx.x, k = threadIdx.x;
    float acc = 0.f;
#pragma unroll
    for (int y = 0; y < PN_DW3_SPLIT; ++y) acc += tmp[((size_t)y * PN_C3 + c) * PN_C2 + k];
    dW3[c * PN_C2 + k] = (max_mean ? dW3[c * PN_C2 + k] : 0.f) + acc;
    float s = 0.f;
    for (int b = k; b < B; b += 256) s += dfeat[(long)b * ldf + c] + (max_mean ? dfeat[(long)b * ldf + PN_C3 + c] : 0.f);
    s = block_sum<float, 256>(s, red);
    if (k == 0) db3[c] = s;
}

#include "pointnet_enc_bwd_bf6.h"

static inline int pn_bwd_grid(int B) { return B < PN_BWD_MAXG ? B : PN_BWD_MAXG; }
static int pn_cu_count() { const int n = pm_cu_count(); return n > PN_BWD_MAXG ? PN_BWD_MAXG : n; }
static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct PnBwdWs {
    size_t off_U, off_H2sum, off_Hg, off_slot, off_parts, off_dw3, off_gemm, off_keys, off_offs, off_Sg, total;
};
static PnBwdWs pn_bwd_layout(int B) {
    PnBwdWs w;
    size_t o = 0;
    w.off_U = o;      o += align256((size_t)B * PN_C2 * 4);
    w.off_H2sum = o;  o += align256((size_t)B * PN_C2 * 4);
    w.off_Hg = o;     o += align256((size_t)B * PN_C3 * PN_C2 * 4);      // worst case: every channel its own point
    w.off_slot = o;   o += align256((size_t)B * PN_C3 * 4);
    w.off_parts = o;  o += align256((size_t)(pn_bwd_grid(B) + PN_RED_SPLIT) * sizeof(PnBwdPart));
    w.off_dw3 = o;    o += align256((size_t)PN_DW3_SPLIT * PN_C3 * PN_C2 * 4);
    w.off_gemm = o;   o += align256(pm_linear_bwd_weight_workspace_bytes(B, PN_C3, PN_C2));
    w.off_keys = o;   o += align256((size_t)B * PN_C3 * 4);               // pn_bwd_prep_kernel: sorted arg-max keys
    w.off_offs = o;   o += align256((size_t)B * (4096 / 2 + 1) * 2);      // ... and their row-block offsets (P <= 4096)
    w.off_Sg = o;     o += align256((size_t)B * PN_C3 * PN_C2 * 4);       // ... and the dh2 contribution of every arg-max point
    w.total = o;
    return w;
}

extern "C" size_t pm_pointnet_enc_bwd_workspace_bytes(int B, int P, int C) {
    (void)P; (void)C;
    return B > 0 ? pn_bwd_layout(B).total : 0;
}

static int pn_enc_bwd_impl(const float* x, long ldx, int B, int P, int C, int sub_mean,
                           const float* W1, const float* b1, const float* b2, const float* W3,
                           const float* packed, int max_mean, const float* dfeat, long ldf,
                           const int32_t* argmax, float* dW1, float* db1, float* dW2, float* db2,
                           float* dW3, float* db3, const float* h2_saved, int act, void* workspace,
                           size_t workspace_bytes, void* stream, const unsigned short* packW2_bf6) {
    PM_REQUIRE(x && W1 && b1 && b2 && W3 && packed && dfeat && argmax && dW1 && db1 && dW2 && db2 && dW3 && db3);
    PM_REQUIRE(act > PM_ACT_NONE && act <= PM_ACT_MAX);
    // the split-bf16 kernel is the saved-layer-2, tanh form only: anything else is refused, not silently run in fp32
    PM_REQUIRE(!packW2_bf6 || (h2_saved && act == PM_ACT_TANH));
    if (packW2_bf6 && ((uintptr_t)packW2_bf6 & 15) != 0) return PM_EALIGN;
    if (h2_saved && ((uintptr_t)h2_saved & 15) != 0) return PM_EALIGN;
    PM_REQUIRE(B > 0 && P > 0 && P % PN_TM == 0 && P <= 4096 && C >= 1 && C <= PN_MAXC && ldx >= (long)P * C);
    PM_REQUIRE(ldf >= PN_C3 * (max_mean ? 2 : 1));
    PM_REQUIRE(!sub_mean || C >= 3);
    if (((uintptr_t)packed & 15) != 0 || ((uintptr_t)W3 & 15) != 0 || ((uintptr_t)workspace & 255) != 0) return PM_EALIGN;
    const PnBwdWs w = pn_bwd_layout(B);
    if (!workspace || workspace_bytes < w.total) return PM_EWORKSPACE;
    char* ws = (char*)workspace;
    float* U = (float*)(ws + w.off_U);
    float* H2sum = (float*)(ws + w.off_H2sum);
    float* Hg = (float*)(ws + w.off_Hg);
    int32_t* slotmap = (int32_t*)(ws + w.off_slot);
    PnBwdPart* parts = (PnBwdPart*)(ws + w.off_parts);
    int rc;
    if (max_mean) {   // U[B,256] = dmean[B,512] * W3[512,256]
        rc = pm_linear_bwd_data_f32(dfeat + PN_C3, ldf, W3, PN_C2, nullptr, 0, U, PN_C2, B, PN_C3, PN_C2, PM_ACT_NONE,
                                    stream);
        if (rc != PM_OK) return rc;
    }
    int G = pn_bwd_grid(B);
    if (packW2_bf6) {                             // saved layer 2, both GEMMs on split-bf16 MFMAs: one 8-wave work-group per CU
        const int ncu = pn_cu_count();
        G = B < ncu ? B : ncu;
        int32_t* keys_g = (int32_t*)(ws + w.off_keys);
        unsigned short* offs_g = (unsigned short*)(ws + w.off_offs);
        float* Sg = (float*)(ws + w.off_Sg);
        hipLaunchKernelGGL(pn_bwd_prep_kernel, dim3(B), dim3(256), 0, pm_stream(stream), argmax, P, keys_g, offs_g, slotmap,
                           dfeat, ldf, W3, Sg);
#define PN_BF6_LAUNCH(CT)                                                                                               \
    hipLaunchKernelGGL((pn_bwd_bf6_kernel<CT>), dim3(G), dim3(PB6_NT), 0, pm_stream(stream), x, ldx, B, P, C, sub_mean, W1, b1, \
                       packW2_bf6, max_mean, U, H2sum, Hg, (const int32_t*)keys_g, (const unsigned short*)offs_g, parts,  \
                       h2_saved, (const float*)Sg)
        if (C == 3) PN_BF6_LAUNCH(3);
        else if (C == 4) PN_BF6_LAUNCH(4);
        else PN_BF6_LAUNCH(0);
#undef PN_BF6_LAUNCH
    } else if (PN_BWD16 && h2_saved) {            // saved layer 2: one 16-wave work-group per CU
        const int ncu = pn_cu_count();
        G = B < ncu ? B : ncu;
        int32_t* keys_g = (int32_t*)(ws + w.off_keys);
        unsigned short* offs_g = (unsigned short*)(ws + w.off_offs);
        float* Sg = (float*)(ws + w.off_Sg);
        hipLaunchKernelGGL(pn_bwd_prep_kernel, dim3(B), dim3(256), 0, pm_stream(stream), argmax, P, keys_g, offs_g, slotmap,
                           dfeat, ldf, W3, Sg);
#define PN_BWD16_LAUNCH_(CT, TH)                                                                                       \
    hipLaunchKernelGGL((pn_bwd16_kernel<CT, TH>), dim3(G), dim3(1024), 0, pm_stream(stream), x, ldx, B, P, C, sub_mean, W1, b1, \
                       W3, packed, max_mean, dfeat, ldf, argmax, U, H2sum, Hg, (const int32_t*)keys_g,                    \
                       (const unsigned short*)offs_g, parts, h2_saved, (const float*)Sg, act)
#define PN_BWD16_LAUNCH(CT)                          \
    do {                                             \
        if (act == PM_ACT_TANH) PN_BWD16_LAUNCH_(CT, true); \
        else PN_BWD16_LAUNCH_(CT, false);            \
    } while (0)
        if (C == 3) PN_BWD16_LAUNCH(3);
        else if (C == 4) PN_BWD16_LAUNCH(4);
        else PN_BWD16_LAUNCH(0);
#undef PN_BWD16_LAUNCH
#undef PN_BWD16_LAUNCH_
    } else {
#define PN_BWD_LAUNCH_(CT, SV, TH)                                                                               \
    hipLaunchKernelGGL((pn_bwd_kernel<CT, SV, TH>), dim3(G), dim3(256), 0, pm_stream(stream), x, ldx, B, P, C, sub_mean, W1, \
                       b1, b2, W3, packed, max_mean, dfeat, ldf, argmax, U, H2sum, Hg, slotmap, parts, h2_saved, act)
#define PN_BWD_LAUNCH(CT)                                                      \
    do {                                                                       \
        if (act == PM_ACT_TANH) {                                              \
            if (h2_saved) PN_BWD_LAUNCH_(CT, true, true);                      \
            else PN_BWD_LAUNCH_(CT, false, true);                              \
        } else {                                                               \
            if (h2_saved) PN_BWD_LAUNCH_(CT, true, false);                     \
            else PN_BWD_LAUNCH_(CT, false, false);                             \
        }                                                                      \
    } while (0)
    if (C == 3) PN_BWD_LAUNCH(3);
    else if (C == 4) PN_BWD_LAUNCH(4);
    else PN_BWD_LAUNCH(0);
#undef PN_BWD_LAUNCH
#undef PN_BWD_LAUNCH_
    }
    const int n = (int)(sizeof(PnBwdPart) / sizeof(float));
    float* red_tmp = (float*)(parts + G);
    hipLaunchKernelGGL(pn_bwd_reduce1_kernel, dim3((n + 255) / 256, PN_RED_SPLIT), dim3(256), 0, pm_stream(stream), parts,
                       G, red_tmp);
    hipLaunchKernelGGL(pn_bwd_reduce2_kernel, dim3((n + 255) / 256), dim3(256), 0, pm_stream(stream), red_tmp, C, dW1,
                       db1, dW2, db2);
    if (max_mean) {   // dW3 = dmean^T * (sum_p h2 / P)
        rc = pm_linear_bwd_weight_f32(dfeat + PN_C3, ldf, H2sum, PN_C2, dW3, PN_C2, nullptr, B, PN_C3, PN_C2,
                                      ws + w.off_gemm, w.total - w.off_gemm, stream);
        if (rc != PM_OK) return rc;
    }
    float* dw3_tmp = (float*)(ws + w.off_dw3);
    hipLaunchKernelGGL(pn_dw3_gather_kernel, dim3(PN_DW3_SPLIT, PN_C3 / PN_DW3_CPB), dim3(256), 0, pm_stream(stream), dfeat,
                       ldf, B, Hg, slotmap, dw3_tmp);
    hipLaunchKernelGGL(pn_dw3_finish_kernel, dim3(PN_C3), dim3(256), 0, pm_stream(stream), dfeat, ldf, B, max_mean,
                       dw3_tmp, dW3, db3);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

extern "C" int pm_pointnet_enc_bwd_f32(const float* x, long ldx, int B, int P, int C, int sub_mean,
                                       const float* W1, const float* b1, const float* b2, const float* W3,
                                       const float* packed, int max_mean, const float* dfeat, long ldf,
                                       const int32_t* argmax, float* dW1, float* db1, float* dW2, float* db2,
                                       float* dW3, float* db3, const float* h2_saved, int act, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    return pn_enc_bwd_impl(x, ldx, B, P, C, sub_mean, W1, b1, b2, W3, packed, max_mean, dfeat, ldf, argmax, dW1, db1, dW2, db2,
                           dW3, db3, h2_saved, act, workspace, workspace_bytes, stream, nullptr);
}

// the same call with dW2 / dh1 on split-bf16 MFMAs (three planes, six products: pointnet_enc_bwd_bf6.h); tanh, saved layer 2
extern "C" int pm_pointnet_enc_bwd_bf6(const float* x, long ldx, int B, int P, int C, int sub_mean,
                                       const float* W1, const float* b1, const float* b2, const float* W3,
                                       const float* packed, const void* packed_w2_bf6, int max_mean, const float* dfeat,
                                       long ldf, const int32_t* argmax, float* dW1, float* db1, float* dW2, float* db2,
                                       float* dW3, float* db3, const float* h2_saved, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    PM_REQUIRE(packed_w2_bf6);
    return pn_enc_bwd_impl(x, ldx, B, P, C, sub_mean, W1, b1, b2, W3, packed, max_mean, dfeat, ldf, argmax, dW1, db1, dW2, db2,
                           dW3, db3, h2_saved, PM_ACT_TANH, workspace, workspace_bytes, stream,
                           (const unsigned short*)packed_w2_bf6);
}
