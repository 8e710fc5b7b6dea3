// OPT-IN variant of the fused PointNet encoder forward (network.py:147-153,172-181) that evaluates the
// two big per-point GEMMs with SPLIT-bf16 MFMAs: every fp32 operand x is split into hi = bf16(x) and
// lo = bf16(x - hi) and a*b is evaluated as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation (the dropped a_lo*b_lo term is <= 2^-16 relative).
// Three bf16 MFMAs (3 x 32 cycles per 32x32x16) replace eight fp32 MFMAs (8 x 64 cycles) -- 5.3x less
// matrix-pipe time -- at ~1e-5 relative accuracy instead of ~1e-7.  The default path stays exact fp32
// (pointnet_enc.hip); this kernel is selected with `net_cfg['precision'] = 'bf16x3'` and its results are
// checked against the oracle at a stated 1e-4 tolerance (tests/test_gpu_learner.py).
//
// Dataflow: one work-group (8 waves) owns one cloud and walks it in tiles of 128 points (the bf16 MFMAs
// need 5x more operand bytes per cycle than the fp32 ones; 128-point tiles halve the L2 weight traffic
// per point):  layer 1 (VALU) -> H1 hi/lo planes [128][136] bf16 in LDS;  layer 2: wave w -> channels
// [32w,32w+32), 4x1 accumulators, A from LDS (ds_read_b128: 8 consecutive k per lane), B from the packed
// hi/lo weight planes in L2;  tanh + split -> H2 hi/lo planes [128][264] (aliasing H1);  layer 3: wave w
// -> channels [64w,64w+64), 4x2 accumulators;  pooling as in the fp32 kernel (rows = points).
// LDS: 135 KB + 4 KB => one work-group per CU, two waves per SIMD.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define B3_TM 128
#define B3_NT 512
#define B3_LD1 (128 + 8)     // halfwords; 272 B rows: 16 rows hit 16 distinct 16-B slots of the 256-B bank row
#define B3_LD2 (256 + 8)
#define B3_MAXC 8
#define B3_C3 512

// packed planes (halfwords): P2h | P2l | P3h | P3l | pad
#define B3_P2 (8 * 8 * 64 * 8)          // [nb 8][step 8][lane 64][8]
#define B3_P3 (16 * 16 * 64 * 8)        // [nb 16][step 16][lane 64][8]
#define B3_OFF_P2H 0
#define B3_OFF_P2L (B3_P2)
#define B3_OFF_P3H (2 * B3_P2)
#define B3_OFF_P3L (2 * B3_P2 + B3_P3)
#define B3_PACKED_HALFS (2 * B3_P2 + 2 * B3_P3 + 4096)

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {        // {bf16(a) | bf16(b) << 16}, RNE
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// x -> (hi, lo) bf16 bit patterns with hi + lo == x to ~2^-17 relative
__device__ __forceinline__ void split_bf16(float x, unsigned short& hi, unsigned short& lo) {
    const unsigned h = cvt_pk_bf16(x, 0.f) & 0xffffu;
    const float xh = __uint_as_float(h << 16);
    const unsigned l = cvt_pk_bf16(x - xh, 0.f) & 0xffffu;
    hi = (unsigned short)h;
    lo = (unsigned short)l;
}

extern "C" size_t pm_pointnet_packed_bf3_bytes(void) { return (size_t)B3_PACKED_HALFS * 2; }

__global__ __launch_bounds__(256) void pn_pack_bf3_kernel(const float* __restrict__ W2, const float* __restrict__ W3,
                                                           unsigned short* __restrict__ packed) {
    const int i = blockIdx.x * 256 + threadIdx.x;                       // one (nb, step, lane, e) slot of P2 or P3
    if (i >= B3_P2 + B3_P3) {
        if (i < B3_P2 + B3_P3 + 4096) packed[2 * B3_P2 + 2 * B3_P3 + (i - B3_P2 - B3_P3)] = 0;
        return;
    }
    const int e = i & 7, lane = (i >> 3) & 63, li = lane & 31, lq = lane >> 5;
    float w;
    int oh, ol;
    if (i < B3_P2) {
        const int step = (i >> 9) & 7, nb = i >> 12;
        w = W2[(nb * 32 + li) * 128 + step * 16 + lq * 8 + e];
        oh = B3_OFF_P2H + i;
        ol = B3_OFF_P2L + i;
    } else {
        const int j = i - B3_P2, step = (j >> 9) & 15, nb = j >> 13;
        w = W3[(nb * 32 + li) * 256 + step * 16 + lq * 8 + e];
        oh = B3_OFF_P3H + j;
        ol = B3_OFF_P3L + j;
    }
    unsigned short h, l;
    split_bf16(w, h, l);
    packed[oh] = h;
    packed[ol] = l;
}

extern "C" int pm_pointnet_pack_weights_bf3(const float* W2, const float* W3, void* packed, void* stream) {
    PM_REQUIRE(W2 && W3 && packed);
    const int n = B3_P2 + B3_P3 + 4096;
    hipLaunchKernelGGL(pn_pack_bf3_kernel, dim3((n + 255) / 256), dim3(256), 0, pm_stream(stream), W2, W3,
                       (unsigned short*)packed);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

#define MFMA_BF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ bf16x8 as_bf(const uint4& v) { return *(const bf16x8*)&v; }

// acc[mb][nb] += A(MB*32 rows, hi/lo planes in LDS, row stride lda halfwords) * B(packed hi/lo planes)
// over NS k-steps of 16.  Ah/Al point at this lane's row li, k-offset lq*8; Bh/Bl at this wave's first
// N-block and this lane.  Next step's B operands are fetched before this step's MFMAs (named ping/pong sets,
// pinned with sched_barrier as in pointnet_enc.hip); A comes from LDS at the top of the step.
template <int MB, int NB, int NS>
__device__ __forceinline__ void bf3_stream(const unsigned short* __restrict__ Ah, const unsigned short* __restrict__ Al,
                                           int lda, const uint4* __restrict__ Bh, const uint4* __restrict__ Bl,
                                           f32x16 (&acc)[MB][NB]) {
    uint4 bh0[NB], bl0[NB], bh1[NB], bl1[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        bh0[nb] = Bh[(size_t)(nb * NS) * 64];
        bl0[nb] = Bl[(size_t)(nb * NS) * 64];
    }
#define B3_STEP(S_, BHC, BLC, BHN, BLN)                                                        \
    {                                                                                          \
        uint4 ah[MB], al[MB];                                                                  \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb) {                                    \
            ah[mb] = *(const uint4*)(Ah + mb * 32 * lda + (S_) * 16);                          \
            al[mb] = *(const uint4*)(Al + mb * 32 * lda + (S_) * 16);                          \
        }                                                                                      \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb) {   /* next step (over-reads one step at the end: padded) */ \
            BHN[nb] = Bh[(size_t)(nb * NS + (S_) + 1) * 64];                                   \
            BLN[nb] = Bl[(size_t)(nb * NS + (S_) + 1) * 64];                                   \
        }                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                      \
            _Pragma("unroll") for (int mb = 0; mb < MB; ++mb) {                                \
                acc[mb][nb] = MFMA_BF(as_bf(ah[mb]), as_bf(BHC[nb]), acc[mb][nb]);             \
                acc[mb][nb] = MFMA_BF(as_bf(ah[mb]), as_bf(BLC[nb]), acc[mb][nb]);             \
                acc[mb][nb] = MFMA_BF(as_bf(al[mb]), as_bf(BHC[nb]), acc[mb][nb]);             \
            }                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    }
#pragma unroll 1
    for (int s = 0; s < NS; s += 2) {
        B3_STEP(s, bh0, bl0, bh1, bl1)
        B3_STEP(s + 1, bh1, bl1, bh0, bl0)
    }
#undef B3_STEP
}

template <int CT>
__global__ __launch_bounds__(B3_NT, 2) void pn_fwd_bf3_kernel(const float* __restrict__ x, long ldx, int P, int C,
                                                               int sub_mean, const float* __restrict__ W1,
                                                               const float* __restrict__ b1,
                                                               const float* __restrict__ b2,
                                                               const float* __restrict__ b3,
                                                               const unsigned short* __restrict__ packed, int max_mean,
                                                               float* __restrict__ feat, long ldf,
                                                               int32_t* __restrict__ argmax,
                                                               float* __restrict__ h2_save) {
    __shared__ __attribute__((aligned(16))) unsigned short Hs[2 * B3_TM * B3_LD2];
    __shared__ __attribute__((aligned(16))) float Xs[B3_TM * B3_MAXC];
    __shared__ double red[16];
    unsigned short* H1h = Hs;
    unsigned short* H1l = Hs + B3_TM * B3_LD1;
    unsigned short* H2h = Hs;
    unsigned short* H2l = Hs + B3_TM * B3_LD2;

    const int b = blockIdx.x, tid = threadIdx.x, lane0 = tid & 63, wave = tid >> 6;
    const float* xb = x + (long)b * ldx;
    const uint4* P2h = (const uint4*)(packed + B3_OFF_P2H);
    const uint4* P2l = (const uint4*)(packed + B3_OFF_P2L);
    const uint4* P3h = (const uint4*)(packed + B3_OFF_P3H);
    const uint4* P3l = (const uint4*)(packed + B3_OFF_P3L);

    float cen[3] = {0.f, 0.f, 0.f};
    if (sub_mean) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        for (int p = tid; p < P; p += B3_NT) {
            s0 += (double)xb[p * C];
            s1 += (double)xb[p * C + 1];
            s2 += (double)xb[p * C + 2];
        }
        s0 = block_sum<double, B3_NT>(s0, red);
        s1 = block_sum<double, B3_NT>(s1, red);
        s2 = block_sum<double, B3_NT>(s2, red);
        cen[0] = (float)(s0 / P);
        cen[1] = (float)(s1 / P);
        cen[2] = (float)(s2 / P);
    }

    float vmax[2], vsum[2];
    int imax[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        vmax[nb] = -INFINITY;
        vsum[nb] = 0.f;
        imax[nb] = 0;
    }

    const int ntiles = P / B3_TM;
    for (int tile = 0; tile < ntiles; ++tile) {
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int li = lane & 31, lq = lane >> 5;
        __syncthreads();                                   // previous tile's layer-3 reads of Hs are done
        for (int i = tid; i < B3_TM * B3_MAXC; i += B3_NT) {
            const int p = i >> 3, d = i & 7;
            float v = 0.f;
            if (d < C) {
                v = xb[(tile * B3_TM + p) * C + d];
                if (sub_mean && d < 3) v -= cen[d];
            }
            Xs[i] = v;
        }
        __syncthreads();
        {   // layer 1: thread (c = tid&127, part = tid>>7) -> 32 points; write hi/lo planes
            const int c = tid & 127, p0 = (tid >> 7) * 32;
            float w[B3_MAXC];
#pragma unroll
            for (int d = 0; d < B3_MAXC; ++d) w[d] = (d < C) ? W1[c * C + d] : 0.f;
            const float b1c = b1[c];
#pragma unroll 4
            for (int p = p0; p < p0 + 32; ++p) {
                const float4 x0 = *(const float4*)(Xs + p * B3_MAXC);
                float s = fmaf(w[0], x0.x, b1c);
                s = fmaf(w[1], x0.y, s);
                s = fmaf(w[2], x0.z, s);
                s = fmaf(w[3], x0.w, s);
                if (CT != 3 && CT != 4) {
                    const float4 x1 = *(const float4*)(Xs + p * B3_MAXC + 4);
                    s = fmaf(w[4], x1.x, s); s = fmaf(w[5], x1.y, s); s = fmaf(w[6], x1.z, s); s = fmaf(w[7], x1.w, s);
                }
                unsigned short h, l;
                split_bf16(pm_tanh(s), h, l);
                H1h[p * B3_LD1 + c] = h;
                H1l[p * B3_LD1 + c] = l;
            }
        }
        __syncthreads();
        {   // layer 2: 128 points x channels [32w, 32w+32), K = 128
            f32x16 acc2[4][1];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[mb][0][r] = 0.f;
            bf3_stream<4, 1, 8>(H1h + li * B3_LD1 + lq * 8, H1l + li * B3_LD1 + lq * 8, B3_LD1,
                                P2h + (size_t)(wave * 8) * 64 + lane, P2l + (size_t)(wave * 8) * 64 + lane, acc2);
            __syncthreads();                               // every wave has finished reading H1
            const float b2c = b2[wave * 32 + li];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lq;
                    unsigned short h, l;
                    const float v = pm_tanh(acc2[mb][0][r] + b2c);
                    split_bf16(v, h, l);
                    H2h[row * B3_LD2 + wave * 32 + li] = h;
                    H2l[row * B3_LD2 + wave * 32 + li] = l;
                    // training forward: the fp32 activation also goes to HBM for the (fp32) backward, see pn_fwd_kernel
                    if (h2_save) h2_save[((long)b * P + (long)tile * B3_TM + row) * 256 + wave * 32 + li] = v;
                }
        }
        __syncthreads();
        // layer 3: 128 points x channels [64w, 64w+64), K = 256
        f32x16 acc[4][2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const float b3c = b3[(wave * 2 + nb) * 32 + li];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][nb][r] = b3c;
        }
        bf3_stream<4, 2, 16>(H2h + li * B3_LD2 + lq * 8, H2l + li * B3_LD2 + lq * 8, B3_LD2,
                             P3h + (size_t)(wave * 2 * 16) * 64 + lane, P3l + (size_t)(wave * 2 * 16) * 64 + lane, acc);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[mb][nb][r];
                    const int p = tile * B3_TM + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lq;
                    if (v > vmax[nb]) {
                        vmax[nb] = v;
                        imax[nb] = p;
                    }
                    vsum[nb] += v;
                }
    }
    const int li = lane0 & 31, lq = lane0 >> 5;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const float ov = __shfl_xor(vmax[nb], 32, 64);
        const int oi = __shfl_xor(imax[nb], 32, 64);
        const float os = __shfl_xor(vsum[nb], 32, 64);
        float v = vmax[nb];
        int i = imax[nb];
        if (ov > v || (ov == v && oi < i)) {
            v = ov;
            i = oi;
        }
        if (lq == 0) {
            const int ch = (wave * 2 + nb) * 32 + li;
            const float sm = vsum[nb] + os;
            if (sm != sm) v = sm;                 // a NaN anywhere in the channel's column: torch.max returns NaN (the strict > above skips it)
            feat[(long)b * ldf + ch] = v;
            if (max_mean) feat[(long)b * ldf + B3_C3 + ch] = sm / (float)P;
            argmax[(long)b * B3_C3 + ch] = i;
        }
    }
}

extern "C" int pm_pointnet_enc_fwd_bf3(const float* x, long ldx, int B, int P, int C, int sub_mean, const float* W1,
                                       const float* b1, const float* b2, const float* b3, const void* packed,
                                       int max_mean, float* feat, long ldf, int32_t* argmax, float* h2_save,
                                       void* stream) {
    PM_REQUIRE(x && W1 && b1 && b2 && b3 && packed && feat && argmax);
    PM_REQUIRE(B > 0 && P > 0 && P % B3_TM == 0 && C >= 1 && C <= B3_MAXC && ldx >= (long)P * C);
    PM_REQUIRE(ldf >= B3_C3 * (max_mean ? 2 : 1));
    PM_REQUIRE(!sub_mean || C >= 3);
    if (((uintptr_t)packed & 15) != 0) return PM_EALIGN;
#define B3_LAUNCH(CT)                                                                                          \
    hipLaunchKernelGGL(pn_fwd_bf3_kernel<CT>, dim3(B), dim3(B3_NT), 0, pm_stream(stream), x, ldx, P, C, sub_mean, W1, \
                       b1, b2, b3, (const unsigned short*)packed, max_mean, feat, ldf, argmax, h2_save)
    if (C == 3) B3_LAUNCH(3);
    else if (C == 4) B3_LAUNCH(4);
    else B3_LAUNCH(0);
#undef B3_LAUNCH
    PM_CHECK_LAUNCH();
    return PM_OK;
}
