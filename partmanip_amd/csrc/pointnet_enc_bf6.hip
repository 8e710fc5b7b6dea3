// OPT-IN variant of the fused PointNet encoder forward (network.py:147-153,172-181) with fp32-CLASS accuracy on the
// bf16 matrix pipe: every fp32 operand is split into THREE bf16 planes (x = x0 + x1 + x2 exactly: 8 + 8 + 8 significand
// bits) and a*b is accumulated as the six products a0b0, a0b1, a1b0, a0b2, a1b1, a2b0 on v_mfma_f32_32x32x16_bf16
// with fp32 accumulation; the dropped products are <= 2^-24 relative, the size of an fp32 multiply's own rounding.
// Emulated on the host for K = 256: mean error against fp64 1.6e-7 of mean |out| for this scheme, 1.8e-7
// for the fp32 MFMA chain, 4.0e-6 for the two-plane bf16x3 variant.  Six bf16 MFMAs (6 x 32 cycles per 32x32x16) replace
// eight fp32 MFMAs (8 x 64 cycles): 2.7x less matrix-pipe time.  Selected with `net_cfg['precision'] = 'bf16x6'`;
// the default path stays the fp32 MFMA kernel (pointnet_enc.hip).
//
// Dataflow as pointnet_enc_bf3.hip with 64-point tiles (three planes of [64][264] halfwords = 101 KB of LDS, one
// work-group of 8 waves per CU): layer 1 (VALU) -> H1 planes; layer 2: wave w -> channels [32w,32w+32), 2x1
// accumulators; tanh + split -> H2 planes (aliasing H1); layer 3: wave w -> channels [64w,64w+64), 2x2 accumulators;
// pooling in-lane over the accumulator registers (rows = points).
//
// Where its 4.1 ms go (A/B ablation builds, 2048 clouds): everything outside the two MFMA loops 0.7 ms; the loops with
// no operand traffic at all 2.4 ms (the bf16 pipe sustains 2.15 PFLOP/s with two waves per SIMD,
// tools/ubench/mfma_bf16_rate.hip: 1.9 ms for these 4.13 PFLOP); the L2 weight stream +1.0 ms; the LDS activation
// reads +0.  The weight stream is a BANDWIDTH limit, not a latency one (three steps of prefetch instead of one, or
// prefetching A as well, change nothing): with 64-row tiles every 1 KB weight fragment feeds two 32-cycle MFMAs per
// product class = 16 B/clk per wave, 64 B/clk per CU at full matrix rate -- the L1 fill rate.  More rows per fragment
// need more than the 160 KB of LDS for three activation planes (128 rows: 203 KB); a 96-row tile (147 KB) is the next
// thing to try -- measured first (round 5): a timing-only build with four row blocks per weight fragment (half the weight bytes
// per MFMA, rows aliased) takes 3.70 ms against 3.97: larger tiles are a 7 % lever, not built.  A 16-wave work-group (four
// waves per SIMD) measured 4.3 ms.  The packed planes are step-major (B6_STEP_MAJOR): -2 %.
#include "common.h"


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define B6_TM 64
#define B6_NT 512
#define B6_LD1 (128 + 8)     // halfwords; 272 B rows: 16 rows hit 16 distinct 16-B slots of the 256-B bank row
#define B6_LD2 (256 + 8)
#define B6_MAXC 8
#define B6_C3 512

// packed planes (halfwords): P2[0..2] | P3[0..2] | pad
#define B6_P2 (8 * 8 * 64 * 8)          // [nb 8][step 8][lane 64][8]
#define B6_P3 (16 * 16 * 64 * 8)        // [nb 16][step 16][lane 64][8]
#define B6_OFF_P2(p) ((p) * B6_P2)
#define B6_OFF_P3(p) (3 * B6_P2 + (p) * B6_P3)
#ifndef B6_STEP_MAJOR
#define B6_STEP_MAJOR 1      // packed order [step][plane][nb][lane][8]: what the 8 waves of every CU request at one k-step is one contiguous 48 KB
#endif
#define B6_PAD (B6_STEP_MAJOR ? 32768 : 4096)    // the last k-step's prefetch over-reads one step (step-major: 3 planes x 16 N-blocks = 24 576 halfwords)
#define B6_PACKED_HALFS (3 * B6_P2 + 3 * B6_P3 + B6_PAD)

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {        // {bf16(a) | bf16(b) << 16}, RNE
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// x -> three bf16 bit patterns with p0 + p1 + p2 == x exactly (each residual is exact in fp32)
__device__ __forceinline__ void split3_bf16(float x, unsigned short& p0, unsigned short& p1, unsigned short& p2) {
    const unsigned a = cvt_pk_bf16(x, 0.f) & 0xffffu;
    const float r1 = x - __uint_as_float(a << 16);
    const unsigned b = cvt_pk_bf16(r1, 0.f) & 0xffffu;
    const float r2 = r1 - __uint_as_float(b << 16);
    const unsigned c = cvt_pk_bf16(r2, 0.f) & 0xffffu;
    p0 = (unsigned short)a;
    p1 = (unsigned short)b;
    p2 = (unsigned short)c;
}

// (x, y) -> three packed bf16 pairs {plane(x) | plane(y) << 16}: one v_cvt_pk_bf16_f32 per plane for both values (the builtin
// conversion, so the scheduler may interleave the chains); the same roundings as split3_bf16
typedef __bf16 b6_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned b6_cvt_pk(float a, float b) {
    const f32x2 v = {a, b};
    const b6_bf16x2 h = __builtin_convertvector(v, b6_bf16x2);
    return *(const unsigned*)&h;
}
__device__ __forceinline__ void split3_pair(float x, float y, unsigned (&q)[3]) {
    q[0] = b6_cvt_pk(x, y);
    const float rx = x - __uint_as_float(q[0] << 16), ry = y - __uint_as_float(q[0] & 0xffff0000u);
    q[1] = b6_cvt_pk(rx, ry);
    const float sx = rx - __uint_as_float(q[1] << 16), sy = ry - __uint_as_float(q[1] & 0xffff0000u);
    q[2] = b6_cvt_pk(sx, sy);
}

extern "C" size_t pm_pointnet_packed_bf6_bytes(void) { return (size_t)B6_PACKED_HALFS * 2; }

__global__ __launch_bounds__(256) void pn_pack_bf6_kernel(const float* __restrict__ W2, const float* __restrict__ W3,
                                                           unsigned short* __restrict__ packed) {
    const int i = blockIdx.x * 256 + threadIdx.x;                       // one (nb, step, lane, e) slot of P2 or P3
    if (i >= B6_P2 + B6_P3) {
        if (i < B6_P2 + B6_P3 + B6_PAD) packed[3 * B6_P2 + 3 * B6_P3 + (i - B6_P2 - B6_P3)] = 0;
        return;
    }
    const int e = i & 7, lane = (i >> 3) & 63, li = lane & 31, lq = lane >> 5;
    float w;
    int o0, stride;
    if (i < B6_P2) {
        const int step = (i >> 9) & 7, nb = i >> 12;
        w = W2[(nb * 32 + li) * 128 + step * 16 + lq * 8 + e];
        o0 = B6_STEP_MAJOR ? B6_OFF_P2(0) + ((step * 3 * 8 + nb) * 64 + lane) * 8 + e : B6_OFF_P2(0) + i;
        stride = B6_STEP_MAJOR ? 8 * 64 * 8 : B6_P2;
    } else {
        const int j = i - B6_P2, step = (j >> 9) & 15, nb = j >> 13;
        w = W3[(nb * 32 + li) * 256 + step * 16 + lq * 8 + e];
        o0 = B6_STEP_MAJOR ? B6_OFF_P3(0) + ((step * 3 * 16 + nb) * 64 + lane) * 8 + e : B6_OFF_P3(0) + j;
        stride = B6_STEP_MAJOR ? 16 * 64 * 8 : B6_P3;
    }
    unsigned short p0, p1, p2;
    split3_bf16(w, p0, p1, p2);
    packed[o0] = p0;
    packed[o0 + stride] = p1;
    packed[o0 + 2 * stride] = p2;
}

extern "C" int pm_pointnet_pack_weights_bf6(const float* W2, const float* W3, void* packed, void* stream) {
    PM_REQUIRE(W2 && W3 && packed);
    const int n = B6_P2 + B6_P3 + B6_PAD;
    hipLaunchKernelGGL(pn_pack_bf6_kernel, dim3((n + 255) / 256), dim3(256), 0, pm_stream(stream), W2, W3,
                       (unsigned short*)packed);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

#define MFMA_BF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ bf16x8 as_bf(const uint4& v) { return *(const bf16x8*)&v; }

// acc[mb][nb] += A(MB*32 rows; three bf16 planes in LDS, plane stride aps, row stride lda halfwords) * B(three packed
// planes, plane stride bps uint4s) over NS k-steps of 16.  A points at this lane's row li, k-offset lq*8 of plane 0;
// B at this wave's first N-block and this lane of plane 0.  The next step's B operands are fetched before this step's
// MFMAs (named ping/pong sets pinned with sched_barrier, as in pointnet_enc.hip); A comes from LDS at the top of the
// step.  Term order: the smallest products first, each product class over all (mb, nb) accumulators before the next
// (consecutive MFMAs never depend on each other).
template <int MB, int NB, int NS, int NBT>
__device__ __forceinline__ void bf6_stream(const unsigned short* __restrict__ A, int aps, int lda,
                                           const uint4* __restrict__ B, f32x16 (&acc)[MB][NB]) {
    // strides of the packed planes in uint4s (NBT = N-blocks of the whole layer): plane, N-block, k-step
    constexpr size_t bps = B6_STEP_MAJOR ? (size_t)NBT * 64 : (size_t)NBT * NS * 64;
    constexpr size_t bnb = B6_STEP_MAJOR ? 64 : (size_t)NS * 64;
    constexpr size_t bst = B6_STEP_MAJOR ? (size_t)3 * NBT * 64 : 64;
    uint4 b0[3][NB], b1[3][NB];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b0[pl][nb] = B[pl * bps + nb * bnb];
#define B6_TERM(pa, pb, BC)                                                                    \
    _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                          \
        _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                      \
            acc[mb][nb] = MFMA_BF(as_bf(a[pa][mb]), as_bf(BC[pb][nb]), acc[mb][nb]);
#define B6_STEP(S_, BC, BN)                                                                    \
    {                                                                                          \
        uint4 a[3][MB];                                                                        \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                       \
            _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                  \
                a[pl][mb] = *(const uint4*)(A + pl * aps + mb * 32 * lda + (S_) * 16);         \
        _Pragma("unroll") for (int pl = 0; pl < 3; ++pl)                                       \
            _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)  /* next step (over-reads one step at the end: padded) */ \
                BN[pl][nb] = B[pl * bps + nb * bnb + (size_t)((S_) + 1) * bst];                \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        B6_TERM(2, 0, BC) B6_TERM(1, 1, BC) B6_TERM(0, 2, BC)                                  \
        B6_TERM(1, 0, BC) B6_TERM(0, 1, BC) B6_TERM(0, 0, BC)                                  \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    }
#pragma unroll 1
    for (int s = 0; s < NS; s += 2) {
        B6_STEP(s, b0, b1)
        B6_STEP(s + 1, b1, b0)
    }
#undef B6_STEP
#undef B6_TERM
}

template <int CT>
__global__ __launch_bounds__(B6_NT, 2) void pn_fwd_bf6_kernel(const float* __restrict__ x, long ldx, int P, int C,
                                                               int sub_mean, const float* __restrict__ W1,
                                                               const float* __restrict__ b1,
                                                               const float* __restrict__ b2,
                                                               const float* __restrict__ b3,
                                                               const unsigned short* __restrict__ packed, int max_mean,
                                                               float* __restrict__ feat, long ldf,
                                                               int32_t* __restrict__ argmax,
                                                               float* __restrict__ h2_save) {
    constexpr int PL1 = B6_TM * B6_LD1, PL2 = B6_TM * B6_LD2;       // plane strides (halfwords) of H1 / H2 (H2 aliases H1)
    __shared__ __attribute__((aligned(16))) unsigned short Hs[3 * PL2];
    __shared__ __attribute__((aligned(16))) float Xs[B6_TM * B6_MAXC];
    __shared__ double red[16];

    const int b = blockIdx.x, tid = threadIdx.x, lane0 = tid & 63, wave = tid >> 6;
    const float* xb = x + (long)b * ldx;
    const uint4* P2 = (const uint4*)(packed + B6_OFF_P2(0));
    const uint4* P3 = (const uint4*)(packed + B6_OFF_P3(0));

    float cen[3] = {0.f, 0.f, 0.f};
    if (sub_mean) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        for (int p = tid; p < P; p += B6_NT) {
            s0 += (double)xb[p * C];
            s1 += (double)xb[p * C + 1];
            s2 += (double)xb[p * C + 2];
        }
        s0 = block_sum<double, B6_NT>(s0, red);
        s1 = block_sum<double, B6_NT>(s1, red);
        s2 = block_sum<double, B6_NT>(s2, red);
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

    const int ntiles = P / B6_TM;
    for (int tile = 0; tile < ntiles; ++tile) {
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int li = lane & 31, lq = lane >> 5;
        __syncthreads();                                   // previous tile's layer-3 reads of Hs are done
        for (int i = tid; i < B6_TM * B6_MAXC; i += B6_NT) {
            const int p = i >> 3, d = i & 7;
            float v = 0.f;
            if (d < C) {
                v = xb[(tile * B6_TM + p) * C + d];
                if (sub_mean && d < 3) v -= cen[d];
            }
            Xs[i] = v;
        }
        __syncthreads();
        {   // layer 1: thread (c = tid&127, part = tid>>7) -> 16 points; write the three planes
            constexpr int PPT = B6_TM * 128 / B6_NT;
            const int c = tid & 127, p0 = (tid >> 7) * PPT;
            float w[B6_MAXC];
#pragma unroll
            for (int d = 0; d < B6_MAXC; ++d) w[d] = (d < C) ? W1[c * C + d] : 0.f;
            const float b1c = b1[c];
#pragma unroll 2
            for (int p = p0; p < p0 + PPT; p += 2) {
                float z[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 x0 = *(const float4*)(Xs + (p + j) * B6_MAXC);
                    float s = fmaf(w[0], x0.x, b1c);
                    s = fmaf(w[1], x0.y, s);
                    s = fmaf(w[2], x0.z, s);
                    s = fmaf(w[3], x0.w, s);
                    if (CT != 3 && CT != 4) {
                        const float4 x1 = *(const float4*)(Xs + (p + j) * B6_MAXC + 4);
                        s = fmaf(w[4], x1.x, s); s = fmaf(w[5], x1.y, s); s = fmaf(w[6], x1.z, s); s = fmaf(w[7], x1.w, s);
                    }
                    z[j] = s;
                }
                const f32x2 t = pm_tanh2(z[0], z[1]);
                unsigned q[3];
                split3_pair(t.x, t.y, q);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    Hs[pl * PL1 + p * B6_LD1 + c] = (unsigned short)q[pl];
                    Hs[pl * PL1 + (p + 1) * B6_LD1 + c] = (unsigned short)(q[pl] >> 16);
                }
            }
        }
        __syncthreads();
        {   // layer 2: 64 points x channels [32w, 32w+32), K = 128
            f32x16 acc2[2][1];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[mb][0][r] = 0.f;
            bf6_stream<2, 1, 8, 8>(Hs + li * B6_LD1 + lq * 8, PL1, B6_LD1, P2 + (size_t)wave * (B6_STEP_MAJOR ? 64 : 8 * 64) + lane, acc2);
            __syncthreads();                               // every wave has finished reading H1
            const float b2c = b2[wave * 32 + li];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 t = pm_tanh2(acc2[mb][0][r] + b2c, acc2[mb][0][r + 1] + b2c);
                    unsigned q[3];
                    split3_pair(t.x, t.y, q);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int row = mb * 32 + ((r + j) & 3) + 8 * ((r + j) >> 2) + 4 * lq;
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            Hs[pl * PL2 + row * B6_LD2 + wave * 32 + li] = (unsigned short)(j ? q[pl] >> 16 : q[pl]);
                        // training forward: the fp32 activation also goes to HBM for the (fp32) backward, see pn_fwd_kernel
                        if (h2_save) h2_save[((long)b * P + (long)tile * B6_TM + row) * 256 + wave * 32 + li] = j ? t.y : t.x;
                    }
                }
        }
        __syncthreads();
        // layer 3: 64 points x channels [64w, 64w+64), K = 256
        f32x16 acc[2][2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const float b3c = b3[(wave * 2 + nb) * 32 + li];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][nb][r] = b3c;
        }
        bf6_stream<2, 2, 16, 16>(Hs + li * B6_LD2 + lq * 8, PL2, B6_LD2, P3 + (size_t)(wave * 2) * (B6_STEP_MAJOR ? 64 : 16 * 64) + lane, acc);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[mb][nb][r];
                    const int p = tile * B6_TM + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lq;
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
            if (max_mean) feat[(long)b * ldf + B6_C3 + ch] = sm / (float)P;
            argmax[(long)b * B6_C3 + ch] = i;
        }
    }
}

extern "C" int pm_pointnet_enc_fwd_bf6(const float* x, long ldx, int B, int P, int C, int sub_mean, const float* W1,
                                       const float* b1, const float* b2, const float* b3, const void* packed,
                                       int max_mean, float* feat, long ldf, int32_t* argmax, float* h2_save,
                                       void* stream) {
    PM_REQUIRE(x && W1 && b1 && b2 && b3 && packed && feat && argmax);
    PM_REQUIRE(B > 0 && P > 0 && P % B6_TM == 0 && C >= 1 && C <= B6_MAXC && ldx >= (long)P * C);
    PM_REQUIRE(ldf >= B6_C3 * (max_mean ? 2 : 1));
    PM_REQUIRE(!sub_mean || C >= 3);
    if (((uintptr_t)packed & 15) != 0) return PM_EALIGN;
#define B6_LAUNCH(CT)                                                                                          \
    hipLaunchKernelGGL(pn_fwd_bf6_kernel<CT>, dim3(B), dim3(B6_NT), 0, pm_stream(stream), x, ldx, P, C, sub_mean, W1, \
                       b1, b2, b3, (const unsigned short*)packed, max_mean, feat, ldf, argmax, h2_save)
    if (C == 3) B6_LAUNCH(3);
    else if (C == 4) B6_LAUNCH(4);
    else B6_LAUNCH(0);
#undef B6_LAUNCH
    PM_CHECK_LAUNCH();
    return PM_OK;
}
