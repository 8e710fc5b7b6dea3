// PointNet++ group-all level, last layer fused with the global max-pool (the reference has no PointNet++ code: BASELINE.json
// cfg 3 words the backbone; structure per the published single-scale-grouping network -- `PointNet2` in
// partmanip_amd/algo_utils/network.py; the tests hold its CPU restatement).
//
// The group-all level runs its shared MLP over the R rows of every cloud (R = centres of the last set-abstraction level, 64)
// and takes the max over the cloud: feat[b, c] = max_r tanh(H[b, r, :] . W[c, :] + bias[c]).  Up to round 4 that was a Linear
// launch on (B*R) x CK -> CO, a max-pool launch, and in the backward a scatter of the pooled gradient into a zero-filled
// (B*R) x CO matrix followed by two dense GEMMs on it (data and weight gradient) -- 2 x 34 GFLOP at 2048 clouds on a matrix
// with ONE non-zero per (cloud, channel).  Here:
//   ga_fwd_kernel      one cloud per tile: H rows -> LDS, 32x32x2 fp32 MFMAs against the packed weights, bias + tanh +
//                      max / arg-max over the rows in the accumulator registers (as pn_fwd_kernel's layer 3)
//   ga_bwd_dh_kernel   dH[b, r, :] = sum over the channels c whose arg-max row is r of dz[b, c] * W[c, :], times tanh' of H: per cloud
//                      CO x CK multiply-adds instead of R x CO x CK x 2 -- the cloud's winners sorted by row in LDS, thread k owns
//                      column k and walks the sorted list with one running sum
//   ga_dw_gather_kernel / ga_dw_finish_kernel   dW[c, :] = sum_b dz[b, c] * H[b, arg[b, c], :], db[c] = sum_b dz[b, c]
//                      (the layout of pn_dw3_gather_kernel: cloud ranges x channel blocks, fixed-order partial sums)
// dz[b, c] = dfeat[b, c] * (1 - feat[b, c]^2).  Ties: the lowest row wins, as maxpool_rows_kernel; a NaN wins (the first one), as torch.max.
#include "mfma_f32.h"

#define GA_TM 64                 // rows per tile
#define GA_NW 8                  // waves per forward work-group
#define GA_DW_SPLIT 16
#define GA_DW_CPB 4

extern "C" int pm_sa_groupall_supported(int CK, int CO, int R) { return CK == 256 && CO == 512 && R > 0 && R % GA_TM == 0 && R <= 64; }
extern "C" size_t pm_sa_groupall_packed_elems(int CK, int CO) { return (size_t)CK * CO + 1024; }   // + 4 KB: the operand stream reads one group past the end

// packed[nb][g][lane][e] = W[(nb*32 + lane%32) * CK + (lane/32) * CK/2 + 4 g + e]   (mfma_f32.h: B operand, k split by lane half)
__global__ __launch_bounds__(256) void ga_pack_kernel(const float* __restrict__ W, int CK, int CO, float* __restrict__ packed) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x, n = (long)CK * CO;
    if (i >= n + 1024) return;
    if (i >= n) {
        packed[i] = 0.f;
        return;
    }
    const int NG = CK / 8;
    const int e = i & 3, lane = (i >> 2) & 63, li = lane & 31, lh = lane >> 5;
    const int g = (int)((i >> 8) % NG), nb = (int)((i >> 8) / NG);
    packed[i] = W[(long)(nb * 32 + li) * CK + lh * (CK / 2) + g * 4 + e];
}

extern "C" int pm_sa_groupall_pack_f32(const float* W, int CK, int CO, float* packed, void* stream) {
    PM_REQUIRE(W && packed && CK % 8 == 0 && CO % 32 == 0);
    const long n = (long)CK * CO + 1024;
    hipLaunchKernelGGL(ga_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, pm_stream(stream), W, CK, CO, packed);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// =================================================================================== forward
template <int CK, int CO>
__global__ __launch_bounds__(GA_NW * 64, GA_NW / 2) void ga_fwd_kernel(const float* __restrict__ H, int B, int R, const float* __restrict__ bias,
                                                                const float* __restrict__ packed, float* __restrict__ feat, long ldf,
                                                                int32_t* __restrict__ argmax) {
    constexpr int NT = GA_NW * 64, NB = CO / 32 / GA_NW, LD = CK + 4, NG = CK / 8;
    __shared__ __attribute__((aligned(16))) float Hs[GA_TM * LD];
    const int tid = threadIdx.x, lane0 = tid & 63, wave = tid >> 6;
    const float4* Pv = (const float4*)packed;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        int lane = lane0;
        asm volatile("" : "+v"(lane));                 // (addresses recomputed per cloud instead of ~20 hoisted lane constants)
        const int li = lane & 31, lh = lane >> 5;
        float vmax[NB];
        int imax[NB];
        float bc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            vmax[nb] = -INFINITY;
            imax[nb] = 0;
            bc[nb] = bias[(wave * NB + nb) * 32 + li];
        }
        for (int t = 0; t < R / GA_TM; ++t) {
            const float* src = H + ((long)b * R + (long)t * GA_TM) * CK;
            __syncthreads();                           // the previous tile's MFMA reads of Hs are done
#pragma unroll 4
            for (int i = 0; i < GA_TM * CK / 4 / NT; ++i) {
                const int q = tid + NT * i, row = q / (CK / 4), c4 = q % (CK / 4);
                *(f32x4*)(Hs + row * LD + 4 * c4) = *(const f32x4*)(src + (long)row * CK + 4 * c4);
            }
            __syncthreads();
            f32x16 acc[2][NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][nb][r] = acc[1][nb][r] = bc[nb];
            mfma_stream<2, NB, NG>(Hs + li * LD + lh * (CK / 2), LD, Pv + (size_t)(wave * NB) * NG * 64 + lane, acc);
            // tanh, then max over this tile's rows in increasing row order (strict >: the lowest row wins a tie)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2 v2 = pm_tanh2(acc[mb][nb][r], acc[mb][nb][r + 1]);
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const float v = j ? v2.y : v2.x;
                            const int p = t * GA_TM + mb * 32 + ((r + j) & 3) + 8 * ((r + j) >> 2) + 4 * lh;
                            if (v > vmax[nb] || (v != v && vmax[nb] == vmax[nb])) {   // (the FIRST NaN sticks, as torch.max: no later > is true against it)
                                vmax[nb] = v;
                                imax[nb] = p;
                            }
                        }
                    }
        }
        // lanes l and l^32 hold the two interleaved row sets of the same channel
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const float ov = __shfl_xor(vmax[nb], 32, 64);
            const int oi = __shfl_xor(imax[nb], 32, 64);
            float v = vmax[nb];
            int i = imax[nb];
            if (ov != ov || v != v) {                  // a NaN in either half: NaN, and the row of the first one
                if (ov != ov && (v == v || oi < i)) {
                    v = ov;
                    i = oi;
                }
            } else if (ov > v || (ov == v && oi < i)) {
                v = ov;
                i = oi;
            }
            if (lh == 0) {
                const int ch = (wave * NB + nb) * 32 + li;
                feat[(long)b * ldf + ch] = v;
                argmax[(long)b * CO + ch] = i;
            }
        }
    }
}

extern "C" int pm_sa_groupall_fwd_f32(const float* H, int B, int R, int CK, int CO, const float* bias, const float* packed, float* feat,
                                      long ldf, int32_t* argmax, void* stream) {
    PM_REQUIRE(H && bias && packed && feat && argmax && B > 0 && ldf >= CO && pm_sa_groupall_supported(CK, CO, R));
    const int grid = B < 2048 ? B : 2048;
    hipLaunchKernelGGL((ga_fwd_kernel<256, 512>), dim3(grid), dim3(GA_NW * 64), 0, pm_stream(stream), H, B, R, bias, packed, feat, ldf,
                       argmax);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// =================================================================================== backward: dH
// One work-group (CK threads: thread k = column k of dH) per cloud.  The cloud's CO (channel, row) winners are put in row
// order by a stable counting sort in LDS (channels ascending inside a row: a fixed summation order); the thread then walks the
// sorted list with ONE running sum, eight W[c][k] loads in flight, and hands the sum over at every row boundary.  A boundary
// issues the load of H[row][k] and parks (row, sum); the NEXT boundary multiplies and stores it -- the load has a whole row of
// time to land -- and zero-fills the rows nobody won.  (First form of this kernel: the R row sums in registers, the register
// chosen per channel by a 6-deep tree of scalar branches on the wave-uniform row -- 250 cycles of taken branches per channel,
// 0.435 ms per 2048 clouds, issue-bound; this one ~30 instructions per 8 channels.)
template <int CK, int CO, int R>
__global__ __launch_bounds__(CK) void ga_bwd_dh_kernel(const float* __restrict__ dfeat, long lddf, const float* __restrict__ feat, long ldf,
                                                       const int32_t* __restrict__ argmax, const float* __restrict__ W,
                                                       const float* __restrict__ H, int B, float* __restrict__ dz_g,
                                                       float* __restrict__ dH) {
    static_assert(CK == 256 && CO % 4 == 0 && R == 64, "the sort below gives every (row, channel quarter) one of the 256 threads");
    constexpr int QC = CO / 4;                         // channels per quarter
    __shared__ float dzs[CO];
    __shared__ int rows[CO];
    __shared__ __attribute__((aligned(16))) int s_c[CO + 8];
    __shared__ __attribute__((aligned(16))) int s_row[CO + 8];
    __shared__ __attribute__((aligned(16))) float s_dz[CO + 8];
    __shared__ int wsum[CK / 64];
    const int k = threadIdx.x, lane = k & 63, wave = k >> 6;
    if (k < 8) {                                       // the tail of the last batch of eight: row R (never flushed into), weight 0
        s_c[CO + k] = 0;
        s_row[CO + k] = R;
        s_dz[CO + k] = 0.f;
    }
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();                               // the previous cloud's reads of the sorted lists are done
        for (int c = k; c < CO; c += CK) {
            const float y = feat[(long)b * ldf + c];
            const float dz = dfeat[(long)b * lddf + c] * (1.0f - y * y);
            dzs[c] = dz;
            dz_g[(long)b * CO + c] = dz;
            rows[c] = argmax[(long)b * CO + c];
        }
        __syncthreads();
        // ---- stable counting sort by row: thread (r, q) owns the channels of quarter q whose row is r
        const int r_ = k >> 2, q_ = k & 3;
        int cnt = 0;
#pragma unroll 8
        for (int i = 0; i < QC; ++i) cnt += rows[q_ * QC + i] == r_;
        int incl = cnt;                                // inclusive scan over the work-group, thread order = (row, quarter)
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int pos = incl - cnt;
        for (int w = 0; w < wave; ++w) pos += wsum[w];
        if (cnt)
            for (int i = 0; i < QC; ++i) {
                const int c = q_ * QC + i;
                if (rows[c] == r_) {
                    s_c[pos] = c;
                    s_row[pos] = r_;
                    s_dz[pos] = dzs[c];
                    ++pos;
                }
            }
        __syncthreads();
        // ---- walk the sorted list
        const float* hb = H + (long)b * R * CK + k;
        float* db = dH + (long)b * R * CK + k;
        const float* wk = W + k;
        int cur = -1, pend_r = -1;                     // (scalar: every lane holds the same values)
        float sum = 0.f, pend_s = 0.f, pend_h = 0.f;
        auto boundary = [&](int next) __attribute__((always_inline)) {
            if (pend_r >= 0) db[(long)pend_r * CK] = pend_s * (1.0f - pend_h * pend_h);
            pend_r = cur;
            if (cur >= 0) {
                pend_s = sum;
                pend_h = hb[(long)cur * CK];
            }
            for (int r = cur + 1; r < next; ++r) db[(long)r * CK] = 0.f;        // rows without a winner
            sum = 0.f;
            cur = next;
        };
#pragma unroll 1
        for (int i = 0; i < CO; i += 8) {
            const int4 c0 = *(const int4*)(s_c + i), c1 = *(const int4*)(s_c + i + 4);
            const int4 r0 = *(const int4*)(s_row + i), r1 = *(const int4*)(s_row + i + 4);
            const float4 z0 = *(const float4*)(s_dz + i), z1 = *(const float4*)(s_dz + i + 4);
            const int cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            const int rs[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
            const float zs[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
            float w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = wk[(long)cs[j] * CK];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int rj = __builtin_amdgcn_readfirstlane(rs[j]);
                if (rj != cur && rj < R) boundary(rj);
                sum += zs[j] * w[j];                   // (tail entries: weight 0 times W[0][k])
            }
        }
        boundary(R);
        if (pend_r >= 0) db[(long)pend_r * CK] = pend_s * (1.0f - pend_h * pend_h);
    }
}

// =================================================================================== backward: dW, db
// One H row (CK floats) read per (cloud, channel); the clouds are split into GA_DW_SPLIT ranges with the range index as
// blockIdx.x (consecutive work-groups go round-robin to the XCDs: a range and its rows stay on one XCD's L2), each
// work-group handles GA_DW_CPB channels; partials are added in fixed order by ga_dw_finish_kernel.
template <int CK, int CO>
__global__ __launch_bounds__(CK) void ga_dw_gather_kernel(const float* __restrict__ dz, int B, const float* __restrict__ H, int R,
                                                          const int32_t* __restrict__ argmax, float* __restrict__ tmp) {
    const int c0 = blockIdx.y * GA_DW_CPB, k = threadIdx.x;
    const int per = (B + GA_DW_SPLIT - 1) / GA_DW_SPLIT;
    const int b0 = blockIdx.x * per, b1 = min(B, b0 + per);
    float acc[GA_DW_CPB];
#pragma unroll
    for (int j = 0; j < GA_DW_CPB; ++j) acc[j] = 0.f;
#pragma unroll 4
    for (int b = b0; b < b1; ++b) {
        const int4 sl = *(const int4*)(argmax + (long)b * CO + c0);
        const float4 g = *(const float4*)(dz + (long)b * CO + c0);
        const float* rows = H + (long)b * R * CK + k;
        acc[0] += g.x * rows[(long)sl.x * CK];
        acc[1] += g.y * rows[(long)sl.y * CK];
        acc[2] += g.z * rows[(long)sl.z * CK];
        acc[3] += g.w * rows[(long)sl.w * CK];
    }
#pragma unroll
    for (int j = 0; j < GA_DW_CPB; ++j) tmp[((size_t)blockIdx.x * CO + c0 + j) * CK + k] = acc[j];
}
template <int CK, int CO>
__global__ __launch_bounds__(CK) void ga_dw_finish_kernel(const float* __restrict__ dz, int B, const float* __restrict__ tmp,
                                                          float* __restrict__ dW, float* __restrict__ dbias) {
    __shared__ float red[CK / 64];
    const int c = blockIdx.x, k = threadIdx.x;
    float acc = 0.f;
#pragma unroll
    for (int y = 0; y < GA_DW_SPLIT; ++y) acc += tmp[((size_t)y * CO + c) * CK + k];
    dW[(long)c * CK + k] = acc;
    float s = 0.f;
    for (int b = k; b < B; b += CK) s += dz[(long)b * CO + c];
    s = block_sum<float, CK>(s, red);
    if (k == 0) dbias[c] = s;
}

extern "C" size_t pm_sa_groupall_bwd_workspace_bytes(int B, int CK, int CO) {
    return ((size_t)B * CO + (size_t)GA_DW_SPLIT * CO * CK) * sizeof(float);      // dz | the weight gradient's partial sums
}

extern "C" int pm_sa_groupall_bwd_f32(const float* dfeat, long lddf, const float* feat, long ldf, const int32_t* argmax, const float* W,
                                      const float* H, int B, int R, int CK, int CO, float* dH, float* dW, float* dbias, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    PM_REQUIRE(dfeat && feat && argmax && W && H && dH && dW && dbias && B > 0 && lddf >= CO && ldf >= CO && R == 64 &&
               pm_sa_groupall_supported(CK, CO, R));
    if (!workspace || workspace_bytes < pm_sa_groupall_bwd_workspace_bytes(B, CK, CO)) return PM_EWORKSPACE;
    // dz / tmp are read and written in 16-byte pieces, H / W / dH rows likewise
    if (((uintptr_t)workspace & 15) != 0 || ((uintptr_t)H & 15) != 0 || ((uintptr_t)W & 15) != 0 || ((uintptr_t)dH & 15) != 0) return PM_EALIGN;
    float* dz = (float*)workspace;
    float* tmp = dz + (size_t)B * CO;
    hipLaunchKernelGGL((ga_bwd_dh_kernel<256, 512, 64>), dim3(B < 2048 ? B : 2048), dim3(256), 0, pm_stream(stream), dfeat, lddf, feat, ldf,
                       argmax, W, H, B, dz, dH);
    hipLaunchKernelGGL((ga_dw_gather_kernel<256, 512>), dim3(GA_DW_SPLIT, 512 / GA_DW_CPB), dim3(256), 0, pm_stream(stream), dz, B, H, R,
                       argmax, tmp);
    hipLaunchKernelGGL((ga_dw_finish_kernel<256, 512>), dim3(512), dim3(256), 0, pm_stream(stream), dz, B, tmp, dW, dbias);
    PM_CHECK_LAUNCH();
    return PM_OK;
}
