// OPT-IN fp32-class split-bf16 variant of the saved-h2 encoder backward (pn_bwd16_kernel, network.py:175-181 backward):
// the two dense GEMMs of a 32-point tile,
//     dW2 (256 x 128) += dz2^T (256 x 32 points) * h1 (32 points x 128)          K = points
//     dh1 (32 points x 128) = dz2 (32 points x 256) * W2 (256 x 128)            K = layer-2 channels
// run on v_mfma_f32_32x32x16_bf16 with every fp32 operand split into THREE bf16 planes (x = x0 + x1 + x2 exactly) and the
// six products a0b0, a0b1, a1b0, a0b2, a1b1, a2b0 accumulated (smallest first) in the MFMA's fp32 accumulator -- the scheme
// of pointnet_enc_bf6.hip: the dropped products are <= 2^-24 relative, an fp32 multiply's own rounding.  Included by
// pointnet_enc.hip (shares PnBwdPart, the prep / reduce launches and the workspace of the fp32 path); selected with
// `net_cfg['precision_bwd'] = 'bf16x6'`; everything outside the two GEMMs (layer 1, tanh', the arg-max rows, bias and
// layer-1 gradients, the dW3 launches) is the fp32 code of the default path.
//
// Layout (one 8-wave work-group per CU, 256 VGPRs per wave; 151 KB of LDS, single-buffered):
//   * bf16 MFMAs want BOTH operands K-contiguous per lane (8 consecutive k = one 16-byte read), and K is the point index
//     for dW2 but the channel index for dh1, so dz2 is written twice by the VALU stage that produces it:
//       Zt [plane][channel row][32 points]   point-contiguous   (A operand of dW2; 64-byte rows)
//       Zc [plane][point][256 channels + 8]  channel-contiguous (A operand of dh1; 528-byte rows)
//       Ht [plane][h1 channel][32 points]    point-contiguous   (B operand of dW2)
//     A thread owns 4 channels of 4 consecutive points, so the point pairs of Zt are two 16-bit halves it already holds
//     (re-paired with and/shift/or from the channel pairs of Zc: the three-plane split runs once per value).
//   * Zt / Ht rows are 64 bytes with the 16-byte chunks rotated by (row >> 2): the 16 lanes of a ds_read_b128 phase hit
//     16 distinct 16-byte slots of the 256-byte bank row without padding (49 + 25 KB instead of 61 + 31 KB).
//     Zt row r holds channel 4 (r & 63) + (r >> 6): the 64 lanes of a wave (4 channels each) then write 64 different rows.
//   * dW2: wave w owns a 64 x 64 block of dW2 (2 x 2 accumulators): 12 fragment reads feed 24 MFMAs per 16 points
//     (16 B/clk per wave of LDS reads; one accumulator block per wave would need 2.5 x that and be LDS-bound).
//   * dh1: wave w owns channels [32 (w & 3), +32) of all 32 points over HALF of K (w >> 2): A from Zc, B = W2 planes
//     pre-packed in fragment order (pm_pointnet_pack_weights_bwd_bf6) streamed from L2 two k-steps ahead in a ring that wraps
//     into the next tile (W2 is the same for every tile: the first steps' fragments arrive under the VALU stage).  The two
//     K-halves of a channel block hand each other half of their accumulator rows through a dedicated 16 KB of LDS and both
//     finish eight rows: dz1 = dh1 .* (1 - h1^2) -- h1 re-assembled from its three planes (exact) in front of the barrier the
//     early waves would idle at -- then dW1 / db1 straight from the registers.
//   * the eight dh1 k-steps carry the dW2 blocks between their MFMAs (one block per step through its six products, the two
//     accumulator chains alternating): back to back, the dh1 loop is bound by the L1 fill rate (eight waves streaming 196 KB
//     of planes at once) while dW2 leaves the L1 idle.
//   Per tile and SIMD: 2 waves x (48 + 48) MFMAs x 32 cycles = 6144 cycles (fp32 path: 16384).
// Two barriers per tile (planes written | planes read): the VALU stage of tile t+1 cannot overlap the MFMAs of tile t the
// way the fp32 kernel's double-buffered fp32 tiles allow -- three planes of two layouts do not fit twice.  MEASURED (round 3,
// 2048 clouds): the call takes 2.10 ms against 2.63 for the fp32 kernel -- far from the 2.7 x of the matrix pipe: the in-kernel
// timeline (-DPN_PROFILE, tools/pn_profile.py with PN_PRECISION_BWD=bf16x6) shows a 15-16 k-cycle tile of which the MFMA
// stage is 7-8 k (6.1 k of MFMAs), the VALU stage 4-4.5 k (~550 VALU instructions per thread and tile: the three-plane split
// alone is 9 per value pair, 12 pairs) and the two barriers' skew 2-2.5 k.  MFMA and VALU cycles are ADDITIVE on a SIMD for the
// bf16 pipe as for the fp32 one (tools/ubench/mfma_bf16_valu_overlap.hip), so the floor of this kernel is 6.1 + 4.4 k cycles
// per tile (1.35 ms); a variant that walks dh1 / dW2 and the two halves of the VALU stage as skewed chunks on the two waves of a
// SIMD was built on the opposite hope, is bit-identical and measured 2.22 ms.

typedef __bf16 pb6_bf16x8 __attribute__((ext_vector_type(8)));
#define PB6_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define PB6_LDZ (PN_C2 + 8)                 // halfwords per Zc row
#define PB6_W2 (4 * 16 * 64 * 8)            // halfwords per packed W2 plane: [n-block 4][k-step 16][lane 64][8]
#define PB6_PACKED_HALFS (3 * PB6_W2 + 4096)
#define PB6_NT 512

typedef __bf16 pb6_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pb6_cvt_pk(float a, float b) {          // {bf16(a) | bf16(b) << 16}, RNE: v_cvt_pk_bf16_f32
    const f32x2 v = {a, b};                                                  // (the builtin conversion, not inline asm: the
    const pb6_bf16x2 h = __builtin_convertvector(v, pb6_bf16x2);             // scheduler may interleave the split chains)
    return *(const unsigned*)&h;
}
// (x, y) -> three packed bf16 pairs q[0..2] with q0 + q1 + q2 == the value exactly in each half (every residual is exact in fp32)
__device__ __forceinline__ void pb6_split_pair(float x, float y, unsigned (&q)[3]) {
    q[0] = pb6_cvt_pk(x, y);
    const float rx = x - __uint_as_float(q[0] << 16), ry = y - __uint_as_float(q[0] & 0xffff0000u);
    q[1] = pb6_cvt_pk(rx, ry);
    const float sx = rx - __uint_as_float(q[1] << 16), sy = ry - __uint_as_float(q[1] & 0xffff0000u);
    q[2] = pb6_cvt_pk(sx, sy);
}
__device__ __forceinline__ pb6_bf16x8 pb6_as_bf(const uint4& v) { return *(const pb6_bf16x8*)&v; }

extern "C" size_t pm_pointnet_packed_bwd_bf6_bytes(void) { return (size_t)PB6_PACKED_HALFS * 2; }

// W2 (256 x 128, row-major [c2][c1]) as the B operand of dh1 = dz2 * W2: B[k = c2][n = c1]; lane (li = lane & 31, lq = lane >> 5)
// of fragment (n-block nb, k-step s) holds W2[16 s + 8 lq + e][32 nb + li], e = 0..7 -- three planes, + a zeroed tail.
__global__ __launch_bounds__(256) void pn_pack_bwd_bf6_kernel(const float* __restrict__ W2, unsigned short* __restrict__ packed) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= PB6_W2) {
        if (i < PB6_W2 + 4096) packed[3 * PB6_W2 + (i - PB6_W2)] = 0;
        return;
    }
    const int e = i & 7, lane = (i >> 3) & 63, li = lane & 31, lq = lane >> 5, s = (i >> 9) & 15, nb = i >> 13;
    const float w = W2[(s * 16 + lq * 8 + e) * PN_C1 + nb * 32 + li];
    unsigned q[3];
    pb6_split_pair(w, 0.f, q);
    packed[i] = (unsigned short)(q[0] & 0xffffu);
    packed[PB6_W2 + i] = (unsigned short)(q[1] & 0xffffu);
    packed[2 * PB6_W2 + i] = (unsigned short)(q[2] & 0xffffu);
}

extern "C" int pm_pointnet_pack_weights_bwd_bf6(const float* W2, void* packed, void* stream) {
    PM_REQUIRE(W2 && packed);
    const int n = PB6_W2 + 4096;
    hipLaunchKernelGGL(pn_pack_bwd_bf6_kernel, dim3((n + 255) / 256), dim3(256), 0, pm_stream(stream), W2, (unsigned short*)packed);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// byte offset of logical 16-byte chunk q (8 points) of row r in a 64-byte-row plane (Zt / Ht)
__device__ __forceinline__ int pb6_chunk(int r, int q) { return r * 64 + (((q + (r >> 2)) & 3) << 4); }

template <int CT>
__global__ __launch_bounds__(PB6_NT, 2) void pn_bwd_bf6_kernel(
    const float* __restrict__ x, long ldx, int B, int P, int C, int sub_mean, const float* __restrict__ W1,
    const float* __restrict__ b1, const unsigned short* __restrict__ packW2, int max_mean, const float* __restrict__ U,
    float* __restrict__ H2sum, float* __restrict__ Hg, const int32_t* __restrict__ keys_g,
    const unsigned short* __restrict__ offs_g, PnBwdPart* __restrict__ parts, const float* __restrict__ h2_saved,
    const float* __restrict__ Sg) {
    constexpr int BT = 32, NT = PB6_NT, NW = 8, RPW = BT / NW;             // 4 rows per wave = two of the prep kernel's 2-point blocks
    constexpr int NXC = (CT == 3 || CT == 4) ? 4 : PN_MAXC;
    constexpr int XSZ = BT * PN_MAXC;
    constexpr int PLT = PN_C2 * 64, PLH = PN_C1 * 64, PLC = BT * PB6_LDZ * 2;   // plane strides in BYTES
    __shared__ __attribute__((aligned(16))) unsigned char Zt[3 * PLT];      // 49152 B; end of cloud / kernel: wave partial sums
    __shared__ __attribute__((aligned(16))) unsigned char Zc[3 * PLC];      // 50688 B
    __shared__ __attribute__((aligned(16))) unsigned char Ht[3 * PLH];      // 24576 B
    __shared__ __attribute__((aligned(16))) float Pp[4 * 16 * 64];          // 16384 B: the dh1 accumulator rows the two K-halves of a channel block hand each other
    __shared__ __attribute__((aligned(16))) float Xs0[3 * XSZ];
    __shared__ __attribute__((aligned(16))) float Us[PN_C2];
    __shared__ int keys[PN_C3];
    __shared__ unsigned short offs[4096 / 2 + 2];
    __shared__ double red[16];
    float* wred = (float*)Zt;

    const int tid = threadIdx.x, lane0 = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float invP = 1.0f / (float)P;
    const uint4* PW = (const uint4*)packW2;
    constexpr size_t PWS = PB6_W2 / 8;                                       // plane stride of PW in uint4s

    const int w2_m0 = (wave & 3) * 2, w2_n0 = (wave >> 2) * 2;               // dW2 blocks: Zt rows [32 (w2_m0 + mb), +32), h1 channels [32 (w2_n0 + nb), +32)
    const int hnb = wave & 3, hk = wave >> 2;                                // dh1: h1 channels [32 hnb, +32), k-steps [8 hk, +8)
    f32x16 accW2[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) accW2[mb][nb][r] = 0.f;
    float4 db2acc = make_float4(0.f, 0.f, 0.f, 0.f);                        // columns 4*lane..+3 over this wave's rows
    float dW1acc[NXC], db1acc = 0.f;                                         // channel 32 hnb + (lane & 31) over this lane's 8 rows per tile
#pragma unroll
    for (int d = 0; d < NXC; ++d) dW1acc[d] = 0.f;
    const int CC = (CT == 3 || CT == 4) ? CT : C;
    // dh1's B stream (this wave's 8 k-steps of the packed W2 planes, the same for every tile) runs two k-steps ahead as a ring
    // that WRAPS: steps 0 and 1 of the next tile are requested by the last two steps of this one and arrive under the VALU stage
    uint4 bq[2][3];
    {
        const uint4* Bp0 = PW + (size_t)(hnb * 16 + hk * 8) * 64 + lane0;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int q = 0; q < 3; ++q) bq[d][q] = Bp0[q * PWS + (size_t)d * 64];
    }
    const int ntiles = P / BT;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const float* xb = x + (long)b * ldx;
        float cen[3] = {0.f, 0.f, 0.f};
        __syncthreads();
        if (sub_mean) cloud_centroid<NT>(xb, P, C, red, cen);
        if (tid < PN_C2) Us[tid] = max_mean ? U[(long)b * PN_C2 + tid] * invP : 0.f;
        keys[tid] = keys_g[(long)b * PN_C3 + tid];
        for (int p = tid; p <= P / 2; p += NT) offs[p] = offs_g[(long)b * (P / 2 + 1) + p];
        __syncthreads();
        float4 h2s = make_float4(0.f, 0.f, 0.f, 0.f);

        // ---- VALU stage of tile tt in two parts, as in pn_bwd16_kernel: valu_issue requests what waits on HBM / L2 (the
        // wave's four saved-h2 rows, the finished dh2 rows of its arg-max points) before the MFMA stage of the previous tile,
        // valu_finish turns them into the three planes of dz2 (both layouts) and of h1 after it.
        float4 hrows[RPW], srow[RPW];
        int slot[RPW];
#ifdef PN_PROFILE
        int prof_tile = -1;
#endif
        auto valu_issue = [&](int tt, int tl) __attribute__((always_inline)) {
            const int lane = tl & 63;
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr)
                hrows[rr] = *(const float4*)(h2_saved + ((long)b * P + tt * BT + wave * RPW + rr) * PN_C2 + 4 * lane);
#pragma unroll
            for (int j = 0; j < RPW / 2; ++j) {                              // the prep kernel's key runs cover 2 points each
                const int p0 = tt * BT + wave * RPW + 2 * j;
                const int e = __builtin_amdgcn_readfirstlane((int)offs[p0 / 2]);
                const int e_end = __builtin_amdgcn_readfirstlane((int)offs[p0 / 2 + 1]);
                slot[2 * j] = slot[2 * j + 1] = -1;
                srow[2 * j] = srow[2 * j + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < e_end) {
                    const int kf = __builtin_amdgcn_readfirstlane(keys[e]);
                    const int kl = __builtin_amdgcn_readfirstlane(keys[e_end - 1]);
                    if (((kf >> 9) & 0x1FFF) == p0) slot[2 * j] = kf >> 22;
                    if (((kl >> 9) & 0x1FFF) == p0 + 1) slot[2 * j + 1] = kl >> 22;
#pragma unroll
                    for (int rr = 2 * j; rr < 2 * j + 2; ++rr)
                        if (slot[rr] >= 0) srow[rr] = *(const float4*)(Sg + ((long)b * PN_C3 + slot[rr]) * PN_C2 + 4 * lane);
                }
            }
        };
        auto valu_finish = [&](int tt, const float* Xs, int tl) __attribute__((always_inline)) {
            const int lane = tl & 63;
            {   // layer 1: thread (c = tl & 127, g = tl >> 7) -> points 8g..8g+7 of channel c = ONE 16-byte chunk of Ht per plane
                const int c = tl & 127, g = tl >> 7;
                const float b1c = b1[c];
                float w[NXC];
#pragma unroll
                for (int d = 0; d < NXC; ++d) w[d] = d < CC ? W1[c * CC + d] : 0.f;
                unsigned hq[4][3];
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    float z[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float4 xv = *(const float4*)(Xs + (8 * g + i + j) * PN_MAXC);
                        float sacc = fmaf(w[0], xv.x, b1c);
                        sacc = fmaf(w[1], xv.y, sacc);
                        sacc = fmaf(w[2], xv.z, sacc);
                        if (CT != 3) sacc = fmaf(w[3], xv.w, sacc);
                        if (CT != 3 && CT != 4) {
                            const float4 x1 = *(const float4*)(Xs + (8 * g + i + j) * PN_MAXC + 4);
                            sacc = fmaf(w[4 % NXC], x1.x, sacc); sacc = fmaf(w[5 % NXC], x1.y, sacc);
                            sacc = fmaf(w[6 % NXC], x1.z, sacc); sacc = fmaf(w[7 % NXC], x1.w, sacc);
                        }
                        z[j] = sacc;
                    }
                    const f32x2 t = pm_tanh2(z[0], z[1]);
                    pb6_split_pair(t.x, t.y, hq[i >> 1]);
                }
                const int at = pb6_chunk(c, g);
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    *(uint4*)(Ht + q * PLH + at) = make_uint4(hq[0][q], hq[1][q], hq[2][q], hq[3][q]);
            }
            {   // rows 4 wave .. 4 wave + 3, channels 4 lane .. 4 lane + 3: h2 -> dz2 -> planes in both layouts
                const float4 u4 = *(const float4*)(Us + 4 * lane);
                unsigned zq[RPW][2][3];                                       // [row][channel pair (4l, 4l+1) | (4l+2, 4l+3)][plane]
#pragma unroll
                for (int rr = 0; rr < RPW; ++rr) {
                    const float4 h = hrows[rr], S = srow[rr];
                    h2s.x += h.x; h2s.y += h.y; h2s.z += h.z; h2s.w += h.w;
                    if (slot[rr] >= 0)                                        // this point is some channel's arg-max
                        *(float4*)(Hg + ((long)b * PN_C3 + slot[rr]) * PN_C2 + 4 * lane) = h;
                    float4 dz;
                    dz.x = (u4.x + S.x) * (1.0f - h.x * h.x);
                    dz.y = (u4.y + S.y) * (1.0f - h.y * h.y);
                    dz.z = (u4.z + S.z) * (1.0f - h.z * h.z);
                    dz.w = (u4.w + S.w) * (1.0f - h.w * h.w);
                    db2acc.x += dz.x; db2acc.y += dz.y; db2acc.z += dz.z; db2acc.w += dz.w;
                    pb6_split_pair(dz.x, dz.y, zq[rr][0]);
                    pb6_split_pair(dz.z, dz.w, zq[rr][1]);
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        *(uint2*)(Zc + q * PLC + (wave * RPW + rr) * (PB6_LDZ * 2) + 8 * lane) = make_uint2(zq[rr][0][q], zq[rr][1][q]);
                }
                // Zt: channel 4 lane + i lives in row lane + 64 i; this wave's 4 points are dwords 2 wave, 2 wave + 1 of the row
                // = logical chunk wave >> 1, bytes 8 (wave & 1) .. +7
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = lane + 64 * i;
                    const int at = pb6_chunk(r, wave >> 1) + 8 * (wave & 1);
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const unsigned a0 = zq[0][i >> 1][q], a1 = zq[1][i >> 1][q], a2 = zq[2][i >> 1][q], a3 = zq[3][i >> 1][q];
                        uint2 v;
                        if (i & 1) {
                            v.x = (a0 >> 16) | (a1 & 0xffff0000u);
                            v.y = (a2 >> 16) | (a3 & 0xffff0000u);
                        } else {
                            v.x = (a0 & 0xffffu) | (a1 << 16);
                            v.y = (a2 & 0xffffu) | (a3 << 16);
                        }
                        *(uint2*)(Zt + q * PLT + at) = v;
                    }
                }
            }
        };
        // ---- pipeline prologue: tile 0 through the VALU stage -------------------------------------------------------------
        stage_points<BT, NT>(xb, 0, C, sub_mean, cen, Xs0);
        if (ntiles > 1) stage_points<BT, NT>(xb, 1, C, sub_mean, cen, Xs0 + XSZ);
        __syncthreads();
        {
            int tl = tid;
            asm volatile("" : "+v"(tl));
            valu_issue(0, tl);
            valu_finish(0, Xs0, tl);
        }
        __syncthreads();
        int ix = 0;                                        // Xs buffer of tile t (t+1: ix+1, t+2: ix+2, mod 3)
        for (int tile = 0; tile < ntiles; ++tile) {
            int tl = tid;                                  // laundered per tile: recompute addresses, don't hoist
            asm volatile("" : "+v"(tl));
            const int ix1 = ix == 2 ? 0 : ix + 1, ix2 = ix1 == 2 ? 0 : ix1 + 1;
            const bool more = tile + 1 < ntiles;
            float xnext = 0.f;                             // the points of tile t+2: requested here, stored at the end of the interval
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
            // ---- MFMA stage of tile t -----------------------------------------------------------------------------------
            const int lane = tl & 63, li = lane & 31, lq = lane >> 5;
            // dW2 += dz2^T * h1 (K = 32 points = two k-steps, LDS operands only) INTERLEAVED with dh1 = dz2 * W2 over this
            // wave's half of K (A from Zc: row = point li, k = channel; B = the packed W2 planes from L2): back to back, the dh1
            // loop alone is bound by the L1 fill rate (all eight waves stream their 24 KB of W2 planes at once: 196 KB per tile
            // at 64 B/clk = 3.1 k cycles for 1.5 k cycles of MFMAs, in-kernel timeline) while dW2 leaves the L1 idle.  Each of
            // the eight dh1 k-steps carries one 32 x 32 block of dW2 through its six products; the two accumulator chains
            // alternate, so no MFMA depends on its predecessor.
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            {
                const unsigned char* Ap = Zc + li * (PB6_LDZ * 2) + (hk * 8) * 32 + lq * 16;
                const uint4* Bp = PW + (size_t)(hnb * 16 + hk * 8) * 64 + lane;
                uint4 wa[3][2], wb[3][2];                  // dW2 fragments of the current k-step: [plane][block]
#define PB6_W_LOAD(S_)                                                                                \
    _Pragma("unroll") for (int q = 0; q < 3; ++q)                                                     \
        _Pragma("unroll") for (int m = 0; m < 2; ++m) {                                               \
            wa[q][m] = *(const uint4*)(Zt + q * PLT + pb6_chunk((w2_m0 + m) * 32 + li, 2 * (S_) + lq)); \
            wb[q][m] = *(const uint4*)(Ht + q * PLH + pb6_chunk((w2_n0 + m) * 32 + li, 2 * (S_) + lq)); \
        }
#define PB6_PAIR(pa, pb, D_, MB_, NB_)                                                                \
    acc = PB6_MFMA(pb6_as_bf(a[pa]), pb6_as_bf(bq[D_][pb]), acc);                                     \
    accW2[MB_][NB_] = PB6_MFMA(pb6_as_bf(wa[pa][MB_]), pb6_as_bf(wb[pb][NB_]), accW2[MB_][NB_]);
#define PB6_STEP(S_, D_, MB_, NB_)                                                                    \
    {                                                                                                 \
        uint4 a[3];                                                                                   \
        _Pragma("unroll") for (int q = 0; q < 3; ++q) a[q] = *(const uint4*)(Ap + q * PLC + (S_) * 32); \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        PB6_PAIR(2, 0, D_, MB_, NB_) PB6_PAIR(1, 1, D_, MB_, NB_) PB6_PAIR(0, 2, D_, MB_, NB_)        \
        PB6_PAIR(1, 0, D_, MB_, NB_) PB6_PAIR(0, 1, D_, MB_, NB_) PB6_PAIR(0, 0, D_, MB_, NB_)        \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        _Pragma("unroll") for (int q = 0; q < 3; ++q)   /* two steps ahead; the last two fetch steps 0, 1 for the NEXT tile */ \
            bq[D_][q] = Bp[q * PWS + (size_t)(((S_) + 2) & 7) * 64];                                  \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }
                PB6_W_LOAD(0)
                PB6_STEP(0, 0, 0, 0) PB6_STEP(1, 1, 1, 0) PB6_STEP(2, 0, 0, 1) PB6_STEP(3, 1, 1, 1)
                PN_STAMP(2);
                PB6_W_LOAD(1)
                PB6_STEP(4, 0, 0, 0) PB6_STEP(5, 1, 1, 0) PB6_STEP(6, 0, 0, 1) PB6_STEP(7, 1, 1, 1)
#undef PB6_STEP
#undef PB6_PAIR
#undef PB6_W_LOAD
            }
            // the two K-halves of a channel block meet through LDS and SHARE the finish: each wave hands the other half of
            // its accumulator rows over and finishes eight rows itself
#pragma unroll
            for (int r = 0; r < 8; ++r) Pp[((hnb * 2 + hk) * 8 + r) * 64 + lane] = hk ? acc[r] : acc[8 + r];
            // tanh' of this lane's eight (point, channel) results, from the h1 planes while they are still intact (p0 + p1 + p2
            // is h1 exactly): read here, in front of the barrier the early waves would otherwise idle at
            float fac[8];
            {
                const int c1 = hnb * 32 + li;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int row0 = 8 * (2 * hk + hf) + 4 * lq;                      // accumulator registers 8 hk + 4 hf .. + 3 = points row0 .. row0 + 3
                    const int at = pb6_chunk(c1, row0 >> 3) + 8 * lq;
                    const uint2 v0 = *(const uint2*)(Ht + at), v1 = *(const uint2*)(Ht + PLH + at), v2 = *(const uint2*)(Ht + 2 * PLH + at);
                    const float h0 = (__uint_as_float(v0.x << 16) + __uint_as_float(v1.x << 16)) + __uint_as_float(v2.x << 16);
                    const float h1v = (__uint_as_float(v0.x & 0xffff0000u) + __uint_as_float(v1.x & 0xffff0000u)) + __uint_as_float(v2.x & 0xffff0000u);
                    const float h2v = (__uint_as_float(v0.y << 16) + __uint_as_float(v1.y << 16)) + __uint_as_float(v2.y << 16);
                    const float h3 = (__uint_as_float(v0.y & 0xffff0000u) + __uint_as_float(v1.y & 0xffff0000u)) + __uint_as_float(v2.y & 0xffff0000u);
                    fac[4 * hf] = 1.0f - h0 * h0;
                    fac[4 * hf + 1] = 1.0f - h1v * h1v;
                    fac[4 * hf + 2] = 1.0f - h2v * h2v;
                    fac[4 * hf + 3] = 1.0f - h3 * h3;
                }
            }
            PN_STAMP(3);
            __syncthreads();                               // every read of the planes is done; the upper K-half's sums are visible
            PN_STAMP(4);
            {
                // dz1 = dh1 .* (1 - h1^2); dW1 / db1 straight from the registers (summed per tile first, then added to the
                // running sums: short fp32 chains)
                const float* Xs = Xs0 + ix * XSZ;
                float tb = 0.f, tw[NXC];
#pragma unroll
                for (int d = 0; d < NXC; ++d) tw[d] = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int r = hk * 8 + i;                                          // accumulator register r = row (r&3) + 8 (r>>2) + 4 lq
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lq;
                    const float4 x0 = *(const float4*)(Xs + row * PN_MAXC);
                    // lower K-half + upper K-half, in that order on both waves
                    const float mine = hk ? acc[8 + i] : acc[i];
                    const float other = Pp[((hnb * 2 + (1 - hk)) * 8 + i) * 64 + lane];
                    const float dz = (hk ? other + mine : mine + other) * fac[i];
                    tb += dz;
                    tw[0] = fmaf(dz, x0.x, tw[0]);
                    tw[1] = fmaf(dz, x0.y, tw[1]);
                    tw[2] = fmaf(dz, x0.z, tw[2]);
                    tw[3] = fmaf(dz, x0.w, tw[3]);
                    if (CT != 3 && CT != 4) {
                        const float4 x1 = *(const float4*)(Xs + row * PN_MAXC + 4);
                        tw[4 % NXC] = fmaf(dz, x1.x, tw[4 % NXC]);
                        tw[5 % NXC] = fmaf(dz, x1.y, tw[5 % NXC]);
                        tw[6 % NXC] = fmaf(dz, x1.z, tw[6 % NXC]);
                        tw[7 % NXC] = fmaf(dz, x1.w, tw[7 % NXC]);
                    }
                }
                db1acc += tb;
#pragma unroll
                for (int d = 0; d < NXC; ++d) dW1acc[d] += tw[d];
            }
            PN_STAMP(5);
            if (more) valu_finish(tile + 1, Xs0 + ix1 * XSZ, tl);
            if (stage2) Xs0[ix2 * XSZ + tl] = xnext;
            PN_STAMP(6);
            __syncthreads();                               // tile t+1's planes are complete
            PN_STAMP(7);
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
        const int li = lane0 & 31, lq = lane0 >> 5;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int R = (w2_m0 + mb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lq;     // Zt row -> layer-2 channel
                    const int out = 4 * (R & 63) + (R >> 6);
                    part->dW2[out * PN_C1 + (w2_n0 + nb) * 32 + li] = accW2[mb][nb][r];
                }
    }
    __syncthreads();
    *(float4*)(wred + wave * PN_C2 + 4 * lane0) = db2acc;
    float* t1 = wred + NW * PN_C2;                      // dW1 / db1: the four row subsets (K-half wave, lane >> 5) of a channel meet in LDS: [4][128][9]
    {
        const int c = hnb * 32 + (lane0 & 31), part_i = hk * 2 + (lane0 >> 5);
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
            const float sum = (t1[tid * (PN_MAXC + 1) + d] + t1[(128 + tid) * (PN_MAXC + 1) + d]) +
                              (t1[(256 + tid) * (PN_MAXC + 1) + d] + t1[(384 + tid) * (PN_MAXC + 1) + d]);
            if (d < PN_MAXC) part->dW1[tid * PN_MAXC + d] = sum;
            else part->db1[tid] = sum;
        }
    }
}
