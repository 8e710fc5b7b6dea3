// Conv3D students (network.py:56-94 `Conv3DNet`: Conv3d(1,16,k5,s3,p2) -> Conv3d(16,32,k3,s3,p1) ->
// Conv3d(32,32,k3,s2,p1), activation after each) as patch gather + the fp32 MFMA Linear kernels (K4/K5):
//   forward   cols = im2col(x)            (this file)      y = act(cols * W^T + b)          (pm_linear_fwd_f32)
//   backward  dW = dy^T * cols, dcols = dy * W            (pm_linear_bwd_*)   dx = col2im(dcols) [.* act'(x)]  (this file)
// with W = conv.weight viewed (Cout, Cin*k^3): column order (c, kd, kh, kw), c slowest -- the order of the
// reference's weight tensor, so `state_dict` tensors are used as they are.  The volumes are tiny (50^3 -> 17^3 ->
// 6^3 -> 3^3: 13.5 MMAC per sample against the PointNet encoder's 168) and the gathers are pure data movement:
// HBM-bound, coalesced over the column index.
// x is addressed through element strides (sb, sc, sd, sh, sw), so the same kernels read the channels-first
// network input and the channels-last (rows x Cout) outputs of the previous layer's Linear kernel.
// The single-channel INPUT layer does not take this route: see "direct conv of the input layer" below.
#include "common.h"

struct Conv3dGeom {
    int C, D, H, W;            // input channels and extent
    int k, stride, pad;
    int Do, Ho, Wo;            // output extent
    long sb, sc, sd, sh, sw;   // input element strides
    int ldc;                   // row stride of the patch matrix (>= C*k^3; the tail is zero-filled)
};

// One thread per (patch row, channel, kd, kh): it decodes its indices once and copies the k taps along w (contiguous
// in the input when sw == 1, contiguous in the patch row) -- the first version decoded every ELEMENT (five div/mod
// chains each) and was integer-VALU-bound: 5.3 ms for conv1's 4 GB patch matrix.  One extra group per row zero-fills
// the ldc - C*k^3 pad columns.
__global__ __launch_bounds__(256) void im2col3d_kernel(const float* __restrict__ x, Conv3dGeom g, long total,
                                                        float* __restrict__ cols) {
    const int kk = g.k * g.k, k3 = kk * g.k, ncol = g.C * k3, gpr = g.C * kk + 1;     // groups per row
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long row = e / gpr;
        const int grp = (int)(e - row * gpr);
        float* dst = cols + row * g.ldc;
        if (grp == g.C * kk) {
            for (int col = ncol; col < g.ldc; ++col) dst[col] = 0.f;
            continue;
        }
        const int c = grp / kk, t = grp - c * kk, kd = t / g.k, kh = t - kd * g.k;
        long r = row;
        const int ow = (int)(r % g.Wo); r /= g.Wo;
        const int oh = (int)(r % g.Ho); r /= g.Ho;
        const int od = (int)(r % g.Do);
        const long b = r / g.Do;
        const int d = od * g.stride - g.pad + kd, h = oh * g.stride - g.pad + kh, w0 = ow * g.stride - g.pad;
        const bool ok = d >= 0 && d < g.D && h >= 0 && h < g.H;
        const float* src = x + b * g.sb + c * g.sc + (long)d * g.sd + (long)h * g.sh;
        dst += c * k3 + kd * kk + kh * g.k;
        for (int kw = 0; kw < g.k; ++kw) {
            const int w = w0 + kw;
            dst[kw] = (ok && w >= 0 && w < g.W) ? src[(long)w * g.sw] : 0.f;
        }
    }
}

// dx[b,c,d,h,w] = sum over the (output position, tap) pairs that read it -- a gather per input element
// (deterministic, no atomics): tap kd contributes iff (d + pad - kd) is a non-negative multiple of stride below Do.
__global__ __launch_bounds__(256) void col2im3d_kernel(const float* __restrict__ dcols, Conv3dGeom g, long total,
                                                        const float* __restrict__ y_tanh, int act, float* __restrict__ dx) {
    const int k3 = g.k * g.k * g.k, kk = g.k * g.k;
#pragma unroll 4                                   // four independent elements in flight per thread: the chain decode -> load ->
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {   // load -> store is latency-bound
        // e enumerates (b, d, h, w, c) with c fastest: consecutive threads read consecutive patch columns' channels
        long r = e;
        const int c = (int)(r % g.C); r /= g.C;
        const int w = (int)(r % g.W); r /= g.W;
        const int h = (int)(r % g.H); r /= g.H;
        const int d = (int)(r % g.D);
        const long b = r / g.D;
        float s = 0.f;
        // only the taps congruent to (x + pad) mod stride can have read this element: start there and step by the stride
        // (with stride == k, the shipped layers, that is ONE tap per dimension instead of k tests); same summation order
        const int pd = d + g.pad, ph = h + g.pad, pw = w + g.pad;
        for (int kd = pd % g.stride; kd < g.k && kd <= pd; kd += g.stride) {
            const int od = (pd - kd) / g.stride;
            if (od >= g.Do) continue;
            for (int kh = ph % g.stride; kh < g.k && kh <= ph; kh += g.stride) {
                const int oh = (ph - kh) / g.stride;
                if (oh >= g.Ho) continue;
                for (int kw = pw % g.stride; kw < g.k && kw <= pw; kw += g.stride) {
                    const int ow = (pw - kw) / g.stride;
                    if (ow >= g.Wo) continue;
                    const long row = ((b * g.Do + od) * g.Ho + oh) * g.Wo + ow;
                    s += dcols[row * g.ldc + c * k3 + kd * kk + kh * g.k + kw];
                }
            }
        }
        const long at = b * g.sb + c * g.sc + d * g.sd + h * g.sh + w * g.sw;
        if (y_tanh) {                                       // x is the previous layer's activation output: fold its derivative in
            const float y = y_tanh[at];
            s *= (act == PM_ACT_TANH) ? (1.0f - y * y) : pm_dact(y, act);
        }
        dx[at] = s;
    }
}

static int conv3d_geom(Conv3dGeom& g, int C, int D, int H, int W, int k, int stride, int pad, long sb, long sc, long sd,
                       long sh, long sw, int ldc) {
    if (C <= 0 || D <= 0 || H <= 0 || W <= 0 || k <= 0 || stride <= 0 || pad < 0) return PM_EINVAL;
    g.C = C; g.D = D; g.H = H; g.W = W; g.k = k; g.stride = stride; g.pad = pad;
    g.Do = (D + 2 * pad - k) / stride + 1;
    g.Ho = (H + 2 * pad - k) / stride + 1;
    g.Wo = (W + 2 * pad - k) / stride + 1;
    g.sb = sb; g.sc = sc; g.sd = sd; g.sh = sh; g.sw = sw; g.ldc = ldc;
    if (g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0 || ldc < C * k * k * k) return PM_EINVAL;
    return PM_OK;
}

extern "C" int pm_im2col3d_f32(const float* x, int B, int C, int D, int H, int W, int k, int stride, int pad, long sb,
                               long sc, long sd, long sh, long sw, float* cols, int ldc, void* stream) {
    PM_REQUIRE(x && cols && B > 0);
    Conv3dGeom g;
    const int rc = conv3d_geom(g, C, D, H, W, k, stride, pad, sb, sc, sd, sh, sw, ldc);
    if (rc != PM_OK) return rc;
    const long total = (long)B * g.Do * g.Ho * g.Wo * (C * k * k + 1);
    long nb = (total + 255) / 256;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(im2col3d_kernel, dim3((unsigned)nb), dim3(256), 0, pm_stream(stream), x, g, total, cols);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

extern "C" int pm_col2im3d_f32(const float* dcols, int B, int C, int D, int H, int W, int k, int stride, int pad,
                               long sb, long sc, long sd, long sh, long sw, const float* y_tanh, int act, float* dx, int ldc,
                               void* stream) {
    PM_REQUIRE(dcols && dx && B > 0);
    PM_REQUIRE(!y_tanh || (act > PM_ACT_NONE && act <= PM_ACT_MAX));
    Conv3dGeom g;
    const int rc = conv3d_geom(g, C, D, H, W, k, stride, pad, sb, sc, sd, sh, sw, ldc);
    if (rc != PM_OK) return rc;
    const long total = (long)B * C * D * H * W;
    long nb = (total + 255) / 256;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(col2im3d_kernel, dim3((unsigned)nb), dim3(256), 0, pm_stream(stream), dcols, g, total, y_tanh, act, dx);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ---------------------------------------------------------------------------------- direct conv of the input layer
// Conv3DNet's first layer (network.py:72,119: Conv3d(1, 16, k=5, stride 3, pad 2) on the 50^3 TSDF) dominated the shipped
// DAgger step through its patch matrix: 4 GB per mini-batch of 1600 volumes, written once (im2col, 3.7 ms) and read
// twice (forward GEMM 2.1 ms, weight-gradient GEMM 1.3 ms).  With ONE input channel the layer is a 125-tap stencil,
// so both directions run straight from the volume (0.8 GB, re-read through L1/L2):
//   forward   one thread per output position, all CO filters in registers; the 125 x CO weights come TAP-major
//             (wt[tap][o]) so that a tap's CO weights are one wave-uniform (scalar) load; bias + tanh fused;
//   gradient  dW[o][tap] = sum_rows dz[row][o] * x[patch(row)][tap]: lane = tap (125 of 128 lanes; lane 125 sees the
//             constant 1 and so accumulates the bias gradient), the row's CO dz values are wave-uniform, CO
//             accumulators per lane; work-groups own contiguous row slices, slabs are added in fixed order.
// The input volume is data, so this layer has no input gradient.
template <int K, int CO>
__global__ __launch_bounds__(256) void conv3d_c1_fwd_kernel(const float* __restrict__ x, Conv3dGeom g,
                                                             const float* __restrict__ wt,
                                                             const float* __restrict__ bias, int act_tanh, long nrows,
                                                             float* __restrict__ y, long ldy,
                                                             const int64_t* __restrict__ bidx, int vec4) {
    // one row per thread, no grid-stride loop: inside a loop the 125 x CO wave-uniform weight loads are loop-invariant
    // and hipcc hoists all of them (2010 spilled SGPRs)
    {
        const long row = (long)blockIdx.x * 256 + threadIdx.x;
        if (row >= nrows) return;
        long r = row;
        const int ow = (int)(r % g.Wo); r /= g.Wo;
        const int oh = (int)(r % g.Ho); r /= g.Ho;
        const int od = (int)(r % g.Do);
        const long b = r / g.Do;
        const int d0 = od * g.stride - g.pad, h0 = oh * g.stride - g.pad, w0 = ow * g.stride - g.pad;
        const float* xb = x + (bidx ? bidx[b] : b) * g.sb;        // bidx: batch row b is row bidx[b] of a larger store (ring)
        float acc[CO];
#pragma unroll
        for (int o = 0; o < CO; ++o) acc[o] = bias ? bias[o] : 0.f;
#pragma unroll 1                                   // ROLLED over (kd, kh): a tap row's 5 x CO wave-uniform weights fit
        for (int kd = 0; kd < K; ++kd) {               // the SGPR file; fully unrolled, hipcc hoists all 125 x CO scalar
            const int d = d0 + kd;                     // loads to the top and spills 1900 SGPRs
            const bool okd = d >= 0 && d < g.D;
#pragma unroll 1
            for (int kh = 0; kh < K; ++kh) {
                const int h = h0 + kh;
                const bool ok = okd && h >= 0 && h < g.H;
                const float* src = xb + (long)d * g.sd + (long)h * g.sh;
#pragma unroll
                for (int kw = 0; kw < K; ++kw) {
                    const int w = w0 + kw;
                    const float v = (ok && w >= 0 && w < g.W) ? src[(long)w * g.sw] : 0.f;
                    const float* wp = wt + ((kd * K + kh) * K + kw) * CO;
#pragma unroll
                    for (int o = 0; o < CO; ++o) acc[o] = fmaf(wp[o], v, acc[o]);
                }
            }
        }
        float* yr = y + row * ldy;
        if (act_tanh) {
#pragma unroll
            for (int o = 0; o < CO; o += 2) {
                const f32x2 t = pm_tanh2(acc[o], acc[o + 1]);
                acc[o] = t.x;
                acc[o + 1] = t.y;
            }
        }
        if (vec4) {                                         // the row's 64 bytes as four 16-byte stores instead of sixteen
#pragma unroll
            for (int o = 0; o < CO; o += 4) *(float4*)(yr + o) = make_float4(acc[o], acc[o + 1], acc[o + 2], acc[o + 3]);
        } else {
#pragma unroll
            for (int o = 0; o < CO; ++o) yr[o] = acc[o];
        }
    }
}

#define C1_WG_SLABS 2048
#define C1_MAXW 64                                           // widest volume row the LDS slab holds
// Weight gradient of the input layer on the matrix pipe: dW^T (128 taps x 16 filters) += patch^T (128 x 4 rows) * dz (4 rows
// x 16) is exactly v_mfma_f32_16x16x4_f32 -- four consecutive output positions of a LINE (fixed b, od, oh; ow = 0..Wo-1) per
// instruction, the tap axis in eight 16-row blocks (125 taps, tap 125 = the constant 1 that collects the bias gradient, two
// idle).  Work-groups own contiguous runs of lines.  A line's patches all lie in the K x K input rows
// x[d0..d0+K)[h0..h0+K)[*]: they are staged in LDS by coalesced loads (zero-filled outside the volume), lane (tap, position)
// reads its value from there -- gathering from global memory, 25-50 cache lines per wave load, was bound by the texture
// addresser.  4 waves share the slab, wave w owns tap blocks 2w, 2w+1 (two accumulators of 4 registers); lines are padded to
// NG groups of 4 positions (the dz of positions >= Wo is 0).  Per-work-group partial sums are added in fixed order.
// History (1600 volumes of 50^3, tools/time_conv3d.py): the round-1 FMA form (lane = tap, 16 FMAs per row and lane with the
// row's dz as ONE wave-uniform scalar load from HBM per row, next line's rows one iteration ahead) 1.49 ms; this kernel with
// the same one-line look-ahead 1.46 (an iteration is ~320 cycles of MFMA per wave, far shorter than an HBM round trip: both ran
// at load latency); two lines ahead 1.17.  Ablations of that: volume loads + LDS stores 0.51, dz loads 0.27 (every wave
// fetches the line's dz), MFMAs + LDS reads 0.31, the rest 0.11 -- they add up: issue-bound.  Measured slower: whole slab rows
// per wave with wave-uniform row arithmetic (1.64: 54 of 64 lanes, twice the load slots), whole rounds of resident groups (no
// change).
typedef float c1_f32x4 __attribute__((ext_vector_type(4)));
#ifndef C1_ABLATE
#define C1_ABLATE 0              // timing probes (wrong results): 1 = no volume loads, 2 = no dz loads, 4 = no MFMAs / LDS reads, 8 = no LDS stores
#endif
template <int K, int CO, int NG>
__global__ __launch_bounds__(256) void conv3d_c1_wgrad_mfma_kernel(const float* __restrict__ dz, long lddz,
                                                                    const float* __restrict__ x, Conv3dGeom g, long nlines,
                                                                    long lines_per_wg, float* __restrict__ slabs,
                                                                    const int64_t* __restrict__ bidx) {
    constexpr int K3 = K * K * K, SW = C1_MAXW + 2 * K;      // NG groups of 4 output positions per line
    constexpr int NST = (K * K * SW + 255) / 256;
    static_assert(CO == 16 && K3 <= 125, "one 16-column MFMA block of filters, taps + bias lane within 128");
    __shared__ float slab[3][K * K][SW];                    // line L lives in buffer L % 3
    const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int WP = g.W + 2 * g.pad;
    // staging map: which (slab row, padded column) this thread copies in slot j (fixed for the whole kernel)
    int srow[NST], scol[NST];
#pragma unroll
    for (int j = 0; j < NST; ++j) {
        const int e = tid + 256 * j;
        srow[j] = e < K * K * WP ? e / WP : -1;
        scol[j] = e - (e / WP) * WP;
    }
    // this lane's two taps and their slab offsets
    int aoff[2];
    float aconst[2];                                        // tap 125: 1 (bias), 126 / 127: 0
    bool areal[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int tap = (wave * 2 + i) * 16 + l16;
        areal[i] = tap < K3;
        aconst[i] = tap == K3 ? 1.0f : 0.0f;
        const int t = areal[i] ? tap : 0;
        aoff[i] = ((t / (K * K)) * K + (t / K) % K) * SW + t % K;
    }
    const long l0 = (long)blockIdx.x * lines_per_wg;
    long l1 = l0 + lines_per_wg;
    if (l1 > nlines) l1 = nlines;
    c1_f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // (b, od, oh) of the line whose loads are issued next, advanced by carry (wave-uniform, no divisions in the loop)
    long rr = l0;
    int oh = (int)(rr % g.Ho); rr /= g.Ho;
    int od = (int)(rr % g.Do);
    long b = rr / g.Do;
    auto advance = [&]() __attribute__((always_inline)) {
        if (++oh == g.Ho) {
            oh = 0;
            if (++od == g.Do) {
                od = 0;
                ++b;
            }
        }
    };
    // Two lines in flight: line L's volume rows and dz are REQUESTED at the top of iteration L-2 (register set L % 2), the rows
    // go to LDS at the end of iteration L-1 and are consumed in iteration L.  (One line ahead, an iteration -- 320 cycles of
    // MFMA per wave -- is far shorter than an HBM round trip: both this kernel and the FMA form then run at load latency.)
    float st[2][NST], bz[3][NG];
    auto loads = [&](long line, float (&sx)[NST], float (&sz)[NG]) __attribute__((always_inline)) {
        const float* xb = x + (bidx ? bidx[b] : b) * g.sb;
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            const int d = od * g.stride - g.pad + srow[j] / K, h = oh * g.stride - g.pad + srow[j] % K, w = scol[j] - g.pad;
            const bool ok = srow[j] >= 0 && d >= 0 && d < g.D && h >= 0 && h < g.H && w >= 0 && w < g.W;
            sx[j] = (ok && !(C1_ABLATE & 1)) ? xb[(long)d * g.sd + (long)h * g.sh + (long)w * g.sw] : 0.f;
        }
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            const int ow = 4 * gq + q;
            sz[gq] = (ow < g.Wo && !(C1_ABLATE & 2)) ? dz[(line * g.Wo + ow) * lddz + l16] : 0.f;
        }
        advance();
    };
    auto stage_store = [&](int buf, const float (&sx)[NST]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NST; ++j)
            if (srow[j] >= 0 && !(C1_ABLATE & 8)) slab[buf][srow[j]][scol[j]] = sx[j];
    };
    // prologue: line l0 loaded and stored, line l0+1 loaded
    if (l0 < l1) {
        loads(l0, st[0], bz[0]);
        stage_store(0, st[0]);
        if (l0 + 1 < l1) loads(l0 + 1, st[1], bz[1]);
    }
    __syncthreads();
    // the loop is unrolled by 6 = lcm(2 register sets, 3 buffers) so that every set / buffer index is a constant
    long line = l0;
    while (line < l1) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            if (line < l1) {                                // wave-uniform
                if (line + 2 < l1) loads(line + 2, st[u % 2], bz[(u + 2) % 3]);
                const float* sp = &slab[u % 3][0][0];
#pragma unroll
                for (int gq = 0; gq < ((C1_ABLATE & 4) ? 0 : NG); ++gq) {
                    int ow = 4 * gq + q;
                    if (ow >= g.Wo) ow = g.Wo - 1;          // padded positions: dz is 0, any finite patch value will do
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const float a = areal[i] ? sp[aoff[i] + ow * g.stride] : aconst[i];
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bz[u % 3][gq], acc[i], 0, 0, 0);
                    }
                }
                if (line + 1 < l1) stage_store((u + 1) % 3, st[(u + 1) % 2]);
                __syncthreads();
                ++line;
            }
        }
    }
    // D block i: rows (taps) (2 wave + i) 16 + 4 q + r, column (filter) l16
    float* out = slabs + (size_t)blockIdx.x * CO * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[l16 * 128 + (wave * 2 + i) * 16 + 4 * q + r] = acc[i][r];
}

// dW[o][tap] = sum_slabs (fixed order), db[o] from lane K^3
__global__ __launch_bounds__(128) void conv3d_c1_wgrad_reduce_kernel(const float* __restrict__ slabs, int nslabs, int CO,
                                                                      int K3, float* __restrict__ dW, long lddw,
                                                                      float* __restrict__ db) {
    const int o = blockIdx.x, tap = threadIdx.x;
    float s = 0.f;
#pragma unroll 8
    for (int z = 0; z < nslabs; ++z) s += slabs[((size_t)z * CO + o) * 128 + tap];
    if (tap < K3) dW[o * lddw + tap] = s;
    else if (tap == K3 && db) db[o] = s;
}

extern "C" int pm_conv3d_c1_supported(int k, int Cout) { return k == 5 && Cout == 16; }
extern "C" size_t pm_conv3d_c1_wgrad_workspace_bytes(int Cout) { return (size_t)C1_WG_SLABS * Cout * 128 * sizeof(float); }

extern "C" int pm_conv3d_c1_fwd_f32(const float* x, int B, int D, int H, int W, int k, int stride, int pad, long sb,
                                    long sd, long sh, long sw, const float* wt, const float* bias, int Cout, int act,
                                    float* y, long ldy, const int64_t* batch_index, void* stream) {
    PM_REQUIRE(x && wt && y && B > 0 && ldy >= Cout);
    if (!pm_conv3d_c1_supported(k, Cout)) return PM_EUNSUPPORTED;
    if (act != PM_ACT_NONE && act != PM_ACT_TANH) return PM_EUNSUPPORTED;
    Conv3dGeom g;
    const int rc = conv3d_geom(g, 1, D, H, W, k, stride, pad, sb, 0, sd, sh, sw, k * k * k);
    if (rc != PM_OK) return rc;
    const long nrows = (long)B * g.Do * g.Ho * g.Wo;
    const long nb = (nrows + 255) / 256;
    if (nb > 0x7fffffffL) return PM_EINVAL;
    hipLaunchKernelGGL((conv3d_c1_fwd_kernel<5, 16>), dim3((unsigned)nb), dim3(256), 0, pm_stream(stream), x, g, wt, bias,
                       act == PM_ACT_TANH, nrows, y, ldy, batch_index, (int)(ldy % 4 == 0 && ((uintptr_t)y & 15) == 0));
    PM_CHECK_LAUNCH();
    return PM_OK;
}

extern "C" int pm_conv3d_c1_wgrad_f32(const float* dz, long lddz, const float* x, int B, int D, int H, int W, int k,
                                      int stride, int pad, long sb, long sd, long sh, long sw, int Cout, float* dW,
                                      long lddw, float* db, const int64_t* batch_index, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    PM_REQUIRE(dz && x && dW && B > 0 && lddz >= Cout && lddw >= k * k * k);
    if (!pm_conv3d_c1_supported(k, Cout)) return PM_EUNSUPPORTED;
    if (!workspace || workspace_bytes < pm_conv3d_c1_wgrad_workspace_bytes(Cout)) return PM_EWORKSPACE;
    Conv3dGeom g;
    const int rc = conv3d_geom(g, 1, D, H, W, k, stride, pad, sb, 0, sd, sh, sw, k * k * k);
    if (rc != PM_OK) return rc;
    if (W > C1_MAXW) return PM_EUNSUPPORTED;
    const long nlines = (long)B * g.Do * g.Ho;
    long per = (nlines + C1_WG_SLABS - 1) / C1_WG_SLABS;
    if (per < 4) per = 4;
    const int nwg = (int)((nlines + per - 1) / per);
#define C1_WGRAD_LAUNCH(NG_)                                                                                             \
    hipLaunchKernelGGL((conv3d_c1_wgrad_mfma_kernel<5, 16, NG_>), dim3(nwg), dim3(256), 0, pm_stream(stream), dz, lddz, x, g, \
                       nlines, per, (float*)workspace, batch_index)
    if (g.Wo <= 20) C1_WGRAD_LAUNCH(5);                     // the shipped geometry: 17 positions per line
    else if (g.Wo <= 32) C1_WGRAD_LAUNCH(8);
    else if (g.Wo <= 64) C1_WGRAD_LAUNCH(16);
    else return PM_EUNSUPPORTED;
#undef C1_WGRAD_LAUNCH
    hipLaunchKernelGGL(conv3d_c1_wgrad_reduce_kernel, dim3(Cout), dim3(128), 0, pm_stream(stream),
                       (const float*)workspace, nwg, Cout, k * k * k, dW, lddw, db);
    PM_CHECK_LAUNCH();
    return PM_OK;
}

// ---------------------------------------------------------------------------------- TSDF integration
// utils/depth2tsdf.py:68-86 `TSDFVolume.integrate`: every voxel looks up its (precomputed, registration-time)
// pixel in every view, turns the depth difference into a truncated signed distance and averages the views
// that see it.  One thread per (env, voxel); the per-view tables (pixel index or -1, camera-frame z) are shared by
// all envs and stay in L2.  Rounding follows the reference's tensor expression op by op:
//   tsdf_m = min(diff / trunc, 1);  w = 1 / n_valid;  vol = sum_m (tsdf_m * w)  [views in order]  + default * [n_valid == 0]
__global__ __launch_bounds__(256) void tsdf_integrate_kernel(const float* __restrict__ depth,
                                                              const int32_t* __restrict__ pix_idx,
                                                              const float* __restrict__ pix_z, int M, long HW, long V,
                                                              float trunc, float default_tsdf, long total,
                                                              float* __restrict__ out) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long b = e / V, v = e - b * V;
        int n_valid = 0;
        for (int m = 0; m < M; ++m) {
            const int pi = pix_idx[m * V + v];
            if (pi < 0) continue;
            const float d = depth[(b * M + m) * HW + pi];
            const float diff = sub_rn(d, pix_z[m * V + v]);
            n_valid += (d > 0.f && diff >= -trunc) ? 1 : 0;
        }
        float acc = 0.f;
        if (n_valid > 0) {
            const float w = __fdiv_rn(1.0f, (float)n_valid);
            for (int m = 0; m < M; ++m) {
                const int pi = pix_idx[m * V + v];
                float term = 0.f;                              // invalid views contribute tsdf * 0
                if (pi >= 0) {
                    const float d = depth[(b * M + m) * HW + pi];
                    const float diff = sub_rn(d, pix_z[m * V + v]);
                    if (d > 0.f && diff >= -trunc) term = mul_rn(fminf(__fdiv_rn(diff, trunc), 1.0f), w);
                }
                acc = add_rn(acc, term);
            }
        } else {
            acc = default_tsdf;
        }
        out[e] = acc;
    }
}

extern "C" int pm_tsdf_integrate_f32(const float* depth, const int32_t* pix_idx, const float* pix_z, int B, int M,
                                     long HW, long V, float trunc, float default_tsdf, float* out, void* stream) {
    PM_REQUIRE(depth && pix_idx && pix_z && out && B > 0 && M > 0 && HW > 0 && V > 0 && trunc > 0.f);
    const long total = (long)B * V;
    long nb = (total + 255) / 256;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(tsdf_integrate_kernel, dim3((unsigned)nb), dim3(256), 0, pm_stream(stream), depth, pix_idx, pix_z,
                       M, HW, V, trunc, default_tsdf, total, out);
    PM_CHECK_LAUNCH();
    return PM_OK;
}
