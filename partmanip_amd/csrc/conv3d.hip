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
                                                        const float* __restrict__ y_tanh, float* __restrict__ dx) {
    const int k3 = g.k * g.k * g.k, kk = g.k * g.k;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        // e enumerates (b, d, h, w, c) with c fastest: consecutive threads read consecutive patch columns' channels
        long r = e;
        const int c = (int)(r % g.C); r /= g.C;
        const int w = (int)(r % g.W); r /= g.W;
        const int h = (int)(r % g.H); r /= g.H;
        const int d = (int)(r % g.D);
        const long b = r / g.D;
        float s = 0.f;
        for (int kd = 0; kd < g.k; ++kd) {
            const int td = d + g.pad - kd;
            if (td < 0 || td % g.stride) continue;
            const int od = td / g.stride;
            if (od >= g.Do) continue;
            for (int kh = 0; kh < g.k; ++kh) {
                const int th = h + g.pad - kh;
                if (th < 0 || th % g.stride) continue;
                const int oh = th / g.stride;
                if (oh >= g.Ho) continue;
                for (int kw = 0; kw < g.k; ++kw) {
                    const int tw = w + g.pad - kw;
                    if (tw < 0 || tw % g.stride) continue;
                    const int ow = tw / g.stride;
                    if (ow >= g.Wo) continue;
                    const long row = ((b * g.Do + od) * g.Ho + oh) * g.Wo + ow;
                    s += dcols[row * g.ldc + c * k3 + kd * kk + kh * g.k + kw];
                }
            }
        }
        const long at = b * g.sb + c * g.sc + d * g.sd + h * g.sh + w * g.sw;
        if (y_tanh) {                                       // x is the previous layer's tanh output: fold its derivative in
            const float y = y_tanh[at];
            s *= (1.0f - y * y);
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
                               long sb, long sc, long sd, long sh, long sw, const float* y_tanh, float* dx, int ldc,
                               void* stream) {
    PM_REQUIRE(dcols && dx && B > 0);
    Conv3dGeom g;
    const int rc = conv3d_geom(g, C, D, H, W, k, stride, pad, sb, sc, sd, sh, sw, ldc);
    if (rc != PM_OK) return rc;
    const long total = (long)B * C * D * H * W;
    long nb = (total + 255) / 256;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(col2im3d_kernel, dim3((unsigned)nb), dim3(256), 0, pm_stream(stream), dcols, g, total, y_tanh, dx);
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
