// Micro-benchmark: what does an operand load cost a SIMD that is busy with fp32 MFMAs?  (gfx950)
// One work-group of 16 waves per CU (4 per SIMD), every wave runs  R x { 4 MFMAs ; L loads }  on private accumulators:
//   kind 0: v_mfma_f32_16x16x4_f32 (32 cycles each), kind 1: v_mfma_f32_32x32x2_f32 (64 cycles each)
//   load 0: none, 1: global_load_dwordx4 from a 64 KB window (L1 / L2 resident), 2: ds_read_b128
// The loads are issued right after the MFMA group and consumed (one add into a sink) two groups later, so no wave ever
// waits for data: what shows up in cycles per MFMA is ISSUE cost (each load also brings 1 v_add + 1 v_and address op and
// 3 v_add for the sink: ~5 plain VALU slots, 20 cycles, are part of the per-load figure).  Prints cycles per MFMA and the extra cycles per load.
// Measured (MI355X, 256 work-groups): 32.5 / 64.0 cycles per bare 16x16x4 / 32x32x2 MFMA per SIMD; one global_load_dwordx4 per
// four MFMAs adds 32-39 cycles per load, of which its VALU companions are 11-18: the load instruction itself takes ~21-23
// cycles out of the SIMD's MFMA issue; a ds_read_b128 ~17.  Neither waits for data.  (pn_bwd16_kernel's dh1 loop: 27 per
// B-fragment load with its 64-bit address add -- DESIGN.md 3.2.)  By width (net of the companions): ds_read_b32 5.6, b64 10.8,
// b128 15.5; global_load_dword 7.1, x2 14.5, x4 21.3 -- about 4-5 cycles per VGPR a load writes, whichever path it takes.
// Build: hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/ubench/mfma_vmem_issue.hip -o gpurun_ab/mfma_vmem_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int LOAD, int L>
__global__ __launch_bounds__(1024, 1) void k(int iters, const float4* __restrict__ g, float* out, long long* cyc, int stride) {
    __shared__ float4 lds[4096];
    __shared__ unsigned long long tmin, tmax;      // a SIMD serves its waves oldest first: time the LAST wave, not wave 0
    if (threadIdx.x == 0) { tmin = ~0ull; tmax = 0; }
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 1024) lds[i] = make_float4(i, 1.f, 2.f, 3.f);
    __syncthreads();
    f32x16 a32[2] = {{0}, {0}};
    f32x4 a16[4] = {{0}, {0}, {0}, {0}};
    float x = 1.0f + lane * 1e-3f, y = 0.5f;
    float4 p0[L > 0 ? L : 1], p1[L > 0 ? L : 1];
    float sink = 0.f;
    const float4* gp = g + (blockIdx.x & 7) * 4096;                // 64 KB window per work-group: stays in L1 / L2
    unsigned ctr = tid;                                            // advanced by a run-time stride: nothing to hoist
#pragma unroll
    for (int j = 0; j < (L > 0 ? L : 1); ++j) p0[j] = p1[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < iters; i += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (KIND == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) a16[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a16[j], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) a32[j & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a32[j & 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            float4* dst = h ? p1 : p0;
#pragma unroll
            for (int j = 0; j < L; ++j) {
                sink += (dst[j].x + dst[j].y) + (dst[j].z + dst[j].w);   // consumes the load issued two groups ago
                ctr += stride;
                if (LOAD == 1) dst[j] = gp[ctr & 4095];
                if (LOAD == 2) dst[j] = lds[ctr & 4095];
                if (LOAD == 3) dst[j].x = __builtin_bit_cast(float, ctr & 4095);      // the companions alone: no load
                if (LOAD == 4) dst[j].x = ((const float*)lds)[ctr & 16383];            // ds_read_b32
                if (LOAD == 5) *(float2*)&dst[j] = ((const float2*)lds)[ctr & 8191];   // ds_read_b64
                if (LOAD == 6) dst[j].x = ((const float*)gp)[ctr & 16383];             // global_load_dword
                if (LOAD == 7) *(float2*)&dst[j] = ((const float2*)gp)[ctr & 8191];    // global_load_dwordx2
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = sink;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += a16[j][0];
    s += a32[0][0] + a32[1][1];
    out[blockIdx.x * 1024 + tid] = s;
    if (lane == 0) {
        atomicMin(&tmin, (unsigned long long)t0);
        atomicMax(&tmax, (unsigned long long)t1);
    }
    __syncthreads();
    if (tid == 0) cyc[blockIdx.x] = (long long)(tmax - tmin);
}

template <int KIND, int LOAD, int L>
static double run(int iters, const float4* g, float* out, long long* cyc, int nblk) {
    hipLaunchKernelGGL((k<KIND, LOAD, L>), dim3(nblk), dim3(1024), 0, 0, iters, g, out, cyc, 1024);
    hipLaunchKernelGGL((k<KIND, LOAD, L>), dim3(nblk), dim3(1024), 0, 0, iters, g, out, cyc, 1024);
    hipDeviceSynchronize();
    long long* h = (long long*)malloc(sizeof(long long) * nblk);
    hipMemcpy(h, cyc, sizeof(long long) * nblk, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < nblk; ++i) s += (double)h[i];
    free(h);
    // 4 waves per SIMD, each `iters` groups of 4 MFMAs: cycles per MFMA as the SIMD sees them
    return s / nblk / (4.0 * 4.0 * iters);
}

int main() {
    const int nblk = 256, iters = 4096;
    float4* g;
    float* out;
    long long* cyc;
    hipMalloc(&g, sizeof(float4) * 4096 * 8 + 65536);
    hipMemset(g, 0, sizeof(float4) * 4096 * 8 + 65536);
    hipMalloc(&out, sizeof(float) * 1024 * nblk);
    hipMalloc(&cyc, sizeof(long long) * nblk);
    const char* kn[2] = {"v_mfma_f32_16x16x4_f32", "v_mfma_f32_32x32x2_f32"};
#define ROW(KIND)                                                                                                      \
    {                                                                                                                  \
        const double b = run<KIND, 0, 0>(iters, g, out, cyc, nblk);                                                    \
        const double g1 = run<KIND, 1, 1>(iters, g, out, cyc, nblk), g2 = run<KIND, 1, 2>(iters, g, out, cyc, nblk);   \
        const double l1 = run<KIND, 2, 1>(iters, g, out, cyc, nblk), l2 = run<KIND, 2, 2>(iters, g, out, cyc, nblk);   \
        const double v1 = run<KIND, 3, 1>(iters, g, out, cyc, nblk), v2 = run<KIND, 3, 2>(iters, g, out, cyc, nblk);   \
        printf("%s: %.1f cycles per MFMA bare | + 1 global_load_dwordx4 per 4 MFMAs: %.1f (%.1f cycles per load) | + 2: %.1f " \
               "(%.1f) | + 1 ds_read_b128: %.1f (%.1f) | + 2: %.1f (%.1f) | the loads' VALU companions alone: %.1f (%.1f) / %.1f (%.1f)\n", \
               kn[KIND], b, g1, (g1 - b) * 4, g2, (g2 - b) * 2, l1, (l1 - b) * 4, l2, (l2 - b) * 2, v1, (v1 - b) * 4, v2,    \
               (v2 - b) * 2);                                                                                            \
    }
    ROW(0)
    ROW(1)
    {   // width sweep beside 32x32x2 MFMAs, two loads per four MFMAs, companions subtracted: cycles per load by bytes per lane
        const double b = run<1, 3, 2>(iters, g, out, cyc, nblk);
        printf("per load, net of the VALU companions: ds_read_b32 %.1f  b64 %.1f  b128 %.1f | global_load_dword %.1f  x2 %.1f  x4 %.1f\n",
               (run<1, 4, 2>(iters, g, out, cyc, nblk) - b) * 2, (run<1, 5, 2>(iters, g, out, cyc, nblk) - b) * 2,
               (run<1, 2, 2>(iters, g, out, cyc, nblk) - b) * 2, (run<1, 6, 2>(iters, g, out, cyc, nblk) - b) * 2,
               (run<1, 7, 2>(iters, g, out, cyc, nblk) - b) * 2, (run<1, 1, 2>(iters, g, out, cyc, nblk) - b) * 2);
    }
    return 0;
}
