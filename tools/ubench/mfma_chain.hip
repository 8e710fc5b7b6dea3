// Does a DEPENDENT chain of v_mfma_f32_32x32x2_f32 on ONE accumulator issue at the 64-cycle rate?  One wave per SIMD
// (256-thread work-groups, one per CU), NACC independent accumulators, 4096 MFMAs per wave.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_chain.hip -o mfma_chain && ./mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, float a0, float b0, unsigned long long* cyc) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 4096 / (8 * NACC); ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC, int WAVES>
void run(float* out, unsigned long long* cyc) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, 1.f, 2.f, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, 1.f, 2.f, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double flops = 256.0 * WAVES * 4096 * 2 * 32 * 32 * 2;
    printf("NACC=%d waves/CU=%d: %.1f us, %.1f TFLOP/s, %.1f counter ticks per MFMA (wave 0)\n", NACC, WAVES, ms * 1e3, flops / ms / 1e9,
           (double)c / 4096);
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    run<1, 4>(out, cyc); run<2, 4>(out, cyc); run<4, 4>(out, cyc);
    run<1, 8>(out, cyc); run<2, 8>(out, cyc); run<1, 16>(out, cyc);
    return 0;
}
