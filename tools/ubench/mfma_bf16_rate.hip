// Micro-benchmark: sustained rate of back-to-back v_mfma_f32_32x32x16_bf16 (and, for comparison, v_mfma_f32_32x32x2_f32)
// with W waves per SIMD, operands in registers (no memory traffic).  Prints achieved TFLOP/s for the whole GPU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_bf16_rate.hip -o gpurun_ab/mfma_bf16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(1024) void k(int iters, float* out) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    bf16x8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(float)(threadIdx.x + j); y[j] = (__bf16)(float)(j + 1); }
    float xf = (float)threadIdx.x, yf = 0.5f;
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, a3, 0, 0, 0);
        } else {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(xf, yf, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(yf, xf, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(xf, xf, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(yf, yf, a3, 0, 0, 0);
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kind = 0; kind < 2; ++kind)
        for (int waves = 4; waves <= 16; waves *= 2) {              // waves per CU: 1, 2, 4 per SIMD
            const int iters = 20000;
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(waves * 64), 0, 0, iters, out);
                else hipLaunchKernelGGL(k<1>, dim3(256), dim3(waves * 64), 0, 0, iters, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            const double flops = (double)256 * waves * iters * 4 * (kind == 0 ? 32768.0 : 4096.0);
            printf("%s, %d wave(s) per SIMD: %.3f ms, %.0f TFLOP/s\n", kind == 0 ? "bf16 32x32x16" : "f32 32x32x2", waves / 4, best,
                   flops / best / 1e9);
        }
    return 0;
}
