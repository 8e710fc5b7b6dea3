// Micro-benchmark: do fp32 MFMA work and plain VALU work of DIFFERENT waves on one SIMD overlap on gfx950?
// One work-group of 16 waves per CU (4 waves per SIMD).  Modes: 0 = every wave runs the MFMA loop, 1 = every wave
// runs the VALU loop, 2 = waves 0-7 MFMA / 8-15 VALU (two of each per SIMD), 3 = one wave per SIMD MFMA + three VALU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_overlap.hip -o gpurun_ab/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void mfma_loop(int iters, float seed, float* out) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = seed, y = seed * 0.5f;
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    out[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
__device__ __forceinline__ void valu_loop(int iters, float seed, float* out) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = seed + j;
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = fmaf(v[j], 1.0001f, 0.5f);      // 64 independent-ish FMAs per trip
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += v[j];
    out[threadIdx.x] = s;
}

__device__ __forceinline__ void int_loop(int iters, float seed, float* out) {
    unsigned v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = (unsigned)seed + j;
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = (v[j] ^ 0x9E3779B9u) + (v[j] >> 3);     // 3 integer VALU ops each
    }
    unsigned s = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += v[j];
    out[threadIdx.x] = (float)s;
}
__device__ __forceinline__ void exp_loop(int iters, float seed, float* out) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = seed * 1e-3f + j;
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v[j]));   // 2 transcendental ops each
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += v[j];
    out[threadIdx.x] = s;
}

__global__ __launch_bounds__(1024, 1) void k(int mode, int it_m, int it_v, float* out, int* simd_of_wave) {
    const int wave = threadIdx.x >> 6;
    if (simd_of_wave && blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        simd_of_wave[wave] = (hw >> 4) & 3;                 // HW_ID[5:4] = SIMD the wave runs on
    }
    float* o = out + (size_t)blockIdx.x * 1024;
    bool do_m;
    if ((mode & 15) == 0) do_m = true;
    else if ((mode & 15) == 1) do_m = false;
    else if ((mode & 15) == 2) do_m = wave < 8;           // waves w -> SIMD w % 4: two MFMA + two VALU waves per SIMD
    else do_m = wave < 4;                          // one MFMA + three VALU waves per SIMD
    const int kind = mode >> 4;                    // VALU flavour: 0 = fp32 FMA, 1 = integer, 2 = exp2 + rcp
    if (do_m) mfma_loop(it_m, (float)threadIdx.x, o);
    else if (kind == 0) valu_loop(it_v, (float)threadIdx.x, o);
    else if (kind == 1) int_loop(it_v, (float)threadIdx.x, o);
    else exp_loop(it_v, (float)threadIdx.x, o);
}

int main() {
    float* out;
    int* simd;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    hipMalloc(&simd, 16 * sizeof(int));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int it_m = 4000, it_v = 4000;            // per wave: 16000 MFMAs (x64 clk) / 256000 VALU ops (x4 clk): ~1 M clk each
    const int modes[] = {0, 1, 2, 3, 16 + 1, 16 + 2, 32 + 1, 32 + 2};
    for (int mi = 0; mi < 8; ++mi) {
        const int mode = modes[mi];
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, mode, it_m, it_v, out, simd);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        int h[16];
        hipMemcpy(h, simd, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (VALU kind %d): %.3f ms   SIMD of waves 0-15:", mode & 15, mode >> 4, best);
        for (int w = 0; w < 16; ++w) printf(" %d", h[w]);
        printf("\n");
    }
    // expectations at 2.4 GHz: mode 0 = 4 waves x 16000 MFMA x 64 clk = 4.1 M clk = 1.71 ms; mode 1 = 4 x 256000 x 4 clk = 4.1 M clk = 1.71 ms
    // mode 2 perfect overlap = max(2 x 1.02 M, 2 x 1.02 M clk) = 0.85 ms; fully serialised = 1.71 ms
    return 0;
}
