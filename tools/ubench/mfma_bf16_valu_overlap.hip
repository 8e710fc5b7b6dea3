// Micro-benchmark: do bf16 MFMA work and plain VALU work overlap on a gfx950 SIMD -- across two waves, and inside one wave?
// One work-group of 8 waves per CU (2 waves per SIMD).  Modes:
//   0  every wave: MFMA loop                      1  every wave: VALU loop
//   2  waves 0-3 MFMA, waves 4-7 VALU (partners w, w + 4 on one SIMD do different work)
//   3  every wave alternates chunks  [6 MFMA][48 VALU]  in the SAME order (lock-step)
//   4  as 3, waves 4-7 start with the VALU chunk (skewed by one chunk)
//   5  every wave: 1 MFMA, 8 VALU, 1 MFMA, 8 VALU ... (fine interleave inside the wave)
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_bf16_valu_overlap.hip -o gpurun_ab/mfma_bf16_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define FENCE __builtin_amdgcn_sched_barrier(0)

__global__ __launch_bounds__(512, 2) void k(int mode, int iters, float* out, int* simd_of_wave) {
    const int wave = threadIdx.x >> 6;
    if (simd_of_wave && blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        simd_of_wave[wave] = (hw >> 4) & 3;
    }
    uint4 ua = make_uint4(threadIdx.x, 0x3f803f80u, 0x3f803f80u, threadIdx.x * 3u), ub = make_uint4(0x3f803f80u, threadIdx.x, 7u, 0x3f803f80u);
    bf16x8 a = *(bf16x8*)&ua, b = *(bf16x8*)&ub;
    f32x16 c0 = {0}, c1 = {0};
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = (float)threadIdx.x + j;
#define MCHUNK c0 = MF(a, b, c0); c1 = MF(b, a, c1); c0 = MF(a, b, c0); c1 = MF(b, a, c1); c0 = MF(a, b, c0); c1 = MF(b, a, c1);
#define VCHUNK _Pragma("unroll") for (int r = 0; r < 3; ++r) _Pragma("unroll") for (int j = 0; j < 16; ++j) v[j] = fmaf(v[j], 1.0001f, 0.5f);
    const bool upper = wave >= 4;
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        if (mode == 0 || (mode == 2 && !upper)) { MCHUNK FENCE; MCHUNK FENCE; }
        else if (mode == 1 || (mode == 2 && upper)) { VCHUNK FENCE; VCHUNK FENCE; }
        else if (mode == 3 || (mode == 4 && !upper)) { MCHUNK FENCE; VCHUNK FENCE; }
        else if (mode == 4) { VCHUNK FENCE; MCHUNK FENCE; }
        else {
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                if (m & 1) c1 = MF(b, a, c1); else c0 = MF(a, b, c0);
                FENCE;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[(m * 8 + j) & 15] = fmaf(v[(m * 8 + j) & 15], 1.0001f, 0.5f);
                FENCE;
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += v[j] + c0[j] + c1[j];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    float* out;
    int* simd;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipMalloc(&simd, 8 * sizeof(int));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 6; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, out, simd);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        int h[8];
        hipMemcpy(h, simd, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d: %.3f ms   SIMD of waves 0-7:", mode, best);
        for (int w = 0; w < 8; ++w) printf(" %d", h[w]);
        printf("\n");
    }
    // per wave and iteration: mode 0: 12 MFMAs x 32 clk = 384 clk (x 2 waves per SIMD = 768); mode 1: 96 FMAs x 4 clk = 384 (x 2 = 768);
    // modes 3-5: 6 MFMAs (192 clk) + 48 FMAs (192 clk) per wave: serialised 768 per SIMD, perfectly overlapped 384
    return 0;
}
