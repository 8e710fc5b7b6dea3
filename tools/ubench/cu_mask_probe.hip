// Which XCDs does a CU-masked stream run on, and does a hipGraph replayed into it keep the mask?  (actor on
// XCDs 0-3, critic on 4-7 through hipExtStreamCreateWithCUMask.)  Every work-group records its XCC id (HW_REG_XCC_ID, low 4 bits)
// and its CU id; the host prints the histogram per mask pattern, for a plain launch and for a captured-graph replay.
// build: hipcc -O2 --offload-arch=gfx950 tools/ubench/cu_mask_probe.hip -o gpurun_ab/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void probe(uint32_t* out, int spin) {
    uint32_t xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 15u) | (hwid << 8);
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);      // keep the CU busy so that all masked CUs get a work-group
}

static void histo(const char* tag, const std::vector<uint32_t>& v) {
    int h[16] = {0};
    for (uint32_t x : v) h[x & 15]++;
    printf("%-44s XCC histogram:", tag);
    for (int i = 0; i < 8; ++i) printf(" %4d", h[i]);
    // distinct (XCC, HW_ID CU / SH / SE fields) = distinct CUs that ran a work-group: HW_ID bits 8-11 CU, 12 SH, 13-15 SE (gfx9)
    std::vector<uint32_t> ids;
    for (uint32_t x : v) {
        const uint32_t hw = x >> 8, key = (x & 15u) | (((hw >> 8) & 0xffu) << 4);
        bool seen = false;
        for (uint32_t k : ids) seen = seen || k == key;
        if (!seen) ids.push_back(key);
    }
    int per[8] = {0};
    for (uint32_t k : ids) per[k & 7]++;
    printf("   distinct CUs %3zu (per XCC:", ids.size());
    for (int i = 0; i < 8; ++i) printf(" %d", per[i]);
    printf(")\n");
}

int main() {
    const int NB = 2048;
    setvbuf(stdout, nullptr, _IONBF, 0);
    uint32_t* d;
    CK(hipMalloc(&d, NB * 4));
    std::vector<uint32_t> h(NB);
    struct { const char* name; uint32_t m[8]; } masks[] = {
        {"bits i with (i % 8) < 4", {0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu}},
        {"bits i with (i % 8) >= 4", {0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u}},
        {"bits 0..127 (low half)", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0}},
        {"bits 128..255 (high half)", {0, 0, 0, 0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}},
        {"bits 0..31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}},
        {"bits i with (i % 8) == 0", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}},
    };
    for (auto& mk : masks) {
        hipStream_t s;
        printf("[%s] creating the masked stream\n", mk.name);
        CK(hipExtStreamCreateWithCUMask(&s, 8, mk.m));
        printf("  created\n");
        CK(hipMemsetAsync(d, 0xff, NB * 4, s));
        hipLaunchKernelGGL(probe, dim3(NB), dim3(64), 0, s, d, 200);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d, NB * 4, hipMemcpyDeviceToHost));
        char tag[96];
        snprintf(tag, sizeof tag, "%s: launch", mk.name);
        histo(tag, h);
        // the same launch captured into a graph and replayed into the masked stream
        static hipStream_t cap = nullptr;
        if (!cap) CK(hipStreamCreate(&cap));
        printf("  capturing\n");
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(probe, dim3(NB), dim3(64), 0, cap, d, 200);
        CK(hipStreamEndCapture(cap, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipMemsetAsync(d, 0xff, NB * 4, s));
        printf("  graph instantiated, launching\n");
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d, NB * 4, hipMemcpyDeviceToHost));
        snprintf(tag, sizeof tag, "%s: graph replay", mk.name);
        histo(tag, h);
    }
    return 0;
}
