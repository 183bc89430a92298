// Which compute units does a stream created with hipExtStreamCreateWithCUMask use on this device?  For masks with the first N bits
// set (and a strided one), 8192 single-wave blocks record (XCC_ID, SE_ID, CU_ID) and the host counts the distinct places; then a
// busy kernel on the masked stream runs beside a timed kernel on an unmasked one.   hipcc -O2 --offload-arch=gfx950 tools/cumask_probe.hip -o tools/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_where(uint32_t* out, int spin) {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 15u) << 16 | (hw & 0xffffu);
}
__global__ void k_busy(float* p, int iters) {
    float a = p[threadIdx.x], b = 1.0001f;
    for (int i = 0; i < iters; ++i) a = a * b + 0.5f;
    p[blockIdx.x * blockDim.x + threadIdx.x] = a;
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    printf("device: %s, %d CUs\n", pr.name, pr.multiProcessorCount);
    const int NB = 8192;
    uint32_t* d;
    CK(hipMalloc(&d, NB * 4));
    std::vector<uint32_t> h(NB);
    auto probe = [&](const char* name, std::vector<uint32_t> mask) -> int {
        hipStream_t s;
        CK(hipExtStreamCreateWithCUMask(&s, uint32_t(mask.size()), mask.data()));
        hipLaunchKernelGGL(k_where, dim3(NB), dim3(64), 0, s, d, 20000);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d, NB * 4, hipMemcpyDeviceToHost));
        std::set<uint32_t> places, xccs;
        int per_xcc[16] = {0};
        for (uint32_t v : h) { if (places.insert(v).second) per_xcc[v >> 16]++; xccs.insert(v >> 16); }
        printf("%-28s distinct (xcc, hw_id) places %4zu on %zu XCCs: per XCC", name, places.size(), xccs.size());
        for (int i = 0; i < 8; ++i) printf(" %d", per_xcc[i]);
        printf("\n");
        CK(hipStreamDestroy(s));
        return 0;
    };
    std::vector<uint32_t> all(8, 0xffffffffu);
    if (probe("all 256 bits", all)) return 1;
    for (int n : {32, 64, 128, 192}) {
        std::vector<uint32_t> m(8, 0);
        for (int i = 0; i < n; ++i) m[i / 32] |= 1u << (i % 32);
        char nm[64];
        snprintf(nm, sizeof nm, "first %d bits", n);
        if (probe(nm, m)) return 1;
    }
    {
        std::vector<uint32_t> m(8, 0);
        for (int i = 0; i < 256; i += 4) m[i / 32] |= 1u << (i % 32);
        if (probe("every 4th bit (64 bits)", m)) return 1;
    }
    // interference: a long busy kernel on a masked stream beside a timed kernel on a plain stream
    float* buf;
    CK(hipMalloc(&buf, size_t(1) << 28));
    CK(hipMemset(buf, 0, size_t(1) << 28));
    hipStream_t plain;
    CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timed = [&]() -> float {
        CK(hipEventRecord(e0, plain));
        hipLaunchKernelGGL(k_busy, dim3(2048), dim3(256), 0, plain, buf, 20000);
        CK(hipEventRecord(e1, plain));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms;
    };
    timed();
    printf("timed kernel alone: %.3f ms\n", timed());
    for (int n : {64, 128, 256}) {
        std::vector<uint32_t> m(8, 0);
        for (int i = 0; i < n; ++i) m[i / 32] |= 1u << (i % 32);
        hipStream_t s;
        CK(hipExtStreamCreateWithCUMask(&s, 8, m.data()));
        hipLaunchKernelGGL(k_busy, dim3(16384), dim3(256), 0, s, buf + (1 << 24), 200000);      // long background load
        const float ms = timed();
        CK(hipStreamSynchronize(s));
        printf("beside a background kernel masked to the first %3d CUs: %.3f ms\n", n, ms);
        CK(hipStreamDestroy(s));
    }
    return 0;
}
