// Attainable fp32 matrix-core rate on this device: register-resident v_mfma_f32_16x16x4_f32 chains, no memory traffic.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak ; run: tools/mfma_peak [waves_per_cu] [ms]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int NACC>
__global__ __launch_bounds__(256) void k_peak(float* out, int iters, float a0, float b0, long long* clk) {
    const long long c0 = clock64(), w0 = wall_clock64();
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 7 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

int main(int argc, char** argv) {
    const int wg_per_cu = argc > 1 ? atoi(argv[1]) : 1;
    const double target_ms = argc > 2 ? atof(argv[2]) : 200.0;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* out;
    hipMalloc(&out, size_t(cus) * wg_per_cu * 256 * sizeof(float));
    long long* clk;
    hipMalloc(&clk, 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    constexpr int NACC = 16;
    int iters = 20000;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_peak<NACC>, dim3(cus * wg_per_cu), dim3(256), 0, 0, out, iters, 1.0f, 0.5f, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = double(cus) * wg_per_cu * 4 /*waves*/ * double(iters) * 4 * NACC * 2048.0;
        long long h[2];
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        printf("cus %d wg/cu %d iters %d: %.2f ms  %.1f TFLOP/s  shader clock (s_memtime / s_memrealtime) %.0f MHz\n", cus, wg_per_cu,
               iters, ms, flops / ms / 1e9, double(h[0]) / (double(h[1]) / 100.0));
        iters = int(iters * target_ms / ms);
    }
    return 0;
}
