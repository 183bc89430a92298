// Issue-rate probe for the fp32 matrix-core instructions of gfx950: cycles per MFMA (s_memtime ticks = shader cycles) for
// register-resident chains of v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 with 1 or 2 waves per SIMD, with and without the
// operand traffic of the CvT block kernel's GEMM loop (per 20 MFMAs: five ds_read_b128 + one 16-byte global load).
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_issue.hip -o tools/mfma_issue ; run: tools/mfma_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// MODE 0: 16x16x4, NACC independent accumulators, bare.  MODE 1: the same + operand traffic.  MODE 2: 32x32x2, NACC accumulators.
// MODE 3: 32x32x2 + operand traffic (per 10 MFMAs = the same flops as 20 of the small ones)
template <int MODE, int NACC>
__global__ __launch_bounds__(512) void k_issue(float* out, const float* __restrict__ g, int iters, long long* clk) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 132];
    for (int i = threadIdx.x; i < 64 * 132; i += blockDim.x) lds[i] = float(i & 7);
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 15, kg = lane >> 4;
    const float* arow = lds + j * 132 + 4 * kg;
    const float* grow = g + (threadIdx.x >> 6) * 4096 + j * 128 + 4 * kg;
    float s = 0.f;
    long long c0 = 0;
    if constexpr (MODE < 2 || MODE >= 4) {
        constexpr int IL = MODE == 7 ? 2 : MODE - 3;         // MODE 4, 5, 6: operand requests interleaved with the MFMAs, IL MFMAs per request
        f32x4 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        float4 a[2][NACC], b[3];
#pragma unroll
        for (int i = 0; i < NACC; ++i) a[0][i] = a[1][i] = *reinterpret_cast<const float4*>(arow + i * 16);
        b[0] = b[1] = b[2] = *reinterpret_cast<const float4*>(grow);
        c0 = clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                if constexpr (MODE == 7) {   // weights as the block kernel reads them: W[n][K = 128], one 64 KB matrix per 8 chunks, 1 MB cycling
                    const int q = it * 6 + c;
                    b[(c + 2) % 3] = *reinterpret_cast<const float4*>(g + size_t((q >> 3) & 15) * 16384 + ((threadIdx.x >> 6) * 16 + j) * 128 + (q & 7) * 16 + 4 * kg);
#pragma unroll
                    for (int i = 0; i < NACC; ++i) a[(c + 1) & 1][i] = *reinterpret_cast<const float4*>(arow + i * 16 * 132 % (48 * 132) + ((it + c) & 7) * 16);
                } else if constexpr (MODE >= 1) {
                    b[(c + 2) % 3] = *reinterpret_cast<const float4*>(grow + ((it * 6 + c) & 7) * 16);
#pragma unroll
                    for (int i = 0; i < NACC; ++i) a[(c + 1) & 1][i] = *reinterpret_cast<const float4*>(arow + i * 16 * 132 % (48 * 132) + ((it + c) & 7) * 16);
                }
                if constexpr (MODE < 4) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < NACC; ++i) {
                        const float4 a4 = a[c & 1][i], b4 = b[c % 3];
                        const float av = e == 0 ? a4.x : (e == 1 ? a4.y : (e == 2 ? a4.z : a4.w));
                        const float bv = e == 0 ? b4.x : (e == 1 ? b4.y : (e == 2 ? b4.z : b4.w));
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
                    }
                if constexpr (MODE >= 4) {
                    __builtin_amdgcn_sched_group_barrier(0x008, IL, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#pragma unroll
                    for (int i = 0; i < NACC; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, IL, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        f32x16 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        float4 a[2][NACC], b[3];
#pragma unroll
        for (int i = 0; i < NACC; ++i) a[0][i] = a[1][i] = *reinterpret_cast<const float4*>(arow + i * 16);
        b[0] = b[1] = b[2] = *reinterpret_cast<const float4*>(grow);
        c0 = clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                if constexpr (MODE == 3) {
                    b[(c + 2) % 3] = *reinterpret_cast<const float4*>(grow + ((it * 6 + c) & 7) * 16);
#pragma unroll
                    for (int i = 0; i < NACC; ++i) a[(c + 1) & 1][i] = *reinterpret_cast<const float4*>(arow + i * 16 * 132 % (48 * 132) + ((it + c) & 7) * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < NACC; ++i) {
                        const float4 a4 = a[c & 1][i], b4 = b[c % 3];
                        const float av = e == 0 ? a4.x : (e == 1 ? a4.y : (e == 2 ? a4.z : a4.w));
                        const float bv = e == 0 ? b4.x : (e == 1 ? b4.y : (e == 2 ? b4.z : b4.w));
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][r];
    }
    __syncthreads();
    const long long c1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 3 && threadIdx.x == 0) clk[0] = c1 - c0;
}

// Waves 0-3: the MFMA stream with operand traffic (1 wave per SIMD).  Waves 4-7 (same SIMDs): VALU work of kind VK for the same
// time - 0: nothing, 1: independent v_fma chains, 2: the attention mix (ds_read_b128, fma, DPP adds, exp, rcp).
template <int VK>
__global__ __launch_bounds__(512) void k_mix(float* out, const float* __restrict__ g, int iters, long long* clk) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 132];
    for (int i = threadIdx.x; i < 64 * 132; i += blockDim.x) lds[i] = float(i & 7) * 0.01f;
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 15, kg = lane >> 4, wave = threadIdx.x >> 6;
    const float* arow = lds + j * 132 + 4 * kg;
    float s = 0.f;
    const long long c0 = clock64();
    long long my = 0;
    if (wave < 4) {
        constexpr int NACC = 5;
        f32x4 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        float4 a[2][NACC], b[3];
#pragma unroll
        for (int i = 0; i < NACC; ++i) a[0][i] = a[1][i] = *reinterpret_cast<const float4*>(arow + i * 16);
        b[0] = b[1] = b[2] = *reinterpret_cast<const float4*>(g + j * 128 + 4 * kg);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const int q = it * 6 + c;
                b[(c + 2) % 3] = *reinterpret_cast<const float4*>(g + size_t((q >> 3) & 15) * 16384 + (wave * 16 + j) * 128 + (q & 7) * 16 + 4 * kg);
#pragma unroll
                for (int i = 0; i < NACC; ++i) a[(c + 1) & 1][i] = *reinterpret_cast<const float4*>(arow + i * 16 * 132 % (48 * 132) + ((it + c) & 7) * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < NACC; ++i) {
                        const float4 a4 = a[c & 1][i], b4 = b[c % 3];
                        const float av = e == 0 ? a4.x : (e == 1 ? a4.y : (e == 2 ? a4.z : a4.w));
                        const float bv = e == 0 ? b4.x : (e == 1 ? b4.y : (e == 2 ? b4.z : b4.w));
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#pragma unroll
                for (int i = 0; i < NACC; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        my = clock64() - c0;
    } else if (VK == 1) {
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = float(lane + i) * 1e-3f;
        for (int it = 0; it < iters * 40; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = fmaf(x[i], 0.999f, 0.001f);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += x[i];
        my = clock64() - c0;
    } else if (VK == 2) {
        const int l4 = (lane & 15) * 4;
        for (int it = 0; it < iters * 4; ++it) {
            const int rr = (it * 4 + (lane >> 4)) % 48;
            const float4 qv = *reinterpret_cast<const float4*>(lds + rr * 132 + l4);
            float4 kv[3], vv[3];
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                kv[jj] = *reinterpret_cast<const float4*>(lds + ((rr + jj + 1) % 48) * 132 + l4);
                vv[jj] = *reinterpret_cast<const float4*>(lds + ((rr + jj + 7) % 48) * 132 + 64 + l4);
            }
            float sc[3];
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                float v = fmaf(qv.w, kv[jj].w, fmaf(qv.z, kv[jj].z, fmaf(qv.y, kv[jj].y, qv.x * kv[jj].x)));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
                sc[jj] = v;
            }
            const float mx = fmaxf(sc[0], fmaxf(sc[1], sc[2]));
            float sum = 0.f;
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) { sc[jj] = __expf((sc[jj] - mx) * 0.125f); sum += sc[jj]; }
            const float inv = __builtin_amdgcn_rcpf(sum);
            float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                const float pj = sc[jj] * inv;
                o4.x = fmaf(pj, vv[jj].x, o4.x); o4.y = fmaf(pj, vv[jj].y, o4.y); o4.z = fmaf(pj, vv[jj].z, o4.z); o4.w = fmaf(pj, vv[jj].w, o4.w);
            }
            s += o4.x + o4.y + o4.z + o4.w;
        }
        my = clock64() - c0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 3 && lane == 0 && (wave == 0 || wave == 4)) clk[wave >> 2] = my;
}
template <int VK>
void run_mix(const char* what, float* out, const float* g, long long* clk) {
    const int iters = 400;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k_mix<VK>), dim3(256), dim3(512), 0, 0, out, g, iters, clk);
        hipDeviceSynchronize();
    }
    long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("MFMA wave + %-28s: %6.2f cycles per MFMA; the other wave ran %lld cycles\n", what, double(h[0]) / (iters * 6.0 * 20), h[1]);
}

template <int MODE, int NACC>
void run(const char* what, int threads, float* out, const float* g, long long* clk) {
    const int iters = 400;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k_issue<MODE, NACC>), dim3(256), dim3(threads), 0, 0, out, g, iters, clk);
        hipDeviceSynchronize();
    }
    long long h;
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    const double per_wave = double(iters) * 6 * 4 * NACC;
    const int waves_per_simd = threads / 256;
    const double cyc = double(h) / (per_wave * waves_per_simd);
    const double ideal = (MODE < 2 || MODE >= 4) ? 32.0 : 64.0;
    printf("%-46s waves/SIMD %d  acc %2d : %6.2f cycles per MFMA per SIMD (%.3f of the %g-cycle rate)\n", what, waves_per_simd, NACC, cyc,
           ideal / cyc, ideal);
}

int main() {
    float *out, *g;
    long long* clk;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipMalloc(&g, 16 * 16384 * sizeof(float));
    hipMemset(g, 0, 16 * 16384 * sizeof(float));
    hipMalloc(&clk, 16);
    run<0, 2>("16x16x4 bare", 256, out, g, clk);
    run<0, 5>("16x16x4 bare", 256, out, g, clk);
    run<0, 5>("16x16x4 bare", 512, out, g, clk);
    run<0, 10>("16x16x4 bare", 256, out, g, clk);
    run<1, 5>("16x16x4 + 5 ds_read_b128 + 1 load / 20", 256, out, g, clk);
    run<1, 5>("16x16x4 + 5 ds_read_b128 + 1 load / 20", 512, out, g, clk);
    run<1, 10>("16x16x4 + 10 ds_read_b128 + 1 load / 40", 256, out, g, clk);
    run<4, 5>("16x16x4 + 5 ds_read + 1 load / 20, 1 MFMA : 1", 256, out, g, clk);
    run<4, 5>("16x16x4 + 5 ds_read + 1 load / 20, 1 MFMA : 1", 512, out, g, clk);
    run<5, 5>("16x16x4 + 5 ds_read + 1 load / 20, 2 MFMA : 1", 256, out, g, clk);
    run<5, 5>("16x16x4 + 5 ds_read + 1 load / 20, 2 MFMA : 1", 512, out, g, clk);
    run<6, 5>("16x16x4 + 5 ds_read + 1 load / 20, 3 MFMA : 1", 256, out, g, clk);
    run<6, 5>("16x16x4 + 5 ds_read + 1 load / 20, 3 MFMA : 1", 512, out, g, clk);
    run<6, 3>("16x16x4 + 3 ds_read + 1 load / 12, 3 MFMA : 1", 512, out, g, clk);
    run<1, 3>("16x16x4 + 3 ds_read + 1 load / 12", 512, out, g, clk);
    run<7, 5>("16x16x4 + 5 ds_read + 1 L2 load / 20, 2 MFMA : 1", 256, out, g, clk);
    run<7, 5>("16x16x4 + 5 ds_read + 1 L2 load / 20, 2 MFMA : 1", 512, out, g, clk);
    run_mix<0>("idle wave", out, g, clk);
    run_mix<1>("v_fma chains", out, g, clk);
    run_mix<2>("attention mix", out, g, clk);
    run<2, 1>("32x32x2 bare", 256, out, g, clk);
    run<2, 2>("32x32x2 bare", 256, out, g, clk);
    run<2, 3>("32x32x2 bare", 512, out, g, clk);
    run<3, 3>("32x32x2 + 3 ds_read_b128 + 1 load / 12", 256, out, g, clk);
    run<3, 3>("32x32x2 + 3 ds_read_b128 + 1 load / 12", 512, out, g, clk);
    return 0;
}
