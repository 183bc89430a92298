// Device kernels of the AFF (CvT) network, the generic fp32-MFMA GEMM and the head glue (gfx950 / CDNA4).
// Included by models.hip only (the non-template kernels below must live in exactly one translation unit).
#pragma once
#include "mfma_common.h"

namespace cto {

enum { ACT_NONE = 0, ACT_GELU = 1, ACT_SELU = 2 };

// --------------------------------------------------------------------------------------------
// Generic fp32-MFMA GEMM:  C[m][n] = act(sum_k A[m][k] W[n][k] + bias[n]) (+ R[m][n])
//   * conv mode (conv_wout > 0): row m = (site, wo) of a 3-tap, stride-2, pad-1 1-D convolution over a
//     channels-last activation x[site][win][cin]; the im2col row is the contiguous run
//     x[site][2wo-1 .. 2wo+1][:] with out-of-range taps zeroed (clairs/model.py:195 - only the middle
//     kernel row of the 3x3 Conv2d is live because H = 1).
//   * split-K (gridDim.z > 1): slice z accumulates k in [z*kslice, (z+1)*kslice) into C + z*slab
//     with no bias/activation; the consumer sums the slabs (deterministic order).
// Block = 256 threads = 4 waves arranged WM x WN, each wave TM x TN tiles of 16x16.
// --------------------------------------------------------------------------------------------
struct GemmArgs {
    const float* A; int64_t lda;
    const float* W; int64_t ldw;
    const float* bias;
    const float* R; int64_t ldr;
    float* C; int64_t ldc;
    int M, N, K;
    int act;
    int conv_win, conv_wout, conv_cin;   // conv mode when conv_wout > 0 (then K = 3 * conv_cin)
    int kslice; int64_t slab;            // split-K
    int vecA;                            // 1: A rows are 16-byte aligned and K % 4 == 0
};

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void k_gemm(GemmArgs g) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16, BK = 16, LDS_S = BK + 2;
    __shared__ float As[BM][LDS_S];
    __shared__ float Ws[BN][LDS_S];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    int kbeg = 0, kend = g.K;
    if (gridDim.z > 1) { kbeg = blockIdx.z * g.kslice; kend = min(g.K, kbeg + g.kslice); }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int A_Q = BM * (BK / 4), W_Q = BN * (BK / 4);   // float4 quads per tile
    constexpr int A_PER = (A_Q + 255) / 256, W_PER = (W_Q + 255) / 256;
    float4 ra[A_PER], rw[W_PER];

    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int q = tid + i * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < A_Q) {
                const int r = q / (BK / 4), kq = (q % (BK / 4)) * 4;
                const int m = m0 + r, k = k0 + kq;
                if (m < g.M) {
                    if (g.conv_wout > 0) {
                        const int site = m / g.conv_wout, wo = m - site * g.conv_wout;
                        const float* row = g.A + (int64_t(site) * g.conv_win + (2 * wo - 1)) * g.conv_cin;
                        const int klo = (wo == 0) ? g.conv_cin : 0;
                        const int khi = (2 * wo + 1 >= g.conv_win) ? 2 * g.conv_cin : 3 * g.conv_cin;
                        float t[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int kk = k + e;
                            t[e] = (kk >= klo && kk < khi && kk < kend) ? row[kk] : 0.f;
                        }
                        v = make_float4(t[0], t[1], t[2], t[3]);
                    } else if (g.vecA && k + 3 < kend) {
                        v = *reinterpret_cast<const float4*>(g.A + int64_t(m) * g.lda + k);
                    } else {
                        const float* row = g.A + int64_t(m) * g.lda;
                        float t[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) t[e] = (k + e < kend) ? row[k + e] : 0.f;
                        v = make_float4(t[0], t[1], t[2], t[3]);
                    }
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < W_PER; ++i) {
            const int q = tid + i * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < W_Q) {
                const int r = q / (BK / 4), kq = (q % (BK / 4)) * 4;
                const int n = n0 + r, k = k0 + kq;
                if (n < g.N) {
                    const float* row = g.W + int64_t(n) * g.ldw;
                    if ((g.ldw & 3) == 0 && k + 3 < kend) {
                        v = *reinterpret_cast<const float4*>(row + k);
                    } else {
                        float t[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) t[e] = (k + e < kend) ? row[k + e] : 0.f;
                        v = make_float4(t[0], t[1], t[2], t[3]);
                    }
                }
            }
            rw[i] = v;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int q = tid + i * 256;
            if (q < A_Q) {
                const int r = q / (BK / 4), kq = (q % (BK / 4)) * 4;
                As[r][kq + 0] = ra[i].x; As[r][kq + 1] = ra[i].y; As[r][kq + 2] = ra[i].z; As[r][kq + 3] = ra[i].w;
            }
        }
#pragma unroll
        for (int i = 0; i < W_PER; ++i) {
            const int q = tid + i * 256;
            if (q < W_Q) {
                const int r = q / (BK / 4), kq = (q % (BK / 4)) * 4;
                Ws[r][kq + 0] = rw[i].x; Ws[r][kq + 1] = rw[i].y; Ws[r][kq + 2] = rw[i].z; Ws[r][kq + 3] = rw[i].w;
            }
        }
    };

    load_tiles(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();          // previous chunk's fragment reads are done
        store_tiles();
        __syncthreads();
        if (k0 + BK < kend) load_tiles(k0 + BK);   // next chunk's global loads fly under the MFMAs
        const int ar = wm * TM * 16 + (lane & 15), wr = wn * TN * 16 + (lane & 15), kg = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[ar + i * 16][ks * 4 + kg];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Ws[wr + j * 16][ks * 4 + kg];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
        }
    }

    float* C = g.C + (gridDim.z > 1 ? int64_t(blockIdx.z) * g.slab : 0);
    const bool plain = gridDim.z > 1;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 16 + (lane & 15);
            if (n >= g.N) continue;
            const float bv = (!plain && g.bias) ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + (wm * TM + i) * 16 + (lane >> 4) * 4 + r;
                if (m >= g.M) continue;
                float v = acc[i][j][r] + bv;
                if (!plain) {
                    if (g.act == ACT_GELU) v = gelu_f(v);
                    else if (g.act == ACT_SELU) v = selu_f(v);
                    if (g.R) v += g.R[int64_t(m) * g.ldr + n];
                }
                C[int64_t(m) * g.ldc + n] = v;
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// Channel LayerNorm of the reference (clairs/model.py:57-67): (x - mean) / (sqrt(var_biased) + eps) * g + b,
// eps added to the standard deviation.  Channels-last rows [M][C]; one wavefront per row.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void k_layernorm(const float* __restrict__ x, float* __restrict__ y,
                                                   const float* __restrict__ gam, const float* __restrict__ bet,
                                                   int M, int C) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + int64_t(row) * C;
    float v0 = lane < C ? xr[lane] : 0.f, v1 = lane + 64 < C ? xr[lane + 64] : 0.f;
    const float mean = wave_sum(v0 + v1) / float(C);
    const float d0 = lane < C ? v0 - mean : 0.f, d1 = lane + 64 < C ? v1 - mean : 0.f;
    const float var = wave_sum(d0 * d0 + d1 * d1) / float(C);
    const float inv = 1.0f / (sqrtf(var) + 1e-5f);
    if (lane < C) y[int64_t(row) * C + lane] = d0 * inv * gam[lane] + bet[lane];
    if (lane + 64 < C) y[int64_t(row) * C + lane + 64] = d1 * inv * gam[lane + 64] + bet[lane + 64];
}

// --------------------------------------------------------------------------------------------
// PreNorm + depth-wise 3-tap conv + BatchNorm(eval) for the q path (stride 1) and the kv path (stride 2)
// of clairs/model.py:102-118.  One block per site; the normalised [W][C] slab lives in LDS.
//   dwq/dwkv [C][3] (middle row of the 3x3 depth-wise kernel), bn* [4][C] = mean, invstd, weight, bias.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ln_dw(const float* __restrict__ h, const float* __restrict__ gam,
                                               const float* __restrict__ bet, const float* __restrict__ dwq,
                                               const float* __restrict__ bnq, const float* __restrict__ dwkv,
                                               const float* __restrict__ bnkv, float* __restrict__ yq,
                                               float* __restrict__ ykv, int W, int Wkv, int C) {
    __shared__ float s_y[17 * 128];
    const int64_t site = blockIdx.x;
    const float* hs = h + site * W * C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int w = wave; w < W; w += 4) {
        const float* xr = hs + w * C;
        float v0 = lane < C ? xr[lane] : 0.f, v1 = lane + 64 < C ? xr[lane + 64] : 0.f;
        const float mean = wave_sum(v0 + v1) / float(C);
        const float d0 = lane < C ? v0 - mean : 0.f, d1 = lane + 64 < C ? v1 - mean : 0.f;
        const float var = wave_sum(d0 * d0 + d1 * d1) / float(C);
        const float inv = 1.0f / (sqrtf(var) + 1e-5f);
        if (lane < C) s_y[w * C + lane] = d0 * inv * gam[lane] + bet[lane];
        if (lane + 64 < C) s_y[w * C + lane + 64] = d1 * inv * gam[lane + 64] + bet[lane + 64];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < W * C; i += 256) {
        const int w = i / C, c = i - w * C;
        const float l = w > 0 ? s_y[(w - 1) * C + c] : 0.f, m = s_y[w * C + c], r = w + 1 < W ? s_y[(w + 1) * C + c] : 0.f;
        const float d = dwq[c * 3 + 0] * l + dwq[c * 3 + 1] * m + dwq[c * 3 + 2] * r;
        yq[(site * W + w) * C + c] = (d - bnq[c]) * bnq[C + c] * bnq[2 * C + c] + bnq[3 * C + c];
    }
    for (int i = threadIdx.x; i < Wkv * C; i += 256) {
        const int wo = i / C, c = i - wo * C, w = 2 * wo;
        const float l = w > 0 ? s_y[(w - 1) * C + c] : 0.f, m = s_y[w * C + c], r = w + 1 < W ? s_y[(w + 1) * C + c] : 0.f;
        const float d = dwkv[c * 3 + 0] * l + dwkv[c * 3 + 1] * m + dwkv[c * 3 + 2] * r;
        ykv[(site * Wkv + wo) * C + c] = (d - bnkv[c]) * bnkv[C + c] * bnkv[2 * C + c] + bnkv[3 * C + c];
    }
}

// --------------------------------------------------------------------------------------------
// Attention core of clairs/model.py:120-131 for one site: softmax(q k^T * 0.125) v per head, dim_head 64.
// q [W][inner], kv [Wkv][2*inner] (k first, then v), out [W][inner]; inner = 64 * heads.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_attention(const float* __restrict__ q, const float* __restrict__ kv,
                                                   float* __restrict__ o, int W, int Wkv, int heads) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int inner = heads * 64;
    float* s_q = smem;                       // [W][inner]
    float* s_kv = s_q + W * inner;           // [Wkv][2*inner]
    float* s_p = s_kv + Wkv * 2 * inner;     // [heads][W][Wkv]
    const int64_t site = blockIdx.x;
    const float* qs = q + site * W * inner;
    const float* kvs = kv + site * Wkv * 2 * inner;
    for (int i = threadIdx.x; i < W * inner; i += 256) s_q[i] = qs[i];
    for (int i = threadIdx.x; i < Wkv * 2 * inner; i += 256) s_kv[i] = kvs[i];
    __syncthreads();
    const int ndots = heads * W * Wkv;
    for (int t = threadIdx.x; t < ndots; t += 256) {
        const int hh = t / (W * Wkv), rem = t - hh * W * Wkv, i = rem / Wkv, j = rem - i * Wkv;
        const float* qv = s_q + i * inner + hh * 64;
        const float* kk = s_kv + j * 2 * inner + hh * 64;
        float s = 0.f;
#pragma unroll 16
        for (int d = 0; d < 64; ++d) s = fmaf(qv[d], kk[d], s);
        s_p[t] = s * 0.125f;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < heads * W; t += 256) {
        float* row = s_p + t * Wkv;
        float mx = row[0];
        for (int j = 1; j < Wkv; ++j) mx = fmaxf(mx, row[j]);
        float sum = 0.f;
        for (int j = 0; j < Wkv; ++j) { const float e = expf(row[j] - mx); row[j] = e; sum += e; }
        const float inv = 1.0f / sum;
        for (int j = 0; j < Wkv; ++j) row[j] *= inv;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < W * inner; t += 256) {
        const int i = t / inner, c = t - i * inner, hh = c >> 6;
        const float* p = s_p + (hh * W + i) * Wkv;
        float s = 0.f;
        for (int j = 0; j < Wkv; ++j) s = fmaf(p[j], s_kv[j * 2 * inner + inner + c], s);
        o[site * W * inner + t] = s;
    }
}

}  // namespace cto
