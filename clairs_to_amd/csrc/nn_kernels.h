// Device kernels of the AFF (CvT) and NEG (BiGRU) networks for gfx950 / CDNA4.
//
// All dense contractions run on the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32: fp32 in, fp32
// accumulate, bit-equal to an fmaf chain), because the parity bar is 1e-4 on probabilities and bf16/fp16
// inputs miss it.  Wave = 64 lanes; one MFMA computes a 16x16 tile over k = 4:
//     lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15],
//     lane l receives  D[row = 4 * (l >> 4) + r][col = l & 15] in register r = 0..3.
// Every weight matrix is consumed in PyTorch's native Linear layout W[n][k] (k contiguous), so
// C[m][n] = sum_k A[m][k] * W[n][k] needs no transposition: lane l reads W[n0 + (l & 15)][k0 + (l >> 4)].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cto {

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- activations (clairs/model.py: nn.SELU, nn.GELU() exact-erf form) ----
__device__ __forceinline__ float selu_f(float x) {
    const float scale = 1.0507009873554804934193349852946f;
    const float alpha = 1.6732632423543772848170429916717f;
    return x > 0.f ? scale * x : scale * alpha * expm1f(x);
}
// exact-erf GELU (nn.GELU() default).  erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32 round-off class):
// libm's erff costs ~45 VALU per call and the FFN epilogues evaluate it 160 times per lane per transformer block.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float r = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

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

// g[b][n] = SELU(sum_s slabs[s][b][n] + bias[n])  -- closes the split-K fc1 of the BiGRU head
__global__ __launch_bounds__(256) void k_sum_bias_selu(const float* __restrict__ slabs, int S, int64_t slab,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       int64_t total, int N) {
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= total) return;
    float s = 0.f;
    for (int z = 0; z < S; ++z) s += slabs[z * slab + i];
    out[i] = selu_f(s + bias[i % N]);
}

// out[k][b][c] = SELU(sum_j u[b][k*128 + j] * W3[k][c][j] + b3[k][c])   (x_fc3 heads, clairs/model.py:250-253)
__global__ __launch_bounds__(256) void k_fc3(const float* __restrict__ u, const float* __restrict__ W3,
                                             const float* __restrict__ b3, float* __restrict__ out, int64_t B, int K) {
    const int64_t b = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    for (int k = 0; k < K; ++k) {
        const float* ur = u + (b * K + k) * 128;
        const float u0 = ur[lane], u1 = ur[lane + 64];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float* wr = W3 + (k * 2 + c) * 128;
            const float s = wave_sum(u0 * wr[lane] + u1 * wr[lane + 64]);
            if (lane == 0) out[(int64_t(k) * B + b) * 2 + c] = selu_f(s + b3[k * 2 + c]);
        }
    }
}

// --------------------------------------------------------------------------------------------
// One direction of one bidirectional GRU layer (torch.nn.GRU semantics, gate order r, z, n;
// clairs/model.py:412-417, 442-443) for a tile of MS*16 sites, all 33 time steps, in one launch.
//
// Work split: the block's 4 waves split the H hidden units (NB = H/64 blocks of 16 per wave); each wave
// owns, for its hidden units, the r / z / n gate columns, so the gate arithmetic is lane-local.
// Per time step the wave accumulates  [x_t | h_{t-1}] (K = KP + H)  against  Wcat[3H][KP + H]
// (= [W_ih | W_hh] per gate row, W_ih zero-padded to KP) with fp32 MFMA:
//     r, z : one accumulator over the whole K;   n : separate accumulators for the x part (gi_n) and
//     the h part (gh_n) because n = tanh(gi_n + r * gh_n).
// A operands: x_t from a double-buffered LDS tile that the whole block fills one step ahead (loads issued at the start
//             of step t for step t+1, written to LDS just before the barrier of step t: the HBM latency of the
//             activations - which every one of the 4 waves needs in full - is paid once per step, off the MFMA path),
//             h_{t-1} from a double-buffered LDS tile [MS*16][H] that all waves rewrite each step.
// B operands (weights) stream from L2 as one 16-byte load per lane per (gate, 16-wide k chunk); they
//             are shared by the MS row-subtiles.  One barrier per time step.
// The K loop is fully unrolled and software-pipelined by hand: the operands of chunk c+1 (and, at the
// end of a step, of chunk 0 of the next step) are requested before the MFMAs of chunk c, so the matrix
// pipe never waits on an L2 round trip except right after the barrier (LDS reads only).
// FUSE_FC1 (layer 2): the head's fc1 (clairs/model.py:445-448, K = 33*384) is accumulated on the fly -
// the h_{t-1} fragments already in registers for the recurrence are multiplied with the matching
// 192-column slice of fc1.weight - and written as one partial [B][128] slab per direction, so the
// [B][33][384] layer output never goes to HBM.
// --------------------------------------------------------------------------------------------
// v_exp_f32 / v_rcp_f32 (1 ulp each): `__fdividef` expands to the full IEEE division sequence (div_scale, fma chain,
// div_fmas, div_fixup: ~10 VALU) - three of those per state element were 40 % of the kernel's non-MFMA instructions.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// MS = 16-row sub-tiles per wave, MH = wave groups along M: the block has 4*MH waves and owns MH*MS*16 sites.
// With MH = 2 every SIMD hosts two waves of the same workgroup, so one wave's gate arithmetic / barrier wait
// is covered by the other's MFMAs.
template <int KIN, int KP, int H, int MS, int MH, bool FUSE_FC1>
__global__ __launch_bounds__(256 * MH) void k_gru_layer(const float* __restrict__ x, const float* __restrict__ Wcat,
                                                   const float* __restrict__ bias, float* __restrict__ out,
                                                   const float* __restrict__ fc1w, float* __restrict__ fc1_part, int B) {
    constexpr int NB = H / 64, T = 33, KT = KP + H, HS = H + 4, NX = KP / 16, NH = H / 16, NC = NX + NH;
    constexpr int FC1_K = T * 2 * H;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TILE = MH * MS * 16, NTHR = 256 * MH;
    constexpr int XS = KP + 4;                                    // row stride of the x tile (16-byte aligned rows)
    constexpr bool XV = (KIN % 4 == 0);                           // stage x as float4 (else scalars)
    constexpr int XQ = XV ? TILE * (KIN / 4) : TILE * KIN;        // staging units per step
    constexpr int XPER = (XQ + NTHR - 1) / NTHR;
    float* hbuf = smem;                       // [2][TILE][HS]
    float* xbuf = smem + 2 * TILE * HS;       // [2][TILE][XS]
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, mh = threadIdx.x >> 8;
    const int rb = mh * MS * 16;          // first tile row of this wave
    const int j = lane & 15, kg = lane >> 4;
    const int dir = blockIdx.x & 1;
    const int site0 = (blockIdx.x >> 1) * TILE;
    const float* Wd = Wcat + int64_t(dir) * 3 * H * KT;
    const float* bd = bias + dir * 4 * H;

    for (int i = threadIdx.x; i < TILE * HS; i += NTHR) hbuf[i] = 0.f;   // h_{-1} = 0
    for (int i = threadIdx.x; i < 2 * TILE * XS; i += NTHR) xbuf[i] = 0.f;   // K padding and rows past the batch stay 0

    float bia[NB][4];
    const float* wrow[NB][3];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int hcol = (wave * NB + nb) * 16 + j;
#pragma unroll
        for (int q = 0; q < 4; ++q) bia[nb][q] = bd[q * H + hcol];
#pragma unroll
        for (int q = 0; q < 3; ++q) wrow[nb][q] = Wd + int64_t(q * H + hcol) * KT + 4 * kg;
    }
    const float* frow[2] = {nullptr, nullptr};
    if constexpr (FUSE_FC1) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) frow[nt] = fc1w + int64_t(wave * 32 + nt * 16 + j) * FC1_K + dir * H + 4 * kg;
    }
    float hprev[MS][NB][4];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) hprev[ms][nb][r] = 0.f;
    f32x4 accf[MS][2];
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) { accf[ms][0] = f32x4{0.f, 0.f, 0.f, 0.f}; accf[ms][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // x staging: unit u of the tile = (row, 4-float group) or (row, scalar); global -> registers -> LDS
    float4 xstage[XPER];
    auto x_fetch = [&](int t) {
#pragma unroll
        for (int q = 0; q < XPER; ++q) {
            const int u = threadIdx.x + q * NTHR;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < XQ) {
                if constexpr (XV) {
                    const int row = u / (KIN / 4), c4 = (u - row * (KIN / 4)) * 4;
                    if (site0 + row < B) v = *reinterpret_cast<const float4*>(x + (int64_t(site0 + row) * T + t) * KIN + c4);
                } else {
                    const int row = u / KIN, c = u - row * KIN;
                    if (site0 + row < B) v.x = x[(int64_t(site0 + row) * T + t) * KIN + c];
                }
            }
            xstage[q] = v;
        }
    };
    auto x_commit = [&](int buf) {
        float* xb = xbuf + buf * (TILE * XS);
#pragma unroll
        for (int q = 0; q < XPER; ++q) {
            const int u = threadIdx.x + q * NTHR;
            if (u < XQ) {
                if constexpr (XV) {
                    const int row = u / (KIN / 4), c4 = (u - row * (KIN / 4)) * 4;
                    *reinterpret_cast<float4*>(xb + row * XS + c4) = xstage[q];
                } else {
                    const int row = u / KIN, c = u - row * KIN;
                    xb[row * XS + c] = xstage[q].x;
                }
            }
        }
    };

    float4 Bq[2][NB][3], Fq[2][2], Aq[2][MS];

    int opq = 0;
    auto load_B = [&](int buf, int c) {   // weights of k chunk c (x chunks first, then h chunks)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 3; ++q) Bq[buf][nb][q] = *reinterpret_cast<const float4*>(wrow[nb][q] + c * 16 + opq);
    };
    auto load_F = [&](int buf, int kh, int tprev) {
        if constexpr (FUSE_FC1) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                Fq[buf][nt] = *reinterpret_cast<const float4*>(frow[nt] + tprev * (2 * H) + kh * 16);
        }
    };
    auto load_Ax = [&](int buf, int c, const float* xc) {
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
            Aq[buf][ms] = *reinterpret_cast<const float4*>(xc + (rb + ms * 16 + j) * XS + c * 16 + 4 * kg);
    };
    auto load_Ah = [&](int buf, int kh, const float* hc) {
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
            Aq[buf][ms] = *reinterpret_cast<const float4*>(hc + (rb + ms * 16 + j) * HS + kh * 16 + 4 * kg);
    };

    // prologue: x tile of step 0 into LDS, weights of chunk 0
    x_fetch(dir == 0 ? 0 : T - 1);
    __syncthreads();              // the zero fill above is complete
    x_commit(0);
    load_B(0, 0);
    __syncthreads();
    load_Ax(0, 0, xbuf);

    for (int step = 0; step < T; ++step) {
        const int t = dir == 0 ? step : T - 1 - step;
        const int tnext = dir == 0 ? t + 1 : t - 1;                          // valid while step + 1 < T
        const int tprev = step == 0 ? t : (dir == 0 ? t - 1 : t + 1);        // step 0: h = 0, any valid slice will do
        const int cur_h = step & 1;
        const float* hc = hbuf + cur_h * (TILE * HS);
        const float* xc = xbuf + cur_h * (TILE * XS);             // x_t (filled during the previous step)
        const float* xn = xbuf + (cur_h ^ 1) * (TILE * XS);       // x_{t+1} (filled during this step)
        // The weight addresses do not depend on `step`; without this the compiler hoists all K chunks of weight
        // loads out of the time loop (hundreds of registers, spills).  An opaque zero keeps them per-step.
        opq = 0;
        asm volatile("" : "+v"(opq));
        f32x4 ar[MS][NB], az[MS][NB], ain[MS][NB], ahn[MS][NB];
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                ar[ms][nb] = f32x4{bia[nb][0], bia[nb][0], bia[nb][0], bia[nb][0]};
                az[ms][nb] = f32x4{bia[nb][1], bia[nb][1], bia[nb][1], bia[nb][1]};
                ain[ms][nb] = f32x4{bia[nb][2], bia[nb][2], bia[nb][2], bia[nb][2]};
                ahn[ms][nb] = f32x4{bia[nb][3], bia[nb][3], bia[nb][3], bia[nb][3]};
            }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int cur = c & 1, nxt = cur ^ 1;
            if (c == 0 && step + 1 < T) x_fetch(tnext);   // next step's activations: in flight under this step's x part
            if (c == NX) {
                if (step + 1 < T) x_commit(cur_h ^ 1);
                __syncthreads();            // h_{t-1} (written during the previous step) and x_{t+1} are complete
                load_Ah(cur, 0, hc);
                load_F(cur, 0, tprev);
            }
            // ---- request the operands of the next chunk before computing this one ----
            if (c + 1 < NC) {
                load_B(nxt, c + 1);
                if (c + 1 < NX) load_Ax(nxt, c + 1, xc);
                else if (c + 1 > NX) { load_Ah(nxt, c + 1 - NX, hc); load_F(nxt, c + 1 - NX, tprev); }
            } else if (step + 1 < T) {
                load_B(nxt, 0);
                load_Ax(nxt, 0, xn);
            }
            // ---- MFMAs of chunk c ----
            // k-step outermost, accumulators innermost: consecutive MFMAs never share an accumulator, so the 40-cycle
            // dependent latency of v_mfma_f32_16x16x4_f32 (issue interval 32) is never exposed.
            const bool xpart = c < NX;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float4 br = Bq[cur][nb][0], bz = Bq[cur][nb][1], bn = Bq[cur][nb][2];
                const float brv[4] = {br.x, br.y, br.z, br.w}, bzv[4] = {bz.x, bz.y, bz.z, bz.w}, bnv[4] = {bn.x, bn.y, bn.z, bn.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int ms = 0; ms < MS; ++ms) {
                        const float4 a4 = Aq[cur][ms];
                        const float av = e == 0 ? a4.x : (e == 1 ? a4.y : (e == 2 ? a4.z : a4.w));
                        ar[ms][nb] = mfma16(av, brv[e], ar[ms][nb]);
                        az[ms][nb] = mfma16(av, bzv[e], az[ms][nb]);
                        if (xpart) ain[ms][nb] = mfma16(av, bnv[e], ain[ms][nb]);
                        else ahn[ms][nb] = mfma16(av, bnv[e], ahn[ms][nb]);
                    }
                }
            }
            if constexpr (FUSE_FC1) {
                if (!xpart) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int ms = 0; ms < MS; ++ms) {
                                const float4 a4 = Aq[cur][ms], f4 = Fq[cur][nt];
                                const float av = e == 0 ? a4.x : (e == 1 ? a4.y : (e == 2 ? a4.z : a4.w));
                                const float fv = e == 0 ? f4.x : (e == 1 ? f4.y : (e == 2 ? f4.z : f4.w));
                                accf[ms][nt] = mfma16(av, fv, accf[ms][nt]);
                            }
                }
            }
            // keep the hand-made pipeline: nothing (in particular no later prefetch) moves across a chunk boundary
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr ((NC & 1) != 0) {   // odd chunk count: next step's chunk 0 landed in buffer 1, it is read from 0
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 3; ++q) Bq[0][nb][q] = Bq[1][nb][q];
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) Aq[0][ms] = Aq[1][ms];
        }
        // ---- gates + state update (lane-local), publish h_t ----
        float* hn = hbuf + (cur_h ^ 1) * (TILE * HS);
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int hcol = (wave * NB + nb) * 16 + j;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float rg = fast_sigmoid(ar[ms][nb][r]);
                    const float zg = fast_sigmoid(az[ms][nb][r]);
                    const float ng = fast_tanh(ain[ms][nb][r] + rg * ahn[ms][nb][r]);
                    const float hv = ng + zg * (hprev[ms][nb][r] - ng);      // (1 - z) * n + z * h
                    hprev[ms][nb][r] = hv;
                    const int row = rb + ms * 16 + kg * 4 + r;
                    hn[row * HS + hcol] = hv;
                    if constexpr (!FUSE_FC1) {
                        const int site = site0 + row;
                        if (site < B) out[(int64_t(site) * T + t) * (2 * H) + dir * H + hcol] = hv;
                    }
                }
            }
    }
    if constexpr (FUSE_FC1) {
        // fc1 contribution of the last state h_{T-1 (fwd) / 0 (bwd)}, then one partial slab per direction
        __syncthreads();
        const float* hl = hbuf + (T & 1) * (TILE * HS);
        const int tl = dir == 0 ? T - 1 : 0;
#pragma unroll
        for (int kh = 0; kh < NH; ++kh) {
            load_Ah(0, kh, hl);
            load_F(0, kh, tl);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int ms = 0; ms < MS; ++ms) {
                    const float4 a = Aq[0][ms], f = Fq[0][nt];
                    accf[ms][nt] = mfma16(a.x, f.x, accf[ms][nt]);
                    accf[ms][nt] = mfma16(a.y, f.y, accf[ms][nt]);
                    accf[ms][nt] = mfma16(a.z, f.z, accf[ms][nt]);
                    accf[ms][nt] = mfma16(a.w, f.w, accf[ms][nt]);
                }
        }
        float* part = fc1_part + int64_t(dir) * B * 128;
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int site = site0 + rb + ms * 16 + kg * 4 + r;
                    if (site < B) part[int64_t(site) * 128 + wave * 32 + nt * 16 + j] = accf[ms][nt][r];
                }
    }
}

}  // namespace cto
