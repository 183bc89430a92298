// BiGRU recurrent kernel (template only: safe to include from several translation units).
#pragma once
#include <type_traits>
#include "mfma_common.h"

namespace cto {

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
// B operands (weights) stream from L2 as one 16-byte load per lane per (gate, 16-wide k chunk) - in the rotated kernel out of a
//             fragment-ordered copy (one contiguous 1 KB per wave request, buffer loads with scalar offsets); they
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
#ifdef CTO_PRECISE_MATH
__device__ __forceinline__ float fast_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return tanhf(x); }
#else
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }
#endif

// MS = 16-row sub-tiles per wave, MH = wave groups along M: the block has 4*MH waves and owns MH*MS*16 sites.
// With MH = 2 every SIMD hosts two waves of the same workgroup, so one wave's gate arithmetic / barrier wait
// is covered by the other's MFMAs.
template <int KIN, int KP, int H, int MS, int MH, bool FUSE_FC1>
__global__ __launch_bounds__(256 * MH) void k_gru_layer(const float* __restrict__ x, const float* __restrict__ Wcat,
                                                   const float* __restrict__ bias, float* __restrict__ out,
                                                   const float* __restrict__ fc1w, float* __restrict__ fc1_part, int B, int site_begin,
                                                       int site_end) {
    constexpr int NB = H / 64, T = 33, KT = KP + H + GRU_WPAD, HS = H + 4, NX = KP / 16, NH = H / 16, NC = NX + NH;
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
    // the launch covers sites [site_begin, site_end) of a batch of B (sub-ranges let the host mix tile heights, see gru.hip)
    const int site0 = site_begin + (blockIdx.x >> 1) * TILE;
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
                    if (site0 + row < site_end) v = *reinterpret_cast<const float4*>(x + (int64_t(site0 + row) * T + t) * KIN + c4);
                } else {
                    const int row = u / KIN, c = u - row * KIN;
                    if (site0 + row < site_end) v.x = x[(int64_t(site0 + row) * T + t) * KIN + c];
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
                        if (site < site_end) out[(int64_t(site) * T + t) * (2 * H) + dir * H + hcol] = hv;
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
                    if (site < site_end) part[int64_t(site) * 128 + wave * 32 + nt * 16 + j] = accf[ms][nt][r];
                }
    }
}

// --------------------------------------------------------------------------------------------
// Rotated schedule of the same recurrence (one wave per SIMD, MH = 1): the x part of step t+1 does not depend on
// h_t, so it is computed *after* the h part of step t, in the same instruction stream as the gate arithmetic of
// step t - the VALU / transcendental work of the gates (and the publication of h_t) then runs in the shadow of
// MFMAs instead of leaving the matrix pipe idle between the last chunk of a step and the barrier.
//     prologue : x_0, x_1 -> LDS;  (r, z, n_x) <- bias + x_0 W_ih^T
//     step t   : barrier;  fetch x_{t+2};  (r, z, n_h) += h_{t-1} W_hh^T  [+ fused fc1 on h_{t-1}];
//                (r', z', n_x') <- bias + x_{t+1} W_ih^T  interleaved with  gates(t) -> h_t -> LDS;  commit x_{t+2}
// Chunk sequence inside a step: h chunks NX..NC-1, then x chunks 0..NX-1 of the next step.
// --------------------------------------------------------------------------------------------
#ifdef CTO_GRU_CLOCKS
__device__ long long g_gru_clk[16];
#endif
// XRAW (layer 1 only): x is the un-rescaled int16 tensor cto_featurize_sites writes (raw_aff / raw_neg) and the coverage rescale of
// clairs/predict.py:172-207 happens where the tile is staged: float(double(v) * scale), scale = min_rescale_cov / depth in double when
// the site's depth (site_info[site][1 + which]) exceeds min_rescale_cov - the expression the tensor kernel itself uses for its fp32
// outputs, so the values that reach the LDS tile are the same bits; the fp32 tensors are then never written or read (18 MB per network
// and 4096-site step).
struct XRawArgs { const int* site_info; int which, min_rescale_cov; };
template <int KIN, int KP, int H, int MS, bool FUSE_FC1, int NW = 4, bool XRAW = false>
__global__ __launch_bounds__(64 * NW) void k_gru_layer_rot(const float* __restrict__ x, const float* __restrict__ Wcat,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       const float* __restrict__ fc1w, float* __restrict__ fc1_part, int B, int site_begin,
                                                       int site_end, XRawArgs xr = XRawArgs{nullptr, 0, 0}) {
    // NW waves share the H columns of the tile: NW = 4 is one wave per SIMD, NW = 8 two (one wave's gate arithmetic then runs
    // under the other's MFMAs as well as under its own)
    static_assert(H % (16 * NW) == 0 && (!FUSE_FC1 || NW == 4), "the fused fc1 slab is laid out for four waves");
    constexpr int NB = H / (16 * NW), T = 33, KT = KP + H + GRU_WPAD, HS = H + 4, NX = KP / 16, NH = H / 16, NC = NX + NH, NP = MS * NB;
    constexpr int FC1_K = T * 2 * H;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TILE = MS * 16, NTHR = 64 * NW;
    constexpr int XS = KP + 4;
    constexpr int XW = (KIN % 4 == 0) ? 4 : ((KIN % 2 == 0) ? 2 : 1);    // floats per staging unit (layer 1: 34 channels -> float2)
    constexpr int XQ = TILE * (KIN / XW);
    constexpr int XPER = (XQ + NTHR - 1) / NTHR;
    float* hbuf = smem;                       // [2][TILE][HS]
    float* xbuf = smem + 2 * TILE * HS;       // [2][TILE][XS]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kg = lane >> 4;
    const int dir = blockIdx.x & 1;
    // the launch covers sites [site_begin, site_end) of a batch of B (sub-ranges let the host mix tile heights, see gru.hip)
    const int site0 = site_begin + (blockIdx.x >> 1) * TILE;
    const float* bd = bias + dir * 4 * H;
    auto t_of = [&](int step) { return dir == 0 ? step : T - 1 - step; };
#ifdef CTO_GRU_CLOCKS
    const long long c0 = clock64(), w0 = wall_clock64();
    long long ph[4] = {0, 0, 0, 0}, tph = 0;     // barrier wait, h part, x part + gates, step tail
#define CTO_PH(i) do { const long long n_ = clock64(); ph[i] += n_ - tph; tph = n_; } while (0)
#else
#define CTO_PH(i) do { } while (0)
#endif

    for (int i = threadIdx.x; i < TILE * HS; i += NTHR) hbuf[i] = 0.f;       // h_{-1} = 0
    for (int i = threadIdx.x; i < 2 * TILE * XS; i += NTHR) xbuf[i] = 0.f;   // K padding and rows past the batch stay 0

    // Weights arrive in FRAGMENT ORDER (models.hip: pack_gru / pack_fc1_fragments): Wcat = [dir][wave][chunk][nb][gate][lane][4],
    // fc1w = [dir][t][wave][kh][nt][lane][4], lane (j, kg) holding W[row .. + j][16 c + 4 kg .. + 3] - a wave's request is one
    // contiguous 1 KB instead of sixteen half cache lines whose other halves the next chunk fetches again
    static_assert(NW == 4, "the fragment-ordered weights are laid out for four waves");
    float bia[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int hcol = (wave * NB + nb) * 16 + j;
#pragma unroll
        for (int q = 0; q < 4; ++q) bia[nb][q] = bd[q * H + hcol];
    }
    (void)KT; (void)FC1_K;
    // requests are buffer loads: wave-uniform resource + scalar offset of the 1 KB unit + one lane offset - no per-lane 64-bit
    // address arithmetic and no address registers in the time loop
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    constexpr unsigned W_WAVE_BYTES = unsigned(KP / 16 + H / 16) * NB * 3 * 1024u, F_T_BYTES = 4u * (H / 16) * 2 * 1024u;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Wcat) + (int64_t(dir) * 4 + wave_u) * (W_WAVE_BYTES / 4), 0, int(W_WAVE_BYTES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(FUSE_FC1 ? fc1w : Wcat) + (FUSE_FC1 ? (int64_t(dir) * T * 4 + wave_u) * ((H / 16) * 2 * 256) : 0), 0,
        FUSE_FC1 ? int(T * F_T_BYTES) : 0, 0x00020000);
    auto buf16 = [&](const __amdgpu_buffer_rsrc_t& r, unsigned voffset, unsigned soffset) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, int(voffset), int(soffset), 0);
        return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
    };
    unsigned voff = unsigned(lane) * 16u;
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

    float4 xstage[XPER];
    // XRAW: the coverage scale of a tile row, in double as clairs/predict.py:172-207 computes it; kept in LDS (registers are what this
    // kernel has none to spare of)
    __shared__ double xscale_row[XRAW ? TILE : 1];
    if constexpr (XRAW) {
        static_assert(!XRAW || XW == 2, "the int16 loader is written for layer 1 (34 channels, two per staging unit)");
        for (int row = threadIdx.x; row < TILE; row += NTHR) {
            double sc = 1.0;
            if (site0 + row < site_end) {
                const int depth = xr.site_info[int64_t(site0 + row) * 12 + 1 + xr.which];
                if (xr.min_rescale_cov > 0 && depth > xr.min_rescale_cov) sc = double(xr.min_rescale_cov) / double(depth);
            }
            xscale_row[row] = sc;
        }
    }
    auto x_fetch = [&](int t) {
#pragma unroll
        for (int q = 0; q < XPER; ++q) {
            const int u = threadIdx.x + q * NTHR;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int row = u / (KIN / XW), c = (u - row * (KIN / XW)) * XW;
            if (u < XQ && site0 + row < site_end) {
                if constexpr (XRAW) {          // two int16 channels in one 4-byte load, kept as bits until the tile is committed
                    const short* src = reinterpret_cast<const short*>(x) + (int64_t(site0 + row) * T + t) * KIN + c;
                    v.x = __uint_as_float(*reinterpret_cast<const unsigned*>(src));
                } else {
                    const float* src = x + (int64_t(site0 + row) * T + t) * KIN + c;     // (site, t) rows are KIN floats: XW-aligned
                    if constexpr (XW == 4) v = *reinterpret_cast<const float4*>(src);
                    else if constexpr (XW == 2) { const float2 w2 = *reinterpret_cast<const float2*>(src); v.x = w2.x; v.y = w2.y; }
                    else v.x = *src;
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
            const int row = u / (KIN / XW), c = (u - row * (KIN / XW)) * XW;
            if (u < XQ) {
                float* dst = xb + row * XS + c;
                if constexpr (XRAW) *reinterpret_cast<float2*>(dst) = make_float2(xstage[q].x, xstage[q].y);       // converted by x_convert
                else if constexpr (XW == 4) *reinterpret_cast<float4*>(dst) = xstage[q];
                else if constexpr (XW == 2) *reinterpret_cast<float2*>(dst) = make_float2(xstage[q].x, xstage[q].y);
                else *dst = xstage[q].x;
            }
        }
    };

    // XRAW: the int16 pairs a step fetched become floats in the middle of the NEXT step's h part - the loads have long landed and the
    // vector unit has nothing else to do under those MFMAs - instead of at the commit, which sits on the step's critical tail
    auto x_convert_one = [&](int q) {
        if constexpr (XRAW) {
            const unsigned w = __float_as_uint(xstage[q].x);
            const int v0 = int(short(w & 0xffffu)), v1 = int(short(w >> 16));
            const int u = threadIdx.x + q * NTHR;
            const double xsc = xscale_row[u < XQ ? u / (KIN / XW) : 0];
            // (not the f64 arithmetic is what this costs - an fp32 stand-in measures the same - but the registers: with the scales in
            // registers layer 1 took 0.292 ms, with them in LDS 0.284, on the fp32 tensor 0.278)
            xstage[q].x = float(double(v0) * xsc);
            xstage[q].y = float(double(v1) * xsc);
        }
    };
    auto x_convert = [&]() {
#pragma unroll
        for (int q = 0; q < XPER; ++q) x_convert_one(q);
    };
    float4 Bq[2][NB][3], Fq[2][2], Aq[2][MS];
    int opq = 0;
    // K of the x part is zero-padded from KIN to KP.  When at most 4 real channels fall into the last 16-wide chunk (layer 1:
    // channels 32, 33 of 34) that chunk is ONE k-step whose four lane groups hold k = 16 (NX - 1) + kg - scalar operand loads -
    // instead of float4 loads of which only two of four k-steps touch a real channel: 9 k-steps for K = 34, not 10.
    // (models.hip: pack_gru lays the last x chunk's fragments out for exactly this rule)
    constexpr bool TAIL1 = (KIN % 16 != 0) && (KIN - 16 * (NX - 1) <= 4);
    auto load_B = [&](int buf, int c) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const unsigned unit = unsigned((c * NB + nb) * 3 + q);      // 1 KB units of this wave's stream
                if (TAIL1 && c == NX - 1)       // W[n][16 c + kg]: the packer put it in element 0
                    Bq[buf][nb][q].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rw, int(voff + (unit & 3u) * 1024u), int((unit & ~3u) * 1024u), 0));
                else Bq[buf][nb][q] = buf16(rw, voff + (unit & 3u) * 1024u, (unit & ~3u) * 1024u);
            }
    };
    auto load_F = [&](int buf, int kh, int tprev) {
        if constexpr (FUSE_FC1) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                Fq[buf][nt] = buf16(rf, voff + unsigned((kh * 2 + nt) & 3) * 1024u,
                                    unsigned(__builtin_amdgcn_readfirstlane(tprev)) * F_T_BYTES + unsigned((kh * 2 + nt) & ~3) * 1024u);
        }
    };
    auto load_Ax = [&](int buf, int c, const float* xc) {
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
            if (TAIL1 && c == NX - 1) Aq[buf][ms].x = xc[(ms * 16 + j) * XS + c * 16 + kg];
            else Aq[buf][ms] = *reinterpret_cast<const float4*>(xc + (ms * 16 + j) * XS + c * 16 + 4 * kg);
        }
    };
    auto load_Ah = [&](int buf, int kh, const float* hc) {
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
            Aq[buf][ms] = *reinterpret_cast<const float4*>(hc + (ms * 16 + j) * HS + kh * 16 + 4 * kg);
    };

    f32x4 ar[MS][NB], az[MS][NB], ain[MS][NB], ahn[MS][NB];     // gates of the current step
    f32x4 nr[MS][NB], nz[MS][NB], nn[MS][NB];                   // x part of the next step, in the making
    auto init_x = [&](f32x4 (&r_)[MS][NB], f32x4 (&z_)[MS][NB], f32x4 (&n_)[MS][NB]) {
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                r_[ms][nb] = f32x4{bia[nb][0], bia[nb][0], bia[nb][0], bia[nb][0]};
                z_[ms][nb] = f32x4{bia[nb][1], bia[nb][1], bia[nb][1], bia[nb][1]};
                n_[ms][nb] = f32x4{bia[nb][2], bia[nb][2], bia[nb][2], bia[nb][2]};
            }
    };
    // one 16-wide k chunk: three gate accumulators per (ms, nb)
    // k-steps of the last x chunk that hold real input channels: K is zero-padded from KIN to KP (layer 1: 34 -> 48) and
    // element e of lane group kg is k = 16 c + 4 kg + e, so with r = KIN - 16 (NX - 1) real channels in the last chunk only
    // the k-steps e < min(4, r) touch any of them (layer 1: r = 2, two of the four k-steps)
    constexpr int NE_LAST = TAIL1 ? 1 : ((KIN - 16 * (NX - 1)) < 4 ? (KIN - 16 * (NX - 1)) : 4);
    auto mfma_chunk = [&](int cur, f32x4 (&r_)[MS][NB], f32x4 (&z_)[MS][NB], f32x4 (&n_)[MS][NB], int ne = 4) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const float4 br = Bq[cur][nb][0], bz = Bq[cur][nb][1], bn = Bq[cur][nb][2];
            const float brv[4] = {br.x, br.y, br.z, br.w}, bzv[4] = {bz.x, bz.y, bz.z, bz.w}, bnv[4] = {bn.x, bn.y, bn.z, bn.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (e >= ne) break;
#pragma unroll
                for (int ms = 0; ms < MS; ++ms) {
                    const float4 a4 = Aq[cur][ms];
                    const float av = e == 0 ? a4.x : (e == 1 ? a4.y : (e == 2 ? a4.z : a4.w));
                    r_[ms][nb] = mfma16(av, brv[e], r_[ms][nb]);
                    z_[ms][nb] = mfma16(av, bzv[e], z_[ms][nb]);
                    n_[ms][nb] = mfma16(av, bnv[e], n_[ms][nb]);
                }
            }
        }
    };
    auto fc1_chunk = [&](int cur) {
        if constexpr (FUSE_FC1) {
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
    };
    // gates + state update of one (ms, nb) pair (lane-local), publish that slice of h_t
    auto gate_pair = [&](int ms, int nb, float* hn, int t) {
        const int hcol = (wave * NB + nb) * 16 + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float rg = fast_sigmoid(ar[ms][nb][r]);
            const float zg = fast_sigmoid(az[ms][nb][r]);
            const float ng = fast_tanh(ain[ms][nb][r] + rg * ahn[ms][nb][r]);
            const float hv = ng + zg * (hprev[ms][nb][r] - ng);      // (1 - z) * n + z * h
            hprev[ms][nb][r] = hv;
            const int row = ms * 16 + kg * 4 + r;
            hn[row * HS + hcol] = hv;
        }
    };
    // Layer output (not fused with fc1): h_t leaves for HBM one step later, as whole 16-byte pieces of its LDS tile - four
    // coalesced stores per lane spread over the next step's h part instead of 16 scalar stores with their address arithmetic
    // in the middle of the gate code (the x part of layer 1 is only 144 MFMAs long and could not hide them).
    constexpr int OPER = (TILE * (H / 4) + NTHR - 1) / NTHR;
    auto store_tile = [&](const float* tile, int tt) {
#pragma unroll
        for (int q = 0; q < OPER; ++q) {
            const int u = threadIdx.x + q * NTHR;
            const int row = u / (H / 4), c4 = (u - row * (H / 4)) * 4;
            if (u < TILE * (H / 4) && site0 + row < site_end)
                *reinterpret_cast<float4*>(out + (int64_t(site0 + row) * T + tt) * (2 * H) + dir * H + c4) =
                    *reinterpret_cast<const float4*>(tile + row * HS + c4);
        }
    };

    // ---- prologue ----
    constexpr int P0 = NX & 1;          // buffer parity such that the weights of the first h chunk land in buffer 0
    x_fetch(t_of(0));
    __syncthreads();                    // zero fill complete
    x_convert();
    x_commit(0);
    x_fetch(t_of(1));
    x_convert();
    x_commit(1);
    load_B(P0, 0);
    __syncthreads();
    load_Ax(P0, 0, xbuf);
    init_x(ar, az, ain);
#pragma unroll
    for (int c = 0; c < NX; ++c) {
        const int cur = (c + P0) & 1, nxt = cur ^ 1;
        if (c + 1 < NX) { load_B(nxt, c + 1); load_Ax(nxt, c + 1, xbuf); }
        else { load_B(nxt, NX); load_F(0, 0, t_of(0)); }
        mfma_chunk(cur, ar, az, ain, c == NX - 1 ? NE_LAST : 4);
        __builtin_amdgcn_sched_barrier(0);
    }

    auto step_body = [&](int step, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int t = t_of(step);
        const int tprev = step == 0 ? t : t_of(step - 1);          // step 0: h = 0, any valid slice will do
        static_assert(!FUSE_FC1 || (NC % 2 == 0), "the fc1 prefetch across the step boundary assumes an even chunk count");
        const int cur_h = step & 1;
        const float* hc = hbuf + cur_h * (TILE * HS);
        float* hn = hbuf + (cur_h ^ 1) * (TILE * HS);
        const float* xnx = xbuf + ((step + 1) & 1) * (TILE * XS);   // x_{t+1}
        opq = 0;
        asm volatile("" : "+v"(opq));      // keeps the (step-invariant) weight loads inside the time loop
        voff = unsigned(lane + opq) * 16u;
#ifdef CTO_GRU_CLOCKS
        tph = clock64();
#endif
        __syncthreads();                   // h_{t-1} and x_{t+1} are complete
        CTO_PH(0);
        if (step + 2 < T) x_fetch(t_of(step + 2));
        load_Ah(0, 0, hc);
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) ahn[ms][nb] = f32x4{bia[nb][3], bia[nb][3], bia[nb][3], bia[nb][3]};
        if constexpr (!LAST) init_x(nr, nz, nn);
#pragma unroll
        for (int sq = 0; sq < (LAST ? NH : NC); ++sq) {
            const int cur = sq & 1, nxt = cur ^ 1;
            // ---- request the operands of the next chunk in the sequence ----
            if (sq + 1 < NH) {
                load_B(nxt, NX + sq + 1);
                load_Ah(nxt, sq + 1, hc);
                load_F(nxt, sq + 1, tprev);
            } else if (!LAST) {
                if (sq + 1 < NC) { load_B(nxt, sq + 1 - NH); load_Ax(nxt, sq + 1 - NH, xnx); }
                else { load_B(nxt, NX); load_F(0, 0, t); }     // first h chunk of the next step (its fc1 slice is that of h_t)
            }
            if (sq < NH) {
                if constexpr (!FUSE_FC1) {
                    if (sq == 1 && step > 0) store_tile(hc, tprev);
                }
                if constexpr (XRAW) {        // x_{t+2}, fetched at the top of this step: one staging unit per chunk, the last chunks of the h part
                    static_assert(!XRAW || XPER <= NH, "one h chunk per staging unit");
                    if (sq >= NH - XPER && step + 2 < T) x_convert_one(sq - (NH - XPER));
                }
                mfma_chunk(cur, ar, az, ahn);
                fc1_chunk(cur);
            } else {
                mfma_chunk(cur, nr, nz, nn, sq - NH == NX - 1 ? NE_LAST : 4);
                // gate arithmetic of step t, spread over the x chunks of step t+1
#pragma unroll
                for (int pq = 0; pq < NP; ++pq)
                    if ((pq * NX) / NP == sq - NH) gate_pair(pq / NB, pq % NB, hn, t);
            }
            // Spread the next chunk's operand requests (NB*3 weight loads, MS LDS reads, 2 fc1 loads) between the MFMAs instead of
            // issuing them as one burst at the top of the chunk: the wave issues in order, so a burst of ~12 memory instructions
            // with their address arithmetic leaves the matrix pipe idle for ~10 % of every chunk.
#ifndef CTO_GRU_IL
#define CTO_GRU_IL 5
#endif
#ifndef CTO_GRU_IL1
#define CTO_GRU_IL1 3       // measured on MI355X (tools/ab.py, layer 1 + tail): 2 -> 0.302, 3 -> 0.300, 4 / 5 / 6 -> 0.306, 8 -> 0.311 ms
#endif
#ifndef CTO_GRU_NO_INTERLEAVE
            // MFMAs between two operand requests: layer 2 (88-MFMA chunks, 13 requests) and layer 1 (48, 8) are tuned separately
            constexpr int IL = FUSE_FC1 ? CTO_GRU_IL : CTO_GRU_IL1;
#pragma unroll
            for (int g = 0; g < NB * 3 + (FUSE_FC1 ? 2 : 0); ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, IL, 0);      // MFMAs
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read
            }
#pragma unroll
            for (int g = 0; g < MS; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, IL, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
            }
            if (!FUSE_FC1 && sq == 1) {
#pragma unroll
                for (int g = 0; g < OPER; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // tile piece from LDS
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);  // ... to HBM
                }
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
            if (sq == NH - 1) CTO_PH(1);
        }
        CTO_PH(2);
        if constexpr (LAST) {
#pragma unroll
            for (int pq = 0; pq < NP; ++pq) gate_pair(pq / NB, pq % NB, hn, t);
        } else {
            if (step + 2 < T) x_commit(step & 1);          // x_{t+2} replaces x_t
            if constexpr ((NC & 1) != 0) {   // odd chunk count: the next step's first weights landed in buffer 1
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int q = 0; q < 3; ++q) Bq[0][nb][q] = Bq[1][nb][q];
            }
#pragma unroll
            for (int ms = 0; ms < MS; ++ms)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) { ar[ms][nb] = nr[ms][nb]; az[ms][nb] = nz[ms][nb]; ain[ms][nb] = nn[ms][nb]; }
        }
        CTO_PH(3);
    };
    for (int step = 0; step + 1 < T; ++step) step_body(step, std::false_type{});
    step_body(T - 1, std::true_type{});
    if constexpr (!FUSE_FC1) {
        __syncthreads();                                  // h_{T-1} is complete in LDS
        store_tile(hbuf + (T & 1) * (TILE * HS), t_of(T - 1));
    }
#ifdef CTO_GRU_CLOCKS
    if (blockIdx.x == 7 && threadIdx.x == 0) {
        const int o = FUSE_FC1 ? 4 : 0;
        g_gru_clk[o] = clock64() - c0; g_gru_clk[o + 1] = wall_clock64() - w0;
        g_gru_clk[8 + o] = ph[0]; g_gru_clk[9 + o] = ph[1]; g_gru_clk[10 + o] = ph[2]; g_gru_clk[11 + o] = ph[3];
    }
#endif

    if constexpr (FUSE_FC1) {
        // fc1 contribution of the last state, then one partial slab per direction
        __syncthreads();
        const float* hl = hbuf + (T & 1) * (TILE * HS);
        const int tl = t_of(T - 1);
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
                    const int site = site0 + ms * 16 + kg * 4 + r;
                    if (site < site_end) part[int64_t(site) * 128 + wave * 32 + nt * 16 + j] = accf[ms][nt][r];
                }
    }
}

}  // namespace cto
