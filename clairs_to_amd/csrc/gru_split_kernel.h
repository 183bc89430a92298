// Split-operand form of the BiGRU layer-2 recurrence + fused fc1 (clairs/model.py:412-417, 442-448).
//
// EXPERIMENT, side channel only: selected with CTO_GRU_SPLIT=f16|bf16 at model creation, never part of the default path, reported
// under its own key by bench.py (`split_mfma`) together with its own max |dP| against the oracle.  The headline kernel
// (gru_kernel.h) multiplies in fp32 on `v_mfma_f32_16x16x4_f32` (256 FLOP / cycle / CU); this one writes every operand as
// hi + lo, both 16-bit floats, and forms a·b ≈ hi·hi + hi·lo + lo·hi with three `v_mfma_f32_16x16x32_{f16,bf16}` passes
// accumulating in fp32 - 48 matrix-pipe cycles for a 16x16x32 product instead of 256.  The state h_t itself, the gate
// arithmetic and all sums stay fp32; what is given up is the lo·lo term and the bits below lo:
//     f16  : hi = RTZ(a) (11 bits), lo = RNE(a - hi) (11 bits): 22 significant bits, products good to ~2^-21.  Operands of this
//            layer are bounded (|h|, |x| < 1; weights far inside the f16 range - checked at packing), small ones go subnormal
//            with an absolute error below 2^-25, and the f16 MFMA does not flush subnormal inputs (measured on MI355X).
//     bf16 : hi, lo = RNE: 16-17 significant bits, products good to ~2^-16; no range condition.
//
// Differences from k_gru_layer_rot (same rotated schedule: the x part of step t+1 and the gates of step t share a stream):
//   * operands swapped: the WEIGHT fragment is the MFMA's A operand and the activations its B operand, so a lane ends up
//     with four consecutive hidden units of ONE site - h_t leaves for LDS as two 8-byte stores (hi, lo) per (tile, block)
//     instead of eight 2-byte ones, and the fc1 slab as 16-byte stores;
//   * weights pre-split and stored in fragment order by cto_bigru_create (`pack_gru_split`): every operand request is one
//     coalesced 1 KB load per wave;
//   * h and x tiles live in LDS as two 16-bit planes each (hi, lo), rows padded by 16 bytes: an operand fragment is one
//     ds_read_b128 per plane;
//   * K advances 32 per chunk: 8 x chunks + 6 h chunks per step.
#pragma once
#include "gru_kernel.h"
#include "split_mfma.h"

namespace cto {

// Power-of-two scales of the f16 form (models.hip: pack_gru_split chooses them; all 1 for bf16, whose exponent range is fp32's).
// An f16 half below 2^-14 is sub-normal: a weight of 0.015 then keeps an ABSOLUTE error of 2^-25, which an unrescaled count of 8 000
// multiplies into 2e-4 of a pre-activation (tests/golden/gen_range.py found it; tools/experiments/split_model.py reproduces it on
// the CPU).  So every operand class is lifted towards the top of the f16 range before it is split - x by sx, h by sh (|h| <= 1:
// 2^14), [W_ih | W_hh] by s_total / sx and s_total / sh, fc1.weight by 1 / (inv_f sh) - the accumulators and the bias they start
// from are in units of s_total, and the gates fold 1 / s_total into the constant their exp2 argument is multiplied with anyway.
struct GruSplitScale { float sx, sh, s_total, inv_s, inv_f; };

// Fragment-ordered operand arrays (16-byte units, index = ... * 64 + lane; lane = (kg << 4) | j holds k = 32 c + 8 kg .. + 7):
//   Wp[dir][wave][chunk c < NX + NH][nb][gate r,z,n][hi,lo][lane] : row  gate * H + (wave * NB + nb) * 16 + j  of [W_ih | W_hh]
//   Fp[dir][t][wave][kh < NH][nt < 2][hi,lo][lane]                : row  wave * 32 + nt * 16 + j  of fc1.weight, k = t 2H + dir H + 32 kh ..
// FUSE_FC1 (layer 2): the head's fc1 accumulated on the fly, one partial [B][128] slab per direction (as in k_gru_layer_rot);
// otherwise (layer 1) h_t goes to `out` [B][33][2H] as fp32, one 16-byte store per lane, tile and block, straight from the gates.
// KP = KIN rounded up to whole 32-wide chunks; the padding columns of the x planes stay zero.
template <int KIN, int KP, int H, int MS, bool F16, bool FUSE_FC1>
__global__ __launch_bounds__(256) void k_gru_split(const float* __restrict__ x, const uint4* __restrict__ Wp,
                                                   const float* __restrict__ bias, const uint4* __restrict__ Fp,
                                                   float* __restrict__ fc1_part, float* __restrict__ out, int B, int site_begin,
                                                   int site_end, GruSplitScale sc) {
    constexpr int NB = H / 64, T = 33, NX = KP / 32, NH = H / 32, NC = NX + NH, NP = MS * NB;
    constexpr int TILE = MS * 16, NTHR = 256;
    constexpr int HSB = H + 8, XSB = KP + 8;             // 16-bit elements per tile row
    static_assert(KP % 32 == 0 && KP >= KIN && H % 64 == 0 && (NC % 2) == 0, "chunking of the split kernel");
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_s[];
    unsigned short* hbuf = smem_s;                             // [2 buffers][hi, lo][TILE][HSB]
    unsigned short* xbuf = smem_s + 4 * TILE * HSB;            // [2 buffers][hi, lo][TILE][XSB]
    float* blds = reinterpret_cast<float*>(xbuf + 4 * TILE * XSB);   // [4][H] biases
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kg = lane >> 4;
    const int dir = blockIdx.x & 1;
    const int site0 = site_begin + (blockIdx.x >> 1) * TILE;
    auto t_of = [&](int step) { return dir == 0 ? step : T - 1 - step; };
#ifdef CTO_GRU_CLOCKS
    const long long c0 = clock64(), w0 = wall_clock64();
    long long ph[4] = {0, 0, 0, 0}, tph = 0;     // barrier wait, h part, x part + gates, step tail
#endif

    for (int i = threadIdx.x; i < 4 * TILE * HSB / 2; i += NTHR) reinterpret_cast<unsigned*>(hbuf)[i] = 0u;   // h_{-1} = 0
    for (int i = threadIdx.x; i < 4 * TILE * XSB / 2; i += NTHR) reinterpret_cast<unsigned*>(xbuf)[i] = 0u;   // rows past the batch
    for (int i = threadIdx.x; i < 4 * H; i += NTHR) blds[i] = bias[dir * 4 * H + i] * sc.s_total;

    // Operand requests are buffer loads: a wave-uniform resource + scalar offset for the (chunk, block) and one lane offset
    // (16 bytes per lane inside a 1 KB unit) - no per-lane 64-bit address arithmetic in the time loop.
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    constexpr unsigned W_WAVE_BYTES = unsigned(NC) * NB * 6 * 1024u, F_T_BYTES = 4u * NH * 4 * 1024u;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4*>(Wp) + (int64_t(dir) * 4 + wave_u) * (W_WAVE_BYTES / 16), 0, int(W_WAVE_BYTES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4*>(FUSE_FC1 ? Fp : Wp) + (FUSE_FC1 ? (int64_t(dir) * T * 4 + wave_u) * (NH * 4 * 64) : 0), 0,
        FUSE_FC1 ? int(T * F_T_BYTES) : 0, 0x00020000);
    auto buf16 = [&](const __amdgpu_buffer_rsrc_t& r, unsigned voffset, unsigned soffset) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, int(voffset), int(soffset), 0);
        return make_uint4(v[0], v[1], v[2], v[3]);
    };

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

    // ---- x staging: fp32 rows of the layer input -> (hi, lo) planes ----
    constexpr int XW = (KIN % 4 == 0) ? 4 : 2;             // floats per staging unit (layer 1: 34 channels -> float2)
    static_assert(KIN % XW == 0, "input rows are staged as float2 / float4");
    constexpr int XQ = TILE * (KIN / XW), XPER = (XQ + NTHR - 1) / NTHR;
    float4 xstage[XPER];
    auto x_fetch = [&](int t) {
#pragma unroll
        for (int q = 0; q < XPER; ++q) {
            const int u = min(int(threadIdx.x) + q * NTHR, XQ - 1);
            const int row = u / (KIN / XW), c = (u - row * (KIN / XW)) * XW;
            // rows past the batch read the batch's last row (and are never looked at): no branch in the time loop
            const int site = min(site0 + row, site_end - 1);
            const float* src = x + (int64_t(site) * T + t) * KIN + c;
            if constexpr (XW == 4) xstage[q] = *reinterpret_cast<const float4*>(src);
            else { const float2 v = *reinterpret_cast<const float2*>(src); xstage[q] = make_float4(v.x, v.y, 0.f, 0.f); }
        }
    };
    auto x_commit = [&](int buf) {
        unsigned short* xb = xbuf + buf * (2 * TILE * XSB);
#pragma unroll
        for (int q = 0; q < XPER; ++q) {
            const int u = threadIdx.x + q * NTHR;
            const int row = u / (KIN / XW), c = (u - row * (KIN / XW)) * XW;
            if (XQ % NTHR == 0 || u < XQ) {
                uint2 hi, lo;
                split_pair<F16>(xstage[q].x * sc.sx, xstage[q].y * sc.sx, hi.x, lo.x);
                if constexpr (XW == 4) {
                    split_pair<F16>(xstage[q].z * sc.sx, xstage[q].w * sc.sx, hi.y, lo.y);
                    *reinterpret_cast<uint2*>(xb + row * XSB + c) = hi;
                    *reinterpret_cast<uint2*>(xb + (TILE + row) * XSB + c) = lo;
                } else {
                    *reinterpret_cast<unsigned*>(xb + row * XSB + c) = hi.x;
                    *reinterpret_cast<unsigned*>(xb + (TILE + row) * XSB + c) = lo.x;
                }
            }
        }
    };

    uint4 Wq[2][NB][3][2], Fq[2][2][2], Aq[2][MS][2];
    int opq = 0;
    unsigned voff = unsigned(lane) * 16u;      // the lane's byte offset inside a 1 KB operand unit
    auto load_W = [&](int buf, int c) {
#if defined(CTO_SPLIT_DBG) && CTO_SPLIT_DBG == 3     // timing probe: the weight stream is requested once (wrong results)
        if (c != 0 && c != NX) return;
#endif
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int unit = q * 2 + p;          // 1 KB units of this (chunk, block): 0..5, the immediate field holds 0..3 KB
#if defined(CTO_SPLIT_DBG) && CTO_SPLIT_DBG == 1     // timing probe: every request hits the same 6 KB (wrong results)
                    Wq[buf][nb][q][p] = buf16(rw, voff + (unit & 3) * 1024u, (unit >> 2) * 4096u);
#else
                    Wq[buf][nb][q][p] = buf16(rw, voff + (unit & 3) * 1024u, unsigned((c * NB + nb) * 6 + (unit & 4)) * 1024u);
#endif
                }
    };
    auto load_F = [&](int buf, int kh, int tprev) {
        if constexpr (!FUSE_FC1) return;
        const unsigned st = unsigned(__builtin_amdgcn_readfirstlane(tprev)) * F_T_BYTES + unsigned(kh) * 4096u;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int p = 0; p < 2; ++p) Fq[buf][nt][p] = buf16(rf, voff + (nt * 2 + p) * 1024u, st);
    };
    auto load_Ax = [&](int buf, int c, const unsigned short* xc) {
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                Aq[buf][ms][p] = *reinterpret_cast<const uint4*>(xc + (p * TILE + ms * 16 + j) * XSB + c * 32 + kg * 8);
    };
    auto load_Ah = [&](int buf, int kh, const unsigned short* hc) {
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                Aq[buf][ms][p] = *reinterpret_cast<const uint4*>(hc + (p * TILE + ms * 16 + j) * HSB + kh * 32 + kg * 8);
    };

    f32x4 ar[MS][NB], az[MS][NB], ain[MS][NB], ahn[MS][NB];     // gates of the current step: lane = (site j, hidden units 4 kg ..)
    f32x4 nr[MS][NB], nz[MS][NB], nn[MS][NB];                   // x part of the next step, in the making
    auto bias4 = [&](int q, int nb) { return *reinterpret_cast<const f32x4*>(blds + q * H + (wave * NB + nb) * 16 + kg * 4); };
    auto init_x = [&](f32x4 (&r_)[MS][NB], f32x4 (&z_)[MS][NB], f32x4 (&n_)[MS][NB]) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const f32x4 b0 = bias4(0, nb), b1 = bias4(1, nb), b2 = bias4(2, nb);
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) { r_[ms][nb] = b0; z_[ms][nb] = b1; n_[ms][nb] = b2; }
        }
    };
    // one 32-wide k chunk: hi·hi, hi·lo, lo·hi for three gate accumulators per (ms, nb); the 6 NB accumulators between two
    // passes over the same one keep the matrix pipe free of back-to-back dependences
    auto mfma_chunk = [&](int cur, f32x4 (&r_)[MS][NB], f32x4 (&z_)[MS][NB], f32x4 (&n_)[MS][NB]) {
#pragma unroll
        for (int pass = 0; pass < 3; ++pass) {
            const int pw = pass == 2 ? 1 : 0, pa = pass == 1 ? 1 : 0;       // (weight part, activation part)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int ms = 0; ms < MS; ++ms) {
                    r_[ms][nb] = mfma_split<F16>(Wq[cur][nb][0][pw], Aq[cur][ms][pa], r_[ms][nb]);
                    z_[ms][nb] = mfma_split<F16>(Wq[cur][nb][1][pw], Aq[cur][ms][pa], z_[ms][nb]);
                    n_[ms][nb] = mfma_split<F16>(Wq[cur][nb][2][pw], Aq[cur][ms][pa], n_[ms][nb]);
                }
        }
    };
    auto fc1_chunk = [&](int cur) {
        if constexpr (!FUSE_FC1) return;
#pragma unroll
        for (int pass = 0; pass < 3; ++pass) {
            const int pw = pass == 2 ? 1 : 0, pa = pass == 1 ? 1 : 0;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int ms = 0; ms < MS; ++ms) accf[ms][nt] = mfma_split<F16>(Fq[cur][nt][pw], Aq[cur][ms][pa], accf[ms][nt]);
        }
    };
    // gates + state update of one (ms, nb) pair (lane-local), publish that slice of h_t as (hi, lo)
    const float c_sig = -1.44269504088896340736f * sc.inv_s, c_tanh = 2.88539008177792681472f * sc.inv_s;
    auto gate_pair = [&](int ms, int nb, unsigned short* hn, int t) {
        float hv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#if defined(CTO_SPLIT_DBG) && CTO_SPLIT_DBG == 2     // timing probe: no transcendentals (wrong results)
            const float rg = ar[ms][nb][r], zg = az[ms][nb][r], ng = ain[ms][nb][r] + rg * ahn[ms][nb][r];
#else
            // fast_sigmoid / fast_tanh of gru_kernel.h on accumulators in units of s_total (the unit rides on the exp2 constant)
            const float rg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ar[ms][nb][r] * c_sig));
            const float zg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(az[ms][nb][r] * c_sig));
            const float ng = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((ain[ms][nb][r] + rg * ahn[ms][nb][r]) * c_tanh));
#endif
            hv[r] = ng + zg * (hprev[ms][nb][r] - ng);      // (1 - z) * n + z * h
            hprev[ms][nb][r] = hv[r];
        }
        uint2 hi, lo;
        split_pair<F16>(hv[0] * sc.sh, hv[1] * sc.sh, hi.x, lo.x);
        split_pair<F16>(hv[2] * sc.sh, hv[3] * sc.sh, hi.y, lo.y);
        const int at = (ms * 16 + j) * HSB + (wave * NB + nb) * 16 + kg * 4;
        *reinterpret_cast<uint2*>(hn + at) = hi;
        *reinterpret_cast<uint2*>(hn + TILE * HSB + at) = lo;
        if constexpr (!FUSE_FC1) {
            const int site = site0 + ms * 16 + j;
            if (site < site_end)
                *reinterpret_cast<f32x4*>(out + (int64_t(site) * T + t) * (2 * H) + dir * H + (wave * NB + nb) * 16 + kg * 4) =
                    f32x4{hv[0], hv[1], hv[2], hv[3]};
        }
    };

    // ---- prologue ----
    x_fetch(t_of(0));
    __syncthreads();                    // zero fill and biases complete
    x_commit(0);
    x_fetch(t_of(1));
    x_commit(1);
    load_W(0, 0);
    __syncthreads();
    load_Ax(0, 0, xbuf);
    init_x(ar, az, ain);
#pragma unroll
    for (int c = 0; c < NX; ++c) {
        const int cur = c & 1, nxt = cur ^ 1;
        if (c + 1 < NX) { load_W(nxt, c + 1); load_Ax(nxt, c + 1, xbuf); }
        else { load_W(nxt, NX); load_F(0, 0, t_of(0)); }
        mfma_chunk(cur, ar, az, ain);
        __builtin_amdgcn_sched_barrier(0);
    }

    auto step_body = [&](int step, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int t = t_of(step);
        const int tprev = step == 0 ? t : t_of(step - 1);          // step 0: h = 0, any valid slice will do
        const int cur_h = step & 1;
        const unsigned short* hc = hbuf + cur_h * (2 * TILE * HSB);
        unsigned short* hn = hbuf + (cur_h ^ 1) * (2 * TILE * HSB);
        const unsigned short* xnx = xbuf + ((step + 1) & 1) * (2 * TILE * XSB);   // x_{t+1}
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
        for (int nb = 0; nb < NB; ++nb) {
            const f32x4 b3 = bias4(3, nb);
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) ahn[ms][nb] = b3;
        }
        if constexpr (!LAST) init_x(nr, nz, nn);
#pragma unroll
        for (int sq = 0; sq < (LAST ? NH : NC); ++sq) {
            const int cur = sq & 1, nxt = cur ^ 1;
            if (sq + 1 < NH) {
                load_W(nxt, NX + sq + 1);
                load_Ah(nxt, sq + 1, hc);
                load_F(nxt, sq + 1, tprev);
            } else if (!LAST) {
                if (sq + 1 < NC) { load_W(nxt, sq + 1 - NH); load_Ax(nxt, sq + 1 - NH, xnx); }
                else { load_W(nxt, NX); load_F(0, 0, t); }     // first h chunk of the next step (its fc1 slice is that of h_t)
            }
            if (sq < NH) {
                mfma_chunk(cur, ar, az, ahn);
                fc1_chunk(cur);
            } else {
                mfma_chunk(cur, nr, nz, nn);
                // gate arithmetic of step t, spread over the x chunks of step t+1
#pragma unroll
                for (int pq = 0; pq < NP; ++pq)
                    if ((pq * NX) / NP == sq - NH) gate_pair(pq / NB, pq % NB, hn, t);
            }
#ifndef CTO_GRU_SPLIT_IL
#define CTO_GRU_SPLIT_IL 3
#endif
#ifndef CTO_GRU_SPLIT_NO_INTERLEAVE
            // spread the next chunk's operand requests between the MFMAs of this one
#pragma unroll
            for (int g = 0; g < NB * 6 + (FUSE_FC1 ? 4 : 0); ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, CTO_GRU_SPLIT_IL, 0);      // MFMAs
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                     // 1 VMEM read
            }
#pragma unroll
            for (int g = 0; g < MS * 2; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, CTO_GRU_SPLIT_IL, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                     // 1 DS read
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
#pragma unroll
            for (int ms = 0; ms < MS; ++ms)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) { ar[ms][nb] = nr[ms][nb]; az[ms][nb] = nz[ms][nb]; ain[ms][nb] = nn[ms][nb]; }
        }
        CTO_PH(3);
    };
    for (int step = 0; step + 1 < T; ++step) step_body(step, std::false_type{});
    step_body(T - 1, std::true_type{});
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
        const unsigned short* hl = hbuf + (T & 1) * (2 * TILE * HSB);
        const int tl = t_of(T - 1);
#pragma unroll
        for (int kh = 0; kh < NH; ++kh) {
            load_Ah(0, kh, hl);
            load_F(0, kh, tl);
            fc1_chunk(0);
        }
        float* part = fc1_part + int64_t(dir) * B * 128;
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int site = site0 + ms * 16 + j;
                if (site < site_end)
                    *reinterpret_cast<f32x4*>(part + int64_t(site) * 128 + wave * 32 + nt * 16 + kg * 4) = accf[ms][nt] * sc.inv_f;
            }
    }
}

}  // namespace cto
