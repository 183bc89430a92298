// EXPERIMENT, not compiled into the product: the ping-pong schedule of the fp32 recurrent kernel (round 4).  It was appended to
// csrc/gru_kernel.h (inside namespace cto) and launched from csrc/gru.hip for 32-site tiles under CTO_GRU_PP=1; results bit-equal
// to k_gru_layer_rot.  Measured on MI355X (tools/ab.py, 4096 sites): layer 2 1.101 vs 1.089 ms, layer 1 + tail 0.306 vs 0.287 ms.
// Phase stamps (-DCTO_GRU_CLOCKS): the same number of shader cycles as the rotated kernel (layer 2: 2.529 M vs 2.536 M per
// workgroup) - the gate arithmetic is hidden, the chunk loop is less efficient (twice the operand requests per MFMA) - at a
// lower ratio of s_memtime ticks to wall time (2.05 vs 2.15 G/s in the instrumented builds; rocm-smi reads 2.39 GHz during the
// default bench, so not a power limit of the default path - not separated further).
// DESIGN.md section 6, round 4.
// --------------------------------------------------------------------------------------------
// Ping-pong schedule of the same recurrence (32-site tiles): the two 16-site sub-tiles of a workgroup are independent recurrences
// that share nothing but the weights, so they run half a step apart - while the matrix pipe works through ALL the chunks (x part
// and h part) of sub-tile s for time step t, the same wave's VALU does the gate arithmetic of the other sub-tile, whose sums were
// completed in the previous half-step, and publishes its new state:
//     half-step k = 2 t + s :   barrier;   MFMA: (r, z, n_x, n_h)(s, t) <- bias + x_t[s] W_ih^T + h_{t-1}[s] W_hh^T  [+ fc1 on h_{t-1}[s]]
//                                          VALU: gates(1 - s, t') -> h_t'[1 - s] -> LDS            (t' = (k - 1) / 2)
// The gates have a whole sub-tile's MFMAs of a step to hide under (the rotated schedule gives them the x part of the next step,
// which in layer 1 is 108 MFMAs), the h tile needs one buffer (a sub-tile's rows are written in a half-step in which nobody reads
// them), the accumulators of "the next step's x part" are gone.  The price: every weight fragment feeds one MFMA per k-step
// instead of two - twice the operand requests per step, affordable since the weights come in fragment order - and two barriers
// per step.
// --------------------------------------------------------------------------------------------
template <int KIN, int KP, int H, bool FUSE_FC1>
__global__ __launch_bounds__(256) void k_gru_layer_pp(const float* __restrict__ x, const float* __restrict__ Wcat,
                                                      const float* __restrict__ bias, float* __restrict__ out,
                                                      const float* __restrict__ fc1w, float* __restrict__ fc1_part, int B, int site_begin,
                                                      int site_end) {
    constexpr int NB = H / 64, T = 33, HS = H + 4, NX = KP / 16, NH = H / 16, NC = NX + NH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TILE = 32, NTHR = 256;
    constexpr int XS = KP + 4;
    constexpr int XW = (KIN % 4 == 0) ? 4 : ((KIN % 2 == 0) ? 2 : 1);
    constexpr int XQ = TILE * (KIN / XW);
    constexpr int XPER = (XQ + NTHR - 1) / NTHR;
    float* hbuf = smem;                   // [TILE][HS]
    float* xbuf = smem + TILE * HS;       // [2][TILE][XS]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kg = lane >> 4;
    const int dir = blockIdx.x & 1;
    const int site0 = site_begin + (blockIdx.x >> 1) * TILE;
    const float* bd = bias + dir * 4 * H;
    auto t_of = [&](int step) { return dir == 0 ? step : T - 1 - step; };
#ifdef CTO_GRU_CLOCKS
    const long long c0 = clock64(), w0 = wall_clock64();
    long long ph[4] = {0, 0, 0, 0}, tph = 0;     // barrier wait, start-up up to the first chunk, chunk loop, half-step tail
#endif

    for (int i = threadIdx.x; i < TILE * HS; i += NTHR) hbuf[i] = 0.f;       // h_{-1} = 0
    for (int i = threadIdx.x; i < 2 * TILE * XS; i += NTHR) xbuf[i] = 0.f;   // K padding and rows past the batch stay 0

    float bia[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int hcol = (wave * NB + nb) * 16 + j;
#pragma unroll
        for (int q = 0; q < 4; ++q) bia[nb][q] = bd[q * H + hcol];
    }
    // fragment-ordered weights through buffer loads, as in k_gru_layer_rot
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    constexpr unsigned W_WAVE_BYTES = unsigned(NC) * NB * 3 * 1024u, F_T_BYTES = 4u * NH * 2 * 1024u;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Wcat) + (int64_t(dir) * 4 + wave_u) * (W_WAVE_BYTES / 4), 0, int(W_WAVE_BYTES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(FUSE_FC1 ? fc1w : Wcat) + (FUSE_FC1 ? (int64_t(dir) * T * 4 + wave_u) * (NH * 2 * 256) : 0), 0,
        FUSE_FC1 ? int(T * F_T_BYTES) : 0, 0x00020000);
    auto buf16 = [&](const __amdgpu_buffer_rsrc_t& r, unsigned voffset, unsigned soffset) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, int(voffset), int(soffset), 0);
        return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
    };
    unsigned voff = unsigned(lane) * 16u;
    int opq = 0;

    float hprev[2][NB][4];
    f32x4 ar[2][NB], az[2][NB], ain[2][NB], ahn[2][NB], accf[2][2];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) hprev[ms][nb][r] = 0.f;
        accf[ms][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        accf[ms][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    float4 xstage[XPER];
    auto x_fetch = [&](int t) {
#pragma unroll
        for (int q = 0; q < XPER; ++q) {
            const int u = min(int(threadIdx.x) + q * NTHR, XQ - 1);
            const int row = u / (KIN / XW), c = (u - row * (KIN / XW)) * XW;
            const int site = min(site0 + row, site_end - 1);     // rows past the batch read its last row and are never looked at
            const float* src = x + (int64_t(site) * T + t) * KIN + c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (XW == 4) v = *reinterpret_cast<const float4*>(src);
            else if constexpr (XW == 2) { const float2 w2 = *reinterpret_cast<const float2*>(src); v.x = w2.x; v.y = w2.y; }
            else v.x = *src;
            xstage[q] = v;
        }
    };
    auto x_commit = [&](int buf) {
        float* xb = xbuf + buf * (TILE * XS);
#pragma unroll
        for (int q = 0; q < XPER; ++q) {
            const int u = threadIdx.x + q * NTHR;
            const int row = u / (KIN / XW), c = (u - row * (KIN / XW)) * XW;
            if (XQ % NTHR == 0 || u < XQ) {
                float* dst = xb + row * XS + c;
                if constexpr (XW == 4) *reinterpret_cast<float4*>(dst) = xstage[q];
                else if constexpr (XW == 2) *reinterpret_cast<float2*>(dst) = make_float2(xstage[q].x, xstage[q].y);
                else *dst = xstage[q].x;
            }
        }
    };

    float4 Bq[2][NB][3], Fq[2][2], Aq[2];
    constexpr bool TAIL1 = (KIN % 16 != 0) && (KIN - 16 * (NX - 1) <= 4);       // see k_gru_layer_rot
    constexpr int NE_LAST = TAIL1 ? 1 : ((KIN - 16 * (NX - 1)) < 4 ? (KIN - 16 * (NX - 1)) : 4);
    auto load_B = [&](int buf, int c) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const unsigned unit = unsigned((c * NB + nb) * 3 + q);
                if (TAIL1 && c == NX - 1)
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
    // A fragment of chunk c for sub-tile s: x chunks from the x tile of this time step, h chunks from the h tile
    auto load_A = [&](int buf, int c, int s, const float* xc) {
        if (c < NX) {
            if (TAIL1 && c == NX - 1) Aq[buf].x = xc[(s * 16 + j) * XS + c * 16 + kg];
            else Aq[buf] = *reinterpret_cast<const float4*>(xc + (s * 16 + j) * XS + c * 16 + 4 * kg);
        } else {
            Aq[buf] = *reinterpret_cast<const float4*>(hbuf + (s * 16 + j) * HS + (c - NX) * 16 + 4 * kg);
        }
    };
    auto gate_nb = [&](int ms, int nb, int t) {
        const int hcol = (wave * NB + nb) * 16 + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float rg = fast_sigmoid(ar[ms][nb][r]);
            const float zg = fast_sigmoid(az[ms][nb][r]);
            const float ng = fast_tanh(ain[ms][nb][r] + rg * ahn[ms][nb][r]);
            const float hv = ng + zg * (hprev[ms][nb][r] - ng);      // (1 - z) * n + z * h
            hprev[ms][nb][r] = hv;
            hbuf[(ms * 16 + kg * 4 + r) * HS + hcol] = hv;
        }
        (void)t;
    };
    // the 16 rows of sub-tile s hold h of time tt and are stable for this half-step: out to HBM as 16-byte pieces
    constexpr int OPER = (16 * (H / 4) + NTHR - 1) / NTHR;
    auto store_rows = [&](int s, int tt) {
#pragma unroll
        for (int q = 0; q < OPER; ++q) {
            const int u = threadIdx.x + q * NTHR;
            const int row = s * 16 + u / (H / 4), c4 = (u % (H / 4)) * 4;
            if (u < 16 * (H / 4) && site0 + row < site_end)
                *reinterpret_cast<float4*>(out + (int64_t(site0 + row) * T + tt) * (2 * H) + dir * H + c4) =
                    *reinterpret_cast<const float4*>(hbuf + row * HS + c4);
        }
    };

#ifndef CTO_GRU_PP_IL
#define CTO_GRU_PP_IL 2
#endif
    // one half-step: sub-tile S, step index `step` (time t); gates of the other sub-tile for step index gstep (>= 0) ride along
    auto half_step = [&](auto s_tag, auto first_tag, int step) {
        constexpr int S = decltype(s_tag)::value, O = 1 - S;
        constexpr bool FIRST = decltype(first_tag)::value;       // step 0: nothing to store yet, and sub-tile 0 has no gates to bring along
        constexpr bool GATES = !(FIRST && S == 0);
        const int gstep = S == 0 ? step - 1 : step;
        const int t = t_of(step);
        const int tprev = step == 0 ? t : t_of(step - 1);
        const float* xc = xbuf + (step & 1) * (TILE * XS);
        opq = 0;
        asm volatile("" : "+v"(opq));      // keeps the (step-invariant) weight loads inside the time loop
        voff = unsigned(lane + opq) * 16u;
#ifdef CTO_GRU_CLOCKS
        tph = clock64();
#endif
        lds_barrier();                     // h_{t-1}[S] (gates of the previous half-step) and x_t (committed a step ago) are in LDS
        CTO_PH(0);
        if (S == 0 && step + 1 < T) x_fetch(t_of(step + 1));
        load_A(0, 0, S, xc);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            ar[S][nb] = f32x4{bia[nb][0], bia[nb][0], bia[nb][0], bia[nb][0]};
            az[S][nb] = f32x4{bia[nb][1], bia[nb][1], bia[nb][1], bia[nb][1]};
            ain[S][nb] = f32x4{bia[nb][2], bia[nb][2], bia[nb][2], bia[nb][2]};
            ahn[S][nb] = f32x4{bia[nb][3], bia[nb][3], bia[nb][3], bia[nb][3]};
        }
        CTO_PH(1);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int cur = c & 1, nxt = cur ^ 1;
            if (c + 1 < NC) {
                load_B(nxt, c + 1);
                load_A(nxt, c + 1, S, xc);
                if (c + 1 >= NX) load_F(nxt, c + 1 - NX, tprev);
            } else {
                load_B(nxt, 0);            // first chunk of the next half-step (NC is odd or even: parity handled below)
            }
            if constexpr (!FUSE_FC1 && !FIRST) {
                if (c == 1) store_rows(S, tprev);
            }
            const int ne = (c == NX - 1) ? NE_LAST : 4;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float4 br = Bq[cur][nb][0], bz = Bq[cur][nb][1], bn = Bq[cur][nb][2];
                const float brv[4] = {br.x, br.y, br.z, br.w}, bzv[4] = {bz.x, bz.y, bz.z, bz.w}, bnv[4] = {bn.x, bn.y, bn.z, bn.w};
                const float4 a4 = Aq[cur];
                const float avv[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (e >= ne) break;
                    ar[S][nb] = mfma16(avv[e], brv[e], ar[S][nb]);
                    az[S][nb] = mfma16(avv[e], bzv[e], az[S][nb]);
                    if (c < NX) ain[S][nb] = mfma16(avv[e], bnv[e], ain[S][nb]);
                    else ahn[S][nb] = mfma16(avv[e], bnv[e], ahn[S][nb]);
                }
            }
            if constexpr (FUSE_FC1) {
                if (c >= NX) {
                    const float4 a4 = Aq[cur];
                    const float avv[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const float4 f4 = Fq[cur][nt];
                            const float fv = e == 0 ? f4.x : (e == 1 ? f4.y : (e == 2 ? f4.z : f4.w));
                            accf[S][nt] = mfma16(avv[e], fv, accf[S][nt]);
                        }
                }
            }
            // gate arithmetic of the other sub-tile, one block of hidden units at a time, spread over the chunks
            if constexpr (GATES) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    if ((nb * NC) / NB == c) gate_nb(O, nb, t_of(gstep));
            }
#ifndef CTO_GRU_NO_INTERLEAVE
#pragma unroll
            for (int g = 0; g < NB * 3 + (FUSE_FC1 ? 2 : 0); ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, CTO_GRU_PP_IL, 0);      // MFMAs
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                  // 1 VMEM read
            }
            __builtin_amdgcn_sched_group_barrier(0x008, CTO_GRU_PP_IL, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                      // 1 DS read
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
        CTO_PH(2);
        if constexpr ((NC & 1) != 0) {      // odd chunk count: the next half-step's first weights landed in buffer 1
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 3; ++q) Bq[0][nb][q] = Bq[1][nb][q];
        }
        if (S == 1 && step + 1 < T) x_commit((step + 1) & 1);
        CTO_PH(3);
    };

    // ---- prologue: x_0 ----
    x_fetch(t_of(0));
    __syncthreads();                    // zero fill complete
    x_commit(0);
    load_B(0, 0);
    if constexpr (FUSE_FC1) load_F(0, 0, t_of(0));
    half_step(std::integral_constant<int, 0>{}, std::true_type{}, 0);
    half_step(std::integral_constant<int, 1>{}, std::true_type{}, 0);             // ... with the gates of (sub-tile 0, step 0)
    for (int step = 1; step < T; ++step) {
        half_step(std::integral_constant<int, 0>{}, std::false_type{}, step);     // ... with the gates of (sub-tile 1, step - 1)
        half_step(std::integral_constant<int, 1>{}, std::false_type{}, step);     // ... with the gates of (sub-tile 0, step)
    }
#ifdef CTO_GRU_CLOCKS
    if (blockIdx.x == 7 && threadIdx.x == 0) {
        const int o = FUSE_FC1 ? 4 : 0;
        g_gru_clk[o] = clock64() - c0; g_gru_clk[o + 1] = wall_clock64() - w0;
        g_gru_clk[8 + o] = ph[0]; g_gru_clk[9 + o] = ph[1]; g_gru_clk[10 + o] = ph[2]; g_gru_clk[11 + o] = ph[3];
    }
#endif
    // gates of (sub-tile 1, T - 1): nothing left to hide them under
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) gate_nb(1, nb, t_of(T - 1));
    __syncthreads();
    if constexpr (!FUSE_FC1) {
        store_rows(0, t_of(T - 1));
        store_rows(1, t_of(T - 1));
    } else {
        // fc1 contribution of the last state, then one partial slab per direction
        const int tl = t_of(T - 1);
#pragma unroll
        for (int kh = 0; kh < NH; ++kh) {
            load_F(0, kh, tl);
#pragma unroll
            for (int ms = 0; ms < 2; ++ms) {
                const float4 a = *reinterpret_cast<const float4*>(hbuf + (ms * 16 + j) * HS + kh * 16 + 4 * kg);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const float4 f = Fq[0][nt];
                    accf[ms][nt] = mfma16(a.x, f.x, accf[ms][nt]);
                    accf[ms][nt] = mfma16(a.y, f.y, accf[ms][nt]);
                    accf[ms][nt] = mfma16(a.z, f.z, accf[ms][nt]);
                    accf[ms][nt] = mfma16(a.w, f.w, accf[ms][nt]);
                }
            }
        }
        float* part = fc1_part + int64_t(dir) * B * 128;
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
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
