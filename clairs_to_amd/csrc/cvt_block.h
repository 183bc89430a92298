// Fused CvT transformer block (clairs/model.py:134-147: x = attn(x) + x; x = ff(x) + x) for gfx950.
//
// One workgroup (8 waves, two per SIMD) owns TS sites = R = TS*W activation rows of C channels and keeps the
// residual stream, every intermediate (LayerNorm output, depth-wise conv outputs, per-head q/k/v, attention
// output, FFN hidden chunk) in LDS; only the weights stream in (from L2, straight into MFMA B registers) and only
// the updated residual stream goes back to HBM.  The unfused path spends ~8 launches and ~10 HBM round trips per
// block on the same work.
//
//   phase 0  h tile -> LDS                                  (CIN > 0, first block of a stage: the stage input tile instead,
//                                                            then the stride-2 conv embedding as an im2col GEMM + LayerNorm)
//   phase 1  y  = LN(h; norm0)                              (model.py:57-76; 16 lanes per row, DPP row reductions)
//   phase 2  yq = BN(DW_s1(y)) in place, ykv = BN(DW_s2(y)) (model.py:91-100, 112-113)
//   phase 3  per head: q_h, k_h, v_h (MFMA) -> LDS; softmax(q_h k_h^T / 8) v_h per site (VALU, <= 9x5
//            scores); out-projection accumulated over heads in registers (MFMA, K = 64 per head)
//   phase 4  h += to_out(o) + bias
//   phase 5  y  = LN(h; norm1)
//   phase 6  per 128-wide chunk of the 4C hidden units: u = GELU(y W1^T + b1) -> LDS, acc += u W2^T
//   phase 7  h += acc + b2 in LDS, then coalesced 16-byte stores to HBM
//   phase 8  (HEAD, last block of the network) fc1 over the LDS image of the tile -> SELU -> K x (fc2, fc3) instead of the store
//
// GEMMs: fp32 MFMA 16x16x4; A fragments from LDS (ds_read_b128 = four k-steps), B fragments from global
// (one 16-byte load per lane per n-tile per 16-wide k chunk, prefetched two chunks ahead); waves 0-3 / 4-7 split the
// M tiles, wave & 3 owns the n-tiles, so each weight fragment is reused for 2-3 m-tiles and one wave's waits and VALU
// epilogues overlap the other's MFMAs on the same SIMD.
#pragma once
#include <type_traits>
#include "nn_kernels.h"

namespace cto {

struct CvtBlockParams {
    const float *n0g, *n0b, *dwq, *bnq, *wq, *dwkv, *bnkv, *wkv, *wo, *bo, *n1g, *n1b, *w1, *b1, *w2, *b2;
    long long* prof;   // debug: phase time stamps (s_memtime) of workgroup 0 / thread 0 when non-null (CTO_BLOCK_PROF=1)
    // first block of a stage (CIN > 0): the stage's conv embedding + LayerNorm run here instead of reading h
    const float *xin, *wembp, *bemb, *lng, *lnb;   // x [B][2W-1][CIN]; wembp [C][KCHE*16] (positions padded to PS)
    // last block of the network (HEAD): fc1 + classifier tail run here instead of writing h
    const float *w1p, *b1h;                        // fc1 [128][KCH1*16] over the LDS image of h (rows padded to RS)
};

// The first two 16-wide k chunks of a GEMM's weights, requested early (before the barriers / VALU phases that
// precede the GEMM) so that the matrix pipe does not start every GEMM with an exposed L2 round trip.
template <int NTW>
struct BPre {
    float4 b0[NTW], b1[NTW];
};
template <int NTW, int KCH>
__device__ __forceinline__ BPre<NTW> prefetch_b(const float* const (&wrow)[NTW]) {
    BPre<NTW> p;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        p.b0[nt] = ldg4(wrow[nt]);
        p.b1[nt] = KCH > 1 ? ldg4(wrow[nt] + 16) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return p;
}

// acc[mt][nt] += A[mt*16 .. +16][0 .. KCH*16) * W[n-tile rows][same k];  wrow[nt] already points at
// W[(n0 + nt*16 + j)][4*kg].  A rows are `lda` floats apart in LDS.  `pre` holds chunks 0 and 1.
template <int MT, int NTW, int KCH>
__device__ __forceinline__ void gemm_lds(const float* __restrict__ A, int lda, const float* const (&wrow)[NTW],
                                         const BPre<NTW>& pre, f32x4 (&acc)[MT][NTW], int j, int kg) {
    float4 Bq[3][NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) { Bq[0][nt] = pre.b0[nt]; Bq[1][nt] = pre.b1[nt]; }
    float4 a[2][MT];     // A fragments are fetched one chunk ahead too (LDS latency is exposed with 1 wave per SIMD)
    const float* Arow = A + j * lda + 4 * kg;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[0][mt] = *reinterpret_cast<const float4*>(Arow + mt * 16 * lda);
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
        if (c + 2 < KCH) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) Bq[(c + 2) % 3][nt] = ldg4(wrow[nt] + (c + 2) * 16);
        }
        if (c + 1 < KCH) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[(c + 1) & 1][mt] = *reinterpret_cast<const float4*>(Arow + mt * 16 * lda + (c + 1) * 16);
        }
        // Without this fence the scheduler sinks the operand requests above to the END of the chunk (nothing here needs them),
        // i.e. right in front of the s_waitcnt of the chunk that does: the "prefetch" then exposes a full LDS / L2 round trip
        // per chunk.  Requests first, then this chunk's MFMAs; the other wave of the SIMD covers the short issue burst.
        __builtin_amdgcn_sched_barrier(0);
        // k-step outermost, tiles innermost: consecutive MFMAs never hit the same accumulator (dependent latency 40 cycles
        // vs issue interval 32 for v_mfma_f32_16x16x4_f32)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const float4 b4 = Bq[c % 3][nt];
                const float bv = e == 0 ? b4.x : (e == 1 ? b4.y : (e == 2 ? b4.z : b4.w));
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float4 a4 = a[c & 1][mt];
                    const float av = e == 0 ? a4.x : (e == 1 ? a4.y : (e == 2 ? a4.z : a4.w));
                    acc[mt][nt] = mfma16(av, bv, acc[mt][nt]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// One 16-row m-tile (the classifier GEMMs: 16 sites per workgroup): weights are used once, so the loop is bound by
// the L2 round trip, not by the matrix pipe, unless many loads are in flight - DEPTH 16-wide k chunks per n-tile are
// requested at a time, one group ahead of the MFMAs; even / odd chunks alternate between two accumulator sets.
template <int NTW, int DEPTH>
struct BGroup {
    float4 b[DEPTH][NTW];
};
template <int NTW, int KCH, int DEPTH>
__device__ __forceinline__ void load_group(BGroup<NTW, DEPTH>& g, const float* const (&wrow)[NTW], int c0) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
            if (c0 + d < KCH) g.b[d][nt] = ldg4(wrow[nt] + (c0 + d) * 16);
}
template <int NTW, int KCH, int DEPTH>
__device__ __forceinline__ void mfma_group(const BGroup<NTW, DEPTH>& g, const float* Arow, int c0, f32x4 (&acc)[2][NTW]) {
    float4 a[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (c0 + d < KCH) a[d] = *reinterpret_cast<const float4*>(Arow + (c0 + d) * 16);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        if (c0 + d >= KCH) break;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float av = e == 0 ? a[d].x : (e == 1 ? a[d].y : (e == 2 ? a[d].z : a[d].w));
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const float4 b4 = g.b[d][nt];
                const float bv = e == 0 ? b4.x : (e == 1 ? b4.y : (e == 2 ? b4.z : b4.w));
                acc[d & 1][nt] = mfma16(av, bv, acc[d & 1][nt]);
            }
        }
    }
}
template <int NTW, int KCH, int DEPTH>
__device__ __forceinline__ void gemm_m1(const float* __restrict__ A, int lda, const float* const (&wrow)[NTW],
                                        const BGroup<NTW, DEPTH>& first, f32x4 (&acc)[2][NTW], int j, int kg) {
    constexpr int NG = (KCH + DEPTH - 1) / DEPTH;
    const float* Arow = A + j * lda + 4 * kg;
    BGroup<NTW, DEPTH> g[2];
    g[0] = first;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        if (i + 1 < NG) load_group<NTW, KCH, DEPTH>(g[(i + 1) & 1], wrow, (i + 1) * DEPTH);
        __builtin_amdgcn_sched_barrier(0);      // keep the next group's loads ahead of this group's MFMAs
        mfma_group<NTW, KCH, DEPTH>(g[i & 1], Arow, i * DEPTH, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- classifier tail shared by both networks (clairs/model.py:245-261, 451-467): K heads of fc2 (128 -> 128) -> SELU ->
// fc3 (128 -> 2) -> SELU on a 16-site tile whose SELU(fc1) activations sit in LDS.  8 waves; wave w owns hidden units
// [16w, 16w+16) of every head, two heads per pass (two independent accumulators keep the matrix pipe at issue rate).
struct HeadTailParams {
    const float *w2, *b2;   // [K*128][128], [K*128]
    const float *w3, *b3;   // [K][2][128], [K][2]
    float* logits;          // [K][B][2]
    int K;
};
constexpr int HEAD_T1S = 132;                       // LDS row stride of the fc1 activations [16][128]
__host__ __device__ constexpr int head_t2s(int K) { return K * 128 + 4; }
__host__ __device__ constexpr int head_lds_floats(int K) { return 16 * HEAD_T1S + 16 * head_t2s(K); }

// t1: [16][HEAD_T1S] (in), t2: [16][head_t2s(K)] scratch.  All 512 threads call; ends without a barrier.
template <int K>
__device__ __forceinline__ void head_tail_k(const float* t1, float* t2, const HeadTailParams& hp, int64_t B, int64_t site0,
                                            int nsite) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kg = lane >> 4;
    constexpr int T2S = head_t2s(K);
    static_assert(K % 2 == 0, "heads are processed in pairs");
    const float* wr[2];
    wr[0] = hp.w2 + int64_t(wave * 16 + j) * 128 + 4 * kg;
    wr[1] = wr[0] + 128 * 128;
    BGroup<2, 8> g[2];
    load_group<2, 8, 8>(g[0], wr, 0);
    const float* Arow = t1 + j * HEAD_T1S + 4 * kg;
#pragma unroll
    for (int pi = 0; pi < K / 2; ++pi) {
        if (pi + 1 < K / 2) {        // the next pair of heads' weights fly under this pair's MFMAs
            const float* wn[2] = {wr[0] + (pi + 1) * 2 * 128 * 128, wr[1] + (pi + 1) * 2 * 128 * 128};
            load_group<2, 8, 8>(g[(pi + 1) & 1], wn, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[a][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        mfma_group<2, 8, 8>(g[pi & 1], Arow, 0, acc);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int col = (pi * 2 + q) * 128 + wave * 16 + j;
            const float bv = hp.b2[col];
#pragma unroll
            for (int r = 0; r < 4; ++r) t2[(4 * kg + r) * T2S + col] = selu_fast(acc[0][q][r] + acc[1][q][r] + bv);
        }
    }
    __syncthreads();
    // fc3: 16 sites x K heads x 2 outputs, four lanes per dot product of length 128
    const int part = tid & 3;
    for (int idx = tid >> 2; idx < 16 * K * 2; idx += blockDim.x >> 2) {
        const int site = idx / (2 * K), rem = idx - site * 2 * K, hh = rem >> 1, o = rem & 1;
        const float4* u = reinterpret_cast<const float4*>(t2 + site * T2S + hh * 128 + part * 32);
        const float4* w = reinterpret_cast<const float4*>(hp.w3 + (hh * 2 + o) * 128 + part * 32);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 a = u[i], b = w[i];
            sum = fmaf(a.x, b.x, sum); sum = fmaf(a.y, b.y, sum); sum = fmaf(a.z, b.z, sum); sum = fmaf(a.w, b.w, sum);
        }
        sum += __shfl_xor(sum, 1, 4);
        sum += __shfl_xor(sum, 2, 4);
        if (part == 0 && site < nsite) hp.logits[(int64_t(hh) * B + site0 + site) * 2 + o] = selu_f(sum + hp.b3[hh * 2 + o]);
    }
}
__device__ __forceinline__ void head_tail(const float* t1, float* t2, const HeadTailParams& hp, int64_t B, int64_t site0,
                                          int nsite) {
    if (hp.K == 4) head_tail_k<4>(t1, t2, hp, B, site0, nsite);
    else head_tail_k<6>(t1, t2, hp, B, site0, nsite);
}

// Stand-alone classifier tail for fc1 partial sums that already sit in HBM (BiGRU: one slab per direction from the fused
// layer-2 kernel; unfused CvT path: split-K slabs): t1 = SELU(sum_z slab_z + b1), then head_tail.
__global__ __launch_bounds__(512) void k_head(const float* __restrict__ slabs, int S, int64_t slab_stride,
                                              const float* __restrict__ b1, HeadTailParams hp, int64_t B) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* t1 = smem;
    float* t2 = smem + 16 * HEAD_T1S;
    const int64_t site0 = int64_t(blockIdx.x) * 16;
    const int nsite = int(min(int64_t(16), B - site0));
    {
        const int site = threadIdx.x >> 5, c4 = (threadIdx.x & 31) * 4;     // 512 threads = 16 sites x 32 float4
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (site < nsite) {
            for (int z = 0; z < S; ++z) {
                const float4 a = *reinterpret_cast<const float4*>(slabs + z * slab_stride + (site0 + site) * 128 + c4);
                v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
            }
        }
        const float4 bb = *reinterpret_cast<const float4*>(b1 + c4);
        *reinterpret_cast<float4*>(t1 + site * HEAD_T1S + c4) =
            make_float4(selu_fast(v.x + bb.x), selu_fast(v.y + bb.y), selu_fast(v.z + bb.z), selu_fast(v.w + bb.w));
    }
    __syncthreads();
    head_tail(t1, t2, hp, B, site0, nsite);
}

// LDS position stride of the stage input [site][2W][PS] for the in-block conv embedding: >= CIN, multiple of 4 (float4
// A fragments) and = 4 mod 16, so that the im2col row stride 2*PS is = 8 mod 32 banks (4-way instead of 16-way conflicts).
__host__ __device__ constexpr int emb_ps(int cin) { return cin > 0 ? ((cin - 4 + 15) / 16) * 16 + 4 : 0; }
__host__ __device__ constexpr int emb_kch(int cin) { return (3 * emb_ps(cin) + 15) / 16; }

template <int C, int W, int WKV, int TS>
struct CvtBlockGeom {
    static constexpr int R = TS * W, RKV = TS * WKV;
    static constexpr int MT = (R + 15) / 16, MTKV = (RKV + 15) / 16;
    static constexpr int RS = C + 4, QS = 68, HC = (4 * C < 128 ? 4 * C : 128), US = HC + 4;
    static constexpr int OFF_H = 0;
    static constexpr int OFF_Y = OFF_H + MT * 16 * RS;
    static constexpr int OFF_YKV = OFF_Y + MT * 16 * RS;
    static constexpr int OFF_Q = OFF_YKV + MTKV * 16 * RS;
    static constexpr int OFF_K = OFF_Q + MT * 16 * QS;
    static constexpr int OFF_V = OFF_K + MTKV * 16 * QS;
    static constexpr int OFF_P = OFF_V + MTKV * 16 * QS;
    static constexpr int SCRATCH = OFF_P - OFF_YKV;            // ykv|q|k|v region, re-used for the FFN hidden chunk
    static constexpr int U_FLOATS = MT * 16 * US;
    static constexpr int TOTAL = OFF_P + TS * W * WKV + (U_FLOATS > SCRATCH ? U_FLOATS - SCRATCH : 0);
    static constexpr size_t LDS_BYTES = size_t(TOTAL) * sizeof(float);
    static constexpr int ALIAS = TOTAL - OFF_Y;                // everything but the residual stream is free in phases 0 and 8
    static constexpr int KCH1 = (W * RS + 15) / 16;            // fc1 over the LDS image of one site's [W][RS] rows
    static constexpr int emb_floats(int cin) { return (MT * 32 + 4) * emb_ps(cin); }
};

// 8 waves per workgroup (two per SIMD): waves 0-3 and 4-7 split the M tiles of every GEMM between them (the
// n-tile owner is wave & 3), so one wave's LDS / L2 waits and VALU phases overlap the other's MFMAs.
constexpr int CVT_BLOCK_THREADS = 512;

// Workgroups of this geometry that fit one CU's 160 KB of LDS (at most 3 are asked for): the register budget follows from
// it through __launch_bounds__ (w = minimum waves per SIMD = 2 per resident 512-thread workgroup), otherwise a kernel that
// fits the LDS twice still runs alone because it was given 132+ registers.
template <int C, int W, int WKV, int TS>
__host__ __device__ constexpr int cvt_blocks_per_cu() {
    return CvtBlockGeom<C, W, WKV, TS>::LDS_BYTES * 3 <= 160 * 1024 ? 3 : (CvtBlockGeom<C, W, WKV, TS>::LDS_BYTES * 2 <= 160 * 1024 ? 2 : 1);
}

template <int C, int W, int WKV, int TS, int CIN, bool HEAD>
__global__ __launch_bounds__(CVT_BLOCK_THREADS, (2 * cvt_blocks_per_cu<C, W, WKV, TS>())) void k_cvt_block(float* __restrict__ h, CvtBlockParams p,
                                                                                                            HeadTailParams hp, int heads, int B) {
    using G = CvtBlockGeom<C, W, WKV, TS>;
    static_assert(CIN == 0 || G::emb_floats(CIN) <= G::ALIAS, "stage input tile does not fit the free LDS");
    static_assert(!HEAD || (TS == 16 && 64 + head_lds_floats(6) <= G::ALIAS), "classifier tail needs a 16-site tile");
    constexpr int NT = CVT_BLOCK_THREADS, NWV = NT / 64;
    constexpr int R = G::R, RKV = G::RKV, MT = G::MT, MTKV = G::MTKV, RS = G::RS, QS = G::QS, HC = G::HC, US = G::US;
    constexpr int MT0 = (MT + 1) / 2, MT1 = MT - MT0;
    // C = 128: every 128-wide output (k|v of a head, out-projection, both FFN GEMMs, the stage embedding) has exactly 8 n-tiles -
    // one per wave, all m-tiles each.  The waves are then balanced (the M split gives waves 0-3 three m-tiles and waves 4-7 two,
    // so half of the workgroup idled a third of every GEMM phase) and a wave requests one weight fragment per chunk, not two.
    constexpr bool NSPLIT = (C == 128);
    constexpr int NTC = NSPLIT ? 1 : (C >= 64 ? C / 64 : 1);   // n-tiles per wave when the output is C wide
    constexpr bool NSPLIT_F = (HC == 128);                     // the same for a 128-wide FFN hidden chunk (any C >= 32)
    constexpr int NTF = NSPLIT_F ? 1 : HC / 64;                // n-tiles per wave of one FFN hidden chunk
    constexpr int MTF = NSPLIT_F ? MT : (MT + 1) / 2;
    constexpr int MTC = NSPLIT ? MT : (MT + 1) / 2;            // m-tiles a wave holds of such an output
    static_assert(C % 16 == 0 && HC % 64 == 0, "channel count must be a multiple of 16");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sh = smem + G::OFF_H;
    float* sy = smem + G::OFF_Y;
    float* sykv = smem + G::OFF_YKV;
    float* sq = smem + G::OFF_Q;
    float* sk = smem + G::OFF_K;
    float* sv = smem + G::OFF_V;
    float* sp = smem + G::OFF_P;
    float* su = smem + G::OFF_YKV;       // alias: FFN hidden chunk [MT*16][US]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kg = lane >> 4;
    const int wn = wave & 3, mh = wave >> 2;                   // n-tile owner, M half
    const bool own_c = NSPLIT || (wn * NTC * 16) < C;          // C < 64: only some waves own columns of a C-wide output
    const int mbase = mh ? MT0 : 0, mcount = mh ? MT1 : MT0;   // this wave's m-tiles of [R]-row operands (M split)
    const int ctile0 = NSPLIT ? wave : wn * NTC;               // first column tile of a C-wide output
    const int ftile0 = NSPLIT_F ? wave : wn * NTF;             // ... of an FFN hidden chunk
    const int mbaseF = NSPLIT_F ? 0 : mbase, mcountF = NSPLIT_F ? MT : mcount;
    const int mbaseC = NSPLIT ? 0 : mbase, mcountC = NSPLIT ? MT : mcount;
    const int site0 = blockIdx.x * TS;
    const int nsite = min(TS, B - site0);
    const int rows_valid = nsite * W;
    const int inner = heads * 64;
    float* hg = h + int64_t(site0) * W * C;
    int nstamp = 0;
    auto stamp = [&]() { if (p.prof && blockIdx.x == 0 && tid == 0) p.prof[nstamp++] = clock64(); };
    stamp();

    // GEMM over this wave's M half; accumulators are sized for the larger half
    auto gemm_r = [&](auto ntw_tag, auto kch_tag, const float* A, int lda, const auto& wr, const auto& pre, auto& acc) {
        constexpr int NTW = decltype(ntw_tag)::value, KCH = decltype(kch_tag)::value;
        if (mh == 0) {
            gemm_lds<MT0, NTW, KCH>(A, lda, wr, pre, acc, j, kg);
        } else if constexpr (MT1 > 0) {
            f32x4 (&a1)[MT1][NTW] = reinterpret_cast<f32x4 (&)[MT1][NTW]>(acc);
            gemm_lds<MT1, NTW, KCH>(A + MT0 * 16 * lda, lda, wr, pre, a1, j, kg);
        }
    };
    auto gemm_c = [&](auto ntw_tag, auto kch_tag, const float* A, int lda, const auto& wr, const auto& pre, auto& acc) {
        constexpr int NTW = decltype(ntw_tag)::value, KCH = decltype(kch_tag)::value;
        if constexpr (NSPLIT) gemm_lds<MT, NTW, KCH>(A, lda, wr, pre, acc, j, kg);
        else gemm_r(ntw_tag, kch_tag, A, lda, wr, pre, acc);
    };
    auto gemm_f = [&](auto ntw_tag, auto kch_tag, const float* A, int lda, const auto& wr, const auto& pre, auto& acc) {
        constexpr int NTW = decltype(ntw_tag)::value, KCH = decltype(kch_tag)::value;
        if constexpr (NSPLIT_F) gemm_lds<MT, NTW, KCH>(A, lda, wr, pre, acc, j, kg);
        else gemm_r(ntw_tag, kch_tag, A, lda, wr, pre, acc);
    };
    using I1 = std::integral_constant<int, 1>;
    using INTC = std::integral_constant<int, NTC>;
    using INTF = std::integral_constant<int, NTF>;
    using KC = std::integral_constant<int, C / 16>;
    using K4 = std::integral_constant<int, 4>;
    using KH = std::integral_constant<int, HC / 16>;

    // ---- phase 0: residual stream tile -> LDS (pad rows zero); first block of a stage: stage input -> LDS instead ----
    if constexpr (HEAD) {   // fc1 sweeps the LDS image of h including the 4 pad columns of every row (zero weights): keep them finite
        for (int i = tid; i < MT * 16; i += NT) *reinterpret_cast<float4*>(sh + i * RS + C) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if constexpr (CIN == 0) {
        constexpr int NIT = (MT * 16 * (C / 4) + NT - 1) / NT;
        float4 stage[NIT];
#pragma unroll
        for (int q = 0; q < NIT; ++q) {
            const int i = tid + q * NT, r = i / (C / 4), c4 = (i - r * (C / 4)) * 4;
            stage[q] = (i < MT * 16 * (C / 4) && r < rows_valid) ? *reinterpret_cast<const float4*>(hg + r * C + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < NIT; ++q) {
            const int i = tid + q * NT, r = i / (C / 4), c4 = (i - r * (C / 4)) * 4;
            if (i < MT * 16 * (C / 4)) *reinterpret_cast<float4*>(sh + r * RS + c4) = stage[q];
        }
        for (int i = tid; i < (MTKV * 16 - RKV) * RS; i += NT) sykv[RKV * RS + i] = 0.f;
    } else {
        // x tile as [site][2W positions][PS]: slot 0 of a site is the conv's left zero pad, the next site's slot 0 doubles as
        // this site's right pad, so the im2col row of output (site, wo) is the contiguous run starting at row*2*PS
        constexpr int WIN = 2 * W - 1, PS = emb_ps(CIN), NPOS = MT * 32 + 4, VW = (CIN % 4 == 0) ? 4 : 2, SLOTS = PS / VW;
        static_assert(CIN % 2 == 0, "stage input channels must be even");
        float* sin = smem + G::OFF_Y;
        const float* xg = p.xin + int64_t(site0) * WIN * CIN;
        // all of a thread's pieces are requested before the first one is written to LDS: one HBM round trip, not one per piece
        constexpr int NIT = (NPOS * SLOTS + NT - 1) / NT;
        float4 stage[NIT];
#pragma unroll
        for (int q = 0; q < NIT; ++q) {
            const int i = tid + q * NT;
            const int pp = i / SLOTS, c = (i - pp * SLOTS) * VW;
            const int s = pp / (2 * W), pos = pp - s * 2 * W - 1;
            const bool ok = i < NPOS * SLOTS && pos >= 0 && s < nsite && c < CIN;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                const float* src = xg + (s * WIN + pos) * CIN + c;
                if constexpr (VW == 4) v = *reinterpret_cast<const float4*>(src);
                else { const float2 w2 = *reinterpret_cast<const float2*>(src); v.x = w2.x; v.y = w2.y; }
            }
            stage[q] = v;
        }
#pragma unroll
        for (int q = 0; q < NIT; ++q) {
            const int i = tid + q * NT;
            const int pp = i / SLOTS, c = (i - pp * SLOTS) * VW;
            if (i < NPOS * SLOTS) {
                if constexpr (VW == 4) *reinterpret_cast<float4*>(sin + pp * PS + c) = stage[q];
                else *reinterpret_cast<float2*>(sin + pp * PS + c) = make_float2(stage[q].x, stage[q].y);
            }
        }
    }
    lds_barrier();

    // Channel LayerNorm sh -> sy.  16 lanes per row (4 rows per wave at a time), each lane owning C/16 contiguous
    // channels: the row reductions are 4 DPP steps inside a 16-lane row instead of 6 cross-lane permutes through the
    // LDS crossbar - the wave-per-row version spent 12.5 k cycles per LayerNorm (10 % of a stage-3 block).
    auto layer_norm = [&](const float* src, float* dst, const float* g, const float* b) {
        constexpr int CPL = C / 16;
        const int l16 = lane & 15, grp = lane >> 4;
        float gv[CPL], bv[CPL];
#pragma unroll
        for (int i = 0; i < CPL; ++i) { gv[i] = g[l16 * CPL + i]; bv[i] = b[l16 * CPL + i]; }
        for (int r = wave * 4 + grp; r < MT * 16; r += NWV * 4) {
            const float* xr = src + r * RS + l16 * CPL;
            float v[CPL], sum = 0.f;
            if constexpr (CPL % 4 == 0) {
#pragma unroll
                for (int i = 0; i < CPL; i += 4) {
                    const float4 q = *reinterpret_cast<const float4*>(xr + i);
                    v[i] = q.x; v[i + 1] = q.y; v[i + 2] = q.z; v[i + 3] = q.w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < CPL; ++i) v[i] = xr[i];
            }
#pragma unroll
            for (int i = 0; i < CPL; ++i) sum += v[i];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
            const float mean = sum / float(C);
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < CPL; ++i) { v[i] -= mean; sq += v[i] * v[i]; }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 16);
            const float inv = 1.0f / (sqrtf(sq / float(C)) + 1e-5f);
            float* yr = dst + r * RS + l16 * CPL;
            if constexpr (CPL % 4 == 0) {
#pragma unroll
                for (int i = 0; i < CPL; i += 4)
                    *reinterpret_cast<float4*>(yr + i) = make_float4(v[i] * inv * gv[i] + bv[i], v[i + 1] * inv * gv[i + 1] + bv[i + 1],
                                                                     v[i + 2] * inv * gv[i + 2] + bv[i + 2], v[i + 3] * inv * gv[i + 3] + bv[i + 3]);
            } else {
#pragma unroll
                for (int i = 0; i < CPL; ++i) yr[i] = v[i] * inv * gv[i] + bv[i];
            }
        }
    };

    if constexpr (CIN > 0) {
        // conv embedding (model.py:195, only the middle kernel row is live) + bias, then the stage's channel LayerNorm
        constexpr int PS = emb_ps(CIN), KE = emb_kch(CIN);
        const float* we_r[NTC];
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) we_r[nt] = p.wembp + int64_t(own_c ? (ctile0 + nt) * 16 + j : j) * (KE * 16) + 4 * kg;
        const BPre<NTC> pre_e = prefetch_b<NTC, KE>(we_r);
        f32x4 acc_e[MTC][NTC];
#pragma unroll
        for (int mt = 0; mt < MTC; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTC; ++nt) acc_e[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (own_c) gemm_c(INTC{}, std::integral_constant<int, KE>{}, smem + G::OFF_Y, 2 * PS, we_r, pre_e, acc_e);
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) {
            if (!own_c) break;
            const int col = (ctile0 + nt) * 16 + j;
            const float bv = p.bemb[col];
#pragma unroll
            for (int mt = 0; mt < MTC; ++mt)
                if (mt < mcountC) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sh[((mbaseC + mt) * 16 + 4 * kg + r) * RS + col] = acc_e[mt][nt][r] + bv;
                }
        }
        lds_barrier();
        for (int i = tid; i < (MTKV * 16 - RKV) * RS; i += NT) sykv[RKV * RS + i] = 0.f;
        layer_norm(sh, sh, p.lng, p.lnb);
        lds_barrier();
    }

    stamp();
    // ---- phase 1 ----
    layer_norm(sh, sy, p.n0g, p.n0b);
    lds_barrier();
    stamp();

    // ---- phase 2: depth-wise 3-tap conv + BatchNorm; q path in place, kv path (stride 2) to sykv ----
    {
        static_assert(NT % C == 0, "column mapping assumes C divides the block size");
        const int c = tid % C;                 // every column this thread handles has the same channel
        const float q0 = p.dwq[c * 3], q1 = p.dwq[c * 3 + 1], q2 = p.dwq[c * 3 + 2];
        const float k0 = p.dwkv[c * 3], k1 = p.dwkv[c * 3 + 1], k2 = p.dwkv[c * 3 + 2];
        const float qm = p.bnq[c], qi = p.bnq[C + c], qw = p.bnq[2 * C + c], qb = p.bnq[3 * C + c];
        const float km = p.bnkv[c], ki = p.bnkv[C + c], kw = p.bnkv[2 * C + c], kb = p.bnkv[3 * C + c];
        for (int s = tid / C; s < TS; s += NT / C) {
            float y[W];
#pragma unroll
            for (int w = 0; w < W; ++w) y[w] = sy[(s * W + w) * RS + c];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const float l = w > 0 ? y[w - 1] : 0.f, r = w + 1 < W ? y[w + 1] : 0.f;
                const float d = q0 * l + q1 * y[w] + q2 * r;
                sy[(s * W + w) * RS + c] = (d - qm) * qi * qw + qb;
            }
#pragma unroll
            for (int wo = 0; wo < WKV; ++wo) {
                const int w = 2 * wo;
                const float l = w > 0 ? y[w - 1] : 0.f, r = w + 1 < W ? y[w + 1] : 0.f;
                const float d = k0 * l + k1 * y[w] + k2 * r;
                sykv[(s * WKV + wo) * RS + c] = (d - km) * ki * kw + kb;
            }
        }
    }
    lds_barrier();

    stamp();
    // ---- phase 3: attention, head by head; out-projection accumulates in registers ----
    f32x4 acc_o[MTC][NTC];
#pragma unroll
    for (int mt = 0; mt < MTC; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) acc_o[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto q_rows = [&](int hh, const float* (&wr)[1]) { wr[0] = p.wq + int64_t(hh * 64 + wn * 16 + j) * C + 4 * kg; };
    auto o_rows = [&](int hh, const float* (&wr)[NTC]) {
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) wr[nt] = p.wo + int64_t(own_c ? (ctile0 + nt) * 16 + j : j) * inner + hh * 64 + 4 * kg;
    };
    auto w1_rows = [&](int cc, const float* (&wr)[NTF]) {
#pragma unroll
        for (int nt = 0; nt < NTF; ++nt) wr[nt] = p.w1 + int64_t(cc * HC + (ftile0 + nt) * 16 + j) * C + 4 * kg;
    };
    auto w2_rows = [&](int cc, const float* (&wr)[NTC]) {
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) wr[nt] = p.w2 + int64_t(own_c ? (ctile0 + nt) * 16 + j : j) * (4 * C) + cc * HC + 4 * kg;
    };

    const float* wq_r[1];
    q_rows(0, wq_r);
    BPre<1> pre_q = prefetch_b<1, C / 16>(wq_r);
    for (int hh = 0; hh < heads; ++hh) {
        // [k_h | v_h] is 128 wide for every stage: wave w computes one of its 8 n-tiles (w < 4: k columns 16 w.., else v columns
        // 16 (w - 4)..) for all m-tiles
        const float* wkv1_r[1] = {p.wkv + int64_t((wave < 4 ? 0 : inner) + hh * 64 + wn * 16 + j) * C + 4 * kg};
        const BPre<1> pre_kv1 = prefetch_b<1, C / 16>(wkv1_r);
        {   // q_h : [R][64], this wave's 16 columns of its M half
            f32x4 aq[MT0][1];
#pragma unroll
            for (int mt = 0; mt < MT0; ++mt) aq[mt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            gemm_r(I1{}, KC{}, sy, RS, wq_r, pre_q, aq);
#pragma unroll
            for (int mt = 0; mt < MT0; ++mt)
                if (mt < mcount) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sq[((mbase + mt) * 16 + 4 * kg + r) * QS + wn * 16 + j] = aq[mt][0][r];
                }
        }
        const float* wo_r[NTC];
        o_rows(hh, wo_r);
        const BPre<NTC> pre_o = prefetch_b<NTC, 4>(wo_r);
        {   // k_h, v_h : [RKV][64] each, one n-tile of the pair per wave, all m-tiles
            f32x4 akv1[MTKV][1];
#pragma unroll
            for (int mt = 0; mt < MTKV; ++mt) akv1[mt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            gemm_lds<MTKV, 1, C / 16>(sykv, RS, wkv1_r, pre_kv1, akv1, j, kg);
            float* dst = wave < 4 ? sk : sv;
#pragma unroll
            for (int mt = 0; mt < MTKV; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(mt * 16 + 4 * kg + r) * QS + wn * 16 + j] = akv1[mt][0][r];
        }
        lds_barrier();
        stamp();
        // scores = q k^T / 8 per site (model.py:126; dim_head = 64)
        for (int t = tid; t < TS * W * WKV; t += NT) {
            const int s = t / (W * WKV), rem = t - s * (W * WKV), i = rem / WKV, jj = rem - i * WKV;
            const float4* qv = reinterpret_cast<const float4*>(sq + (s * W + i) * QS);
            const float4* kv = reinterpret_cast<const float4*>(sk + (s * WKV + jj) * QS);
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < 16; ++d) {
                const float4 a = qv[d], b = kv[d];
                acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
            }
            sp[t] = acc * 0.125f;
        }
        lds_barrier();
        for (int t = tid; t < TS * W; t += NT) {
            float* row = sp + t * WKV;
            float mx = row[0];
#pragma unroll
            for (int jj = 1; jj < WKV; ++jj) mx = fmaxf(mx, row[jj]);
            float e[WKV], sum = 0.f;
#pragma unroll
            for (int jj = 0; jj < WKV; ++jj) { e[jj] = expf(row[jj] - mx); sum += e[jj]; }
            const float inv = 1.0f / sum;
#pragma unroll
            for (int jj = 0; jj < WKV; ++jj) row[jj] = e[jj] * inv;
        }
        lds_barrier();
        // o_h = P v_h, overwrites q_h
        for (int t = tid; t < R * 16; t += NT) {
            const int row = t >> 4, d4 = (t & 15) * 4, s = row / W;
            const float* pr = sp + row * WKV;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int jj = 0; jj < WKV; ++jj) {
                const float pj = pr[jj];
                const float4 vv = *reinterpret_cast<const float4*>(sv + (s * WKV + jj) * QS + d4);
                o.x = fmaf(pj, vv.x, o.x); o.y = fmaf(pj, vv.y, o.y); o.z = fmaf(pj, vv.z, o.z); o.w = fmaf(pj, vv.w, o.w);
            }
            *reinterpret_cast<float4*>(sq + row * QS + d4) = o;
        }
        if (hh + 1 < heads) {        // next head's q weights fly under the barrier and the out-projection
            q_rows(hh + 1, wq_r);
            pre_q = prefetch_b<1, C / 16>(wq_r);
        }
        lds_barrier();
        stamp();
        if (own_c) gemm_c(INTC{}, K4{}, sq, QS, wo_r, pre_o, acc_o);   // acc_o += o_h Wo[:, hh*64 .. +64]^T
        lds_barrier();   // sq / sk / sv are rewritten by the next head
        stamp();
    }

    // first FFN weights are requested before the residual update and the second LayerNorm
    const float* w1_r[NTF];
    w1_rows(0, w1_r);
    BPre<NTF> pre_w1 = prefetch_b<NTF, C / 16>(w1_r);

    // ---- phase 4: h += to_out(o) + bias ----
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt) {
        if (!own_c) break;
        const int col = (ctile0 + nt) * 16 + j;
        const float bv = p.bo[col];
#pragma unroll
        for (int mt = 0; mt < MTC; ++mt)
            if (mt < mcountC) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sh[((mbaseC + mt) * 16 + 4 * kg + r) * RS + col] += acc_o[mt][nt][r] + bv;
            }
    }
    lds_barrier();

    stamp();
    // ---- phase 5 ----
    layer_norm(sh, sy, p.n1g, p.n1b);
    lds_barrier();
    stamp();

    // ---- phase 6: feed-forward, hidden units in chunks of HC ----
    f32x4 acc_f[MTC][NTC];
#pragma unroll
    for (int mt = 0; mt < MTC; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) acc_f[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int cc = 0; cc < 4 * C / HC; ++cc) {
        const float* w2_r[NTC];
        w2_rows(cc, w2_r);
        const BPre<NTC> pre_w2 = prefetch_b<NTC, HC / 16>(w2_r);
        {
            f32x4 au[MTF][NTF];
#pragma unroll
            for (int mt = 0; mt < MTF; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTF; ++nt) au[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            gemm_f(INTF{}, KC{}, sy, RS, w1_r, pre_w1, au);
            const int n0 = cc * HC + ftile0 * 16;
#pragma unroll
            for (int nt = 0; nt < NTF; ++nt) {
                const float bv = p.b1[n0 + nt * 16 + j];
#pragma unroll
                for (int mt = 0; mt < MTF; ++mt)
                    if (mt < mcountF) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            su[((mbaseF + mt) * 16 + 4 * kg + r) * US + (ftile0 + nt) * 16 + j] = gelu_f(au[mt][nt][r] + bv);
                    }
            }
        }
        if (cc + 1 < 4 * C / HC) {
            w1_rows(cc + 1, w1_r);
            pre_w1 = prefetch_b<NTF, C / 16>(w1_r);
        }
        lds_barrier();
        if (own_c) gemm_c(INTC{}, KH{}, su, US, w2_r, pre_w2, acc_f);
        lds_barrier();   // su is rewritten by the next chunk
        stamp();
    }

    // ---- phase 7: h += ff(y) + bias in LDS, then -> HBM (last block of the network: the classifier consumes it in LDS) ----
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt) {
        if (!own_c) break;
        const int col = (ctile0 + nt) * 16 + j;
        const float bv = p.b2[col];
#pragma unroll
        for (int mt = 0; mt < MTC; ++mt)
            if (mt < mcountC) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = (mbaseC + mt) * 16 + 4 * kg + r;
                    sh[row * RS + col] += acc_f[mt][nt][r] + bv;     // the tile leaves through LDS: 16-byte coalesced stores below
                }
            }
    }
    if constexpr (!HEAD) {
        lds_barrier();
        for (int i = tid; i < rows_valid * (C / 4); i += NT) {
            const int r = i / (C / 4), c4 = (i - r * (C / 4)) * 4;
            *reinterpret_cast<float4*>(hg + r * C + c4) = *reinterpret_cast<const float4*>(sh + r * RS + c4);
        }
    }
    if constexpr (HEAD) {
        // ---- phase 8: fc1 over the flattened [W][C] features of each site (model.py:239-247; torch's c*W + w order is folded
        // into w1p, whose k axis follows the LDS image w*RS + c with zero weights on the pad columns), SELU, classifier tail
        constexpr int K1 = G::KCH1;
        float* t1 = smem + G::OFF_Y + 64;       // the last site's k run ends up to 12 floats past its rows
        float* t2 = t1 + 16 * HEAD_T1S;
        const float* w1_r1[1] = {p.w1p + int64_t(wave * 16 + j) * (K1 * 16) + 4 * kg};
        BGroup<1, 7> g0;
        load_group<1, K1, 7>(g0, w1_r1, 0);
        lds_barrier();
        stamp();
        f32x4 a1[2][1] = {{f32x4{0.f, 0.f, 0.f, 0.f}}, {f32x4{0.f, 0.f, 0.f, 0.f}}};
        gemm_m1<1, K1, 7>(sh, W * RS, w1_r1, g0, a1, j, kg);
        const float bv = p.b1h[wave * 16 + j];
#pragma unroll
        for (int r = 0; r < 4; ++r) t1[(4 * kg + r) * HEAD_T1S + wave * 16 + j] = selu_fast(a1[0][0][r] + a1[1][0][r] + bv);
        lds_barrier();
        stamp();
        head_tail(t1, t2, hp, B, site0, nsite);
    }
    stamp();
}

}  // namespace cto
