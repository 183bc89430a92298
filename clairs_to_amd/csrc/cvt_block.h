// Fused CvT transformer block (clairs/model.py:134-147: x = attn(x) + x; x = ff(x) + x) for gfx950.
//
// One workgroup (8 waves, two per SIMD) owns TS sites = R = TS*W activation rows of C channels and keeps every
// intermediate (LayerNorm output, depth-wise conv outputs, per-head q/k/v, attention output, FFN hidden chunk) in LDS;
// the RESIDUAL STREAM lives in the accumulator registers of the two GEMMs that update it (out-projection, second FFN GEMM),
// only the weights stream in (from L2, straight into MFMA B registers) and only the updated residual stream goes back to HBM.
// The unfused path spends ~8 launches and ~10 HBM round trips per block on the same work.
//
//   phase 0  h tile -> sy                                   (CIN > 0, first block of a stage: the stage input tile instead,
//                                                            then the stride-2 conv embedding as an im2col GEMM + LayerNorm)
//   phase 1  acc_o = h + bias_o (accumulator layout);  t = LN(h; norm0) -> tmp      (model.py:57-76; 16 lanes per row)
//   phase 2  yq = BN(DW_s1(t)) -> sy, ykv = BN(DW_s2(t)) -> sykv                     (model.py:91-100, 112-113)
//   phase 3  per head: q_h, k_h, v_h (MFMA) -> LDS; softmax(q_h k_h^T / 8) v_h per query row in ONE pass (16 lanes per
//            row: partial dots, DPP reductions, softmax in registers, o_h over q_h in place); acc_o += o_h Wo_h^T (MFMA)
//   phase 4  h' = acc_o -> tmp;  acc_f = h' + bias_2
//   phase 5  y  = LN(h'; norm1) -> sy
//   phase 6  per 128-wide chunk of the 4C hidden units: u = GELU(y W1^T + b1) -> LDS, acc_f += u W2^T
//   phase 7  h'' = acc_f -> sy, then coalesced 16-byte stores to HBM
//   phase 8  (HEAD, last block of the network) fc1 over the LDS image of the tile -> SELU -> K x (fc2, fc3) instead of the store
//
// Up to CVT_MAX_BLK consecutive blocks of a stage run in ONE launch (CvtStageParams): between two of them the tile goes from the
// second FFN GEMM's accumulators to sy (for the next LayerNorm) and, still in registers, + bias_o into the next out-projection's
// accumulators - no store, no load, no launch ramp, and no CU waits for the slowest one at a kernel boundary.
//
// Tile ownership of an N-wide GEMM output (ColOwn): min(8, N/16) waves own distinct 16-column tiles and the remaining
// factor of the 8 waves splits the m-tiles, so that every wave owns MFMAs whatever N is (N = 16: eight m-tile groups).
#pragma once
#include "cvt_gemm.h"

namespace cto {

// LDS position stride of the stage input [site][2W][PS] for the in-block conv embedding: >= CIN, multiple of 4 (float4
// A fragments) and = 4 mod 16, so that the im2col row stride 2*PS is = 8 mod 32 banks (4-way instead of 16-way conflicts).
__host__ __device__ constexpr int emb_ps(int cin) { return cin > 0 ? ((cin - 4 + 15) / 16) * 16 + 4 : 0; }
__host__ __device__ constexpr int emb_kch(int cin) { return (3 * emb_ps(cin) + 15) / 16; }

template <int C, int W, int WKV, int TS>
struct CvtBlockGeom {
    static constexpr int R = TS * W, RKV = TS * WKV;
    static constexpr int MT = (R + 15) / 16, MTKV = (RKV + 15) / 16;
    static constexpr int RS = C + 4, QS = 68, HC = (4 * C < 128 ? 4 * C : 128), US = HC + 4;
    static constexpr int OFF_Y = 0;                            // h -> yq -> y (FFN input) -> h'' : [MT*16][RS]
    static constexpr int OFF_YKV = OFF_Y + MT * 16 * RS;       // [MTKV*16][RS]
    static constexpr int OFF_Q = OFF_YKV + MTKV * 16 * RS;     // q_h, then o_h : [MT*16][QS]
    static constexpr int OFF_K = OFF_Q + MT * 16 * QS;
    static constexpr int OFF_V = OFF_K + MTKV * 16 * QS;
    static constexpr int OFF_END = OFF_V + MTKV * 16 * QS;
    static constexpr int SCRATCH = OFF_END - OFF_YKV;          // ykv|q|k|v region, re-used for the FFN hidden chunk
    static constexpr int U_FLOATS = MT * 16 * US;
    static constexpr int TOTAL = OFF_END + (U_FLOATS > SCRATCH ? U_FLOATS - SCRATCH : 0);
    static constexpr size_t LDS_BYTES = size_t(TOTAL) * sizeof(float);
    static constexpr int TMP_FLOATS = OFF_END - OFF_Q;         // q|k|v region doubles as the [MT*16][RS] LayerNorm staging tile
    static constexpr int ALIAS = TOTAL - OFF_YKV;              // everything but sy is free in phases 0 and 8
    static constexpr int KCH1 = (W * RS + 15) / 16;            // fc1 over the LDS image of one site's [W][RS] rows
    static constexpr int emb_floats(int cin) { return (MT * 32 + 4) * emb_ps(cin); }
    static constexpr bool HEAD_OK = (TS == 16) && (64 + head_lds_floats(6) <= ALIAS);
};

// no pad rows in the q-path and kv-path tiles (rows of a split tile that nobody writes would be read as arbitrary 16-bit floats)
template <int C, int W, int WKV, int TS>
__host__ __device__ constexpr bool MT_EXACT() { return (TS * W) % 16 == 0 && (TS * WKV) % 16 == 0; }

// 8 waves per workgroup (two per SIMD), so that one wave's LDS / L2 waits and VALU epilogues overlap the other's MFMAs
constexpr int CVT_BLOCK_THREADS = 512;
constexpr int CVT_WAVES = CVT_BLOCK_THREADS / 64;

// Who computes what of an [MTx*16] x N GEMM output.
template <int N>
struct ColOwn {
    static constexpr int TILES = N / 16;
    static constexpr int NOWN = TILES < CVT_WAVES ? TILES : CVT_WAVES;   // waves owning distinct column tiles
    static constexpr int MSPLIT = CVT_WAVES / NOWN;                      // wave groups that split the m-tiles
    static constexpr int NTW = TILES / NOWN;                             // column tiles per wave
    static_assert(TILES * 16 == N && CVT_WAVES % NOWN == 0 && NTW * NOWN == TILES, "unsupported GEMM width");
};
template <int MTx, int MSPLIT>
struct MSplit {
    static constexpr int MTG = (MTx + MSPLIT - 1) / MSPLIT;    // m-tiles of a full group
    static constexpr int FULL = MTx / MTG;                     // groups that own MTG tiles
    static constexpr int LAST = MTx - FULL * MTG;              // tiles of the group after them (0: none); later groups own nothing
};
struct TileSpan { int tile0, mbase, mcount; };
template <int N, int MTx>
__device__ __forceinline__ TileSpan tile_span(int wave) {
    using O = ColOwn<N>;
    constexpr int MTG = MSplit<MTx, O::MSPLIT>::MTG;
    const int mb = (wave / O::NOWN) * MTG;
    const int left = MTx - mb;
    return TileSpan{(wave % O::NOWN) * O::NTW, mb, left < 0 ? 0 : (left > MTG ? MTG : left)};
}
// this wave's rows of an N-wide output: a group owns MTG m-tiles, one group may own fewer (LAST), the rest none
template <int N, int MTx, int KCH>
__device__ __forceinline__ void gemm_span(const TileSpan& sp, const float* A, int lda, const float* const (&wrow)[ColOwn<N>::NTW],
                                          const BPre<ColOwn<N>::NTW>& pre,
                                          f32x4 (&acc)[MSplit<MTx, ColOwn<N>::MSPLIT>::MTG][ColOwn<N>::NTW], int j, int kg) {
    using S = MSplit<MTx, ColOwn<N>::MSPLIT>;
    constexpr int NTW = ColOwn<N>::NTW;
    const float* A0 = A + sp.mbase * 16 * lda;
    if (ColOwn<N>::MSPLIT == 1 || sp.mcount == S::MTG) {
        gemm_lds<S::MTG, NTW, KCH>(A0, lda, wrow, pre, acc, j, kg);
    } else if constexpr (S::LAST > 0) {
        if (sp.mcount == S::LAST) {
            f32x4 (&sub)[S::LAST][NTW] = reinterpret_cast<f32x4 (&)[S::LAST][NTW]>(acc);
            gemm_lds<S::LAST, NTW, KCH>(A0, lda, wrow, pre, sub, j, kg);
        }
    }
}

template <int N, int MTx, int KCH, bool F16>
__device__ __forceinline__ void gemm_span_split(const TileSpan& sp, const float* A, int lda, int klo,
                                                const unsigned short* const (&wrow)[ColOwn<N>::NTW],
                                                const BPreS<ColOwn<N>::NTW>& pre,
                                                f32x4 (&acc)[MSplit<MTx, ColOwn<N>::MSPLIT>::MTG][ColOwn<N>::NTW], int j, int kg) {
    using S = MSplit<MTx, ColOwn<N>::MSPLIT>;
    constexpr int NTW = ColOwn<N>::NTW;
    const float* A0 = A + sp.mbase * 16 * lda;
    if (ColOwn<N>::MSPLIT == 1 || sp.mcount == S::MTG) {
        gemm_lds_split<S::MTG, NTW, KCH, F16>(A0, lda, klo, wrow, pre, acc, j, kg);
    } else if constexpr (S::LAST > 0) {
        if (sp.mcount == S::LAST) {
            f32x4 (&sub)[S::LAST][NTW] = reinterpret_cast<f32x4 (&)[S::LAST][NTW]>(acc);
            gemm_lds_split<S::LAST, NTW, KCH, F16>(A0, lda, klo, wrow, pre, sub, j, kg);
        }
    }
}

// Workgroups of this geometry that fit one CU's 160 KB of LDS (at most 3 are asked for): the register budget follows from
// it through __launch_bounds__ (w = minimum waves per SIMD = 2 per resident 512-thread workgroup), otherwise a kernel that
// fits the LDS twice still runs alone because it was given 132+ registers.
template <int C, int W, int WKV, int TS>
__host__ __device__ constexpr int cvt_blocks_per_cu() {
    return CvtBlockGeom<C, W, WKV, TS>::LDS_BYTES * 3 <= 160 * 1024 ? 3 : (CvtBlockGeom<C, W, WKV, TS>::LDS_BYTES * 2 <= 160 * 1024 ? 2 : 1);
}

// SPLIT (experiment, side channel: CTO_CVT_SPLIT=f16|bf16; 0 = the fp32 product kernel, 1 = f16, 2 = bf16): the five weight GEMMs
// of a block (q, k|v, out-projection, both FFN GEMMs) run on split 16-bit operands (split_mfma.h) - their LDS input tiles are
// written as [hi | lo] rows by the phase that produces them (depth-wise conv + BN, attention, second LayerNorm, GELU epilogue)
// at the fp32 row pitch, the weights come pre-split.  The residual stream, LayerNorm, softmax, the embedding and the classifier
// are unchanged fp32.
template <int C, int W, int WKV, int TS, int CIN, bool HEAD, int SPLIT = 0>
__global__ __launch_bounds__(CVT_BLOCK_THREADS, (2 * cvt_blocks_per_cu<C, W, WKV, TS>())) void k_cvt_block(float* __restrict__ h, CvtStageParams sp,
                                                                                                            HeadTailParams hp, int heads, int B) {
    using G = CvtBlockGeom<C, W, WKV, TS>;
    constexpr bool SP = SPLIT != 0, F16 = SPLIT == 1;
    static_assert(!SP || (C % 64 == 0 && MT_EXACT<C, W, WKV, TS>()), "split GEMMs: 32-wide k chunks, 16-byte LayerNorm pieces, no pad rows");
    using WPtr = std::conditional_t<SP, const unsigned short*, const float*>;
    static_assert(CIN == 0 || G::emb_floats(CIN) <= G::ALIAS, "stage input tile does not fit the free LDS");
    static_assert(!HEAD || G::HEAD_OK, "classifier tail needs a 16-site tile and room for its scratch");
    static_assert(G::MT * 16 * G::RS <= G::TMP_FLOATS, "LayerNorm staging tile does not fit the q|k|v region");
    constexpr int NT = CVT_BLOCK_THREADS, NWV = CVT_WAVES;
    constexpr int R = G::R, MT = G::MT, MTKV = G::MTKV, RKV = G::RKV, RS = G::RS, QS = G::QS, HC = G::HC, US = G::US;
    static_assert(C % 16 == 0 && C <= 128 && NT % C == 0, "channel count must be 16, 32, 64 or 128");
    using OC = ColOwn<C>;       // C-wide outputs: embedding, out-projection, second FFN GEMM (= the residual stream's owners)
    using OQ = ColOwn<64>;      // q of one head
    using OKV = ColOwn<128>;    // [k_h | v_h]
    using OF = ColOwn<HC>;      // one FFN hidden chunk
    constexpr int MGC = MSplit<MT, OC::MSPLIT>::MTG, MGQ = MSplit<MT, OQ::MSPLIT>::MTG, MGF = MSplit<MT, OF::MSPLIT>::MTG;
    constexpr int NTC = OC::NTW, NTF = OF::NTW;
    static_assert(OQ::NTW == 1 && OKV::NTW == 1 && OKV::MSPLIT == 1, "q / kv tiles: one column tile per wave");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sy = smem + G::OFF_Y;
    float* sykv = smem + G::OFF_YKV;
    float* sq = smem + G::OFF_Q;
    float* sk = smem + G::OFF_K;
    float* sv = smem + G::OFF_V;
    float* stmp = smem + G::OFF_Q;       // alias: LayerNorm staging tile [MT*16][RS] (phases 1-2 and 4-5)
    float* su = smem + G::OFF_YKV;       // alias: FFN hidden chunk [MT*16][US]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kg = lane >> 4;
    const TileSpan spc = tile_span<C, MT>(wave), spq = tile_span<64, MT>(wave), spf = tile_span<HC, MT>(wave);
    const int site0 = blockIdx.x * TS;
    const int nsite = min(TS, B - site0);
    const int rows_valid = nsite * W;
    const int inner = heads * 64;
    float* hg = h + int64_t(site0) * W * C;
    int nstamp = 0;
    long long* const prof = sp.blk[0].prof;
    auto stamp = [&]() { if (prof && blockIdx.x == 0 && tid == 0 && nstamp < 250) prof[nstamp++] = clock64(); };
    stamp();
    const CvtBlockParams& pe = sp.blk[0];       // the stage's embedding, when this launch starts the stage


    // the embedding GEMM's first weight chunks are requested before the stage input is: their L2 round trip runs under the HBM one
    constexpr int KE_ = CIN > 0 ? emb_kch(CIN) : 1;
    const float* we_r[NTC];
    BPre<NTC> pre_e;
    if constexpr (CIN > 0) {
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) we_r[nt] = pe.wembp + int64_t(spc.tile0 + nt) * (KE_ * FRAG_CS) + 4 * lane;
        pre_e = prefetch_b<NTC, KE_>(we_r);
    }
    // ---- phase 0: residual stream tile -> sy (pad rows zero); first block of a stage: stage input -> LDS instead ----
    if constexpr (HEAD) {   // fc1 sweeps the LDS image of the tile including the 4 pad columns of every row (zero weights): keep them finite
        for (int i = tid; i < MT * 16; i += NT) *reinterpret_cast<float4*>(sy + i * RS + C) = make_float4(0.f, 0.f, 0.f, 0.f);
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
            if (i < MT * 16 * (C / 4)) *reinterpret_cast<float4*>(sy + r * RS + c4) = stage[q];
        }
        for (int i = tid; i < (MTKV * 16 - RKV) * RS; i += NT) sykv[RKV * RS + i] = 0.f;
    } else {
        // x tile as [site][2W positions][PS]: slot 0 of a site is the conv's left zero pad, the next site's slot 0 doubles as
        // this site's right pad, so the im2col row of output (site, wo) is the contiguous run starting at row*2*PS
        constexpr int WIN = 2 * W - 1, PS = emb_ps(CIN), NPOS = MT * 32 + 4, VW = (CIN % 4 == 0) ? 4 : 2, SLOTS = PS / VW;
        static_assert(CIN % 2 == 0, "stage input channels must be even");
        float* sin = smem + G::OFF_YKV;
        const float* xg = pe.xin + int64_t(site0) * WIN * CIN;
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
                if (pe.xraw) {              // (workgroup-uniform) int16 tensor: the bits now, the conversion when all pieces are in flight
                    const short* src = pe.xraw + (int64_t(site0 + s) * WIN + pos) * CIN + c;
                    if constexpr (VW == 4) { const uint2 w = *reinterpret_cast<const uint2*>(src); v.x = __uint_as_float(w.x); v.y = __uint_as_float(w.y); }
                    else v.x = __uint_as_float(*reinterpret_cast<const unsigned*>(src));
                } else {
                    const float* src = xg + (s * WIN + pos) * CIN + c;
                    if constexpr (VW == 4) v = *reinterpret_cast<const float4*>(src);
                    else { const float2 w2 = *reinterpret_cast<const float2*>(src); v.x = w2.x; v.y = w2.y; }
                }
            }
            stage[q] = v;
        }
        if (pe.xraw) {
            // the coverage scale of a site, in double as clairs/predict.py:172-207 computes it: one division per site (lane l of every wave
            // holds site l's), not one per piece
            static_assert(TS <= 64, "one lane per site of the tile");
            double sc_lane = 1.0;
            if (lane < nsite) {
                const int depth = pe.xinfo[int64_t(site0 + lane) * 12 + 1 + pe.xwhich];
                if (pe.xcov > 0 && depth > pe.xcov) sc_lane = double(pe.xcov) / double(depth);
            }
            const int sc_lo = __double2loint(sc_lane), sc_hi = __double2hiint(sc_lane);
#pragma unroll
            for (int q = 0; q < NIT; ++q) {
                const int i = tid + q * NT;
                const int pp = i / SLOTS, sx = pp / (2 * W), pos = pp - sx * 2 * W - 1;
                const int from = sx < TS ? sx : 0;         // every lane takes part in the exchange: a lane that sits out may be another's source
                const double sc = __hiloint2double(__shfl(sc_hi, from), __shfl(sc_lo, from));
                if (i < NPOS * SLOTS && pos >= 0 && sx < nsite) {
                    const unsigned w0 = __float_as_uint(stage[q].x), w1 = __float_as_uint(stage[q].y);
                    stage[q].x = float(double(int(short(w0 & 0xffffu))) * sc);
                    stage[q].y = float(double(int(short(w0 >> 16))) * sc);
                    if constexpr (VW == 4) { stage[q].z = float(double(int(short(w1 & 0xffffu))) * sc); stage[q].w = float(double(int(short(w1 >> 16))) * sc); }
                }
            }
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

    // Channel LayerNorm src -> dst (may be the same tile).  16 lanes per row (4 rows per wave and pass), each lane owning C/16
    // contiguous channels: the row reductions are 4 DPP steps inside a 16-lane row.  All passes are in flight together - one
    // pass is a dependent chain of an LDS read, two 4-step reductions, a square root and a division, and with two waves per
    // SIMD nothing else hides it.
    auto layer_norm = [&](const float* src, float* dst, const float* g, const float* b, auto split_out) {
#if CTO_CVT_ABL == 2
        return;
#endif
        constexpr bool SPO = decltype(split_out)::value;      // dst feeds a split GEMM: rows of [hi | lo]
        constexpr int CPL = C / 16, NP = (MT * 16 + NWV * 4 - 1) / (NWV * 4);
        const int l16 = lane & 15, grp = lane >> 4;
        float gv[CPL], bv[CPL];
#pragma unroll
        for (int i = 0; i < CPL; ++i) { gv[i] = g[l16 * CPL + i]; bv[i] = b[l16 * CPL + i]; }
        float v[NP][CPL];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int r = min(q * NWV * 4 + wave * 4 + grp, MT * 16 - 1);
            const float* xr = src + r * RS + l16 * CPL;
            if constexpr (CPL % 4 == 0) {
#pragma unroll
                for (int i = 0; i < CPL; i += 4) {
                    const float4 t = *reinterpret_cast<const float4*>(xr + i);
                    v[q][i] = t.x; v[q][i + 1] = t.y; v[q][i + 2] = t.z; v[q][i + 3] = t.w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < CPL; ++i) v[q][i] = xr[i];
            }
        }
        float inv[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < CPL; ++i) sum += v[q][i];
            sum = row16_sum(sum);
            const float mean = sum / float(C);
            float sq2 = 0.f;
#pragma unroll
            for (int i = 0; i < CPL; ++i) { v[q][i] -= mean; sq2 += v[q][i] * v[q][i]; }
            sq2 = row16_sum(sq2);
            inv[q] = 1.0f / (sqrtf(sq2 / float(C)) + 1e-5f);
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int r = q * NWV * 4 + wave * 4 + grp;
            if (r < MT * 16) {
                float* yr = dst + r * RS + l16 * CPL;
                if constexpr (SPO) {
                    static_assert(!SPO || CPL % 4 == 0, "split rows are written four channels at a time");
#pragma unroll
                    for (int i = 0; i < CPL; i += 4)
                        put_split4<F16>(dst + r * RS, C, l16 * CPL + i, v[q][i] * inv[q] * gv[i] + bv[i], v[q][i + 1] * inv[q] * gv[i + 1] + bv[i + 1],
                                        v[q][i + 2] * inv[q] * gv[i + 2] + bv[i + 2], v[q][i + 3] * inv[q] * gv[i + 3] + bv[i + 3]);
                } else if constexpr (CPL % 4 == 0) {
#pragma unroll
                    for (int i = 0; i < CPL; i += 4)
                        *reinterpret_cast<float4*>(yr + i) =
                            make_float4(v[q][i] * inv[q] * gv[i] + bv[i], v[q][i + 1] * inv[q] * gv[i + 1] + bv[i + 1],
                                        v[q][i + 2] * inv[q] * gv[i + 2] + bv[i + 2], v[q][i + 3] * inv[q] * gv[i + 3] + bv[i + 3]);
                } else {
#pragma unroll
                    for (int i = 0; i < CPL; ++i) yr[i] = v[q][i] * inv[q] * gv[i] + bv[i];
                }
            }
        }
    };

    if constexpr (CIN > 0) {
        // conv embedding (model.py:195, only the middle kernel row is live) + bias, then the stage's channel LayerNorm
        constexpr int PS = emb_ps(CIN), KE = emb_kch(CIN);
        f32x4 acc_e[MGC][NTC];
#pragma unroll
        for (int mt = 0; mt < MGC; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTC; ++nt) acc_e[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        gemm_span<C, MT, KE>(spc, smem + G::OFF_YKV, 2 * PS, we_r, pre_e, acc_e, j, kg);
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) {
            const int col = (spc.tile0 + nt) * 16 + j;
            const float bv = pe.bemb[col];
#pragma unroll
            for (int mt = 0; mt < MGC; ++mt)
                if (mt < spc.mcount) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sy[((spc.mbase + mt) * 16 + 4 * kg + r) * RS + col] = acc_e[mt][nt][r] + bv;
                }
        }
        lds_barrier();
        for (int i = tid; i < (MTKV * 16 - RKV) * RS; i += NT) sykv[RKV * RS + i] = 0.f;    // the input tile lay over these rows
        layer_norm(sy, sy, pe.lng, pe.lnb, std::false_type{});
        lds_barrier();
    }

    f32x4 acc_f[MGC][NTC];       // the residual stream at the end of a block (second FFN GEMM's accumulators)
    for (int bi = 0; bi < sp.nblk; ++bi) {
    const CvtBlockParams& p = sp.blk[bi];
    stamp();
    // ---- phase 1: the residual stream moves into the out-projection's accumulators; LayerNorm -> staging tile ----
    f32x4 acc_o[MGC][NTC];
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt) {
        const int col = (spc.tile0 + nt) * 16 + j;
        const float bv = p.bo[col];
#pragma unroll
        for (int mt = 0; mt < MGC; ++mt) {
            const int row0 = (spc.mbase + (mt < spc.mcount ? mt : 0)) * 16 + 4 * kg;
            if (bi == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc_o[mt][nt][r] = sy[(row0 + r) * RS + col] + bv;
            } else {             // a later block of the launch: this wave still holds its part of the tile in registers
#pragma unroll
                for (int r = 0; r < 4; ++r) acc_o[mt][nt][r] = acc_f[mt][nt][r] + bv;
            }
        }
    }
    layer_norm(sy, stmp, p.n0g, p.n0b, std::false_type{});
    lds_barrier();
    stamp();

    // ---- phase 2: depth-wise 3-tap conv + BatchNorm of the staging tile; q path -> sy, kv path (stride 2) -> sykv ----
#if CTO_CVT_ABL != 2
    {
        constexpr int NSLOT = NT / C;                                   // (site, position group) slots of one pass
        constexpr int PG = NSLOT > TS ? NSLOT / TS : 1;                 // more slots than sites: split a site's positions
        constexpr int SL = PG > 1 ? TS : NSLOT, NIT = (TS + SL - 1) / SL;    // slots past SL * PG (TS does not divide NSLOT) idle
        constexpr int WQ = (W + PG - 1) / PG, WK = (WKV + PG - 1) / PG;
        const int c = tid % C, slot = tid / C, pg = slot / SL, s0 = slot - pg * SL;
        const int wlo = pg * WQ, whi = min(W, wlo + WQ), klo = pg * WK, khi = min(WKV, klo + WK);
        const float q0 = p.dwq[c * 3], q1 = p.dwq[c * 3 + 1], q2 = p.dwq[c * 3 + 2];
        const float k0 = p.dwkv[c * 3], k1 = p.dwkv[c * 3 + 1], k2 = p.dwkv[c * 3 + 2];
        const float qm = p.bnq[c], qi = p.bnq[C + c], qw = p.bnq[2 * C + c], qb = p.bnq[3 * C + c];
        const float km = p.bnkv[c], ki = p.bnkv[C + c], kw = p.bnkv[2 * C + c], kb = p.bnkv[3 * C + c];
        float y[NIT][W];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int s = min(s0 + it * SL, TS - 1);
#pragma unroll
            for (int w = 0; w < W; ++w) y[it][w] = stmp[(s * W + w) * RS + c];
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int s = s0 + it * SL;
            if (s < TS && pg < PG) {
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    if (PG > 1 && (w < wlo || w >= whi)) continue;
                    const float l = w > 0 ? y[it][w - 1] : 0.f, r = w + 1 < W ? y[it][w + 1] : 0.f;
                    const float d = q0 * l + q1 * y[it][w] + q2 * r;
                    if constexpr (SP) put_split1<F16>(sy + (s * W + w) * RS, C, c, (d - qm) * qi * qw + qb);
                    else sy[(s * W + w) * RS + c] = (d - qm) * qi * qw + qb;
                }
#pragma unroll
                for (int wo = 0; wo < WKV; ++wo) {
                    if (PG > 1 && (wo < klo || wo >= khi)) continue;
                    const int w = 2 * wo;
                    const float l = w > 0 ? y[it][w - 1] : 0.f, r = w + 1 < W ? y[it][w + 1] : 0.f;
                    const float d = k0 * l + k1 * y[it][w] + k2 * r;
                    if constexpr (SP) put_split1<F16>(sykv + (s * WKV + wo) * RS, C, c, (d - km) * ki * kw + kb);
                    else sykv[(s * WKV + wo) * RS + c] = (d - km) * ki * kw + kb;
                }
            }
        }
        // pad rows of the q-path tile (rows R .. MT*16) keep what phase 0 left there: zeros or the embedding of zeros, finite
    }
#endif
    lds_barrier();

    stamp();
    // ---- phase 3: attention, head by head; out-projection accumulates on top of the residual stream in registers ----
    // this lane's fragments (cvt_gemm.h: fragment order): n-tile `tile`, first chunk `c0` of a matrix with `kch` chunks per row
    auto frag = [&](const float* w, const unsigned short* ws, int tile, int kch, int c0) -> WPtr {
        if constexpr (SP) return ws + (int64_t(tile) * kch + c0) * SPLIT_CS + 8 * lane;
        else return w + (int64_t(tile) * kch + c0) * FRAG_CS + 4 * lane;
    };
    constexpr int KD = SP ? 32 : 16;            // k elements per chunk
    auto q_rows = [&](int hh, WPtr (&wr)[1]) { wr[0] = frag(p.wq, p.wq_s, hh * 4 + spq.tile0, C / KD, 0); };
    auto o_rows = [&](int hh, WPtr (&wr)[NTC]) {
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) wr[nt] = frag(p.wo, p.wo_s, spc.tile0 + nt, inner / KD, hh * (64 / KD));
    };
    auto w1_rows = [&](int cc, WPtr (&wr)[NTF]) {
#pragma unroll
        for (int nt = 0; nt < NTF; ++nt) wr[nt] = frag(p.w1, p.w1_s, cc * (HC / 16) + spf.tile0 + nt, C / KD, 0);
    };
    auto w2_rows = [&](int cc, WPtr (&wr)[NTC]) {
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) wr[nt] = frag(p.w2, p.w2_s, spc.tile0 + nt, 4 * C / KD, cc * (HC / KD));
    };
    WPtr wq_r[1];
    q_rows(0, wq_r);
    auto pre_q = prefetch_b<1, C / KD>(wq_r);
    for (int hh = 0; hh < heads; ++hh) {
        // [k_h | v_h] is 128 wide for every stage: wave w computes one of its 8 n-tiles (w < 4: k columns 16 w.., else v columns
        // 16 (w - 4)..) for all m-tiles
        const int wn = wave & 3;
        WPtr wkv1_r[1] = {frag(p.wkv, p.wkv_s, (wave < 4 ? 0 : inner / 16) + hh * 4 + wn, C / KD, 0)};
        const auto pre_kv1 = prefetch_b<1, C / KD>(wkv1_r);
        {   // q_h : [R][64], this wave's 16 columns of its m-tile group
            f32x4 aq[MGQ][1];
#pragma unroll
            for (int mt = 0; mt < MGQ; ++mt) aq[mt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (SP) gemm_span_split<64, MT, C / 32, F16>(spq, sy, RS, C, wq_r, pre_q, aq, j, kg);
            else gemm_span<64, MT, C / 16>(spq, sy, RS, wq_r, pre_q, aq, j, kg);
#pragma unroll
            for (int mt = 0; mt < MGQ; ++mt)
                if (mt < spq.mcount) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) sq[((spq.mbase + mt) * 16 + 4 * kg + r) * QS + spq.tile0 * 16 + j] = aq[mt][0][r];
                }
        }
        WPtr wo_r[NTC];
        o_rows(hh, wo_r);
        const auto pre_o = prefetch_b<NTC, 64 / KD>(wo_r);
        {   // k_h, v_h : [RKV][64] each, one n-tile of the pair per wave, all m-tiles
            f32x4 akv1[MTKV][1];
#pragma unroll
            for (int mt = 0; mt < MTKV; ++mt) akv1[mt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (SP) gemm_lds_split<MTKV, 1, C / 32, F16>(sykv, RS, C, wkv1_r, pre_kv1, akv1, j, kg);
            else gemm_lds<MTKV, 1, C / 16>(sykv, RS, wkv1_r, pre_kv1, akv1, j, kg);
            float* dst = wave < 4 ? sk : sv;
#pragma unroll
            for (int mt = 0; mt < MTKV; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(mt * 16 + 4 * kg + r) * QS + wn * 16 + j] = akv1[mt][0][r];
        }
        lds_barrier();
        stamp();
        // softmax(q k^T / 8) v of one query row per 16-lane group (model.py:126-131; dim_head = 64): lane l owns dimensions
        // 4l .. 4l+3 of q, of every k and v row of the site and of the output, which replaces q in place.  Scores are partial
        // dots reduced inside the 16-lane row; no score matrix in LDS, no barrier between scores, softmax and P v.
#if CTO_CVT_ABL != 2
        {
            constexpr int NPA = (R + NT / 16 - 1) / (NT / 16); constexpr int UF = WKV <= 5 ? 2 : 1;
            const int l4 = (lane & 15) * 4;
#pragma unroll UF
            for (int q = 0; q < NPA; ++q) {
                const int row = q * (NT / 16) + (tid >> 4);
                const int rr = row < R ? row : 0;
                const int s = rr / W;
                const float4 qv = *reinterpret_cast<const float4*>(sq + rr * QS + l4);
                float4 kv[WKV], vv[WKV];
#pragma unroll
                for (int jj = 0; jj < WKV; ++jj) {
                    kv[jj] = *reinterpret_cast<const float4*>(sk + (s * WKV + jj) * QS + l4);
                    vv[jj] = *reinterpret_cast<const float4*>(sv + (s * WKV + jj) * QS + l4);
                }
                float sc[WKV];
#pragma unroll
                for (int jj = 0; jj < WKV; ++jj) sc[jj] = fmaf(qv.w, kv[jj].w, fmaf(qv.z, kv[jj].z, fmaf(qv.y, kv[jj].y, qv.x * kv[jj].x)));
#pragma unroll
                for (int jj = 0; jj < WKV; ++jj) sc[jj] = row16_sum(sc[jj]);
                float mx = sc[0];
#pragma unroll
                for (int jj = 1; jj < WKV; ++jj) mx = fmaxf(mx, sc[jj]);
                float sum = 0.f;
#pragma unroll
                for (int jj = 0; jj < WKV; ++jj) { sc[jj] = exp_le0((sc[jj] - mx) * 0.125f); sum += sc[jj]; }
                float inv = rcp_fast(sum);
                inv = fmaf(fmaf(-sum, inv, 1.0f), inv, inv);      // one Newton step: 0.5 ulp
                float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int jj = 0; jj < WKV; ++jj) {
                    const float pj = sc[jj] * inv;
                    o4.x = fmaf(pj, vv[jj].x, o4.x); o4.y = fmaf(pj, vv[jj].y, o4.y); o4.z = fmaf(pj, vv[jj].z, o4.z); o4.w = fmaf(pj, vv[jj].w, o4.w);
                }
                if (row < R) {      // o_h over q_h in place (split: the row's first 256 bytes become [hi | lo])
                    if constexpr (SP) put_split4<F16>(sq + rr * QS, 64, l4, o4.x, o4.y, o4.z, o4.w);
                    else *reinterpret_cast<float4*>(sq + rr * QS + l4) = o4;
                }
            }
        }
#endif
        if (hh + 1 < heads) {        // next head's q weights fly under the barrier and the out-projection
            q_rows(hh + 1, wq_r);
            pre_q = prefetch_b<1, C / KD>(wq_r);
        }
        lds_barrier();
        stamp();
        if constexpr (SP) gemm_span_split<C, MT, 2, F16>(spc, sq, QS, 64, wo_r, pre_o, acc_o, j, kg);
        else gemm_span<C, MT, 4>(spc, sq, QS, wo_r, pre_o, acc_o, j, kg);   // acc_o += o_h Wo[:, hh*64 .. +64]^T
        lds_barrier();   // sq / sk / sv are rewritten by the next head
        stamp();
    }

    // first FFN weights are requested before the residual hand-over and the second LayerNorm
    WPtr w1_r[NTF];
    w1_rows(0, w1_r);
    auto pre_w1 = prefetch_b<NTF, C / KD>(w1_r);

    // ---- phase 4: h' = h + to_out(o) + bias sits in acc_o: a copy -> staging tile for the LayerNorm, and h' + b2 seeds the
    //      second FFN GEMM's accumulators ----
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt) {
        const int col = (spc.tile0 + nt) * 16 + j;
        const float bv = p.b2[col];
#pragma unroll
        for (int mt = 0; mt < MGC; ++mt) {
            if (mt < spc.mcount) {
#pragma unroll
                for (int r = 0; r < 4; ++r) stmp[((spc.mbase + mt) * 16 + 4 * kg + r) * RS + col] = acc_o[mt][nt][r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) acc_f[mt][nt][r] = acc_o[mt][nt][r] + bv;
        }
    }
    lds_barrier();

    stamp();
    // ---- phase 5 ----
    layer_norm(stmp, sy, p.n1g, p.n1b, std::bool_constant<SP>{});
    lds_barrier();
    stamp();

    // ---- phase 6: feed-forward, hidden units in chunks of HC ----
    for (int cc = 0; cc < 4 * C / HC; ++cc) {
        WPtr w2_r[NTC];
        w2_rows(cc, w2_r);
        const auto pre_w2 = prefetch_b<NTC, HC / KD>(w2_r);
        {
            f32x4 au[MGF][NTF];
#pragma unroll
            for (int mt = 0; mt < MGF; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTF; ++nt) au[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (SP) gemm_span_split<HC, MT, C / 32, F16>(spf, sy, RS, C, w1_r, pre_w1, au, j, kg);
            else gemm_span<HC, MT, C / 16>(spf, sy, RS, w1_r, pre_w1, au, j, kg);
            const int n0 = cc * HC + spf.tile0 * 16;
#pragma unroll
            for (int nt = 0; nt < NTF; ++nt) {
                const float bv = p.b1[n0 + nt * 16 + j];
#pragma unroll
                for (int mt = 0; mt < MGF; ++mt)
                    if (mt < spf.mcount) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float* urow = su + ((spf.mbase + mt) * 16 + 4 * kg + r) * US;
                            if constexpr (SP) put_split1<F16>(urow, HC, (spf.tile0 + nt) * 16 + j, gelu_f(au[mt][nt][r] + bv));
#if CTO_CVT_ABL == 2
                            else urow[(spf.tile0 + nt) * 16 + j] = au[mt][nt][r] + bv;
#else
                            else urow[(spf.tile0 + nt) * 16 + j] = gelu_f(au[mt][nt][r] + bv);
#endif
                        }
                    }
            }
        }
        if (cc + 1 < 4 * C / HC) {
            w1_rows(cc + 1, w1_r);
            pre_w1 = prefetch_b<NTF, C / KD>(w1_r);
        }
        lds_barrier();
        if constexpr (SP) gemm_span_split<C, MT, HC / 32, F16>(spc, su, US, HC, w2_r, pre_w2, acc_f, j, kg);
        else gemm_span<C, MT, HC / 16>(spc, su, US, w2_r, pre_w2, acc_f, j, kg);
        if (cc + 1 < 4 * C / HC) lds_barrier();   // su is rewritten by the next chunk
        stamp();
    }

    // ---- phase 7: h'' = h' + ff(y) + bias sits in acc_f -> sy (its last reader was the first GEMM of the last chunk, a barrier
    //      ago), then -> HBM (last block of the network: the classifier consumes it in LDS) ----
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt) {
        const int col = (spc.tile0 + nt) * 16 + j;
#pragma unroll
        for (int mt = 0; mt < MGC; ++mt)
            if (mt < spc.mcount) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sy[((spc.mbase + mt) * 16 + 4 * kg + r) * RS + col] = acc_f[mt][nt][r];
            }
    }
    if (bi + 1 < sp.nblk) {
        lds_barrier();       // the next block's LayerNorm reads whole rows of sy; every wave is past its last read of the hidden chunk
        if constexpr (MTKV * 16 > RKV)
            for (int i = tid; i < (MTKV * 16 - RKV) * RS; i += NT) sykv[RKV * RS + i] = 0.f;    // the hidden chunks lay over these pad rows
    }
    }   // blocks of this launch
    const CvtBlockParams& p = sp.blk[sp.nblk - 1];
    if constexpr (!HEAD) {
        lds_barrier();
        for (int i = tid; i < rows_valid * (C / 4); i += NT) {
            const int r = i / (C / 4), c4 = (i - r * (C / 4)) * 4;
            *reinterpret_cast<float4*>(hg + r * C + c4) = *reinterpret_cast<const float4*>(sy + r * RS + c4);
        }
    }
    if constexpr (HEAD) {
        // ---- phase 8: fc1 over the flattened [W][C] features of each site (model.py:239-247; torch's c*W + w order is folded
        // into w1p, whose k axis follows the LDS image w*RS + c with zero weights on the pad columns), SELU, classifier tail.
        // The scratch starts 64 floats into the free region: the last site's k run ends up to 12 floats past its rows, on
        // floats of the last FFN hidden chunk (finite, zero weights).  The other waves may still read that chunk: barrier first.
        constexpr int K1 = G::KCH1;
        float* t1 = smem + G::OFF_YKV + 64;
        float* t2 = t1 + 16 * HEAD_T1S;
        const float* w1_r1[1] = {p.w1p + int64_t(wave) * (K1 * 256) + 4 * lane};       // this wave's n-tile, fragment order
        BGroup<1, 7> g0;
        load_group<1, K1, 7, 256>(g0, w1_r1, 0);
        lds_barrier();
        stamp();
        f32x4 a1[2][1] = {{f32x4{0.f, 0.f, 0.f, 0.f}}, {f32x4{0.f, 0.f, 0.f, 0.f}}};
        HeadPre hpre;
        gemm_m1<1, K1, 7, 256>(sy, W * RS, w1_r1, g0, a1, j, kg, [&] { head_prefetch(hpre, hp); });
        const float bv = p.b1h[wave * 16 + j];
#pragma unroll
        for (int r = 0; r < 4; ++r) t1[(4 * kg + r) * HEAD_T1S + wave * 16 + j] = selu_fast(a1[0][0][r] + a1[1][0][r] + bv);
        lds_barrier();
        stamp();
        head_tail(t1, t2, hp, B, site0, nsite, hpre);
    }
    stamp();
}

}  // namespace cto
