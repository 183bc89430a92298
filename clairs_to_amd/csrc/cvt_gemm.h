// LDS-resident GEMM building blocks of the fused CvT block kernel (cvt_block.h) and the classifier tail shared by both networks.
//
// GEMMs: fp32 MFMA 16x16x4; A fragments from LDS (ds_read_b128 = four k-steps), B fragments from global
// (one 16-byte load per lane per n-tile per 16-wide k chunk, prefetched two chunks ahead).
// Weights are stored in FRAGMENT ORDER (models.hip: pack_fragments): [n-tile][16-wide k chunk][lane = (kg << 4) | j][4] with lane
// (j, kg) holding W[tile * 16 + j][16 c + 4 kg .. + 3], so a wave's request is one contiguous 1 KB = eight whole cache lines; a
// row-major W[n][k] makes it sixteen half lines whose other halves the next chunk fetches again (they have left the 32 KB L1 by
// then).  wrow[nt] = panel + (tile * chunks_per_row + first chunk) * 256 + 4 * lane; consecutive chunks are FRAG_CS floats apart.
constexpr int FRAG_CS = 256;
#pragma once
// CTO_CVT_ABL (tools/ builds only, wrong results by design - DESIGN.md 7.1): 1 = every GEMM of the fused CvT kernel returns at once
// (what is left is LayerNorm, depth-wise conv, softmax, epilogues, barriers); 2 = those phases do nothing (what is left is the GEMMs, their
// operand traffic, their epilogue stores and the barriers).  Timing the two beside the product kernel bounds what ANY schedule that overlaps the
// two kinds of work inside a workgroup can reach.
#ifndef CTO_CVT_ABL
#define CTO_CVT_ABL 0
#endif
#include <type_traits>
#include "nn_kernels.h"
#include "split_mfma.h"

namespace cto {

struct CvtBlockParams {
    const float *n0g, *n0b, *dwq, *bnq, *wq, *dwkv, *bnkv, *wkv, *wo, *bo, *n1g, *n1b, *w1, *b1, *w2, *b2;
    long long* prof;   // debug: phase time stamps (s_memtime) of workgroup 0 / thread 0 when non-null (CTO_BLOCK_PROF=1)
    // first block of a stage (CIN > 0): the stage's conv embedding + LayerNorm run here instead of reading h
    const float *xin, *wembp, *bemb, *lng, *lnb;   // x [B][2W-1][CIN]; wembp [C][KCHE*16] (positions padded to PS)
    // last block of the network (HEAD): fc1 + classifier tail run here instead of writing h
    const float *w1p, *b1h;                        // fc1 over the LDS image of h (rows padded to RS), fragment order [8][KCH1][64][4]
    // split-operand experiment (CTO_CVT_SPLIT): the five GEMM weights as [hi plane | lo plane] of 16-bit values, row-major [N][K]
    const unsigned short *wq_s, *wkv_s, *wo_s, *w1_s, *w2_s;
    // stage input as the un-rescaled int16 tensor (xraw != null replaces xin): the rescale of predict.py:172-207 happens at the load,
    // float(double(v) * min_rescale_cov / depth) with the site's depth = xinfo[site][1 + xwhich], exactly the tensor kernel's expression
    const short* xraw;
    const int* xinfo;
    int xwhich, xcov;
};

// Consecutive blocks of ONE stage run in one launch (the tile never leaves the CU between them): blk[0] may carry the stage's
// embedding, blk[nblk - 1] the classifier.
constexpr int CVT_MAX_BLK = 4;
struct CvtStageParams {
    CvtBlockParams blk[CVT_MAX_BLK];
    int nblk;
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
        p.b1[nt] = KCH > 1 ? ldg4(wrow[nt] + FRAG_CS) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return p;
}

// acc[mt][nt] += A[mt*16 .. +16][0 .. KCH*16) * W[n-tile rows][same k];  wrow[nt] points at this lane's fragment of the n-tile's
// first chunk.  A rows are `lda` floats apart in LDS.  `pre` holds chunks 0 and 1.
template <int MT, int NTW, int KCH>
__device__ __forceinline__ void gemm_lds(const float* __restrict__ A, int lda, const float* const (&wrow)[NTW],
                                         const BPre<NTW>& pre, f32x4 (&acc)[MT][NTW], int j, int kg) {
#if CTO_CVT_ABL == 1
    return;
#endif
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
            for (int nt = 0; nt < NTW; ++nt) Bq[(c + 2) % 3][nt] = ldg4(wrow[nt] + (c + 2) * FRAG_CS);
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

// ---- split-operand forms of the two building blocks above (experiment, side channel: CTO_CVT_SPLIT) ----
// A tile rows keep their fp32 pitch `lda` and hold [hi: klo 16-bit][lo: klo 16-bit] (split_mfma.h: put_split*); weights are in
// fragment order too - [n-tile][32-wide k chunk][hi, lo][lane][8 x 16-bit], lane (j, kg) holding W[tile * 16 + j][32 c + 8 kg .. + 7]
// (models.hip: upload_split_fragments) - wrow[nt] points at the lane's hi fragment of the first chunk, the lo fragment is
// SPLIT_LO elements further, the next chunk SPLIT_CS.  KCH counts 32-wide k chunks.  Same output layout as gemm_lds.
constexpr int SPLIT_LO = 512, SPLIT_CS = 1024;
template <int NTW>
struct BPreS {
    uint4 b[2][NTW][2];      // chunks 0 and 1; hi, lo
};
template <int NTW, int KCH>
__device__ __forceinline__ BPreS<NTW> prefetch_b(const unsigned short* const (&wrow)[NTW]) {
    BPreS<NTW> p;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (c < KCH) { p.b[c][nt][0] = ldg16(wrow[nt] + c * SPLIT_CS); p.b[c][nt][1] = ldg16(wrow[nt] + SPLIT_LO + c * SPLIT_CS); }
            else { p.b[c][nt][0] = make_uint4(0u, 0u, 0u, 0u); p.b[c][nt][1] = make_uint4(0u, 0u, 0u, 0u); }
        }
    return p;
}
template <int MT, int NTW, int KCH, bool F16>
__device__ __forceinline__ void gemm_lds_split(const float* __restrict__ A, int lda, int klo, const unsigned short* const (&wrow)[NTW],
                                               const BPreS<NTW>& pre, f32x4 (&acc)[MT][NTW], int j, int kg) {
    uint4 Bq[3][NTW][2];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int p = 0; p < 2; ++p) { Bq[0][nt][p] = pre.b[0][nt][p]; Bq[1][nt][p] = pre.b[1][nt][p]; }
    // A fragments one chunk ahead while they fit (MT <= 5: 80 registers); taller tiles read each chunk's fragments right before its
    // MFMAs - the SIMD's other wave covers the LDS latency, 144 registers of fragments would spill
    constexpr bool ADB = MT <= 5;
    uint4 a[ADB ? 2 : 1][MT][2];
    const unsigned short* Arow = reinterpret_cast<const unsigned short*>(A + j * lda) + 8 * kg;
    auto load_a = [&](int buf, int c) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                a[buf][mt][p] = *reinterpret_cast<const uint4*>(Arow + mt * 16 * lda * 2 + p * klo + c * 32);
    };
    if (ADB) load_a(0, 0);
#pragma unroll
    for (int c = 0; c < KCH; ++c) {
        if (c + 2 < KCH) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                Bq[(c + 2) % 3][nt][0] = ldg16(wrow[nt] + (c + 2) * SPLIT_CS);
                Bq[(c + 2) % 3][nt][1] = ldg16(wrow[nt] + SPLIT_LO + (c + 2) * SPLIT_CS);
            }
        }
        if (ADB) { if (c + 1 < KCH) load_a((c + 1) & 1, c + 1); }
        else load_a(0, c);
        __builtin_amdgcn_sched_barrier(0);      // requests first, then this chunk's MFMAs (see gemm_lds)
        // pass outermost: the MT * NTW accumulators between two passes over the same one keep dependent MFMAs apart
#pragma unroll
        for (int pass = 0; pass < 3; ++pass) {
            const int pa = pass == 1 ? 1 : 0, pw = pass == 2 ? 1 : 0;       // (activation part, weight part)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = mfma_split<F16>(a[ADB ? (c & 1) : 0][mt][pa], Bq[c % 3][nt][pw], acc[mt][nt]);
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
// CS = floats between a lane's fragments of consecutive k chunks: 16 for a row-major W[n][k] (wrow = &W[n0 + j][4 kg]); 256 for a
// panel stored in FRAGMENT ORDER [n-tile][chunk][lane][4] (wrow = panel + tile * KCH * 256 + 4 lane), where a wave's request is one
// contiguous 1 KB = eight whole cache lines.  Row-major requests touch sixteen half lines whose other halves the next chunk asks
// for again after they left the 32 KB L1: a panel that is used once (the classifier's: M = 16 rows per workgroup) then crosses
// the L2 -> CU path twice.
template <int NTW, int KCH, int DEPTH, int CS = 16>
__device__ __forceinline__ void load_group(BGroup<NTW, DEPTH>& g, const float* const (&wrow)[NTW], int c0) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
            if (c0 + d < KCH) g.b[d][nt] = ldg4(wrow[nt] + (c0 + d) * CS);
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
// `under_last` runs right after the last group's loads have been requested: whatever it requests (the classifier tail's fc2
// weights) queues behind this GEMM's own stream and arrives while its last MFMA groups run.
template <int NTW, int KCH, int DEPTH, int CS, typename F>
__device__ __forceinline__ void gemm_m1(const float* __restrict__ A, int lda, const float* const (&wrow)[NTW],
                                        const BGroup<NTW, DEPTH>& first, f32x4 (&acc)[2][NTW], int j, int kg, F&& under_last) {
    constexpr int NG = (KCH + DEPTH - 1) / DEPTH;
    const float* Arow = A + j * lda + 4 * kg;
    // groups are requested TWO ahead of their MFMAs (every workgroup of the launch streams the same panel at the same time:
    // an L2 round trip under that load is longer than one group's 28 MFMAs)
    BGroup<NTW, DEPTH> g[3];
    g[0] = first;
    if (NG > 1) load_group<NTW, KCH, DEPTH, CS>(g[1], wrow, DEPTH);
    if (NG <= 2) under_last();
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        if (i + 2 < NG) {
            load_group<NTW, KCH, DEPTH, CS>(g[(i + 2) % 3], wrow, (i + 2) * DEPTH);
            if (i + 3 == NG) under_last();
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the requests ahead of this group's MFMAs
        mfma_group<NTW, KCH, DEPTH>(g[i % 3], Arow, i * DEPTH, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int NTW, int KCH, int DEPTH>
__device__ __forceinline__ void gemm_m1(const float* __restrict__ A, int lda, const float* const (&wrow)[NTW],
                                        const BGroup<NTW, DEPTH>& first, f32x4 (&acc)[2][NTW], int j, int kg) {
    gemm_m1<NTW, KCH, DEPTH, 16>(A, lda, wrow, first, acc, j, kg, [] {});
}

// ---- classifier tail shared by both networks (clairs/model.py:245-261, 451-467): K heads of fc2 (128 -> 128) -> SELU ->
// fc3 (128 -> 2) -> SELU on a 16-site tile whose SELU(fc1) activations sit in LDS.  8 waves; wave w owns hidden units
// [16w, 16w+16) of every head, two heads per pass (two independent accumulators keep the matrix pipe at issue rate).
struct HeadTailParams {
    const float *w2, *b2;   // fc2 in fragment order [K heads][8 n-tiles][8 chunks][64 lanes][4] (pack_fragments), [K*128]
    const float *w3, *b3;   // [K][2][128], [K][2]
    float* logits;          // [K][B][2]
    int K;
};
constexpr int HEAD_T1S = 132;                       // LDS row stride of the fc1 activations [16][128]
__host__ __device__ constexpr int head_t2s(int K) { return K * 128 + 4; }
__host__ __device__ constexpr int head_lds_floats(int K) { return 16 * HEAD_T1S + 16 * head_t2s(K); }

// The fc2 weights of the first two head pairs, requested ahead of the tail (they depend on nothing the tail computes: a caller
// asks for them under whatever precedes it - fc1's last MFMA groups, the slab sums)
struct HeadPre {
    BGroup<2, 8> g[2];
};
__device__ __forceinline__ void head_prefetch(HeadPre& hpre, const HeadTailParams& hp) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* w0 = hp.w2 + int64_t(wave) * (8 * 256) + 4 * lane;      // this wave's n-tile of head 0
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
        const float* wr[2] = {w0 + pi * 2 * 128 * 128, w0 + (pi * 2 + 1) * 128 * 128};
        load_group<2, 8, 8, 256>(hpre.g[pi], wr, 0);
    }
}

// t1: [16][HEAD_T1S] (in), t2: [16][head_t2s(K)] scratch.  All 512 threads call; ends without a barrier.
template <int K>
__device__ __forceinline__ void head_tail_k(const float* t1, float* t2, const HeadTailParams& hp, int64_t B, int64_t site0,
                                            int nsite, HeadPre& hpre) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kg = lane >> 4;
    constexpr int T2S = head_t2s(K);
    static_assert(K % 2 == 0 && K >= 4, "heads are processed in pairs; two pairs arrive prefetched");
    const float* wr[2];
    wr[0] = hp.w2 + int64_t(wave) * (8 * 256) + 4 * lane;
    wr[1] = wr[0] + 128 * 128;
    const float* Arow = t1 + j * HEAD_T1S + 4 * kg;
#pragma unroll
    for (int pi = 0; pi < K / 2; ++pi) {
        f32x4 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q = 0; q < 2; ++q) acc[a][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        mfma_group<2, 8, 8>(hpre.g[pi & 1], Arow, 0, acc);
        __builtin_amdgcn_sched_barrier(0);
        if (pi + 2 < K / 2) {        // a third pair (K = 6): its weights take the slot this pair just freed
            const float* wn[2] = {wr[0] + (pi + 2) * 2 * 128 * 128, wr[1] + (pi + 2) * 2 * 128 * 128};
            load_group<2, 8, 8, 256>(hpre.g[pi & 1], wn, 0);
        }
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
                                          int nsite, HeadPre& hpre) {
    if (hp.K == 4) head_tail_k<4>(t1, t2, hp, B, site0, nsite, hpre);
    else head_tail_k<6>(t1, t2, hp, B, site0, nsite, hpre);
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
    HeadPre hpre;
    head_prefetch(hpre, hp);       // the fc2 weights fly under the slab sums
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
    lds_barrier();                 // LDS hand-over only: the prefetched weights stay in flight
    head_tail(t1, t2, hp, B, site0, nsite, hpre);
}


}  // namespace cto
