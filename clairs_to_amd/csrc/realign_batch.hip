// Illumina read realignment, every window of a run at once (SURVEY.md 8f #4b; the `realign_reads` leg of BASELINE configs[3]).
//
// The reference hands its native realigner ONE window per call (src/realign_reads.py:582-595 -> realign_reads(...),
// src/realign/realigner.cpp:782-857) from one Python process per low-QUAL call.  Three of that call's stages are data-parallel over
// (haplotype, read) pairs and carry nearly all of its arithmetic; cto_realign_windows runs them for ALL windows handed over at once:
//   k_fast_pass   realigner.cpp:129-229 (FastPassAligner): a read is placed on a haplotype where one of its 32-mers matches
//                 exactly and the whole read has <= 2 mismatches.  The reference walks a hash index of the reads' k-mers; here
//                 one workgroup per (window, haplotype) tries the diagonals of every read against the haplotype held in LDS (a
//                 diagonal is walked when one 16-aligned block of the read matches on it: a run of 32 contains one) and reproduces
//                 the order-dependent parts of the original (which start a read keeps on a score tie, when a haplotype position
//                 counts as covered) from the time (haplotype position, read offset) each candidate would have been visited first.
//   k_sw*<byte>,  ssw.c:118-529 (sw_sse2_byte / sw_sse2_word) as ssw_align runs them (:781-830): forward pass, word-mode rerun
//   k_sw*<word>   on overflow, backward pass.  One DPP row is one SSE2 register - 16 lanes in the 8-bit kernel (four alignments per
//                 wavefront), 8 lanes in the 16-bit kernel (eight per wavefront): lane l holds the stripe positions q = l * seg + j
//                 of the query exactly as the striped layout of Farrar's kernel does, the byte shift _mm_slli_si128 is row_shr:1.
//                 The lazy-F loops are computed as what they amount to (a scan over the lanes, one correction applied where the
//                 next column loads H: row_pass) - their corrections are not fed back into E, and output CIGARs depend on that
//                 (csrc/realign.cpp header), which the closed form keeps.  Stripes of up to 128 positions keep their H / E columns
//                 in registers and the score profile in LDS (k_sw_regs / row_pass_regs); longer ones all three in LDS (k_sw).
//   k_banded      ssw.c:531-741 (banded_sw): the traceback between the end points, one wavefront per alignment, for the pairs the
//                 windows predict they will need (every haplotype against the reference, the pair each unplaced read picks).
// What is left - haplotype order, the picks, CIGAR composition - is strings and stays on the host (csrc/realign.cpp:
// Window::finish, which also runs any traceback it finds missing), so device results and host results meet in the same code and
// are held byte-equal by tests/test_gpu_realign.py against oracle/_ref (the reference's own realigner.cpp + SSW).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <pthread.h>
#include <condition_variable>
#include <cstring>
#include <exception>
#include <functional>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include "common.h"
#include "realign_internal.h"

using cto_realign::Ends;
using cto_realign::Window;

namespace {

constexpr int kKmer = 32, kMaxMism = 2;
constexpr int FP_LMAX = 2048;       // haplotype / reference bytes a window may have on the device path
constexpr int FP_RMAX = 512;        // read bytes
constexpr int FP_NT = 256;

template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
// A "row" is one SSE2 register of the reference's kernels: LW = 16 lanes (sw_sse2_byte) or 8 lanes (sw_sse2_word) of a wavefront.
template <int LW>
__device__ __forceinline__ int row_max(int v) {
    v = max(v, dpp_i<0xB1>(v));                      // quad_perm [1,0,3,2]
    v = max(v, dpp_i<0x4E>(v));                      // quad_perm [2,3,0,1]
    v = max(v, dpp_i<0x141>(v));                     // row_half_mirror: the other quad of the 8 lanes
    if (LW == 16) v = max(v, dpp_i<0x140>(v));       // row_mirror: the other 8 lanes
    return v;
}
template <int LW>
__device__ __forceinline__ int row_min(int v) { return -row_max<LW>(-v); }
// _mm_slli_si128(x, one element): lane l takes lane l - 1 of its row, lane 0 takes 0 (row_shr:1 works on 16 lanes: the first lane
// of an 8-lane row in the upper half must not see its neighbour row's last lane)
template <int LW>
__device__ __forceinline__ int row_shl1(int v, int l) {
    v = dpp_i<0x111>(v);
    if (LW == 8 && l == 0) v = 0;
    return v;
}

// ------------------------------------------------------------------------------------------------------------------------
// Fast pass.  One workgroup per haplotype.  Notation of realigner.cpp:147-229: i = haplotype position of a k-mer, o = its offset in
// the read, start = max(0, i - o).  A diagonal d = i - o of a (haplotype, read) pair is walked once: runs of >= 32 equal bytes are
// the k-mer hits on it (exact string equality, as the hash lookup), the N-tolerant mismatch count of the whole read on it is
// FastAlignStrings (:231-251).  Diagonals d < 0 all mean start 0 (the reference clips; the comparison then runs on diagonal 0).
//   hit of read r      = the accepted start (whole read inside the haplotype, <= 2 mismatches) of largest score; on a tie the one the
//                        reference visits first, i.e. smallest (i, o) over its k-mer hits (reads are visited in order, offsets ascending,
//                        and only a strictly larger score replaces a hit)
//   coverage[i] > 0 at the time position i is tested  <=>  some accepted (read, start) with start <= i < start + span was first
//                        visited at a position <= i
//   the test itself runs only at positions whose k-mer some read holds (`continue` at :162-165), inside [prefix, L - suffix) and
//   not for the reference haplotype
struct FpArgs {
    const unsigned char* hap_bytes; const int* hap_off; const int* hap_win; const unsigned char* hap_isref; const long long* hit_off;
    const unsigned char* read_bytes; const int* read_off; const int* win_read0; const int* win_prefix; const int* win_suffix;
    int* hit_score; int* hit_pos; int* hap_score;
};

__global__ __launch_bounds__(FP_NT) void k_fast_pass(FpArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char s_hap[FP_LMAX + 16], s_read[FP_RMAX];      // (+16: the block test reads whole dwords)
    __shared__ unsigned char s_seed[FP_LMAX];
    __shared__ int s_cov[FP_LMAX];
    __shared__ unsigned long long s_best;
    __shared__ unsigned s_neg, s_d0_visit;
    __shared__ int s_d0_mism, s_dropped;
    // diagonals with a k-mer hit in sight are not walked by the thread that found them - one thread walking 150 positions and marking
    // 150 covered positions while 255 wait - but listed, and every listed diagonal is walked by a wavefront
    constexpr int FP_CAND = 64;
    __shared__ int s_cand[FP_CAND], s_ncand;
    const int h = blockIdx.x, tid = threadIdx.x;
    const int h0 = a.hap_off[h], L = a.hap_off[h + 1] - h0, w = a.hap_win[h];
    const int r0 = a.win_read0[w], n = a.win_read0[w + 1] - r0;
    const long long hb = a.hit_off[h];
    for (int i = tid; i < L; i += FP_NT) { s_hap[i] = a.hap_bytes[h0 + i]; s_seed[i] = 0; s_cov[i] = 0x7fffffff; }
    int score_sum = 0;                    // thread 0's copy is the one used
    // an accepted candidate: remember the best, mark what it covers and since when
    auto emit = [&](int start, int span, int mism, int t, int o) {
        const int sc = (span - mism) * 4 - mism * 6;
        const unsigned long long key = (static_cast<unsigned long long>(sc) << 32) | (static_cast<unsigned long long>(0xffff - t) << 16) |
                                       static_cast<unsigned long long>(0xffff - o);
        atomicMax(&s_best, key);
        for (int p = start; p < start + span; ++p) atomicMin(&s_cov[p], t);
    };
    auto emit_all = [&](int start, int span, int mism, int t, int o) {            // the same, called by every thread of the workgroup
        const int sc = (span - mism) * 4 - mism * 6;
        const unsigned long long key = (static_cast<unsigned long long>(sc) << 32) | (static_cast<unsigned long long>(0xffff - t) << 16) |
                                       static_cast<unsigned long long>(0xffff - o);
        if (tid == 0) atomicMax(&s_best, key);
        for (int p = start + tid; p < start + span; p += FP_NT) atomicMin(&s_cov[p], t);
    };
    for (int r = 0; r < n; ++r) {
        const int q0 = a.read_off[r0 + r], span = a.read_off[r0 + r + 1] - q0;
        __syncthreads();                  // the previous read's result has been taken
        if (span <= kKmer) {              // BuildIndex skips reads of <= 32 bases (:437-440)
            if (tid == 0) { a.hit_score[hb + r] = 0; a.hit_pos[hb + r] = -1; }
            continue;
        }
        for (int i = tid; i < span; i += FP_NT) s_read[i] = a.read_bytes[q0 + i];
        if (tid == 0) { s_best = 0ull; s_neg = 0xffffffffu; s_d0_visit = 0xffffffffu; s_d0_mism = 99; s_ncand = 0; }
        __syncthreads();
        const int dmin = -(span - kKmer), dmax = L - kKmer;
        for (int d = dmin + tid; d <= dmax; d += FP_NT) {
            const int q_lo = d < 0 ? -d : 0, q_hi = min(span, L - d);
            const bool full = d >= 0 && d + span <= L;
            int run = 0, first = -1, mism = 0;
            // A run of >= 32 equal bytes contains a whole 16-aligned block of the read, so a diagonal on which no such block matches
            // has no k-mer hit: nothing to mark, nothing to emit - and only diagonal 0 needs its mismatch count without a hit of its
            // own (the clipped diagonals may supply one).  A block of unrelated sequence fails after 1.3 compares on average.
            bool maybe = d == 0;
            // (sixteen bytes a side in one LDS round trip: the read's block is one aligned 16-byte read - the same address in every lane -,
            // the haplotype's the five dwords around it shifted into place; byte by byte a block cost ~3.5 dependent round trips, the
            // longest lane's)
            for (int qb = (q_lo + 15) & ~15; qb + 16 <= q_hi && !maybe; qb += 16) {
                const int at = qb + d, sh = at & 3;
                const unsigned* hp = reinterpret_cast<const unsigned*>(s_hap + (at & ~3));
                const unsigned w0 = hp[0], w1 = hp[1], w2 = hp[2], w3 = hp[3], w4 = hp[4];
                const uint4 rb = *reinterpret_cast<const uint4*>(s_read + qb);
                const unsigned diff = (__builtin_amdgcn_alignbyte(w1, w0, sh) ^ rb.x) | (__builtin_amdgcn_alignbyte(w2, w1, sh) ^ rb.y) |
                                      (__builtin_amdgcn_alignbyte(w3, w2, sh) ^ rb.z) | (__builtin_amdgcn_alignbyte(w4, w3, sh) ^ rb.w);
                maybe = diff == 0u;
            }
            if (!maybe) continue;
            {
                const int slot = atomicAdd(&s_ncand, 1);
                if (slot < FP_CAND) { s_cand[slot] = d; continue; }               // (a full list: this thread walks its diagonal itself)
            }
            for (int q = q_lo; q < q_hi; ++q) {
                const unsigned char x = s_hap[q + d], y = s_read[q];
                const bool eq = x == y;
                run = eq ? run + 1 : 0;
                if (run >= kKmer) {
                    s_seed[q - (kKmer - 1) + d] = 1;
                    if (first < 0) first = q - (kKmer - 1);
                }
                mism += (!eq && x != 'N' && y != 'N') ? 1 : 0;
            }
            const unsigned visit = first >= 0 ? (static_cast<unsigned>(first + d) << 16) | static_cast<unsigned>(first) : 0xffffffffu;
            if (d == 0) { s_d0_mism = full ? mism : 99; s_d0_visit = visit; }
            else if (d > 0) { if (first >= 0 && full && mism <= kMaxMism) emit(d, span, mism, first + d, first); }
            else if (first >= 0) atomicMin(&s_neg, visit);
        }
        __syncthreads();
        const int n_cand = min(s_ncand, FP_CAND);
        // a wavefront per listed diagonal (no workgroup barrier inside: everything a diagonal contributes goes through atomics that
        // commute - s_best, s_cov, s_neg - or belongs to diagonal 0 alone).  64 positions a step: the equality bits are a ballot, a
        // run of 32 that starts at bit p is p of M & M>>1 & ... folded five times over two steps' masks - scalar arithmetic.
        for (int c = tid >> 6; c < n_cand; c += FP_NT / 64) {
            const int lane = tid & 63;
            const int d = s_cand[c];
            const int q_lo = d < 0 ? -d : 0, q_hi = min(span, L - d);
            const bool full = d >= 0 && d + span <= L;
            int first = -1, mism = 0;
            unsigned long long cur = 0ull;                      // equality bits of positions [qs, qs + 64)
            {
                const int q = lane;
                const bool in = q >= q_lo && q < q_hi;
                const unsigned char x = in ? s_hap[q + d] : 0, y = in ? s_read[q] : 1;
                cur = __ballot(in && x == y);
                mism += __popcll(__ballot(in && x != y && x != 'N' && y != 'N'));
            }
            for (int qs = 0; qs < q_hi; qs += 64) {
                unsigned long long nxt = 0ull;
                {
                    const int q = qs + 64 + lane;
                    const bool in = q >= q_lo && q < q_hi;
                    const unsigned char x = in ? s_hap[q + d] : 0, y = in ? s_read[q] : 1;
                    nxt = __ballot(in && x == y);
                    mism += __popcll(__ballot(in && x != y && x != 'N' && y != 'N'));
                }
                // starts p in [0, 64) of a run of 32 set bits in the 128 bits (nxt : cur)
                unsigned long long lo = cur, hi = nxt;
#pragma unroll
                for (int k = 1; k < 32; k <<= 1) {
                    const unsigned long long slo = (lo >> k) | (hi << (64 - k)), shi = hi >> k;
                    lo &= slo; hi &= shi;
                }
                if (lo) {
                    if (first < 0) first = qs + __builtin_ctzll(lo);
                    if ((lo >> lane) & 1ull) s_seed[qs + lane + d] = 1;
                }
                cur = nxt;
            }
            const unsigned visit = first >= 0 ? (static_cast<unsigned>(first + d) << 16) | static_cast<unsigned>(first) : 0xffffffffu;
            auto emit_wave = [&](int start, int t, int o) {
                const int sc = (span - mism) * 4 - mism * 6;
                const unsigned long long key = (static_cast<unsigned long long>(sc) << 32) | (static_cast<unsigned long long>(0xffff - t) << 16) |
                                               static_cast<unsigned long long>(0xffff - o);
                if (lane == 0) atomicMax(&s_best, key);
                for (int p = start + lane; p < start + span; p += 64) atomicMin(&s_cov[p], t);
            };
            if (d == 0) { if (lane == 0) { s_d0_mism = full ? mism : 99; s_d0_visit = visit; } }
            else if (d > 0) { if (first >= 0 && full && mism <= kMaxMism) emit_wave(d, first + d, first); }
            else if (first >= 0 && lane == 0) atomicMin(&s_neg, visit);
        }
        __syncthreads();
        {                                 // start 0: k-mer hits of diagonal 0 and of every clipped diagonal
            const unsigned visit = min(s_d0_visit, s_neg);
            if (visit != 0xffffffffu && s_d0_mism <= kMaxMism) emit_all(0, span, s_d0_mism, int(visit >> 16), int(visit & 0xffffu));
        }
        __syncthreads();
        if (tid == 0) {
            const unsigned long long best = s_best;
            if (best) {
                const int sc = int(best >> 32), t = 0xffff - int((best >> 16) & 0xffffull), o = 0xffff - int(best & 0xffffull);
                a.hit_score[hb + r] = sc;
                a.hit_pos[hb + r] = max(0, t - o);
                score_sum += sc;
            } else { a.hit_score[hb + r] = 0; a.hit_pos[hb + r] = -1; }
        }
    }
    if (tid == 0) s_dropped = 0;
    __syncthreads();
    if (!a.hap_isref[h]) {
        const int prefix = a.win_prefix[w];
        const unsigned long long hi = static_cast<unsigned long long>(L) - static_cast<unsigned long long>(static_cast<long long>(a.win_suffix[w]));
        for (int i = tid; i + kKmer <= L; i += FP_NT)
            if (s_seed[i] && i >= prefix && static_cast<unsigned long long>(i) < hi && s_cov[i] > i) s_dropped = 1;
    }
    __syncthreads();
    __shared__ int s_score;
    if (tid == 0) { s_score = s_dropped ? 0 : score_sum; a.hap_score[h] = s_score; }
    __syncthreads();
    if (s_score == 0)
        for (int r = tid; r < n; r += FP_NT) { a.hit_score[hb + r] = 0; a.hit_pos[hb + r] = -1; }
}

// ------------------------------------------------------------------------------------------------------------------------
// Striped Smith-Waterman end points.  k_sw<true>: the 8-bit passes, 16-lane rows (forward; backward when the forward pass did not
// overflow).  k_sw<false>: the 16-bit passes of the alignments whose 8-bit pass overflowed (score >= 249: any read with 63 matching
// bases in a row, every haplotype against its reference), 8-lane rows.  Every lane reads and writes only its own LDS entries (layout
// at sw_sp below), so there is no barrier inside a pass - lanes meet in DPP shifts and row reductions only.
struct SwDesc { int ref_off, R, q_off, Q; };
struct RowPass { int score, ref_end, read_end; bool overflow; };

constexpr int kBias = 6, kGapO = 8, kGapE = 2;

// LDS of a row, lane-private and contiguous: lane l owns stripe positions j = 0 .. seg-1 at [l * SP + j] of every array (SP = the
// launch's largest seg rounded up to 16 positions, an odd multiple of 8 so that the lanes' 16-byte reads fall on different banks):
//   P [LW][SP] shorts    the query profile of SSW (ssw.c:64-116) as one 16-bit word per position: five 3-bit fields f, one per
//                        reference code (field 4: a reference base that matches nothing) from bit 1 up, score = 2 f - 6, i.e. f = 5
//                        match (4), 0 mismatch (-6), 3 padding (0) - a column shifts its field down and reads 2 f = score + bias for a pair of positions
//   H, E [LW][SP] shorts the H column - ONE array, updated in place: the old value of a position is the next position's diagonal and is
//                        read before the new one is written - and E; 8 positions = one ds_read_b128
// 6 bytes per stripe position.  Round 5 kept five signed-byte profile planes and two H columns, 11 bytes: the LDS footprint is what
// bounds the number of resident wavefronts of the long classes (a 590-base haplotype against its reference, four to a wavefront: 31 KB,
// five wavefronts per CU - one per SIMD, which then runs a dependent chain at 7-9 clocks per instruction with nothing to overlap it).
// A lane never touches another lane's entries: lanes meet in DPP shifts and row reductions only, and there is no barrier inside a pass.
__host__ __device__ inline int sw_sp(int segcap) { int sp = (segcap + 15) / 16 * 16; if (((sp / 8) & 1) == 0) sp += 8; return sp; }
// (the operands themselves stay in HBM: a pass reads one reference code per column, two columns ahead of its use, and the query once,
// when the profile is built)
__host__ __device__ inline size_t sw_row_bytes(int segcap, int LW) { return size_t(sw_sp(segcap)) * LW * (3 * sizeof(short)); }

// The rows' arrays are addressed as LDS explicitly (address space 3): row_pass is a function of its own, and through generic pointers its
// reads and writes were FLAT instructions - an address-space check and the global path's latency in front of every LDS access, on the
// critical path of every group of positions (round 5's kernels: 13 clocks per instruction for a wavefront alone on its SIMD).
typedef __attribute__((address_space(3))) short* lds_i16;
typedef __attribute__((address_space(3))) unsigned short* lds_u16;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <class P> __device__ __forceinline__ u32x4 lds_ld16(P p) { return *(const __attribute__((address_space(3))) u32x4*)p; }
template <class P> __device__ __forceinline__ u32x2 lds_ld8(P p) { return *(const __attribute__((address_space(3))) u32x2*)p; }
template <class P> __device__ __forceinline__ unsigned lds_ld4(P p) { return *(const __attribute__((address_space(3))) unsigned*)p; }
template <class P> __device__ __forceinline__ void lds_st16(P p, unsigned a, unsigned b, unsigned c, unsigned d) {
    u32x4 v; v.x = a; v.y = b; v.z = c; v.w = d;
    *(__attribute__((address_space(3))) u32x4*)p = v;
}
template <class P> __device__ __forceinline__ void lds_st8(P p, unsigned a, unsigned b) {
    u32x2 v; v.x = a; v.y = b;
    *(__attribute__((address_space(3))) u32x2*)p = v;
}
template <class P> __device__ __forceinline__ void lds_st4(P p, unsigned a) { *(__attribute__((address_space(3))) unsigned*)p = a; }

// `rev_from` >= 0: the query is qraw[rev_from], qraw[rev_from - 1], ... (the reversed prefix of the backward pass)
// BEHIND_NEG: the entries j >= seg score -6 against every reference base (row_pass_regs computes those positions) instead of padding's 0
template <int LW, bool BEHIND_NEG = false>
__device__ __forceinline__ void build_profile(lds_u16 prof, const signed char* qraw, int Q, int seg, int SP, int l, int rev_from) {
    for (int j = 0; j < SP; ++j) {
        const int q = l * seg + j;
        const int c = (j < seg && q < Q) ? int(rev_from >= 0 ? qraw[rev_from - q] : qraw[q]) : 7;
        unsigned w = 0u;
#pragma unroll
        for (int rc = 0; rc < 5; ++rc) w |= (c == 7 ? 3u : ((c == rc && rc < 4) ? 5u : 0u)) << (3 * rc + 1);      // bits 1 .. 15: a field reads as 2 f
        if (BEHIND_NEG && j >= seg) w = 0u;
        prof[l * SP + j] = static_cast<unsigned short>(w);
    }
}

// packed pairs of 16-bit values (all values here are 0 .. 32767, so signed and unsigned order agree): SSE2's own operations
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_subs(unsigned a, unsigned b) {      // _mm_subs_epu16
    return __builtin_bit_cast(unsigned, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ unsigned pk_max(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, a), __builtin_bit_cast(i16x2, b)));
}
__device__ __forceinline__ unsigned pk2(int lo, int hi) { return (unsigned(lo) & 0xffffu) | (unsigned(hi) << 16); }

// lane l takes lane l - S of its row, the first S lanes take 0
template <int LW, int S>
__device__ __forceinline__ int row_shl(int v, int l) {
    v = dpp_i<0x110 + S>(v);                         // row_shr:S, out-of-row lanes read 0
    if (LW == 8 && l < S) v = 0;
    return v;
}

// The lazy-F step of a column.  The reference's two loops (ssw.c:207-241 for bytes: test, then correct, wrapping around the stripes;
// :446-459 for words: correct, then test, at most `lanes` rounds) carry the F that leaves lane l - 1's stripe into lane l's, sweep
// after sweep, and stop at the first position where no lane's carried F can matter any more.  Run to the end they compute
//     Fin[l]  = max over k >= 1 of ( F_out[l - k] - ext * seg * (k - 1) )          (a max-plus scan over the lanes, saturating at 0)
//     H[l][j] = max( H[l][j], Fin[l] - ext * j )
// and their exits are pure shortcuts: when a loop stops, every lane's carried F is 0 or lies gap_open below an H that a stronger
// chain (the main pass, or an earlier sweep - already applied) put there, so everything not yet applied changes nothing
// (oracle/ssw_model.cpp states both forms; tests/test_realign.py::test_lazy_f_closed_form_equals_the_loops holds them equal column by
// column).  So a column costs ONE pass over the stripe: the scan is log2(lanes) DPP steps on the outgoing F, and the sweep is not done
// at all - the column stays in LDS as the main loop wrote it and its Fin stays in a register: the next column corrects H as it loads
// it (two packed instructions per pair of positions), the column maximum is max(main-loop maximum, Fin) since the correction is
// largest at j = 0, and the search for the best cell corrects on the fly as well.  E never sees the corrections, as in the reference.
template <bool BYTE>
__device__ RowPass row_pass(const signed char* refc, int r_begin, int r_end, int r_step, lds_u16 prof, int Q, int seg, int SP,
                            lds_i16 Hc, lds_i16 E, int terminate, int l) {
    constexpr int LW = BYTE ? 16 : 8;
    const int lb = l * SP;
    {
        for (int j = 0; j < SP; j += 8) { lds_st16(Hc + lb + j, 0u, 0u, 0u, 0u); lds_st16(E + lb + j, 0u, 0u, 0u, 0u); }
    }
    const lds_i16 store = Hc;           // the column being written ...
    const lds_i16 load = Hc;            // ... over the one before it (read first, position by position)
    int best = 0, ref_end = BYTE ? -1 : 0, best_q = 0x7fffffff;
    bool overflow = false;
    int fin = 0;                                     // Fin of the column in `store` (the last one written)
    const int D = kGapE * seg;
    const int n_col = (r_end - r_begin) * r_step;
    int rc = n_col > 0 ? refc[r_begin] : 0, rc_next = n_col > 1 ? refc[r_begin + r_step] : 0;
    int col = 0;
    for (int i = r_begin; i != r_end; i += r_step, ++col) {
        const int rc_next2 = col + 2 < n_col ? refc[i + 2 * r_step] : 0;       // from HBM, two columns ahead of its use
        int f = 0, colmax = 0, garg = 0;                       // garg: the first group of eight whose maximum is the lane's
        int h = row_shl1<LW>(max(int(store[lb + seg - 1]), max(fin - kGapE * (seg - 1), 0)), l);
        // (the array holds column i - 1 as its main loop left it - `fin` completes it - and becomes column i group by group)
        const lds_u16 pr = prof + lb;
        const int sh0 = 3 * (unsigned(rc) < 4u ? rc : 4);                                // field 4: matches nothing
        int fg = fin;                                          // Fin - ext * j0
        // G = positions a call handles: 8, or - for what is left of a stripe behind its last full eight, which for a short query is the
        // whole stripe - 4 or 2.  (A 30-base read on 16 lanes is 2 positions per lane: as a predicated group of eight, three quarters of the
        // column's instructions computed nothing, and half of the batch's alignments are such reads - the stage is bound by instruction issue.)
        auto group = [&](int j0, auto g_const, auto is_tail) {
            constexpr int G = decltype(g_const)::value;
            constexpr bool TAIL = decltype(is_tail)::value;
            unsigned ew[4] = {0u, 0u, 0u, 0u}, hw[4] = {0u, 0u, 0u, 0u}, pw[4] = {0u, 0u, 0u, 0u};
            if (G == 8) {
                const u32x4 e8 = lds_ld16(E + lb + j0), h8 = lds_ld16(load + lb + j0), p8 = lds_ld16(pr + j0);
                ew[0] = e8.x; ew[1] = e8.y; ew[2] = e8.z; ew[3] = e8.w;
                hw[0] = h8.x; hw[1] = h8.y; hw[2] = h8.z; hw[3] = h8.w;
                pw[0] = p8.x; pw[1] = p8.y; pw[2] = p8.z; pw[3] = p8.w;
            } else if (G == 4) {
                const u32x2 e4 = lds_ld8(E + lb + j0), h4 = lds_ld8(load + lb + j0), p4 = lds_ld8(pr + j0);
                ew[0] = e4.x; ew[1] = e4.y; hw[0] = h4.x; hw[1] = h4.y; pw[0] = p4.x; pw[1] = p4.y;
            } else {
                ew[0] = lds_ld4(E + lb + j0);
                hw[0] = lds_ld4(load + lb + j0);
                pw[0] = lds_ld4(pr + j0);
            }
            const unsigned fg2 = pk2(fg, fg);
#pragma unroll
            for (int k = 0; k < G / 2; ++k) hw[k] = pk_max(hw[k], pk_subs(fg2, pk2(kGapE * 2 * k, kGapE * (2 * k + 1))));
            fg = max(fg - kGapE * G, 0);
            // Two positions to an instruction wherever the recurrence allows it.  With t = max(diagonal + score, e) - which does not depend on
            // F - a position is h = max(t, f), f' = max(f - ext, max(h - open, 0)) = max(f - ext, max(t - open, 0)) (f - open < f - ext): only
            // the two operations that carry F from position to position are a chain; the diagonal, the score, t, u = max(t - open, 0) before it
            // and h, the new E, the group's maximum after it work on the PAIRS of 16-bit values the arrays hold anyway (v_pk_*: SSE2's own
            // operations - all values here are 0 .. 32 767, saturating adds and subtractions as in ssw.c).  Round 5 unpacked every position into
            // a 32-bit register: ~17 vector instructions per position, ~10 now.
            unsigned sw[4] = {0u, 0u, 0u, 0u}, nw[4] = {ew[0], ew[1], ew[2], ew[3]};
            unsigned tp[G / 2], up[G / 2];
            const unsigned shv = unsigned(sh0);
#pragma unroll
            for (int k = 0; k < G / 2; ++k) {
                // the diagonals of positions 2k, 2k + 1: the old (corrected) H of 2k - 1 and 2k
                const unsigned dg = k == 0 ? ((unsigned(h) & 0xffffu) | (hw[0] << 16)) : __builtin_amdgcn_alignbit(hw[k], hw[k - 1], 16);
                const unsigned fld = (pw[k] >> shv) & 0x000e000eu;                 // score + 6 of both
                unsigned x;
                static_assert(kBias == 6, "the profile's fields hold (score + bias) / 2");
                const u16x2 s2 = __builtin_bit_cast(u16x2, dg) + __builtin_bit_cast(u16x2, fld);                               // h + score + bias
                // (16 bits: the reference adds with signed saturation at 32 767 and takes the maximum with E >= 0 next - scores here stay
                // below 4 x 2 048, and a sum below the bias ends at 0 either way)
                if (BYTE) x = pk_subs(__builtin_bit_cast(unsigned, __builtin_elementwise_min(s2, (u16x2){255, 255})), pk2(kBias, kBias));
                else x = pk_subs(__builtin_bit_cast(unsigned, s2), pk2(kBias, kBias));
                tp[k] = pk_max(x, ew[k]);
                up[k] = pk_subs(tp[k], pk2(kGapO, kGapO));
            }
            h = int(hw[G / 2 - 1] >> 16);                                          // the diagonal of the next group's first position
            int gm = 0;
            unsigned gm2 = 0u;
#pragma unroll
            for (int k = 0; k < G / 2; ++k) {
                const bool v0 = !TAIL || j0 + 2 * k < seg, v1 = !TAIL || j0 + 2 * k + 1 < seg;
                const int f0 = f;
                if (v0) f = max(f - kGapE, int(up[k] & 0xffffu));
                const int f1 = f;
                if (v1) f = max(f - kGapE, int(up[k] >> 16));
                unsigned hh = pk_max(tp[k], pk2(f0, f1));
                unsigned en = pk_max(pk_subs(ew[k], pk2(kGapE, kGapE)), pk_subs(hh, pk2(kGapO, kGapO)));            // E never sees the lazy-F corrections
                if (TAIL) {
                    const unsigned m = (v0 ? 0xffffu : 0u) | (v1 ? 0xffff0000u : 0u);
                    hh &= m;
                    en = (en & m) | (ew[k] & ~m);
                }
                gm2 = pk_max(gm2, hh);
                sw[k] = hh;
                nw[k] = en;
            }
            gm = max(int(gm2 & 0xffffu), int(gm2 >> 16));
            if (gm > colmax) { colmax = gm; garg = j0; }
            if (G == 8) {
                lds_st16(store + lb + j0, sw[0], sw[1], sw[2], sw[3]);
                lds_st16(E + lb + j0, nw[0], nw[1], nw[2], nw[3]);
            } else if (G == 4) {
                lds_st8(store + lb + j0, sw[0], sw[1]);
                lds_st8(E + lb + j0, nw[0], nw[1]);
            } else {
                lds_st4(store + lb + j0, sw[0]);
                lds_st4(E + lb + j0, nw[0]);
            }
        };
        int j0 = 0;
        for (; j0 + 8 <= seg; j0 += 8) group(j0, std::integral_constant<int, 8>{}, std::false_type{});
        {
            const int left = seg - j0;                         // 0 .. 7 (wave-uniform per row only through seg: lanes of a wavefront's rows may differ)
            if (left > 4) group(j0, std::integral_constant<int, 8>{}, std::true_type{});
            else if (left > 2) group(j0, std::integral_constant<int, 4>{}, std::true_type{});
            else if (left > 0) group(j0, std::integral_constant<int, 2>{}, std::true_type{});
        }
        // Fin of this column: lane l - 1's outgoing F, or an earlier lane's after whole stripes of decay
        fin = row_shl1<LW>(f, l);
        fin = max(fin, max(row_shl<LW, 1>(fin, l) - D, 0));
        fin = max(fin, max(row_shl<LW, 2>(fin, l) - 2 * D, 0));
        fin = max(fin, max(row_shl<LW, 4>(fin, l) - 4 * D, 0));
        if (LW == 16) fin = max(fin, max(row_shl<LW, 8>(fin, l) - 8 * D, 0));
        const int lane_max = colmax;                       // of the main loop's values of this lane's stripe
        colmax = row_max<LW>(max(colmax, fin));
        if (colmax > best) {
            best = colmax;
            if (BYTE && best + kBias >= 255) { overflow = true; break; }
            ref_end = i;
            // the smallest linear query position that holds the new maximum.  In a lane that is position 0 when its Fin is the maximum (the
            // correction Fin - ext * j is largest there and cannot reach the maximum anywhere else), otherwise a position of the FIRST group
            // of eight whose main-loop maximum is the lane's - the groups before it stay below it - and of no group when the lane's maximum
            // is not the row's.  One group per lane: a haplotype against its reference sets a new maximum in nearly every column, and a
            // search through the whole stripe was a third of the pass's instructions.
            int mq = 0x7fffffff;
            if (fin == best) mq = l * seg;
            else if (lane_max == best) {
                const unsigned bb = pk2(best, best);
                const int s0 = garg;
                const u32x4 a8 = lds_ld16(store + lb + s0);
                const unsigned sv[4] = {a8.x, a8.y, a8.z, a8.w};
#pragma unroll
                for (int k = 3; k >= 0; --k) {
                    const unsigned x = sv[k] ^ bb;
                    if (s0 + 2 * k + 1 < seg && (x >> 16) == 0u) mq = l * seg + s0 + 2 * k + 1;
                    if (s0 + 2 * k < seg && (x & 0xffffu) == 0u) mq = l * seg + s0 + 2 * k;
                }
            }
            best_q = row_min<LW>(mq);
        }
        if (colmax == terminate) break;
        rc = rc_next; rc_next = rc_next2;
    }
    int read_end = Q - 1;
    if (best == 0) read_end = min(read_end, 0);          // the zeroed hmax matches a maximum of 0 at position 0
    else if (best_q < read_end) read_end = best_q;
    return RowPass{overflow ? 255 : best, ref_end, read_end, overflow};
}

// A run of 63 matching bases on one diagonal carries H to 249 - the 8-bit pass's overflow threshold (ssw.c:165, 243-247: max + bias >= 255)
// - whatever else the matrix holds: a column computes H = max(diagonal + score, E, F) with saturating bytes, so H never falls below the
// diagonal's chain.  A haplotype begins and ends like its window's reference, so for the haplotype-length classes the 8-bit pass is
// known to give up before it starts: the row says so and leaves (the 16-bit pass computes the alignment either way).
__device__ __forceinline__ bool sure_overflow16(const signed char* refc, int R, const signed char* qraw, int Q, int l) {
    if (R < 64 || Q < 64) return false;
    int okp = 1, oks = 1;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int i = 4 * l + t;
        const int a = refc[i], b = qraw[i], c = refc[R - 64 + i], d = qraw[Q - 64 + i];
        okp &= int(a == b && unsigned(a) < 4u);
        oks &= int(c == d && unsigned(c) < 4u);
    }
    return row_min<16>(okp) != 0 || row_min<16>(oks) != 0;
}

// The same pass with the H column and E in REGISTERS (round 6, late): NG groups of eight stripe positions per lane, the loop over a
// stripe unrolled so that every group has registers of its own; the LDS holds the profile only (2 bytes per position instead of 6).  Why:
// the three long classes' 16-bit launches are the stage (5-6 ms of its 8.5), their LDS footprint (31 KB a wavefront at 590 bases) leaves a
// CU five wavefronts - one to a SIMD, which then issues a chain of dependent instructions at 5-6 clocks each with nothing beside it -
// and while they hold the LDS the short classes' launches wait.  The arithmetic is row_pass's, position for position.  (Stage: 8.5 -> 5.6 ms
// together with the stream layout of sw_ends_pool and sure_overflow16)  What differs is
// the end of a stripe: row_pass masks the positions j >= seg of the last group (TAIL); here they are computed like any other, against a
// profile entry that scores -6 for every reference base (build_profile<.., true>), and are inert -
//   * their H never exceeds the best score seen before this column (diagonal: an H of the previous column - 6; E: an earlier H - 8) or lies
//     below an H of the same lane and column (F), so they neither set a maximum nor equal a new one;
//   * they feed positions j >= seg only (diagonal and F run towards higher j), except through the two values a lane hands on: its outgoing
//     F and the H of its last position - both are taken where position seg - 1 is computed (`fout`, `hlast`), not at the end of the loop.
template <bool BYTE, int NG, int GW>
__device__ RowPass row_pass_regs(const signed char* refc, int r_begin, int r_end, int r_step, lds_u16 prof, int Q, int seg, int SP, int terminate, int l) {
    constexpr int LW = BYTE ? 16 : 8;
    const int lb = l * SP;
    static_assert(GW == 8 || GW == 4, "a group is eight positions, or four for the shortest queries");
    constexpr int GP = GW / 2;                                   // packed pairs of a group
    unsigned Hr[NG][GP], Er[NG][GP];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int k = 0; k < GP; ++k) { Hr[g][k] = 0u; Er[g][k] = 0u; }
    int best = 0, ref_end = BYTE ? -1 : 0, best_q = 0x7fffffff;
    bool overflow = false;
    int fin = 0, hlast = 0;
    const int D = kGapE * seg;
    const int gl = (seg - 1) / GW, rl_ = (seg - 1) % GW;          // the group and the place in it of the stripe's last position
    const int n_col = (r_end - r_begin) * r_step;
    int rc = n_col > 0 ? refc[r_begin] : 0, rc_next = n_col > 1 ? refc[r_begin + r_step] : 0;
    int col = 0;
    const lds_u16 pr = prof + lb;
    for (int i = r_begin; i != r_end; i += r_step, ++col) {
        const int rc_next2 = col + 2 < n_col ? refc[i + 2 * r_step] : 0;
        int f = 0, colmax = 0, garg = 0, fout = 0;
        int h = row_shl1<LW>(max(hlast, max(fin - kGapE * (seg - 1), 0)), l);
        const unsigned shv = unsigned(3 * (unsigned(rc) < 4u ? rc : 4));
        int fg = fin;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (GW * g < seg) {
                const int j0 = GW * g;
                unsigned pw[GP], hw[GP], ew[GP];
                if (GW == 8) {
                    const u32x4 p8 = lds_ld16(pr + j0);
                    pw[0] = p8.x; pw[1] = p8.y; pw[GP - 2] = p8.z; pw[GP - 1] = p8.w;
                } else {
                    const u32x2 p4 = lds_ld8(pr + j0);
                    pw[0] = p4.x; pw[1] = p4.y;
                }
#pragma unroll
                for (int k = 0; k < GP; ++k) { hw[k] = Hr[g][k]; ew[k] = Er[g][k]; }
                const unsigned fg2 = pk2(fg, fg);
#pragma unroll
                for (int k = 0; k < GP; ++k) hw[k] = pk_max(hw[k], pk_subs(fg2, pk2(kGapE * 2 * k, kGapE * (2 * k + 1))));
                fg = max(fg - kGapE * GW, 0);
                unsigned tp[GP], up[GP];
#pragma unroll
                for (int k = 0; k < GP; ++k) {
                    const unsigned dg = k == 0 ? ((unsigned(h) & 0xffffu) | (hw[0] << 16)) : __builtin_amdgcn_alignbit(hw[k], hw[k - 1], 16);
                    const unsigned fld = (pw[k] >> shv) & 0x000e000eu;
                    unsigned x;
                    const u16x2 s2 = __builtin_bit_cast(u16x2, dg) + __builtin_bit_cast(u16x2, fld);
                    if (BYTE) x = pk_subs(__builtin_bit_cast(unsigned, __builtin_elementwise_min(s2, (u16x2){255, 255})), pk2(kBias, kBias));
                    else x = pk_subs(__builtin_bit_cast(unsigned, s2), pk2(kBias, kBias));
                    tp[k] = pk_max(x, ew[k]);
                    up[k] = pk_subs(tp[k], pk2(kGapO, kGapO));
                }
                h = int(hw[GP - 1] >> 16);
                unsigned gm2 = 0u;
                int fb[GW + 1];                                // F in front of position j0 + t (fb[GW]: behind the group)
                fb[0] = f;
#pragma unroll
                for (int k = 0; k < GP; ++k) {
                    fb[2 * k + 1] = max(fb[2 * k] - kGapE, int(up[k] & 0xffffu));
                    fb[2 * k + 2] = max(fb[2 * k + 1] - kGapE, int(up[k] >> 16));
                    const unsigned hh = pk_max(tp[k], pk2(fb[2 * k], fb[2 * k + 1]));
                    Er[g][k] = pk_max(pk_subs(ew[k], pk2(kGapE, kGapE)), pk_subs(hh, pk2(kGapO, kGapO)));
                    Hr[g][k] = hh;
                    gm2 = pk_max(gm2, hh);
                }
                f = fb[GW];
                const int gm = max(int(gm2 & 0xffffu), int(gm2 >> 16));
                if (gm > colmax) { colmax = gm; garg = j0; }
                if (g == gl) {                                 // the stripe ends in this group: what the lane hands on
                    int fo = fb[1];
                    unsigned w = Hr[g][0];
#pragma unroll
                    for (int t = 1; t < GW; ++t) if (rl_ == t) { fo = fb[t + 1]; w = Hr[g][t >> 1]; }
                    fout = fo;
                    hlast = int((rl_ & 1) ? (w >> 16) : (w & 0xffffu));
                }
            }
        }
        fin = row_shl1<LW>(fout, l);
        fin = max(fin, max(row_shl<LW, 1>(fin, l) - D, 0));
        fin = max(fin, max(row_shl<LW, 2>(fin, l) - 2 * D, 0));
        fin = max(fin, max(row_shl<LW, 4>(fin, l) - 4 * D, 0));
        if (LW == 16) fin = max(fin, max(row_shl<LW, 8>(fin, l) - 8 * D, 0));
        const int lane_max = colmax;
        colmax = row_max<LW>(max(colmax, fin));
        if (colmax > best) {
            best = colmax;
            if (BYTE && best + kBias >= 255) { overflow = true; break; }
            ref_end = i;
            int mq = 0x7fffffff;
            if (fin == best) mq = l * seg;
            else if (lane_max == best) {
                const unsigned bb = pk2(best, best);
                const int s0 = garg;
                unsigned sv[GP];
#pragma unroll
                for (int k = 0; k < GP; ++k) sv[k] = Hr[0][k];
#pragma unroll
                for (int g = 1; g < NG; ++g)
                    if (s0 == GW * g) {
#pragma unroll
                        for (int k = 0; k < GP; ++k) sv[k] = Hr[g][k];
                    }
#pragma unroll
                for (int k = GP - 1; k >= 0; --k) {
                    const unsigned x = sv[k] ^ bb;
                    if (s0 + 2 * k + 1 < seg && (x >> 16) == 0u) mq = l * seg + s0 + 2 * k + 1;
                    if (s0 + 2 * k < seg && (x & 0xffffu) == 0u) mq = l * seg + s0 + 2 * k;
                }
            }
            best_q = row_min<LW>(mq);
        }
        if (colmax == terminate) break;
        rc = rc_next; rc_next = rc_next2;
    }
    int read_end = Q - 1;
    if (best == 0) read_end = min(read_end, 0);
    else if (best_q < read_end) read_end = best_q;
    return RowPass{overflow ? 255 : best, ref_end, read_end, overflow};
}

template <bool BYTE, int NG, int GW = 8>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(NG <= 4 ? 4 : (NG <= 12 ? 3 : 2)))) void k_sw_regs(const signed char* pool, const SwDesc* desc, const int* order, int n, Ends* out,
                                                  unsigned char* overflowed, int segcap) {
    constexpr int LW = BYTE ? 16 : 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int row = threadIdx.x / LW, l = threadIdx.x % LW;
    const int slot = blockIdx.x * (blockDim.x / LW) + row;
    bool live = slot < n;
    const int k = live ? order[slot] : 0;
    if (!BYTE && live && !overflowed[k]) live = false;
    const int SP = sw_sp(segcap);
    const lds_u16 prof = (lds_u16)(lds + size_t(row) * (size_t(SP) * LW * sizeof(short)));
    if (!live) return;
    // the longest alignments are the end of the stage: their wavefronts go first where a SIMD's wavefronts compete for issue slots
    __builtin_amdgcn_s_setprio(NG >= 12 ? 3 : (NG >= 8 ? 2 : (NG >= 4 ? 1 : 0)));
    const SwDesc d = desc[k];
    const signed char* refc = pool + d.ref_off;
    const signed char* qraw = pool + d.q_off;
    Ends e{0, 0, 0, 0, 0, 16};
    bool ovf = false;
    if (BYTE && sure_overflow16(refc, d.R, qraw, d.Q, l)) {
        if (l == 0) overflowed[k] = 1;
        return;
    }
    if (d.R > 0 && d.Q > 0) {
        const int seg = (d.Q + LW - 1) / LW;
        build_profile<LW, true>(prof, qraw, d.Q, seg, SP, l, -1);
        const RowPass fw = row_pass_regs<BYTE, NG, GW>(refc, 0, d.R, 1, prof, d.Q, seg, SP, BYTE ? 255 : 65535, l);
        if (BYTE && fw.overflow) ovf = true;
        else if (fw.score > 0) {
            const int Q2 = fw.read_end + 1, seg2 = (Q2 + LW - 1) / LW;
            build_profile<LW, true>(prof, qraw, Q2, seg2, SP, l, fw.read_end);
            const RowPass bw = row_pass_regs<BYTE, NG, GW>(refc, fw.ref_end, -1, -1, prof, Q2, seg2, SP, fw.score, l);
            e = Ends{fw.score, fw.ref_end, fw.read_end, bw.ref_end, bw.read_end, LW};
        }
    }
    if (l == 0) {
        if (BYTE) overflowed[k] = ovf ? 1 : 0;
        if (!ovf) out[k] = e;
    }
}

template <bool BYTE>
__global__ __launch_bounds__(64) void k_sw(const signed char* pool, const SwDesc* desc, const int* order, int n, Ends* out,
                                             unsigned char* overflowed, int segcap) {
    constexpr int LW = BYTE ? 16 : 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int row = threadIdx.x / LW, l = threadIdx.x % LW;
    const int slot = blockIdx.x * (blockDim.x / LW) + row;
    bool live = slot < n;
    const int k = live ? order[slot] : 0;
    // the 16-bit kernel takes only what the 8-bit pass gave up on.  (Round 6 tried the dense form - the 8-bit launch appends the alignments
    // that overflow to a list, the 16-bit launch runs full wavefronts over it: 10.3 -> 12.0 ms.  A wavefront lasts as long as its longest row
    // whether four or eight of its rows are live, so the list halves the number of wavefronts that share the work and loses the
    // longest-first order on top.)
    if (!BYTE && live && !overflowed[k]) live = false;
    const int SP = sw_sp(segcap);
    unsigned char* base = lds + size_t(row) * ((sw_row_bytes(segcap, LW) + 15) / 16 * 16);
    const lds_i16 Hc = (lds_i16)base;
    const lds_i16 E = Hc + SP * LW;
    const lds_u16 prof = (lds_u16)(E + SP * LW);
    if (!live) return;
    const SwDesc d = desc[k];
    const signed char* refc = pool + d.ref_off;
    const signed char* qraw = pool + d.q_off;
    Ends e{0, 0, 0, 0, 0, 16};
    bool ovf = false;
    if (BYTE && sure_overflow16(refc, d.R, qraw, d.Q, l)) {
        if (l == 0) overflowed[k] = 1;
        return;
    }
    if (d.R > 0 && d.Q > 0) {
        const int seg = (d.Q + LW - 1) / LW;
        build_profile<LW>(prof, qraw, d.Q, seg, SP, l, -1);
        const RowPass fw = row_pass<BYTE>(refc, 0, d.R, 1, prof, d.Q, seg, SP, Hc, E, BYTE ? 255 : 65535, l);
        if (BYTE && fw.overflow) ovf = true;
        else if (fw.score > 0) {
            const int Q2 = fw.read_end + 1, seg2 = (Q2 + LW - 1) / LW;
            build_profile<LW>(prof, qraw, Q2, seg2, SP, l, fw.read_end);
            const RowPass bw = row_pass<BYTE>(refc, fw.ref_end, -1, -1, prof, Q2, seg2, SP, Hc, E, fw.score, l);
            e = Ends{fw.score, fw.ref_end, fw.read_end, bw.ref_end, bw.read_end, LW};
        }
    }
    if (l == 0) {
        if (BYTE) overflowed[k] = ovf ? 1 : 0;
        if (!ovf) out[k] = e;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Device memory of one call: a bump allocator over a block the library keeps between calls (grow-only; one call at a time has the kept
// one, a concurrent call - or one on another device - a block of its own that it frees).  A call makes ~25 small allocations; as
// hipMalloc / hipFree pairs they cost it ~3 ms (hipFree waits for the device each time).
struct DeviceArena {
    struct Block { char* p; size_t cap, used; };
    std::vector<Block> blocks;
    size_t asked = 0;                  // bytes taken since the last reset
    int device = -1;
    void* take(size_t bytes) {
        bytes = (std::max<size_t>(bytes, 1) + 255) & ~size_t(255);
        asked += bytes;
        if (!blocks.empty() && blocks.back().used + bytes <= blocks.back().cap) {
            void* r = blocks.back().p + blocks.back().used;
            blocks.back().used += bytes;
            return r;
        }
        const size_t cap = std::max<size_t>(bytes, std::max<size_t>(size_t(32) << 20, blocks.empty() ? 0 : 2 * blocks.back().cap));
        char* p = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&p), cap) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        blocks.push_back(Block{p, cap, bytes});
        return p;
    }
    void free_all() { for (Block& b : blocks) (void)hipFree(b.p); blocks.clear(); asked = 0; }
    // end of a call: one block large enough for what this call took, so that the next one of its size allocates nothing
    void reset() {
        if (blocks.size() > 1) {
            const size_t want = asked + asked / 4;
            free_all();
            char* p = nullptr;
            if (hipMalloc(reinterpret_cast<void**>(&p), want) == hipSuccess) blocks.push_back(Block{p, want, 0});
            else (void)hipGetLastError();
        } else if (!blocks.empty()) {
            blocks.back().used = 0;
        }
        asked = 0;
    }
};
// the same for page-locked host memory: what a call copies up and down (operand bytes, descriptors, results) is built in and landed on
// pinned blocks, so that hipMemcpyAsync is a DMA and not a staged copy through the runtime's bounce buffers
struct HostArena {
    struct Block { char* p; size_t cap, used; };
    std::vector<Block> blocks;
    size_t asked = 0;
    void* take(size_t bytes) {
        bytes = (std::max<size_t>(bytes, 1) + 255) & ~size_t(255);
        asked += bytes;
        if (!blocks.empty() && blocks.back().used + bytes <= blocks.back().cap) {
            void* r = blocks.back().p + blocks.back().used;
            blocks.back().used += bytes;
            return r;
        }
        const size_t cap = std::max<size_t>(bytes, std::max<size_t>(size_t(16) << 20, blocks.empty() ? 0 : 2 * blocks.back().cap));
        char* p = nullptr;
        if (hipHostMalloc(reinterpret_cast<void**>(&p), cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        blocks.push_back(Block{p, cap, bytes});
        return p;
    }
    bool owns(const void* q) const {
        for (const Block& b : blocks) if (q >= b.p && q < b.p + b.cap) return true;
        return false;
    }
    void free_all() { for (Block& b : blocks) (void)hipHostFree(b.p); blocks.clear(); asked = 0; }
    void reset() {
        if (blocks.size() > 1) {
            const size_t want = asked + asked / 4;
            free_all();
            char* p = nullptr;
            if (hipHostMalloc(reinterpret_cast<void**>(&p), want, hipHostMallocDefault) == hipSuccess) blocks.push_back(Block{p, want, 0});
            else (void)hipGetLastError();
        } else if (!blocks.empty()) {
            blocks.back().used = 0;
        }
        asked = 0;
    }
};
thread_local HostArena* t_harena = nullptr;
// std::vector over the call's pinned arena (plain heap outside a call or when the arena cannot grow)
template <class T>
struct PinnedAlloc {
    typedef T value_type;
    PinnedAlloc() = default;
    template <class U> PinnedAlloc(const PinnedAlloc<U>&) {}
    T* allocate(size_t n) {
        if (t_harena) { void* p = t_harena->take(n * sizeof(T)); if (p) return static_cast<T*>(p); }
        return static_cast<T*>(::operator new(n * sizeof(T)));
    }
    void deallocate(T* p, size_t) { if (!(t_harena && t_harena->owns(p))) ::operator delete(p); }
    template <class U> bool operator==(const PinnedAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const PinnedAlloc<U>&) const { return false; }
};
template <class T> using pinned_vector = std::vector<T, PinnedAlloc<T>>;

thread_local DeviceArena* t_arena = nullptr;
struct ArenaLease {
    static std::mutex& lock() { static std::mutex m; return m; }
    static DeviceArena& kept() { static DeviceArena a; return a; }
    static HostArena& kept_host() { static HostArena a; return a; }
    static bool& busy() { static bool b = false; return b; }
    DeviceArena* a = nullptr;
    DeviceArena* prev = nullptr;
    HostArena* h = nullptr;
    HostArena* hprev = nullptr;
    bool from_kept = false;
    ArenaLease() {
        int dev = -1;
        (void)hipGetDevice(&dev);
        {
            std::lock_guard<std::mutex> g(lock());
            if (!busy() && (kept().device < 0 || kept().device == dev)) { busy() = true; kept().device = dev; a = &kept(); from_kept = true; }
        }
        if (!a) a = new DeviceArena();
        h = from_kept ? &kept_host() : new HostArena();
        prev = t_arena; hprev = t_harena;
        t_arena = a; t_harena = h;
    }
    ~ArenaLease() {
        t_arena = prev; t_harena = hprev;
        if (from_kept) { a->reset(); h->reset(); std::lock_guard<std::mutex> g(lock()); busy() = false; }
        else { a->free_all(); delete a; h->free_all(); delete h; }
    }
};

template <class T>
struct DevBuf {
    T* p = nullptr;
    bool owned = true;                 // false: the call's arena owns the bytes
    ~DevBuf() { if (p && owned) (void)hipFree(p); }
    int alloc(size_t n) {
        if (t_arena) {
            p = static_cast<T*>(t_arena->take(std::max<size_t>(n, 1) * sizeof(T)));
            owned = false;
            CTO_REQUIRE(p != nullptr, CTO_EHIP, "cto_realign_windows: out of device memory");
            return CTO_OK;
        }
        CTO_HIP(hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(n, 1) * sizeof(T)));
        return CTO_OK;
    }
    template <class A>
    int put(const std::vector<T, A>& v, hipStream_t s) {
        int rc = alloc(v.size());
        if (rc != CTO_OK) return rc;
        if (!v.empty()) CTO_HIP(hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
        return CTO_OK;
    }
    void swap(DevBuf& o) { std::swap(p, o.p); std::swap(owned, o.owned); }
};

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Worker threads that outlive a call.  A call runs a dozen parallel loops over its windows (packing, collecting pairs, filing results,
// planning and composing tracebacks) of a millisecond or less each; as std::threads created and joined per loop, sixteen at a time, the
// creation alone was a third of a millisecond per loop.  The pool is made once (grown on demand, never destroyed: its threads sleep on a
// condition variable and end with the process); one call at a time uses it, a concurrent one falls back to threads of its own.
class WorkerPool {
    std::mutex m;
    std::condition_variable cv_start, cv_done;
    std::vector<std::thread> th;
    const std::function<void()>* job = nullptr;
    unsigned long long gen = 0;
    int want = 0, pending = 0;
    void worker(int id) {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void()>* mine = nullptr;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_start.wait(lk, [&] { return gen != seen; });
                seen = gen;
                if (id < want) mine = job;
            }
            if (!mine) continue;
            (*mine)();
            std::lock_guard<std::mutex> lk(m);
            if (--pending == 0) cv_done.notify_all();
        }
    }
public:
    std::mutex use;                                    // held by the call that runs loops on the pool
    bool run(int helpers, const std::function<void()>& f) {
        try {
            std::lock_guard<std::mutex> lk(m);
            while (int(th.size()) < helpers) { th.emplace_back(&WorkerPool::worker, this, int(th.size())); th.back().detach(); }
        } catch (...) {
            return false;                              // no more threads to be had
        }
        {
            std::lock_guard<std::mutex> lk(m);
            job = &f; want = helpers; pending = helpers; ++gen;
        }
        cv_start.notify_all();
        f();
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return pending == 0; });
        job = nullptr;
        return true;
    }
    // (a child of fork() has none of the parent's threads: it starts with a pool of its own; the parent's object is left as it is)
    static WorkerPool*& slot() { static WorkerPool* p = nullptr; return p; }
    static WorkerPool& get() {
        static std::once_flag once;
        std::call_once(once, [] { (void)pthread_atfork(nullptr, nullptr, [] { slot() = new WorkerPool(); }); slot() = new WorkerPool(); });
        return *slot();
    }
};

template <class F>
void parallel_for(size_t n, int threads, F&& f) {
    // an exception on a worker (std::bad_alloc in a window's vectors) must not reach std::terminate: the first one is kept, every
    // worker stops taking items, and the caller rethrows it after the join - where the C boundary's CTO_CATCH turns it into an error code
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    std::exception_ptr first;
    std::mutex first_lock;
    const std::function<void()> work = [&]() {
        try {
            for (size_t i = next++; i < n && !failed.load(std::memory_order_relaxed); i = next++) f(i);
        } catch (...) {
            std::lock_guard<std::mutex> g(first_lock);
            if (!first) first = std::current_exception();
            failed.store(true);
        }
    };
    const int nt = int(std::min<size_t>(size_t(std::max(1, threads)), n));
    bool done = false;
    if (nt > 1) {
        WorkerPool& pool = WorkerPool::get();
        std::unique_lock<std::mutex> mine(pool.use, std::try_to_lock);
        if (mine.owns_lock()) done = pool.run(nt - 1, work);
    }
    if (!done) {
        std::vector<std::thread> own;
        try {
            for (int t = 1; t < nt; ++t) own.emplace_back(work);
        } catch (...) {                               // no more threads to be had: the ones that started and this one do the work
        }
        work();
        for (std::thread& t : own) t.join();
    }
    if (first) std::rethrow_exception(first);
}

struct Reaper {
    std::mutex m;
    std::thread t;
    void wait() { if (t.joinable()) t.join(); }
    ~Reaper() { wait(); }
};
Reaper g_reaper;
void reap(std::vector<Window>&& ws) {
    std::lock_guard<std::mutex> lock(g_reaper.m);
    g_reaper.wait();
    auto* gone = new std::vector<Window>(std::move(ws));
    g_reaper.t = std::thread([gone]() { delete gone; });
}

bool device_eligible(const Window& w) {
    if (w.reference.size() > size_t(FP_LMAX) || w.haps.empty()) return false;
    for (const std::string& h : w.haps) if (h.size() > size_t(FP_LMAX)) return false;
    for (const std::string& r : w.reads) if (r.size() > size_t(FP_RMAX)) return false;
    return true;
}

// CTO_REALIGN_TRACE=1: the classes of the Smith-Waterman stage and the wall time of every stage of a call, on stderr
bool trace_on() { static const bool on = std::getenv("CTO_REALIGN_TRACE") != nullptr; return on; }
struct StageClock {
    double t = now_ms();
    void lap(const char* what) { if (trace_on()) { const double n = now_ms(); std::fprintf(stderr, "[realign] %-28s %8.3f ms\n", what, n - t); t = n; } }
};

int fast_pass_device(std::vector<Window*>& ws, hipStream_t s, int threads, cto_realign_stats* st) {
    StageClock clk;
    // sizes first, then every window fills its own slices (on the workers)
    const size_t nw = ws.size();
    std::vector<size_t> r0(nw + 1, 0), h0(nw + 1, 0), rb0(nw + 1, 0), hb0(nw + 1, 0);
    std::vector<long long> hits0(nw + 1, 0);
    for (size_t wi = 0; wi < nw; ++wi) {
        const Window& w = *ws[wi];
        size_t rb = 0, hb = 0;
        for (const std::string& r : w.reads) rb += r.size();
        for (const std::string& h : w.haps) hb += h.size();
        r0[wi + 1] = r0[wi] + w.reads.size(); h0[wi + 1] = h0[wi] + w.haps.size();
        rb0[wi + 1] = rb0[wi] + rb; hb0[wi + 1] = hb0[wi] + hb;
        hits0[wi + 1] = hits0[wi] + (long long)w.haps.size() * w.n_reads();
    }
    const long long hits = hits0[nw];
    pinned_vector<unsigned char> hap_bytes(hb0[nw]), read_bytes(rb0[nw]);            // (page-locked: the two large uploads of the stage)
    std::vector<unsigned char> hap_isref(h0[nw]);
    std::vector<int> hap_off(h0[nw] + 1, 0), hap_win(h0[nw]), read_off(r0[nw] + 1, 0), win_read0(nw + 1, 0), win_prefix(nw), win_suffix(nw);
    std::vector<long long> hit_off(h0[nw]);
    CTO_REQUIRE(hb0[nw] < (size_t(1) << 31) && rb0[nw] < (size_t(1) << 31), CTO_EUNSUPPORTED,
                "cto_realign_windows: more than 2 GiB of sequence in one call; split it");
    parallel_for(nw, threads, [&](size_t wi) {
        const Window& w = *ws[wi];
        size_t at = rb0[wi];
        for (size_t r = 0; r < w.reads.size(); ++r) {
            if (!w.reads[r].empty()) memcpy(read_bytes.data() + at, w.reads[r].data(), w.reads[r].size());
            at += w.reads[r].size();
            read_off[r0[wi] + r + 1] = int(at);
        }
        win_read0[wi + 1] = int(r0[wi + 1]);
        win_prefix[wi] = w.ref_prefix;
        win_suffix[wi] = w.ref_suffix;
        at = hb0[wi];
        long long hit = hits0[wi];
        for (size_t h = 0; h < w.haps.size(); ++h) {
            if (!w.haps[h].empty()) memcpy(hap_bytes.data() + at, w.haps[h].data(), w.haps[h].size());
            at += w.haps[h].size();
            hap_off[h0[wi] + h + 1] = int(at);
            hap_win[h0[wi] + h] = int(wi);
            hap_isref[h0[wi] + h] = w.haps[h] == w.reference ? 1 : 0;
            hit_off[h0[wi] + h] = hit;
            hit += w.n_reads();
        }
    });
    const int nh = int(hap_win.size());
    if (nh == 0) return CTO_OK;
    clk.lap("  fast pass: pack");
    DevBuf<unsigned char> d_hap, d_read, d_isref;
    DevBuf<int> d_hap_off, d_hap_win, d_read_off, d_win_read0, d_prefix, d_suffix, d_hit_score, d_hit_pos, d_hap_score;
    DevBuf<long long> d_hit_off;
    int rc;
    if ((rc = d_hap.put(hap_bytes, s)) || (rc = d_read.put(read_bytes, s)) || (rc = d_isref.put(hap_isref, s)) || (rc = d_hap_off.put(hap_off, s)) ||
        (rc = d_hap_win.put(hap_win, s)) || (rc = d_read_off.put(read_off, s)) || (rc = d_win_read0.put(win_read0, s)) ||
        (rc = d_prefix.put(win_prefix, s)) || (rc = d_suffix.put(win_suffix, s)) || (rc = d_hit_off.put(hit_off, s)) ||
        (rc = d_hit_score.alloc(size_t(hits))) || (rc = d_hit_pos.alloc(size_t(hits))) || (rc = d_hap_score.alloc(size_t(nh))))
        return rc;
    clk.lap("  fast pass: alloc + H2D");
    FpArgs a{d_hap.p, d_hap_off.p, d_hap_win.p, d_isref.p, d_hit_off.p, d_read.p, d_read_off.p, d_win_read0.p, d_prefix.p, d_suffix.p,
             d_hit_score.p, d_hit_pos.p, d_hap_score.p};
    hipEvent_t e0, e1;
    CTO_HIP(hipEventCreate(&e0)); CTO_HIP(hipEventCreate(&e1));
    CTO_HIP(hipEventRecord(e0, s));
    hipLaunchKernelGGL(k_fast_pass, dim3(unsigned(nh)), dim3(FP_NT), 0, s, a);
    CTO_HIP(hipGetLastError());
    CTO_HIP(hipEventRecord(e1, s));
    pinned_vector<int> hit_score(static_cast<size_t>(hits), 0), hit_pos(static_cast<size_t>(hits), 0);
    std::vector<int> hap_score(static_cast<size_t>(nh), 0);
    if (hits) {
        CTO_HIP(hipMemcpyAsync(hit_score.data(), d_hit_score.p, size_t(hits) * sizeof(int), hipMemcpyDeviceToHost, s));
        CTO_HIP(hipMemcpyAsync(hit_pos.data(), d_hit_pos.p, size_t(hits) * sizeof(int), hipMemcpyDeviceToHost, s));
    }
    CTO_HIP(hipMemcpyAsync(hap_score.data(), d_hap_score.p, size_t(nh) * sizeof(int), hipMemcpyDeviceToHost, s));
    CTO_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    CTO_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (st) { st->fast_pass_ms += ms; st->fast_pairs += hits; }
    clk.lap("  fast pass: kernel + D2H");
    std::vector<size_t> hfirst(ws.size() + 1, 0);
    for (size_t wi = 0; wi < ws.size(); ++wi) hfirst[wi + 1] = hfirst[wi] + size_t(ws[wi]->n_haps());
    parallel_for(ws.size(), threads, [&](size_t wi) {
        const size_t hcur = hfirst[wi];
        if (ws[wi]->n_haps() > 0) ws[wi]->set_fast_pass(hit_score.data() + hit_off[hcur], hit_pos.data() + hit_off[hcur], hap_score.data() + hcur);
    });
    return CTO_OK;
}

template <bool BYTE>
int launch_sw(hipStream_t s, const signed char* pool, const SwDesc* desc, const int* order, int n, Ends* out, unsigned char* overflowed,
              int Rcap, int Qcap) {
    if (n == 0) return CTO_OK;
    constexpr int LW = BYTE ? 16 : 8;
    // A pass is a chain of dependent steps: a wavefront runs it at the pace of ONE row whatever the number of rows it holds, so a
    // class with few alignments (the haplotype-length one: ~2 500 of the bench's batch) can spread them over more wavefronts - rows per
    // wavefront halve until the class has CTO_SW_MIN_WAVES wavefronts or one row per wavefront.  Round 5 measured 0 -> 17.4 ms, 256 / 512
    // -> 15.5, 1 024 -> 17.3, 2 048 -> 24.5 and used 512.  Round 6, after the best-cell search stopped re-reading the whole stripe in
    // every column of a matching pair (the longest class's 16-bit launch: 10.6 -> 7.0 ms), the stage is bound by instruction issue and a
    // wavefront with idle lanes costs the others its slots: 0 -> 10.1-10.3 ms, 256 -> 10.9, 512 -> 11.2 - full wavefronts are the default
    int ROWS = 64 / LW;
    static const int min_waves = std::getenv("CTO_SW_MIN_WAVES") ? atoi(std::getenv("CTO_SW_MIN_WAVES")) : 0;
    while (ROWS > 1 && (n + ROWS - 1) / ROWS < min_waves) ROWS /= 2;
    (void)Rcap;
    Qcap = (Qcap + 15) & ~15;
    const int segcap = (Qcap + LW - 1) / LW;
    const size_t row_bytes = (sw_row_bytes(segcap, LW) + 15) / 16 * 16;
    while (ROWS > 1 && row_bytes * ROWS > size_t(160) * 1024) ROWS /= 2;          // queries near 2 048 bases: fewer rows share the LDS
    const size_t smem = row_bytes * ROWS;
    CTO_REQUIRE(smem <= size_t(160) * 1024, CTO_EUNSUPPORTED, "cto_realign_windows: an alignment of %d x %d does not fit the LDS", Rcap, Qcap);
    // once per kernel and for the whole LDS: launches of different classes (and of concurrent calls) must not lower each other's limit
    // (a function attribute belongs to the device it was set on: once per kernel AND device)
    {
        static std::atomic<unsigned char> done[64];
        int dev = 0;
        CTO_HIP(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !done[dev].load(std::memory_order_acquire)) {
            CTO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sw<BYTE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            if (dev >= 0 && dev < 64) done[dev].store(1, std::memory_order_release);
        }
    }
    // stripes of 17 .. 128 positions: the register form (row_pass_regs) - the LDS holds the profile only
    static const bool regs_on = !(std::getenv("CTO_SW_REGS") && atoi(std::getenv("CTO_SW_REGS")) == 0);
    static const int regs_min = std::getenv("CTO_SW_REGS_MIN") ? atoi(std::getenv("CTO_SW_REGS_MIN")) : 0;
    if (regs_on && segcap > regs_min && segcap <= 128 && ROWS == 64 / LW) {
        const size_t pm = size_t(sw_sp(segcap)) * LW * sizeof(short) * ROWS;
        const dim3 grid(unsigned((n + ROWS - 1) / ROWS)), block(unsigned(LW * ROWS));
        if (segcap <= 4) hipLaunchKernelGGL((k_sw_regs<BYTE, 1, 4>), grid, block, pm, s, pool, desc, order, n, out, overflowed, segcap);
        else if (segcap <= 8) hipLaunchKernelGGL((k_sw_regs<BYTE, 1>), grid, block, pm, s, pool, desc, order, n, out, overflowed, segcap);
        else if (segcap <= 16) hipLaunchKernelGGL((k_sw_regs<BYTE, 2>), grid, block, pm, s, pool, desc, order, n, out, overflowed, segcap);
        else if (segcap <= 32) hipLaunchKernelGGL((k_sw_regs<BYTE, 4>), grid, block, pm, s, pool, desc, order, n, out, overflowed, segcap);
        else if (segcap <= 64) hipLaunchKernelGGL((k_sw_regs<BYTE, 8>), grid, block, pm, s, pool, desc, order, n, out, overflowed, segcap);
        else if (segcap <= 96) hipLaunchKernelGGL((k_sw_regs<BYTE, 12>), grid, block, pm, s, pool, desc, order, n, out, overflowed, segcap);
        else hipLaunchKernelGGL((k_sw_regs<BYTE, 16>), grid, block, pm, s, pool, desc, order, n, out, overflowed, segcap);
        CTO_HIP(hipGetLastError());
        return CTO_OK;
    }
    hipLaunchKernelGGL((k_sw<BYTE>), dim3(unsigned((n + ROWS - 1) / ROWS)), dim3(unsigned(LW * ROWS)), smem, s, pool, desc, order, n, out, overflowed, segcap);
    CTO_HIP(hipGetLastError());
    return CTO_OK;
}

// Side streams kept between calls (per device; one call at a time holds them, a concurrent call makes and destroys its own): creating and
// destroying four streams and their events cost a call ~1 ms.
struct SideStreams {
    static constexpr int kMax = 5;
    hipStream_t sx[kMax] = {};
    hipEvent_t join[kMax] = {};
    int n = 0;
    bool cached = false;
    static std::mutex& lock() { static std::mutex m; return m; }
    struct Kept { hipStream_t sx[kMax]; hipEvent_t join[kMax]; int n = 0; int device = -1; bool busy = false; };
    static Kept& kept() { static Kept k; return k; }
    // stream i (made on first use)
    int get(int i, hipStream_t* out) {
        if (i >= kMax) return CTO_EINVAL;
        while (n <= i) {
            // at the device's highest priority: the runtime keeps a pool of hardware queues PER PRIORITY and hands a new stream the least used
            // queue of its pool - beside a process's ordinary streams (torch's, the pipeline's) two of these could land on one queue and their
            // classes would run one after the other (bench.py's process: 7.2 ms for the stage the stand-alone tool runs in 5.6); a pool of their own
            // gives the four of them a queue each
            int lo = 0, hi = 0;
            if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
            if (hipStreamCreateWithPriority(&sx[n], hipStreamNonBlocking, hi) != hipSuccess) return CTO_EHIP;
            if (hipEventCreateWithFlags(&join[n], hipEventDisableTiming) != hipSuccess) { (void)hipStreamDestroy(sx[n]); return CTO_EHIP; }
            ++n;
        }
        *out = sx[i];
        return CTO_OK;
    }
    SideStreams() {
        int dev = -1;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> g(lock());
        Kept& k = kept();
        if (!k.busy && (k.device < 0 || k.device == dev)) {
            k.busy = true; k.device = dev; cached = true;
            n = k.n;
            for (int i = 0; i < n; ++i) { sx[i] = k.sx[i]; join[i] = k.join[i]; }
        }
    }
    ~SideStreams() {
        for (int i = 0; i < n; ++i) (void)hipStreamSynchronize(sx[i]);
        if (cached) {
            std::lock_guard<std::mutex> g(lock());
            Kept& k = kept();
            k.n = n;
            for (int i = 0; i < n; ++i) { k.sx[i] = sx[i]; k.join[i] = join[i]; }
            k.busy = false;
        } else {
            for (int i = 0; i < n; ++i) { (void)hipStreamDestroy(sx[i]); (void)hipEventDestroy(join[i]); }
        }
    }
};

// Both passes of every alignment of `desc` (operands = base codes in `pool`): the end points, in desc order
template <class PoolVec>
int sw_ends_pool(const PoolVec& pool, const std::vector<SwDesc>& desc, hipStream_t s, cto_realign_stats* st, std::vector<Ends>& ends,
                 DevBuf<signed char>* keep_pool = nullptr) {
    StageClock clk;
    const int n = int(desc.size());
    ends.assign(static_cast<size_t>(n), Ends{0, 0, 0, 0, 0, 16});
    if (n == 0) return CTO_OK;
    // Classes by query length: a launch's LDS footprint is sized by its longest query (H, E and the profile are per stripe position),
    // and the footprint is what bounds the wavefronts a CU holds - one class for everything would run the 100-base reads at the
    // occupancy of the haplotype-length queries.  Inside a class by descending work, so that the rows of a wavefront - and the
    // waves of a round - run for about as long.
    constexpr int kClasses = 11;
    // (CTO_SW_FINE_CLASSES: eleven classes in steps of 1.5 - smaller LDS footprints, more launches: measured slower, 17-19.5 ms against 15.3)
    static const bool fine = std::getenv("CTO_SW_FINE_CLASSES") != nullptr;
    const int qcap_fine[kClasses] = {64, 96, 128, 192, 256, 384, FP_RMAX, 768, 1024, 1536, 0x7fffffff};
    const int qcap_coarse[kClasses] = {64, 128, 256, FP_RMAX, 0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    const int* qcap = fine ? qcap_fine : qcap_coarse;
    std::vector<int> cls[kClasses];
    int Rc[kClasses] = {}, Qc[kClasses] = {};
    long long cells = 0;
    for (int k = 0; k < n; ++k) {
        cells += (long long)desc[k].R * desc[k].Q;
        int c = 0;
        while (desc[k].Q > qcap[c]) ++c;
        cls[c].push_back(k); Rc[c] = std::max(Rc[c], desc[k].R); Qc[c] = std::max(Qc[c], desc[k].Q);
    }
    // inside a class by descending work: one 64-bit key per alignment ((2^40 - 1 - work) << 24 | index), sorted as numbers
    std::vector<int> order;
    order.reserve(static_cast<size_t>(n));
    size_t at[kClasses + 1] = {};
    CTO_REQUIRE(n < (1 << 24), CTO_EUNSUPPORTED, "cto_realign_windows: more than 16 M alignments in one call; split it");
    parallel_for(size_t(kClasses), kClasses, [&](size_t c) {           // the classes' sorts side by side
        std::vector<unsigned long long> keys;
        keys.reserve(cls[c].size());
        for (int k : cls[c]) keys.push_back(((((1ull << 40) - 1ull) - (unsigned long long)desc[k].Q * (unsigned long long)desc[k].R) << 24) | (unsigned long long)k);
        std::sort(keys.begin(), keys.end());
        for (size_t i = 0; i < keys.size(); ++i) cls[c][i] = int(keys[i] & 0xffffffull);
    });
    for (int c = 0; c < kClasses; ++c) {
        at[c] = order.size();
        order.insert(order.end(), cls[c].begin(), cls[c].end());
    }
    at[kClasses] = order.size();
    if (trace_on()) {
        for (int c = 0; c < kClasses; ++c) {
            const std::vector<int>& v = cls[c];
            if (v.empty()) continue;
            long long w = 0, sr = 0, sq = 0;
            for (int k : v) { w += (long long)desc[k].R * desc[k].Q; sr += desc[k].R; sq += desc[k].Q; }
            std::fprintf(stderr, "[sw class %d] n %zu cells %lld mean R %lld Q %lld; largest %d x %d, median %d x %d, smallest %d x %d\n", c, v.size(), w,
                         sr / (long long)v.size(), sq / (long long)v.size(), desc[v[0]].R, desc[v[0]].Q, desc[v[v.size() / 2]].R, desc[v[v.size() / 2]].Q,
                         desc[v.back()].R, desc[v.back()].Q);
        }
    }
    DevBuf<signed char> d_pool;
    DevBuf<SwDesc> d_desc;
    DevBuf<int> d_order;
    DevBuf<Ends> d_out;
    DevBuf<unsigned char> d_ovf;
    int rc;
    if ((rc = d_pool.put(pool, s)) || (rc = d_desc.put(desc, s)) || (rc = d_order.put(order, s)) || (rc = d_out.alloc(size_t(n))) ||
        (rc = d_ovf.alloc(size_t(n))))
        return rc;
    clk.lap("  SW: classes, alloc + H2D");
    // 8-bit passes, then the 16-bit passes of what overflowed (same slots, same order: a row without overflow leaves at once).  The
    // classes are independent chains of two launches each and every one ends in a tail (the longest alignment of the class), so each
    // runs on a stream of its own, longest queries first.
    hipEvent_t e0, e1, fork;
    SideStreams side;
    CTO_HIP(hipEventCreate(&e0)); CTO_HIP(hipEventCreate(&e1));
    CTO_HIP(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CTO_HIP(hipEventRecord(e0, s));
    CTO_HIP(hipEventRecord(fork, s));
    int made = 0, used = 0;
    rc = CTO_OK;
    // at most FOUR streams, the caller's included: the runtime has four hardware queues for a process's streams, and one more stream shares
    // one - its launches then wait behind another class's two launches instead of running beside them (measured: the two shortest classes
    // started when the longest one's 16-bit launch ended, 3 ms of the stage's 8.5 - and again, late in round 6, with four side streams beside
    // the caller's idle one).  The three longest classes get a side stream each, the others follow one another on the caller's stream, which
    // waits for the side streams after its own launches.
    constexpr int kSide = 3;
    for (int c = kClasses - 1; c >= 0 && rc == CTO_OK; --c) {
        if (cls[c].empty()) continue;
        const bool own = used < kSide;
        hipStream_t t = s;
        if (own) {
            if ((rc = side.get(made, &t)) != CTO_OK) break;
            ++made;
            if (hipStreamWaitEvent(t, fork, 0) != hipSuccess) { rc = CTO_EHIP; break; }
        }
        ++used;
        const int m = int(cls[c].size());
        if ((rc = launch_sw<true>(t, d_pool.p, d_desc.p, d_order.p + at[c], m, d_out.p, d_ovf.p, Rc[c], Qc[c])) ||
            (rc = launch_sw<false>(t, d_pool.p, d_desc.p, d_order.p + at[c], m, d_out.p, d_ovf.p, Rc[c], Qc[c])))
            break;
        if (own && hipEventRecord(side.join[made - 1], t) != hipSuccess) rc = CTO_EHIP;
    }
    for (int i = 0; i < made && rc == CTO_OK; ++i)
        if (hipStreamWaitEvent(s, side.join[i], 0) != hipSuccess) rc = CTO_EHIP;
    auto drop = [&]() {
        for (int i = 0; i < made; ++i) (void)hipStreamSynchronize(side.sx[i]);      // (the streams themselves go back to the cache with `side`)
        (void)hipEventDestroy(fork);
    };
    if (rc != CTO_OK) { drop(); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return rc; }
    CTO_HIP(hipEventRecord(e1, s));
    pinned_vector<Ends> landed(static_cast<size_t>(n));
    CTO_HIP(hipMemcpyAsync(landed.data(), d_out.p, size_t(n) * sizeof(Ends), hipMemcpyDeviceToHost, s));
    CTO_HIP(hipStreamSynchronize(s));
    memcpy(ends.data(), landed.data(), size_t(n) * sizeof(Ends));
    float ms = 0.f;
    CTO_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    drop();
    if (st) { st->sw_ms += ms; st->sw_pairs += n; st->sw_cells += cells; }
    if (keep_pool) keep_pool->swap(d_pool);
    clk.lap("  SW: launches + D2H");
    return CTO_OK;
}

struct SwStage { std::vector<SwDesc> desc; std::vector<size_t> first; DevBuf<signed char> d_pool; };     // what the traceback stage re-uses

int ends_device(std::vector<Window*>& ws, hipStream_t s, int threads, cto_realign_stats* st, SwStage& stage) {
    StageClock clk;
    // code pool: per window the reference, its haplotypes, the reads that need Smith-Waterman - each once
    pinned_vector<signed char> pool;
    std::vector<SwDesc>& desc = stage.desc;
    std::vector<size_t>& first = stage.first;
    desc.clear();
    first.assign(ws.size() + 1, 0);
    // sizes first, so that every window fills its own slice of the pool and of the descriptors (on the workers)
    std::vector<size_t> pool_at(ws.size() + 1, 0);
    for (size_t wi = 0; wi < ws.size(); ++wi) {
        const Window& w = *ws[wi];
        size_t bytes = w.refc.size();
        for (const std::vector<int8_t>& h : w.hapc) bytes += h.size();
        for (int r : w.todo) bytes += w.readc[size_t(r)].size();
        pool_at[wi + 1] = pool_at[wi] + bytes;
        first[wi + 1] = first[wi] + w.sw_pairs().size();
    }
    CTO_REQUIRE(pool_at[ws.size()] < (size_t(1) << 31), CTO_EUNSUPPORTED, "cto_realign_windows: more than 2 GiB of sequence in one call; split it");
    pool.resize(pool_at[ws.size()]);
    desc.resize(first[ws.size()]);
    std::atomic<int> unknown{0};
    parallel_for(ws.size(), threads, [&](size_t wi) {
        const Window& w = *ws[wi];
        size_t at = pool_at[wi];
        auto put = [&](const std::vector<int8_t>& v) { const int off = int(at); if (!v.empty()) memcpy(pool.data() + at, v.data(), v.size()); at += v.size(); return off; };
        const int ref_off = put(w.refc);
        std::vector<int> hap_at(w.hapc.size()), read_at(w.readc.size(), -1);
        for (size_t h = 0; h < w.hapc.size(); ++h) hap_at[h] = put(w.hapc[h]);
        for (int r : w.todo) read_at[size_t(r)] = put(w.readc[size_t(r)]);
        size_t k = first[wi];
        for (const cto_realign::SwPair& p : w.sw_pairs()) {
            // identify the operands by address (the pairs point into refc / hapc / readc)
            int roff = -1, qoff = -1;
            if (p.ref == w.refc.data()) roff = ref_off;
            else for (size_t h = 0; h < w.hapc.size(); ++h) if (p.ref == w.hapc[h].data()) { roff = hap_at[h]; break; }
            for (size_t h = 0; h < w.hapc.size() && qoff < 0; ++h) if (p.query == w.hapc[h].data()) qoff = hap_at[h];
            if (qoff < 0) for (int r : w.todo) if (p.query == w.readc[size_t(r)].data()) { qoff = read_at[size_t(r)]; break; }
            if (roff < 0 || qoff < 0) { unknown = 1; roff = qoff = 0; }
            desc[k++] = SwDesc{roff, p.R, qoff, p.Q};
        }
    });
    CTO_REQUIRE(unknown == 0, CTO_EINVAL, "cto_realign_windows: internal: unknown operand");
    if (desc.empty()) return CTO_OK;
    clk.lap("  SW: pool + descriptors");
    std::vector<Ends> ends;
    const int rc = sw_ends_pool(pool, desc, s, st, ends, &stage.d_pool);
    if (rc != CTO_OK) return rc;
    parallel_for(ws.size(), threads, [&](size_t wi) { ws[wi]->set_ends(ends.data() + first[wi]); });
    return CTO_OK;
}


// ------------------------------------------------------------------------------------------------------------------------
// Banded traceback (ssw.c:531-741 as restated in realign.cpp's banded_path, which is what the reference's compiled code is held to):
// one wavefront per alignment.  A row of the band is computed by the lanes side by side - E and the diagonal term of a cell depend on
// the previous row only, and the F chain along the row, f[j] = max(h[j-1] - open, f[j-1] - ext), is a max-plus scan because an h that
// owes its value to f cannot win that max (f - open < f - ext): f[j] = max over k < j of (h'[k] - open - ext * (j-1-k)) with h' = the cell
// without its F term.  The three rolling arrays (previous H, E, current H) live in LDS with the reference's own indices, zeroed edge cells
// included; direction choices (one byte per cell: E's, F's, H's) go to a scratch region in HBM, rows the band does not reach written
// as 0 = "not a cell" as the host's zero-filled array has them.  The band doubles until the best cell reaches the score the striped passes
// found; lane 0 then walks back from the last cell.  Runs leave in walk order (the host reverses them).
constexpr int TB_RUNS = 29;          // runs an alignment may have here (2 x gaps + 1): 128 bytes of result per alignment; more: the host's
struct TbDesc { int ref_off, q_off, R, Q, score, band, band_cap, pad; long long dir_off; };
struct TbOut { int status, n_runs, band; int runs[TB_RUNS]; };          // status 1 = done, 0 = the reference's traceback fails, 2 = not done here
constexpr int TB_NEG = -(1 << 28);

// wavefront-wide steps of a traceback row as DPP operations (a few clocks each; as ds_bpermute shuffles they were a chain of ~100-clock LDS
// crossbar trips, and a row of k_banded is nothing but that chain: 2.8 of the stage's 3.2 ms)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_keep(int identity, int v) { return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xf, false); }
// inclusive max-scan over the 64 lanes: row_shr 1, 2, 4, 8 inside the rows of 16, then row_bcast:15 / :31 carry a row's total into the rows above
__device__ __forceinline__ int wave_scan_max(int v, int identity) {
    v = max(v, dpp_keep<0x111, 0xf>(identity, v));
    v = max(v, dpp_keep<0x112, 0xf>(identity, v));
    v = max(v, dpp_keep<0x114, 0xf>(identity, v));
    v = max(v, dpp_keep<0x118, 0xf>(identity, v));
    v = max(v, dpp_keep<0x142, 0xa>(identity, v));          // row_bcast:15 into rows 1 and 3
    v = max(v, dpp_keep<0x143, 0xc>(identity, v));          // row_bcast:31 into rows 2 and 3
    return v;
}
// lane l takes lane l - 1, lane 0 takes `fill` (wave_shr:1)
__device__ __forceinline__ int wave_shr1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }

__global__ __launch_bounds__(64) void k_banded(const signed char* __restrict__ pool, const TbDesc* __restrict__ desc, const int* __restrict__ order,
                                                int n, unsigned char* dirbuf, TbOut* __restrict__ out, int W) {
    extern __shared__ int tb_lds[];
    if (int(blockIdx.x) >= n) return;
    const int k = order[blockIdx.x];
    const int lane = threadIdx.x;
    const TbDesc d = desc[k];
    const signed char* ref = pool + d.ref_off;
    const signed char* read = pool + d.q_off;
    unsigned char* dir = dirbuf + d.dir_off;
    int* hb = tb_lds;
    int* eb = tb_lds + W;
    int* hc = tb_lds + 2 * W;
    const int R = d.R, Q = d.Q;
    for (int i = lane; i < 3 * W; i += 64) tb_lds[i] = 0;
    // The workgroup is ONE wavefront: its LDS operations execute in order, so a row's steps need no s_barrier - and must not have
    // __syncthreads()'s fence, which also drains the row's direction bytes on their way to HBM (a microsecond per row, twice).
    auto lds_sync = [] { __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
    int band = d.band, best = 0, status = 1, width_d = 0;
    for (;;) {
        const int width = band * 2 + 3;
        width_d = band * 2 + 1;
        lds_sync();
        for (int j = 1 + lane; j < width - 1; j += 64) hb[j] = 0;
        // operands one step ahead of their use: the read's base of the next row, the reference bases of the next row's first 64 cells
        // and of the next chunk of this row - a row is a chain of LDS steps, and a load from HBM in it costs more than the rest
        int ri_next = Q > 0 ? read[0] : 0;
        int rj_row = (lane <= min(R - 1, band)) ? ref[lane] : 0;          // row 0: x = 0, cells 0 .. min(R - 1, band)
        for (int i = 0; i < Q; ++i) {
            const int x = max(0, i - band), dx = x - max(0, i - 1 - band);
            const int end = min(R - 1, i + band), nc = end - x + 1;             // cells of this row: j = x .. end
            const int edge = min(end + 1, width - 1);
            lds_sync();
            if (lane == 0) { hb[0] = 0; eb[0] = 0; hb[edge] = 0; eb[edge] = 0; hc[0] = 0; }
            lds_sync();
            const int ri = ri_next;
            int rj_cur = rj_row;
            if (i + 1 < Q) {
                ri_next = read[i + 1];
                const int xn = max(0, i + 1 - band), endn = min(R - 1, i + 1 + band);
                rj_row = (xn + lane <= endn) ? ref[xn + lane] : 0;
            }
            int carry = kGapE * x - 4;                       // the F chain that enters the row: f = 0 before the first cell, h[-1] = 0
            int g_prev = -kGapO, f_prev = 0;
            unsigned char* line = dir + size_t(i) * size_t(width_d);
            for (int c0 = 0; c0 < width_d; c0 += 64) {
                const int jj = c0 + lane;
                const bool valid = jj < nc;
                const int j = x + jj, u = jj + 1, up = u + dx;
                const int hbu = valid ? hb[up] : 0, ebu = valid ? eb[up] : 0, hbd = valid ? hb[up - 1] : 0;
                const int t1 = i == 0 ? -kGapO : hbu - kGapO, t2 = i == 0 ? -kGapE : ebu - kGapE;
                const int e = max(t1, t2);
                const bool de3 = t1 > t2;
                const int rj = valid ? rj_cur : 0;
                if (c0 + 64 < width_d) rj_cur = (jj + 64 < nc) ? ref[j + 64] : 0;          // the next chunk's
                const int sc = (rj == ri && rj < 4) ? 4 : -6;
                const int t2h = hbd + sc, e1 = max(e, 0), hp = max(e1, t2h), g = hp - kGapO;
                int incl = valid ? g + kGapE * j : TB_NEG;
                const int span = min(64, width_d - c0);            // lanes of this chunk that can hold a cell: most bands are a few cells wide
                incl = wave_scan_max(incl, TB_NEG);                // (lanes behind the row's cells hold TB_NEG: the scan may run over all 64)
                const int excl = wave_shr1(incl, TB_NEG);
                const int f = max(carry, excl) - kGapE * (j - 1);
                const int gp = wave_shr1(g, g_prev), fp = wave_shr1(f, f_prev);
                const bool df5 = gp > fp - kGapE;
                const int f1 = max(f, 0), tt1 = max(e1, f1), h = max(tt1, t2h);
                const int dh = tt1 <= t2h ? 1 : (e1 > f1 ? (de3 ? 3 : 2) : (df5 ? 5 : 4));
                if (valid) { eb[u] = e; hc[u] = h; best = max(best, h); }
                if (jj < width_d) line[jj] = valid ? static_cast<unsigned char>(int(de3) | (int(df5) << 1) | (dh << 2)) : static_cast<unsigned char>(0);
                carry = max(carry, __builtin_amdgcn_readlane(incl, __builtin_amdgcn_readfirstlane(span - 1)));
                g_prev = __builtin_amdgcn_readlane(g, 63); f_prev = __builtin_amdgcn_readlane(f, 63);
            }
            { int* t = hb; hb = hc; hc = t; }              // the row just written is the next row's "row above" (no copy: the two cells
                                                           // a row reads outside what the last one wrote - [0] and [edge] - it zeroes first)
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o));
        if (best >= d.score) break;
        band *= 2;
        if (band > d.band_cap) { status = 2; break; }
    }
    __threadfence();
    __syncthreads();
    if (lane != 0) return;
    TbOut& o = out[k];
    int nr = 0;
    if (status == 1) {
        int i = Q - 1, j = R - 1, state = 2, count = 0, op = cto_realign::kTraceM, prev = cto_realign::kTraceM;
        const volatile unsigned char* vdir = dir;
        while (i > 0) {
            const int x = i - band > 0 ? i - band : 0;
            if (j < x || j - x >= width_d) { status = 0; break; }
            const int b = vdir[size_t(i) * size_t(width_d) + size_t(j - x)];
            if (b == 0) { status = 0; break; }
            const int dd = state == 0 ? 2 + (b & 1) : (state == 1 ? 4 + ((b >> 1) & 1) : (b >> 2));
            if (dd == 1) { --i; --j; state = 2; op = cto_realign::kTraceM; }
            else if (dd == 2) { --i; state = 0; op = cto_realign::kTraceI; }
            else if (dd == 3) { --i; state = 2; op = cto_realign::kTraceI; }
            else if (dd == 4) { --j; state = 1; op = cto_realign::kTraceD; }
            else { --j; state = 2; op = cto_realign::kTraceD; }
            if (op == prev) ++count;
            else {
                if (nr >= TB_RUNS) { status = 2; break; }
                o.runs[nr++] = (count << 2) | prev;
                prev = op; count = 1;
            }
        }
        if (status == 1) {
            if (nr + 2 > TB_RUNS) status = 2;
            else if (op == cto_realign::kTraceM) o.runs[nr++] = ((count + 1) << 2) | op;
            else { o.runs[nr++] = (count << 2) | op; o.runs[nr++] = (1 << 2) | cto_realign::kTraceM; }
        }
    }
    o.status = status; o.n_runs = nr; o.band = band;
}

// grow-only direction scratch kept by the library between calls
struct DirScratch {
    static std::mutex& lock() { static std::mutex m; return m; }
    static unsigned char*& kept() { static unsigned char* p = nullptr; return p; }
    static size_t& kept_cap() { static size_t c = 0; return c; }
    static bool& busy() { static bool b = false; return b; }
    static int& kept_device() { static int d = -1; return d; }
    unsigned char* p = nullptr;
    bool from_kept = false;
    bool get(size_t bytes) {
        bytes = std::max<size_t>(bytes, 1);
        int dev = -1;
        (void)hipGetDevice(&dev);
        {
            // the kept buffer belongs to the device of the first call (as the kept arena does): a call on another device allocates its own
            std::lock_guard<std::mutex> g(lock());
            if (!busy() && (kept_device() < 0 || kept_device() == dev)) {
                kept_device() = dev;
                if (kept_cap() < bytes) {
                    if (kept()) (void)hipFree(kept());
                    kept() = nullptr; kept_cap() = 0;
                    const size_t want = bytes + bytes / 4;
                    if (hipMalloc(reinterpret_cast<void**>(&kept()), want) == hipSuccess) kept_cap() = want;
                    else { (void)hipGetLastError(); kept() = nullptr; }
                }
                if (kept()) { busy() = true; from_kept = true; p = kept(); return true; }
            }
        }
        if (hipMalloc(reinterpret_cast<void**>(&p), bytes) != hipSuccess) { (void)hipGetLastError(); p = nullptr; return false; }
        return true;
    }
    ~DirScratch() {
        if (from_kept) { std::lock_guard<std::mutex> g(lock()); busy() = false; }
        else if (p) (void)hipFree(p);
    }
};

constexpr int kBandMax = 1024;                          // bands beyond are the host's
constexpr size_t kDirMax = size_t(6) << 30;             // direction scratch of one call

// band cap and scratch offset of a traceback; false = not taken (the caller leaves it to the host)
bool tb_place(const cto_realign::TraceJob& j, int ref_off, int q_off, size_t& dir_bytes, TbDesc& d) {
    if (j.band > kBandMax) return false;
    const int cap = std::min(kBandMax, j.band * 4);
    const size_t need = (size_t(j.subQ) * size_t(2 * cap + 1) + 63) & ~size_t(63);
    if (dir_bytes + need > kDirMax) return false;
    d = TbDesc{ref_off + j.ref_begin, q_off + j.read_begin, j.subR, j.subQ, j.score, j.band, cap, 0, (long long)dir_bytes};
    dir_bytes += need;
    return true;
}

// the tracebacks of `desc` (operands = base codes in the device pool): out[k] in desc order
int traceback_pool(const signed char* d_pool, const std::vector<TbDesc>& desc, size_t dir_bytes, hipStream_t s, cto_realign_stats* st, pinned_vector<TbOut>& out) {
    const int n = int(desc.size());
    out.clear();
    if (n == 0) return CTO_OK;
    // two classes by the widest band an alignment may reach (the LDS footprint), longest first inside a class
    // (one 64-bit key per alignment, sorted as numbers: class bit | (2^38 - 1 - work) << 24 | index - a comparator that looks both
    // descriptors up took 2 ms for the bench batch's 23 000)
    CTO_REQUIRE(n < (1 << 24), CTO_EUNSUPPORTED, "cto_realign_windows: more than 16 M tracebacks in one call; split it");
    auto wide = [&](int k) { return desc[size_t(k)].band_cap > 62; };
    std::vector<unsigned long long> keys(static_cast<size_t>(n));
    for (int k = 0; k < n; ++k) {
        const unsigned long long work = std::min<unsigned long long>((unsigned long long)desc[size_t(k)].Q * (unsigned long long)desc[size_t(k)].band, (1ull << 38) - 1ull);
        keys[size_t(k)] = (wide(k) ? 0ull : 1ull << 63) | ((((1ull << 38) - 1ull) - work) << 24) | (unsigned long long)k;
    }
    std::sort(keys.begin(), keys.end());
    std::vector<int> order(static_cast<size_t>(n));
    for (int k = 0; k < n; ++k) order[size_t(k)] = int(keys[size_t(k)] & 0xffffffull);
    int n_wide = 0, cap_wide = 0;
    for (int k = 0; k < n; ++k) if (wide(k)) { ++n_wide; cap_wide = std::max(cap_wide, desc[size_t(k)].band_cap); }
    DevBuf<TbDesc> d_desc;
    DevBuf<int> d_order;
    DevBuf<TbOut> d_out;
    int rc;
    // the direction scratch is the one large allocation of a call (~40 KB per haplotype traceback): kept between calls (one call at a
    // time uses the kept one, a concurrent call allocates its own); without it every traceback is the host's
    DirScratch dir_scratch;
    if (!dir_scratch.get(dir_bytes)) {
        out.assign(static_cast<size_t>(n), TbOut{});
        for (TbOut& o : out) o.status = 2;
        return CTO_OK;
    }
    struct { unsigned char* p; } d_dir{dir_scratch.p};
    if (trace_on()) std::fprintf(stderr, "[realign]   traceback: %d alignments, %.1f MB of direction scratch\n", n, double(dir_bytes) / 1e6);
    if ((rc = d_desc.put(desc, s)) || (rc = d_order.put(order, s)) || (rc = d_out.alloc(size_t(n)))) return rc;
    hipEvent_t e0, e1;
    CTO_HIP(hipEventCreate(&e0)); CTO_HIP(hipEventCreate(&e1));
    CTO_HIP(hipEventRecord(e0, s));
    // the few wide-band alignments are a tail of their own: on a second stream beside the many narrow ones
    hipStream_t s2 = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    SideStreams side;                                  // (declared before the launches: its destructor waits for the side stream)
    if (n_wide > 0 && n_wide < n) {
        if ((rc = side.get(0, &s2)) != CTO_OK) return rc;
        join = side.join[0];
        CTO_HIP(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        CTO_HIP(hipEventRecord(fork, s)); CTO_HIP(hipStreamWaitEvent(s2, fork, 0));
    }
    hipStream_t s_wide = s2 ? s2 : s;
    auto launch_on = [&](hipStream_t t, int first, int count, int cap) -> int {
        if (count == 0) return CTO_OK;
        const int W = 2 * cap + 4;
        hipLaunchKernelGGL(k_banded, dim3(unsigned(count)), dim3(64), size_t(3) * W * sizeof(int), t, d_pool, d_desc.p, d_order.p + first, count,
                           d_dir.p, d_out.p, W);
        CTO_HIP(hipGetLastError());
        return CTO_OK;
    };
    rc = launch_on(s_wide, 0, n_wide, cap_wide);
    if (rc == CTO_OK) rc = launch_on(s, n_wide, n - n_wide, 62);
    if (s2) {
        if (rc == CTO_OK && (hipEventRecord(join, s2) != hipSuccess || hipStreamWaitEvent(s, join, 0) != hipSuccess)) rc = CTO_EHIP;
        if (rc != CTO_OK) (void)hipStreamSynchronize(s2);
    }
    auto drop = [&]() {
        if (s2) { (void)hipStreamSynchronize(s2); (void)hipEventDestroy(fork); s2 = nullptr; }
    };
    if (rc != CTO_OK) { drop(); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return rc; }
    CTO_HIP(hipEventRecord(e1, s));
    out.resize(static_cast<size_t>(n));
    CTO_HIP(hipMemcpyAsync(out.data(), d_out.p, size_t(n) * sizeof(TbOut), hipMemcpyDeviceToHost, s));
    CTO_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    CTO_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    drop();
    if (st) { st->traceback_ms += ms; st->tracebacks += n; }
    return CTO_OK;
}

// the tracebacks the windows are going to ask for (Window::plan_tracebacks), on the device; what the device declines stays for finish()
int traceback_device(std::vector<Window*>& ws, const SwStage& stage, hipStream_t s, int threads, cto_realign_stats* st) {
    StageClock clk;
    std::vector<std::vector<cto_realign::TraceJob>> jobs(ws.size());
    parallel_for(ws.size(), threads, [&](size_t i) { ws[i]->plan_tracebacks(jobs[i]); });
    std::vector<TbDesc> desc;
    std::vector<std::pair<int, int>> who;                   // (window, job)
    size_t dir_bytes = 0;
    for (size_t wi = 0; wi < ws.size(); ++wi)
        for (size_t ji = 0; ji < jobs[wi].size(); ++ji) {
            const cto_realign::TraceJob& j = jobs[wi][ji];
            const SwDesc& p = stage.desc[stage.first[wi] + size_t(j.pair)];
            TbDesc d;
            if (!tb_place(j, p.ref_off, p.q_off, dir_bytes, d)) continue;       // finish() runs it on the host
            desc.push_back(d);
            who.emplace_back(int(wi), int(ji));
        }
    const int n = int(desc.size());
    if (n == 0) return CTO_OK;
    clk.lap("  traceback: plan + descriptors");
    pinned_vector<TbOut> out;
    const int rc = traceback_pool(stage.d_pool.p, desc, dir_bytes, s, st, out);
    if (rc != CTO_OK) return rc;
    clk.lap("  traceback: launches + D2H");
    // a window's results are installed by one worker (set_traced appends to the window's own vectors)
    std::vector<size_t> wfirst(ws.size() + 1, size_t(n));
    for (int k = n - 1; k >= 0; --k) wfirst[size_t(who[size_t(k)].first)] = size_t(k);
    for (size_t wi = ws.size(); wi-- > 0;) if (wfirst[wi] == size_t(n)) wfirst[wi] = wfirst[wi + 1];
    long long declined = 0;
    for (int k = 0; k < n; ++k) declined += out[size_t(k)].status == 2;
    if (st) st->tracebacks_declined += declined;
    parallel_for(ws.size(), threads, [&](size_t wi) {
        for (size_t k = wfirst[wi]; k < wfirst[wi + 1]; ++k) {
            const TbOut& o = out[k];
            if (o.status == 2) continue;
            int32_t runs[TB_RUNS];
            for (int i = 0; i < o.n_runs; ++i) runs[i] = o.runs[o.n_runs - 1 - i];
            ws[wi]->set_traced(jobs[wi][size_t(who[k].second)], o.status == 1, runs, o.n_runs);
        }
    });
    clk.lap("  traceback: CIGARs");
    return CTO_OK;
}

}  // namespace

extern "C" int cto_sw_ends_batch(int n, const int8_t* codes, size_t n_codes, const int32_t* desc, int where, int host_threads, void* stream,
                                 int32_t* out) try {
    CTO_REQUIRE(n >= 0 && (n == 0 || (codes && desc && out)) && (where == CTO_REALIGN_HOST || where == CTO_REALIGN_DEVICE), CTO_EINVAL,
                "cto_sw_ends_batch: bad argument");
    CTO_REQUIRE(n_codes < (size_t(1) << 31), CTO_EUNSUPPORTED, "cto_sw_ends_batch: more than 2 GiB of sequence in one call; split it");
    std::vector<SwDesc> d(static_cast<size_t>(n));
    for (int k = 0; k < n; ++k) {
        d[size_t(k)] = SwDesc{desc[4 * k], desc[4 * k + 1], desc[4 * k + 2], desc[4 * k + 3]};
        const SwDesc& x = d[size_t(k)];
        CTO_REQUIRE(x.R >= 0 && x.Q >= 0 && x.ref_off >= 0 && x.q_off >= 0 && size_t(x.ref_off) + size_t(x.R) <= n_codes &&
                    size_t(x.q_off) + size_t(x.Q) <= n_codes, CTO_EINVAL, "cto_sw_ends_batch: alignment %d reaches outside the codes", k);
    }
    std::vector<Ends> ends(static_cast<size_t>(n), Ends{0, 0, 0, 0, 0, 16});
    if (where == CTO_REALIGN_DEVICE) {
        ArenaLease lease;
        const std::vector<signed char> pool(reinterpret_cast<const signed char*>(codes), reinterpret_cast<const signed char*>(codes) + n_codes);
        const int rc = sw_ends_pool(pool, d, static_cast<hipStream_t>(stream), nullptr, ends);
        if (rc != CTO_OK) return rc;
    } else {
        const int threads = host_threads > 0 ? host_threads : cto_realign::get_threads();
        parallel_for(size_t(n), threads, [&](size_t k) { ends[k] = cto_realign::ends_of_pair(codes + d[k].ref_off, d[k].R, codes + d[k].q_off, d[k].Q); });
    }
    for (int k = 0; k < n; ++k) {
        const Ends& e = ends[size_t(k)];
        int32_t* o = out + 6 * size_t(k);
        o[0] = e.score; o[1] = e.ref_end; o[2] = e.read_end; o[3] = e.ref_begin; o[4] = e.bw_read_end; o[5] = e.lanes;
    }
    return CTO_OK;
}
CTO_CATCH("cto_sw_ends_batch", int)

extern "C" int cto_ssw_align_batch(int n, const int8_t* codes, size_t n_codes, const int32_t* desc, int where, int host_threads, void* stream,
                                  int32_t* score, int32_t* ref_begin, char* cigar_buf, size_t cigar_cap, int64_t* cigar_off) try {
    CTO_REQUIRE(n >= 0 && (n == 0 || (codes && desc && score && ref_begin)) && cigar_off && (cigar_buf || cigar_cap == 0) &&
                (where == CTO_REALIGN_HOST || where == CTO_REALIGN_DEVICE), CTO_EINVAL, "cto_ssw_align_batch: bad argument");
    CTO_REQUIRE(n_codes < (size_t(1) << 31), CTO_EUNSUPPORTED, "cto_ssw_align_batch: more than 2 GiB of sequence in one call; split it");
    std::vector<SwDesc> d(static_cast<size_t>(n));
    for (int k = 0; k < n; ++k) {
        d[size_t(k)] = SwDesc{desc[4 * k], desc[4 * k + 1], desc[4 * k + 2], desc[4 * k + 3]};
        const SwDesc& x = d[size_t(k)];
        CTO_REQUIRE(x.R >= 0 && x.Q >= 0 && x.ref_off >= 0 && x.q_off >= 0 && size_t(x.ref_off) + size_t(x.R) <= n_codes &&
                    size_t(x.q_off) + size_t(x.Q) <= n_codes, CTO_EINVAL, "cto_ssw_align_batch: alignment %d reaches outside the codes", k);
    }
    const int threads = host_threads > 0 ? host_threads : cto_realign::get_threads();
    std::vector<cto_realign::SwAlignment> al(static_cast<size_t>(n));
    auto on_host = [&](size_t k, const Ends& e) { al[k] = cto_realign::alignment_of_pair(codes + d[k].ref_off, d[k].R, codes + d[k].q_off, d[k].Q, e); };
    if (where == CTO_REALIGN_DEVICE && n > 0) {
        ArenaLease lease;
        hipStream_t s = static_cast<hipStream_t>(stream);
        const std::vector<signed char> pool(reinterpret_cast<const signed char*>(codes), reinterpret_cast<const signed char*>(codes) + n_codes);
        std::vector<Ends> ends;
        DevBuf<signed char> d_pool;
        int rc = sw_ends_pool(pool, d, s, nullptr, ends, &d_pool);
        if (rc != CTO_OK) return rc;
        std::vector<cto_realign::TraceJob> jobs(static_cast<size_t>(n));
        std::vector<int> at(static_cast<size_t>(n), -1);                  // alignment -> traceback, -1 = none on the device
        std::vector<char> planned(static_cast<size_t>(n), 0);
        std::vector<TbDesc> tb;
        size_t dir_bytes = 0;
        for (int k = 0; k < n; ++k) {
            jobs[size_t(k)].pair = k;
            planned[size_t(k)] = cto_realign::plan_pair(d[size_t(k)].R, d[size_t(k)].Q, ends[size_t(k)], jobs[size_t(k)]) ? 1 : 0;
            TbDesc t;
            if (planned[size_t(k)] && tb_place(jobs[size_t(k)], d[size_t(k)].ref_off, d[size_t(k)].q_off, dir_bytes, t)) { at[size_t(k)] = int(tb.size()); tb.push_back(t); }
        }
        pinned_vector<TbOut> out;
        rc = traceback_pool(d_pool.p, tb, dir_bytes, s, nullptr, out);
        if (rc != CTO_OK) return rc;
        parallel_for(size_t(n), threads, [&](size_t k) {
            if (!planned[k]) return;                                     // no alignment: score 0, empty CIGAR
            if (at[k] < 0 || out[size_t(at[k])].status == 2) { on_host(k, ends[k]); return; }
            const TbOut& o = out[size_t(at[k])];
            if (o.status != 1) return;                                   // the reference's traceback fails
            int32_t runs[TB_RUNS];
            for (int i = 0; i < o.n_runs; ++i) runs[i] = o.runs[o.n_runs - 1 - i];
            al[k] = cto_realign::alignment_from_device_runs(codes + d[k].ref_off, codes + d[k].q_off, d[k].Q, ends[k], jobs[k], runs, o.n_runs);
        });
    } else {
        parallel_for(size_t(n), threads, [&](size_t k) { on_host(k, cto_realign::ends_of_pair(codes + d[k].ref_off, d[k].R, codes + d[k].q_off, d[k].Q)); });
    }
    std::vector<std::string> text(static_cast<size_t>(n));
    for (int k = 0; k < n; ++k) {
        score[k] = al[size_t(k)].score;
        ref_begin[k] = al[size_t(k)].ref_begin;
        for (const cto_realign::Op& o : al[size_t(k)].cigar) { text[size_t(k)] += std::to_string(o.len); text[size_t(k)] += o.op; }
    }
    return cto_realign_write_cigars(text, cigar_buf, cigar_cap, cigar_off);
}
CTO_CATCH("cto_ssw_align_batch", int)

extern "C" int cto_realign_windows(int n_jobs, cto_realign_job* jobs, int where, int host_threads, void* stream, cto_realign_stats* stats) try {
    CTO_REQUIRE(n_jobs >= 0 && (jobs || n_jobs == 0) && (where == CTO_REALIGN_HOST || where == CTO_REALIGN_DEVICE), CTO_EINVAL,
                "cto_realign_windows: bad argument");
    if (stats) memset(stats, 0, sizeof(*stats));
    const double t0 = now_ms();
    const int threads = host_threads > 0 ? host_threads : cto_realign::get_threads();
    std::vector<Window> ws(static_cast<size_t>(n_jobs));
    std::vector<int> status(static_cast<size_t>(n_jobs), CTO_OK);
    std::vector<std::string> errors(static_cast<size_t>(n_jobs));
    parallel_for(size_t(n_jobs), threads, [&](size_t i) {
        cto_realign_job& j = jobs[i];
        if (!(j.out_positions && j.cigar_off && (j.cigar_buf || j.cigar_cap == 0))) { status[i] = CTO_EINVAL; errors[i] = "cto_realign_windows: a job without output buffers"; return; }
        std::vector<const char*> ps, pc;         // the joined forms: n_reads strings back to back
        auto split = [&](const char* joined, std::vector<const char*>& v) {
            v.resize(static_cast<size_t>(std::max(j.n_reads, 0)));
            for (int r = 0; r < j.n_reads; ++r) { v[size_t(r)] = joined; joined += strlen(joined) + 1; }
            return v.data();
        };
        const char* const* seqs = j.seqs ? j.seqs : (j.seqs_joined ? split(j.seqs_joined, ps) : nullptr);
        const char* const* cigars = j.cigars ? j.cigars : (j.cigars_joined ? split(j.cigars_joined, pc) : nullptr);
        status[i] = ws[i].init(j.n_reads, seqs, j.positions, cigars, j.reference, j.haplotypes, j.ref_start, j.ref_prefix, j.ref_suffix);
        if (status[i] != CTO_OK) errors[i] = cto_last_error();
    });
    StageClock clk;
    clk.t = t0;
    clk.lap("windows: parse, de Bruijn");
    std::vector<Window*> dev, host;
    for (size_t i = 0; i < ws.size(); ++i) {
        if (status[i] != CTO_OK) continue;
        if (where == CTO_REALIGN_DEVICE && device_eligible(ws[i])) dev.push_back(&ws[i]);
        else host.push_back(&ws[i]);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!dev.empty()) {
        ArenaLease lease;                                    // (declared before the stage objects: they go first)
        int rc = fast_pass_device(dev, s, threads, stats);
        if (rc != CTO_OK) return rc;
        clk.lap("fast pass (device, copies)");
        parallel_for(dev.size(), threads, [&](size_t i) { dev[i]->collect_pairs(); });
        clk.lap("collect pairs");
        SwStage stage;
        rc = ends_device(dev, s, threads, stats, stage);
        if (rc != CTO_OK) return rc;
        clk.lap("SW ends (device, copies)");
        static const bool host_traceback = std::getenv("CTO_REALIGN_HOST_TRACEBACK") != nullptr;      // A/B switch: every traceback in finish()
        if (!host_traceback) {
            rc = traceback_device(dev, stage, s, threads, stats);
            if (rc != CTO_OK) return rc;
            clk.lap("tracebacks (device, copies)");
        }
    }
    const double t1 = now_ms();
    // CTO_REALIGN_PLAN_HOST=1 (tests): the host windows' tracebacks go the way the device stage's do - planned, run, installed - before finish()
    const bool plan_host = std::getenv("CTO_REALIGN_PLAN_HOST") != nullptr;
    parallel_for(host.size(), threads, [&](size_t i) {
        host[i]->fast_pass_host(); host[i]->collect_pairs(); host[i]->ends_host();
        if (!plan_host) return;
        std::vector<cto_realign::TraceJob> jobs;
        host[i]->plan_tracebacks(jobs);
        std::vector<int32_t> runs;
        for (const cto_realign::TraceJob& j : jobs) {
            const bool ok = cto_realign::trace_runs_host(*host[i], j, runs);
            host[i]->set_traced(j, ok, runs.data(), int(runs.size()));
        }
    });
    parallel_for(ws.size(), threads, [&](size_t i) {
        if (status[i] != CTO_OK) return;
        cto_realign_job& j = jobs[i];
        std::vector<std::string> out;
        int rc = ws[i].finish(j.out_positions, out);
        if (rc == CTO_OK) rc = cto_realign_write_cigars(out, j.cigar_buf, j.cigar_cap, j.cigar_off);
        if (rc != CTO_OK) { status[i] = rc; errors[i] = cto_last_error(); }
    });
    clk.lap("host windows, traceback, CIGARs");
    int first_bad = -1;
    for (size_t i = 0; i < ws.size(); ++i) {
        jobs[i].status = status[i];
        if (status[i] != CTO_OK && first_bad < 0) first_bad = int(i);
    }
    if (stats) {
        stats->windows = n_jobs;
        stats->host_windows = (long long)host.size();
        for (const Window& w : ws) { stats->reads += w.n_reads(); stats->haplotypes += w.n_haps(); }
        stats->device_stage_ms = t1 - t0;
    }
    // A window owns thousands of small vectors (a hit per read and haplotype) and giving ~100 k of them back costs as much as the
    // traceback stage (and more when many threads free into each other's arenas): the windows are handed to a thread that does it
    // behind the caller's back; the next call (or the library's unloading) waits for it.
    static const bool inline_teardown = std::getenv("CTO_REALIGN_INLINE_TEARDOWN") != nullptr;      // A/B switch
    if (inline_teardown) parallel_for(ws.size(), threads, [&](size_t i) { ws[i] = Window(); });
    if (stats) stats->host_ms = now_ms() - t1;
    if (!inline_teardown) reap(std::move(ws));
    if (first_bad >= 0) { cto::set_error("window %d: %s", first_bad, errors[size_t(first_bad)].c_str()); return status[size_t(first_bad)]; }
    return CTO_OK;
}
CTO_CATCH("cto_realign_windows", int)
