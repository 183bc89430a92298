// Illumina read realignment (SURVEY.md 8f #4b): the native half of `realign_reads`, written from scratch.
//
// What the reference does (src/realign/realigner.cpp, a descendant of DeepVariant's fast-pass aligner, + the SSW library):
// the reads of one window are aligned to every candidate haplotype - first by an exact k-mer seeded "fast pass" that accepts
// <= 2 mismatches (realigner.cpp:147-229), then, for reads no haplotype took, by Smith-Waterman (:351-384) - the haplotypes are
// aligned to the reference by Smith-Waterman (:315-349), and every read's read->haplotype alignment is composed with its best
// haplotype's haplotype->reference alignment into a new read->reference CIGAR and position (:386-433, :653-778).
//
// This file restates that contract, not its data structures: flat op arrays walked with cursors instead of std::list<CigarOp>
// splicing and regex parsing, one integer cell model for both SSE2 kernels.  Two places have to follow the reference to the letter because their
// *tie-breaking* decides the output bytes:
//   (1) Smith-Waterman.  The reference's scores come from the SSE2 striped kernels of SSW (ssw.c:118-529).  Their "lazy F" loop
//       does not feed corrected H values back into E, so whether an insertion may directly follow a deletion depends on where the
//       stripe boundaries fall (segment length = ceil(query / 16) in 8-bit mode, ceil(query / 8) in 16-bit mode).  `striped_pass`
//       is a scalar model of exactly that recurrence (per-stripe F chains, the two lazy-F loop shapes and their exit tests, the
//       8-bit overflow rule that switches to 16 bit), so end points and begin points agree cell for cell.  The CIGAR comes from the
//       banded traceback of ssw.c:531-741, whose band bookkeeping (one zeroed edge cell per row, doubled band until the score is
//       reached, E before F before diagonal on ties) is reproduced with the same band coordinates.
//   (2) Haplotype order.  The reference std::sort()s haplotypes by score (realigner.cpp:108) and breaks read-score ties by that
//       order (:515-539); the same std::sort over the same keys gives the same permutation.
// Pinned against the reference compiled here (oracle/_ref/librealigner_ref.so, `make -C oracle ref`): tests/test_realign.py and the
// committed windows of tests/golden/realign.json.gz.
//
// Deviations (all unreachable from `reads_realignment`, flagged): a haplotype shorter than the k-mer makes the reference index
// past the string (unsigned wrap at realigner.cpp:157) - here CTO_EINVAL; an alignment of score 0 makes SSW read ref[-1] - here
// "no alignment"; bytes >= 0x80 index the reference's translation table out of bounds - here they are N.
#include <emmintrin.h>
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include "common.h"
#include "realign_internal.h"

namespace {
using cto_realign::Ends;
using cto_realign::Op;
using cto_realign::ReadHit;
using cto_realign::HapState;
using cto_realign::SwPair;
using cto_realign::SwAlignment;
using cto_realign::TraceJob;

// ---- scoring: realigner.cpp:63-73 (set_options) and the default SSW aligner it ends up using (ssw_cpp.cpp:230-242; `InitSswLib`
// at realigner.cpp:121-127 builds a local object, so the member keeps its defaults - which are the same numbers)
constexpr int kMatch = 4, kMismatch = 6, kGapOpen = 8, kGapExt = 2;
constexpr int kKmer = 32;
constexpr int kMaxMismatches = 2;
constexpr int kSswThreshold = 1;      // CalculateSswAlignmentScoreThreshold (:75-85): 4*250*0.16934 - 6*250*(1-0.16934) < 0 -> 1

inline int8_t base_code(char c) {      // ssw_cpp.cpp:37-50 (A C G T U, either case; everything else 4)
    switch (c) {
        case 'A': case 'a': case 'U': case 'u': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}
inline int sub_score(int8_t a, int8_t b) { return (a == b && a < 4) ? kMatch : -kMismatch; }   // ssw_cpp.cpp:52-76

struct Codes {
    std::vector<int8_t> v;
    explicit Codes(const char* s, size_t n) : v(n) { for (size_t i = 0; i < n; ++i) v[i] = base_code(s[i]); }
};

// ------------------------------------------------------------------------------------------------------------------------
// One pass of the striped recurrence (Farrar 2007, as ssw.c:118-311 runs it on 16 unsigned bytes and :341-529 on 8 signed words):
// the query occupies linear positions q = lane * seg + j, vector j holds the 16 (8) lanes of stripe position j.  ref is walked
// forwards (reverse = false) or backwards over [0, ref_len).  What has to be reproduced bit for bit, because the caller's
// CIGARs depend on it: byte arithmetic saturates and carries a bias of |mismatch|, a column whose maximum reaches 255 - bias makes the
// caller repeat the pass on words; E never sees the corrections of the lazy-F loop; the two lazy-F loops differ (the byte one tests
// before it corrects and wraps around the stripes, the word one corrects first and runs at most `lanes` rounds); the end position is
// the first column that RAISES the maximum and, in that column, the smallest linear query position that holds it.
// oracle/ssw_model.cpp states the same recurrence lane by lane in scalar code; tests hold the two and the compiled reference together.
struct PassEnd { int score, ref_end, read_end; bool overflow; };

std::atomic<int> g_threads{[] { const char* e = getenv("CTO_REALIGN_THREADS"); const int n = e ? atoi(e) : 1; return n > 0 ? n : 1; }()};

struct PassScratch { std::vector<__m128i> prof, h0, h1, e, hmax; };
thread_local PassScratch t_scratch;

template <bool BYTE>
PassEnd striped_pass_t(const int8_t* ref, int ref_len, bool reverse, const int8_t* read, int read_len, int terminate) {
    constexpr int lanes = BYTE ? 16 : 8;
    const int seg = (read_len + lanes - 1) / lanes;
    constexpr int bias = kMismatch;                   // ssw_init: |most negative matrix entry|
    PassScratch& S = t_scratch;
    const __m128i zero = _mm_setzero_si128();
    S.prof.resize(size_t(5) * seg);
    S.h0.assign(seg, zero);
    S.h1.assign(seg, zero);
    S.e.assign(seg, zero);
    S.hmax.assign(seg, zero);
    for (int c = 0; c < 5; ++c)                        // query profile: padding rows score 0 (profile = bias / 0)
        for (int j = 0; j < seg; ++j) {
            alignas(16) int8_t b8[16];
            alignas(16) int16_t b16[8];
            for (int l = 0; l < lanes; ++l) {
                const int q = l * seg + j;
                const int sc = q < read_len ? sub_score(int8_t(c), read[q]) : 0;
                if (BYTE) b8[l] = int8_t(sc + bias); else b16[l] = int16_t(sc);
            }
            S.prof[size_t(c) * seg + j] = BYTE ? _mm_load_si128(reinterpret_cast<const __m128i*>(b8)) : _mm_load_si128(reinterpret_cast<const __m128i*>(b16));
        }
    __m128i* store = S.h0.data();
    __m128i* load = S.h1.data();
    __m128i* pe = S.e.data();
    const __m128i gap_o = BYTE ? _mm_set1_epi8(kGapOpen) : _mm_set1_epi16(kGapOpen);
    const __m128i gap_e = BYTE ? _mm_set1_epi8(kGapExt) : _mm_set1_epi16(kGapExt);
    const __m128i vbias = _mm_set1_epi8(bias);
    int best = 0, ref_end = BYTE ? -1 : 0;
    bool overflow = false;
    const int begin = reverse ? ref_len - 1 : 0, end = reverse ? -1 : ref_len, step = reverse ? -1 : 1;
    for (int i = begin; i != end; i += step) {
        const __m128i* prof = S.prof.data() + size_t(ref[i]) * seg;
        __m128i f = zero, colmax_v = zero;
        __m128i h = BYTE ? _mm_slli_si128(store[seg - 1], 1) : _mm_slli_si128(store[seg - 1], 2);   // H(i-1, q-1) of the stripes' first rows
        std::swap(store, load);                       // load = column i-1 (final), store = column i
        int colmax;
        if (BYTE) {
            for (int j = 0; j < seg; ++j) {
                h = _mm_subs_epu8(_mm_adds_epu8(h, prof[j]), vbias);
                __m128i e = pe[j];
                h = _mm_max_epu8(_mm_max_epu8(h, e), f);
                colmax_v = _mm_max_epu8(colmax_v, h);
                store[j] = h;
                h = _mm_subs_epu8(h, gap_o);
                pe[j] = _mm_max_epu8(_mm_subs_epu8(e, gap_e), h);                   // E never sees the lazy-F corrections below
                f = _mm_max_epu8(_mm_subs_epu8(f, gap_e), h);
                h = load[j];
            }
            // lazy F: the F chain that leaves stripe k enters stripe k + 1 (ssw.c:207-241)
            f = _mm_slli_si128(f, 1);
            int j = 0;
            while (_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_subs_epu8(f, _mm_subs_epu8(store[j], gap_o)), zero)) != 0xffff) {
                h = _mm_max_epu8(store[j], f);
                colmax_v = _mm_max_epu8(colmax_v, h);
                store[j] = h;
                f = _mm_subs_epu8(f, gap_e);
                if (++j >= seg) { j = 0; f = _mm_slli_si128(f, 1); }
            }
            __m128i m = _mm_max_epu8(colmax_v, _mm_srli_si128(colmax_v, 8));
            m = _mm_max_epu8(m, _mm_srli_si128(m, 4));
            m = _mm_max_epu8(m, _mm_srli_si128(m, 2));
            m = _mm_max_epu8(m, _mm_srli_si128(m, 1));
            colmax = _mm_cvtsi128_si32(m) & 0xff;
        } else {
            for (int j = 0; j < seg; ++j) {
                h = _mm_adds_epi16(h, prof[j]);
                __m128i e = pe[j];
                h = _mm_max_epi16(_mm_max_epi16(h, e), f);
                colmax_v = _mm_max_epi16(colmax_v, h);
                store[j] = h;
                h = _mm_subs_epu16(h, gap_o);
                pe[j] = _mm_max_epi16(_mm_subs_epu16(e, gap_e), h);
                f = _mm_max_epi16(_mm_subs_epu16(f, gap_e), h);
                h = load[j];
            }
            bool done = false;                        // ssw.c:446-459
            for (int k = 0; k < lanes && !done; ++k) {
                f = _mm_slli_si128(f, 2);
                for (int j = 0; j < seg; ++j) {
                    h = _mm_max_epi16(store[j], f);
                    colmax_v = _mm_max_epi16(colmax_v, h);
                    store[j] = h;
                    h = _mm_subs_epu16(h, gap_o);
                    f = _mm_subs_epu16(f, gap_e);
                    if (!_mm_movemask_epi8(_mm_cmpgt_epi16(f, h))) { done = true; break; }
                }
            }
            __m128i m = _mm_max_epi16(colmax_v, _mm_srli_si128(colmax_v, 8));
            m = _mm_max_epi16(m, _mm_srli_si128(m, 4));
            m = _mm_max_epi16(m, _mm_srli_si128(m, 2));
            colmax = int16_t(_mm_cvtsi128_si32(m) & 0xffff);
        }
        if (colmax > best) {
            best = colmax;
            if (BYTE && best + bias >= 255) { overflow = true; break; }
            ref_end = i;
            memcpy(S.hmax.data(), store, size_t(seg) * sizeof(__m128i));
        }
        if (colmax == terminate) break;
    }
    int read_end = read_len - 1;
    bool found = false;
    for (int l = 0; l < lanes && !found; ++l)
        for (int j = 0; j < seg; ++j) {
            const int v = BYTE ? int(reinterpret_cast<const uint8_t*>(S.hmax.data())[j * 16 + l])
                               : int(reinterpret_cast<const int16_t*>(S.hmax.data())[j * 8 + l]);
            if (v == best) { const int q = l * seg + j; if (q < read_end) read_end = q; found = true; break; }
        }
    return {overflow ? 255 : best, ref_end, read_end, overflow};
}

PassEnd striped_pass(const int8_t* ref, int ref_len, bool reverse, const int8_t* read, int read_len, int lanes, int terminate) {
    return lanes == 16 ? striped_pass_t<true>(ref, ref_len, reverse, read, read_len, terminate)
                       : striped_pass_t<false>(ref, ref_len, reverse, read, read_len, terminate);
}

// ------------------------------------------------------------------------------------------------------------------------
// Banded traceback (ssw.c:531-741).  Returns run-length ops over {'M','I','D'} in query order; empty = the reference would fail.
struct Run { char op; int len; };

bool banded_path(const int8_t* ref, const int8_t* read, int R, int Q, int score, int band, std::vector<Run>& out) {
    static thread_local std::vector<int> hb, eb, hc;            // scratch of the calling worker: no allocation per alignment
    static thread_local std::vector<int8_t> dir;
    hb.clear(); eb.clear(); hc.clear();                         // contents as of fresh vectors, the storage kept
    // ... up to a bound: one wide-band traceback (dir may reach 1 GiB under the guard below) must not stay pinned to this worker thread
    struct Shrink {
        std::vector<int8_t>& d;
        ~Shrink() { if (d.capacity() > (size_t(64) << 20)) std::vector<int8_t>().swap(d); }
    } shrink{dir};
    int best = 0, width_d = 0;
    auto bu = [](int w, int i, int j) { int x = i - w; if (x < 0) x = 0; return j - x + 1; };            // set_u
    auto bd = [](int w, int i, int j, int p) { int x = i - w; if (x < 0) x = 0; return (j - x) * 3 + p; };  // set_d
    for (;;) {
        const int width = band * 2 + 3;
        width_d = band * 2 + 1;
        if ((int)hb.size() < width + 1) { hb.resize(width + 1, 0); eb.resize(width + 1, 0); hc.resize(width + 1, 0); }
        if ((int64_t)width_d * Q * 3 > (int64_t)1 << 30) return false;
        dir.assign((size_t)width_d * Q * 3 + 3, 0);
        for (int j = 1; j < width - 1; ++j) hb[j] = 0;
        for (int i = 0; i < Q; ++i) {
            const int beg = std::max(0, i - band), end = std::min(R - 1, i + band);
            const int edge = std::min(end + 1, width - 1);
            int f = 0, u = 0;
            hb[0] = eb[0] = hb[edge] = eb[edge] = hc[0] = 0;
            int8_t* line = dir.data() + (size_t)width_d * i * 3;
            for (int j = beg; j <= end; ++j) {
                u = bu(band, i, j);
                const int up = bu(band, i - 1, j), left = bu(band, i, j - 1), diag = bu(band, i - 1, j - 1);
                int t1 = i == 0 ? -kGapOpen : hb[up] - kGapOpen;
                int t2 = i == 0 ? -kGapExt : eb[up] - kGapExt;
                eb[u] = std::max(t1, t2);
                const int8_t de = t1 > t2 ? 3 : 2;
                line[bd(band, i, j, 0)] = de;
                t1 = hc[left] - kGapOpen;
                t2 = f - kGapExt;
                f = std::max(t1, t2);
                const int8_t df = t1 > t2 ? 5 : 4;
                line[bd(band, i, j, 1)] = df;
                const int e1 = std::max(eb[u], 0), f1 = std::max(f, 0);
                t1 = std::max(e1, f1);
                t2 = hb[diag] + sub_score(ref[j], read[i]);
                hc[u] = std::max(t1, t2);
                if (hc[u] > best) best = hc[u];
                line[bd(band, i, j, 2)] = t1 <= t2 ? (int8_t)1 : (e1 > f1 ? de : df);
            }
            for (int j = 1; j <= u; ++j) hb[j] = hc[j];
        }
        if (best >= score) break;
        band *= 2;
    }
    // trace back from the last cell in the H state
    int i = Q - 1, j = R - 1, state = 2, count = 0;
    char op = 'M', prev = 'M';
    std::vector<Run> rev;
    while (i > 0) {
        const int x = i - band > 0 ? i - band : 0;
        if (j < x || j - x >= width_d) return false;
        const int8_t d = dir[(size_t)width_d * i * 3 + (j - x) * 3 + state];
        switch (d) {
            case 1: --i; --j; state = 2; op = 'M'; break;
            case 2: --i; state = 0; op = 'I'; break;
            case 3: --i; state = 2; op = 'I'; break;
            case 4: --j; state = 1; op = 'D'; break;
            case 5: --j; state = 2; op = 'D'; break;
            default: return false;
        }
        if (op == prev) ++count;
        else { rev.push_back({prev, count}); prev = op; count = 1; }
    }
    if (op == 'M') rev.push_back({op, count + 1});
    else { rev.push_back({op, count}); rev.push_back({'M', 1}); }
    out.assign(rev.rbegin(), rev.rend());
    return true;
}

// ------------------------------------------------------------------------------------------------------------------------
// ssw_align (ssw.c:781-867) + Aligner::Align / ConvertAlignment / CalculateNumberMismatch (ssw_cpp.cpp:78-215, :302-337):
// local alignment of `query` to `ref`, CIGAR over {S,=,X,I,D}.

// the two striped passes (ssw.c:781-830): where the best local alignment ends and begins
Ends sw_ends(const int8_t* ref, int R, const int8_t* query, int Q) {
    Ends z{0, 0, 0, 0, 0, 16};
    if (R == 0 || Q == 0) return z;
    int lanes = 16;
    PassEnd fw = striped_pass(ref, R, false, query, Q, 16, 255);
    if (fw.overflow) { lanes = 8; fw = striped_pass(ref, R, false, query, Q, 8, 65535); }
    if (fw.score <= 0) return z;
    std::vector<int8_t> rq(query, query + fw.read_end + 1);
    std::reverse(rq.begin(), rq.end());
    const PassEnd bw = striped_pass(ref, fw.ref_end + 1, true, rq.data(), fw.read_end + 1, lanes, fw.score);
    return Ends{fw.score, fw.ref_end, fw.read_end, bw.ref_end, bw.read_end, lanes};
}

// banded traceback between those points and the CIGAR SSW's C++ wrapper prints (ssw.c:831-867, ssw_cpp.cpp:78-215)
// the sub-problem of the traceback; false = the reference returns no alignment before it gets there
bool trace_job(int R, int Q, const Ends& e, TraceJob& j) {
    if (R == 0 || Q == 0 || e.score <= 0) return false;
    j.ref_begin = e.ref_begin; j.read_begin = e.read_end - e.bw_read_end;
    if (j.ref_begin < 0 || j.read_begin < 0) return false;
    j.subR = e.ref_end - j.ref_begin + 1; j.subQ = e.read_end - j.read_begin + 1;
    if (j.subR > 32768 || j.subQ > 32768) return false;                             // distance_filter 32767: no CIGAR
    j.score = e.score;
    j.band = std::abs(j.subR - j.subQ) + 1;
    return true;
}

// the runs of the banded walk ({'M','I','D'} in query order) as Align prints them: soft clips, '=' / 'X' runs
SwAlignment alignment_from_runs(const int8_t* ref, const int8_t* query, int Q, const Ends& e, const TraceJob& j, const std::vector<Run>& runs) {
    SwAlignment al;
    al.score = e.score;
    al.ref_begin = j.ref_begin;
    if (j.read_begin > 0) al.cigar.push_back({'S', j.read_begin});
    const int8_t* r = ref + j.ref_begin;
    const int8_t* q = query + j.read_begin;
    char cur = 0;
    int len = 0;
    auto flush = [&] { if (cur) al.cigar.push_back({cur, len}); cur = 0; len = 0; };
    for (const Run& run : runs) {
        if (run.op == 'M') {
            for (int k = 0; k < run.len; ++k, ++r, ++q) {
                const char c = *r != *q ? 'X' : '=';
                if (c != cur) { flush(); cur = c; }
                ++len;
            }
        } else {
            flush();
            al.cigar.push_back({run.op, run.len});
            if (run.op == 'I') q += run.len; else r += run.len;
        }
    }
    flush();
    const int tail = Q - e.read_end - 1;
    if (tail > 0) al.cigar.push_back({'S', tail});
    return al;
}

SwAlignment sw_finish(const int8_t* ref, int R, const int8_t* query, int Q, const Ends& e) {
    TraceJob j{};
    if (!trace_job(R, Q, e, j)) return SwAlignment();
    std::vector<Run> runs;
    if (!banded_path(ref + j.ref_begin, query + j.read_begin, j.subR, j.subQ, j.score, j.band, runs)) return SwAlignment();
    return alignment_from_runs(ref, query, Q, e, j, runs);
}

SwAlignment sw_align(const std::vector<int8_t>& ref, const std::vector<int8_t>& query) {
    const int R = (int)ref.size(), Q = (int)query.size();
    return sw_finish(ref.data(), R, query.data(), Q, sw_ends(ref.data(), R, query.data(), Q));
}

// ------------------------------------------------------------------------------------------------------------------------
// CIGAR helpers.  Internal op kinds follow the reference's enum (realigner.h:48-56): only these four survive its regex.
enum Kind : int8_t { K_MATCH = 1, K_INS = 2, K_DEL = 3, K_SOFT = 5 };
struct Cop { int8_t kind; int len; };

inline bool kind_of(char c, int8_t& k) {
    switch (c) {
        case '=': case 'X': case 'x': k = K_MATCH; return true;
        case 'S': case 's': k = K_SOFT; return true;
        case 'D': case 'd': k = K_DEL; return true;
        case 'I': case 'i': k = K_INS; return true;
        default: return false;
    }
}
std::vector<Cop> to_cops(const std::vector<Op>& c) {
    std::vector<Cop> v;
    for (const Op& o : c) { int8_t k; if (kind_of(o.op, k)) v.push_back({k, o.len}); }
    return v;
}

// MergeCigarOp (realigner.cpp:552-575)
struct Composed {
    std::vector<Cop> ops;
    int aligned = 0;                   // sum of the non-deletion lengths
    void merge(int8_t kind, int len, int read_len) {
        int n = kind != K_DEL ? std::min(len, read_len - aligned) : len;
        if (n <= 0 || aligned == read_len) return;
        if (!ops.empty() && ops.back().kind == kind) ops.back().len += n;
        else ops.push_back({kind, n});
        if (kind != K_DEL) aligned += n;
    }
};

inline bool is_m(int8_t k) { return k == K_MATCH || k == K_SOFT; }

// CalculateReadToRefAlignment (:653-778) with LeftTrimHaplotypeToRefAlignment (:579-609): two op queues walked with cursors.
bool compose(const std::vector<Cop>& read_to_hap_in, const std::vector<Cop>& hap_to_ref_in, int read_to_hap_pos, int read_len,
             std::vector<Cop>& out) {
    std::vector<Cop> a = read_to_hap_in, b = hap_to_ref_in;     // a: read -> haplotype, b: haplotype -> reference
    size_t ia = 0, ib = 0;
    // left trim b to the read's start on the haplotype
    int cur = 0;
    while (cur != read_to_hap_pos) {
        if (ib >= b.size()) return false;                      // the reference would pop an empty list
        Cop op = b[ib++];
        if (op.kind == K_MATCH || op.kind == K_SOFT || op.kind == K_INS) {
            if (op.len + cur > read_to_hap_pos) { --ib; b[ib] = {op.kind, op.len - (read_to_hap_pos - cur)}; }
            cur = std::min(op.len + cur, read_to_hap_pos);
        }
    }
    if (ib < b.size() && b[ib].kind == K_DEL) ++ib;
    Composed c;
    if (ia < a.size() && a[ia].kind == K_SOFT) { c.merge(K_SOFT, a[ia].len, read_len); ++ia; }
    // put-back of a shortened head: the slot just consumed is free, so a cursor step back stands in for push_front
    while ((ia < a.size() || ib < b.size()) && c.aligned < read_len) {
        if (ia < a.size() && ib >= b.size()) { c.merge(a[ia].kind, a[ia].len, read_len); ++ia; continue; }
        if (ia >= a.size()) break;
        Cop ra = a[ia++], hb = b[ib++];
        if (is_m(ra.kind) && is_m(hb.kind)) {
            const int n = std::min(ra.len, hb.len);
            c.merge(ra.kind == K_SOFT || hb.kind == K_SOFT ? K_SOFT : K_MATCH, n, read_len);
            ra.len -= n; hb.len -= n;
            if (ra.len > 0) a[--ia] = ra;
            if (hb.len > 0) b[--ib] = hb;
        } else if (ra.kind == K_DEL && is_m(hb.kind)) {
            c.merge(K_DEL, ra.len, read_len);
            hb.len -= ra.len;
            if (hb.len > 0) b[--ib] = hb;
        } else if (hb.kind == K_DEL && is_m(ra.kind)) {
            c.merge(K_DEL, hb.len, read_len);
            if (ra.len > 0) a[--ia] = ra;
        } else if (ra.kind == K_DEL && hb.kind == K_DEL) {
            c.merge(K_DEL, ra.len + hb.len, read_len);
        } else if (ra.kind == K_INS && is_m(hb.kind)) {
            ra.len = std::min(read_len - c.aligned, ra.len);
            c.merge(K_INS, ra.len, read_len);
            if (hb.len > 0) b[--ib] = hb;
        } else if (hb.kind == K_INS && is_m(ra.kind)) {
            hb.len = std::min(read_len - c.aligned, hb.len);
            c.merge(K_INS, hb.len, read_len);
            ra.len = std::max(0, ra.len - hb.len);
            if (ra.len > 0) a[--ia] = ra;
        } else if (ra.kind == K_INS && hb.kind == K_INS) {
            c.merge(K_INS, ra.len + hb.len, read_len);
        } else {
            out.clear();
            return true;
        }
    }
    out.swap(c.ops);
    return true;
}

// SetPositionsMap (:454-505)
std::vector<int> positions_map(const std::vector<Op>& cigar, size_t hap_len) {
    std::vector<int> m(hap_len, 0);
    int shift = 0;
    size_t p = 0;
    auto put = [&](int v) { if (p < hap_len) m[p] = v; ++p; };
    for (const Op& o : cigar) {
        switch (o.op) {
            case '=': case 'X': for (int k = 0; k < o.len; ++k) put(shift); break;
            case 'S': shift -= o.len; for (int k = 0; k < o.len; ++k) put(shift); break;
            case 'D': shift += o.len; break;
            case 'I': for (int k = 0; k < o.len; ++k) { put(shift); --shift; } break;
            default: break;
        }
    }
    return m;
}

// ------------------------------------------------------------------------------------------------------------------------

// k-mer index over the reads (BuildIndex, :435-452): occurrences of one k-mer keep read order, then offset order
struct KmerIndex {
    std::unordered_map<std::string, std::vector<std::pair<int, int>>> map;   // key = the 32 bytes
    void add(const std::string& read, int id) {
        if ((int)read.size() <= kKmer) return;
        for (size_t i = 0; i + kKmer <= read.size(); ++i) map[read.substr(i, kKmer)].push_back({id, (int)i});
    }
};

// FastAlignStrings (:231-251): N on either side is a match; the third mismatch ends the comparison with score 0
inline int fast_score(const char* hap, const char* read, int n, int& mism) {
    int matches = 0;
    mism = 0;
    for (int i = 0; i < n; ++i) {
        const char a = hap[i], b = read[i];
        if (a != b && a != 'N' && b != 'N') {
            if (++mism == kMaxMismatches + 1) return 0;
        } else ++matches;
    }
    return matches * kMatch - mism * kMismatch;
}

void append_cigar(std::string& s, const std::vector<Cop>& ops) {   // CigarVectorToString (:292-316): a match prints as X
    for (const Cop& o : ops) {
        s += std::to_string(o.len);
        switch (o.kind) { case K_MATCH: s += 'X'; break; case K_INS: s += 'I'; break; case K_DEL: s += 'D'; break; case K_SOFT: s += 'S'; break; }
    }
}

}  // namespace

int cto_realign_write_cigars(const std::vector<std::string>& out, char* cigar_buf, size_t cigar_cap, int64_t* cigar_off) {
    size_t used = 0;
    const int n = (int)out.size();
    for (int i = 0; i < n; ++i) {
        cigar_off[i] = (int64_t)used;
        CTO_REQUIRE(used + out[i].size() + 1 <= cigar_cap, CTO_ENOMEM, "cto_realign_reads: cigar buffer too small");
        memcpy(cigar_buf + used, out[i].c_str(), out[i].size() + 1);
        used += out[i].size() + 1;
    }
    cigar_off[n] = (int64_t)used;
    return CTO_OK;
}

namespace cto_realign {

int get_threads() { return std::max(1, g_threads.load()); }

static std::vector<std::string> split_ws(const char* s) {             // `in >> t` (:786-793)
    std::vector<std::string> v;
    const char* p = s;
    while (*p) {
        while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == '\f' || *p == '\v') ++p;
        const char* q = p;
        while (*q && !(*q == ' ' || *q == '\t' || *q == '\n' || *q == '\r' || *q == '\f' || *q == '\v')) ++q;
        if (q > p) v.emplace_back(p, q);
        p = q;
    }
    return v;
}

int Window::init(int n_reads, const char* const* seqs, const int32_t* pos, const char* const* cig, const char* ref,
                 const char* haplotypes, int ref_start_, int ref_prefix_, int ref_suffix_) {
    CTO_REQUIRE(n_reads >= 0 && ref && haplotypes, CTO_EINVAL, "cto_realign_reads: bad argument");
    CTO_REQUIRE(n_reads == 0 || (seqs && pos && cig), CTO_EINVAL, "cto_realign_reads: bad argument");
    reads.resize(n_reads);
    cigars.resize(n_reads);
    positions.assign(pos, pos + n_reads);
    for (int i = 0; i < n_reads; ++i) { reads[i] = seqs[i]; cigars[i] = cig[i]; }
    reference = ref;
    haps = split_ws(haplotypes);
    ref_start = ref_start_; ref_prefix = ref_prefix_; ref_suffix = ref_suffix_;
    for (const std::string& h : haps)
        CTO_REQUIRE((int)h.size() >= kKmer, CTO_EINVAL, "cto_realign_reads: a haplotype is shorter than the %d-mer seed", kKmer);
    const int n = n_reads, H = (int)haps.size();
    hs.assign(H, HapState());
    for (int h = 0; h < H; ++h) { hs[h].index = h; hs[h].hits.assign(n, ReadHit()); }
    return CTO_OK;
}

// fast pass, haplotype by haplotype (:129-229)
void Window::fast_pass_host() {
    const int n = (int)reads.size(), H = (int)haps.size();
    KmerIndex index;
    for (int r = 0; r < n; ++r) index.add(reads[r], r);
    for (int h = 0; h < H; ++h) {
        HapState& st = hs[h];
        const std::string& hap = haps[h];
        const bool is_ref = hap == reference;
        const int L = (int)hap.size();
        std::vector<int> coverage(L, 0);
        int score = 0;
        bool dropped = false;
        for (int i = 0; i + kKmer <= L; ++i) {
            auto it = index.map.find(hap.substr(i, kKmer));
            if (it == index.map.end()) continue;               // ... which also skips the coverage test below (:162-165)
            for (const auto& occ : it->second) {
                const int r = occ.first, start = std::max(0, i - occ.second), span = (int)reads[r].size();
                if (start + span > L) continue;
                ReadHit& hit = st.hits[r];
                if (hit.position == start) continue;
                int mism;
                const int sc = fast_score(hap.data() + start, reads[r].data(), span, mism);
                if (mism > kMaxMismatches) continue;
                for (int p = start; p < start + span; ++p) ++coverage[p];
                if (hit.score < sc) {
                    score += sc - hit.score;
                    hit.score = sc;
                    hit.position = start;
                    hit.exact = true;
                }
            }
            // a seeded position inside the consensus part that no read covers with <= 2 mismatches disqualifies the haplotype
            // (the upper bound is computed in size_t by the reference: a suffix longer than the haplotype wraps instead of going negative)
            if (coverage[i] == 0 && i >= ref_prefix && (uint64_t)i < (uint64_t)L - (uint64_t)(int64_t)ref_suffix && !is_ref) { dropped = true; break; }
        }
        if (dropped) score = 0;
        if (score == 0) st.hits.assign(n, ReadHit());
        st.score = score;
    }
}

// The same state from the device's fast pass (csrc/realign_batch.hip: k_fast_pass)
void Window::set_fast_pass(const int32_t* hit_score, const int32_t* hit_pos, const int32_t* hap_score) {
    const int n = (int)reads.size(), H = (int)haps.size();
    for (int h = 0; h < H; ++h) {
        HapState& st = hs[h];
        st.score = hap_score[h];
        st.hits.assign(n, ReadHit());
        if (st.score == 0) continue;
        for (int r = 0; r < n; ++r) {
            const int sc = hit_score[size_t(h) * n + r];
            if (sc <= 0) continue;
            ReadHit& hit = st.hits[r];
            hit.score = sc;
            hit.position = hit_pos[size_t(h) * n + r];
            hit.exact = true;
        }
    }
}

// haplotypes against the reference (:315-349) and Smith-Waterman for the reads no haplotype took (:351-384): the list of alignments
void Window::collect_pairs() {
    const int n = (int)reads.size(), H = (int)haps.size();
    refc = Codes(reference.data(), reference.size()).v;
    hapc.clear();
    for (const std::string& h : haps) hapc.push_back(Codes(h.data(), h.size()).v);
    todo.clear();
    for (int r = 0; r < n; ++r) {
        bool taken = false;
        for (const HapState& st : hs) if (st.hits[r].score > 0) { taken = true; break; }
        if (!taken) todo.push_back(r);
    }
    readc.assign(n, std::vector<int8_t>());
    for (int r : todo) readc[r] = Codes(reads[r].data(), reads[r].size()).v;
    pairs.clear();
    for (int h = 0; h < H; ++h) pairs.push_back({refc.data(), (int)refc.size(), hapc[h].data(), (int)hapc[h].size()});
    for (int r : todo)
        for (const HapState& st : hs) {
            if (st.score == 0) continue;
            pairs.push_back({hapc[st.index].data(), (int)hapc[st.index].size(), readc[r].data(), (int)readc[r].size()});
        }
    ends.clear();
    traced_at.clear();
    traced.clear();
}

// Reads are independent, so the pairs are dealt to g_threads workers when the caller allows more than one
// (cto_set_realign_threads; default 1, as the reference's one process per chunk).
void Window::ends_host() {
    ends.assign(pairs.size(), Ends{0, 0, 0, 0, 0, 16});
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (size_t k = next++; k < pairs.size(); k = next++) {
            const SwPair& p = pairs[k];
            ends[k] = sw_ends(p.ref, p.R, p.query, p.Q);
        }
    };
    const int nt = int(std::min<size_t>(size_t(get_threads()), pairs.size() > (size_t)haps.size() ? pairs.size() - haps.size() : 1));
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (std::thread& t : pool) t.join();
}

Ends ends_of_pair(const int8_t* ref, int R, const int8_t* query, int Q) { return sw_ends(ref, R, query, Q); }

bool plan_pair(int R, int Q, const Ends& e, TraceJob& j) { return trace_job(R, Q, e, j); }
SwAlignment alignment_of_pair(const int8_t* ref, int R, const int8_t* query, int Q, const Ends& e) { return sw_finish(ref, R, query, Q, e); }
SwAlignment alignment_from_device_runs(const int8_t* ref, const int8_t* query, int Q, const Ends& e, const TraceJob& j, const int32_t* runs, int n_runs) {
    std::vector<Run> rr(static_cast<size_t>(n_runs));
    for (int i = 0; i < n_runs; ++i) rr[size_t(i)] = Run{"MID"[runs[i] & 3], int(runs[i] >> 2)};
    return alignment_from_runs(ref, query, Q, e, j, rr);
}

bool trace_runs_host(const Window& w, const TraceJob& job, std::vector<int32_t>& runs) {
    const SwPair& p = w.sw_pairs()[size_t(job.pair)];
    std::vector<Run> rr;
    runs.clear();
    if (!banded_path(p.ref + job.ref_begin, p.query + job.read_begin, job.subR, job.subQ, job.score, job.band, rr)) return false;
    for (const Run& r : rr) runs.push_back((r.len << 2) | (r.op == 'M' ? kTraceM : (r.op == 'I' ? kTraceI : kTraceD)));
    return true;
}

// the rule of GetBestReadAlignment (:514-538) over the haplotypes in their sorted order; score(h) = the read's score on haplotype h
template <class Score, class IsRef>
int pick_haplotype(const std::vector<int>& order, Score&& score, IsRef&& is_reference) {
    int best = 0, pick = -1;
    for (int h : order) {
        const int sc = score(h);
        if (sc > best || (best > 0 && sc == best && !is_reference(h))) { best = sc; pick = h; }
    }
    return pick;
}

// the reference's std::sort by haplotype score (:108); ties keep whatever order that algorithm leaves them in
std::vector<int> haplotype_order(const std::vector<HapState>& hs) {
    std::vector<int> order(hs.size());
    for (size_t h = 0; h < hs.size(); ++h) order[h] = int(h);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return hs[a].score < hs[b].score; });
    return order;
}

void Window::plan_tracebacks(std::vector<TraceJob>& jobs) const {
    const int n = (int)reads.size(), H = (int)haps.size();
    if (ends.size() != pairs.size()) return;
    std::vector<char> is_ref(size_t(H), 0);
    for (int h = 0; h < H; ++h) {
        TraceJob j{};
        j.pair = h;
        if (!trace_job(pairs[h].R, pairs[h].Q, ends[h], j)) continue;
        jobs.push_back(j);
        is_ref[size_t(h)] = ends[h].score == kMatch * pairs[h].Q;      // every base of the haplotype matched: one '=' run of its length
    }
    if (todo.empty()) return;
    std::vector<int> noted(size_t(H) * size_t(n), -1);
    size_t k = size_t(H);
    for (int r : todo)
        for (int hi = 0; hi < H; ++hi) {
            const HapState& st = hs[size_t(hi)];
            if (st.score == 0) continue;
            const Ends& e = ends[k];
            if (e.score > 0 && e.score >= kSswThreshold && st.hits[r].score < e.score) noted[size_t(hi) * n + r] = int(k);
            ++k;
        }
    const std::vector<int> order = haplotype_order(hs);
    for (int r : todo) {
        const int pick = pick_haplotype(order, [&](int h) { const int at = noted[size_t(h) * n + r]; return at >= 0 ? ends[size_t(at)].score : hs[size_t(h)].hits[r].score; },
                                        [&](int h) { return is_ref[size_t(h)] != 0; });
        if (pick < 0 || noted[size_t(pick) * n + r] < 0) continue;
        TraceJob j{};
        j.pair = noted[size_t(pick) * n + r];
        if (trace_job(pairs[size_t(j.pair)].R, pairs[size_t(j.pair)].Q, ends[size_t(j.pair)], j)) jobs.push_back(j);
    }
}

void Window::set_traced(const TraceJob& job, bool ok, const int32_t* runs, int n_runs) {
    if (traced_at.empty()) traced_at.assign(pairs.size(), -1);
    SwAlignment al;
    if (ok) {
        const SwPair& p = pairs[size_t(job.pair)];
        al = alignment_from_device_runs(p.ref, p.query, p.Q, ends[size_t(job.pair)], job, runs, n_runs);
    }
    traced_at[size_t(job.pair)] = int(traced.size());
    traced.push_back(std::move(al));
}

int Window::finish(int32_t* out_pos, std::vector<std::string>& out_cigar) {
    const int n = (int)reads.size(), H = (int)haps.size();
    CTO_REQUIRE(ends.size() == pairs.size(), CTO_EINVAL, "cto_realign_reads: stage 2 did not run");
    // the traceback of pair k: taken from set_traced when it was done elsewhere, run here otherwise
    auto finish_pair = [&](size_t k) -> SwAlignment {
        if (!traced_at.empty() && traced_at[k] >= 0) return traced[size_t(traced_at[k])];
        const SwPair& p = pairs[k];
        return sw_finish(p.ref, p.R, p.query, p.Q, ends[k]);
    };
    size_t k = 0;
    // haplotypes against the reference (:315-349), position maps (:507-512)
    for (HapState& st : hs) {
        const SwAlignment al = finish_pair(k);
        ++k;
        if (al.score > 0) {
            st.is_reference = al.cigar.size() == 1 && al.cigar[0].op == '=' && al.cigar[0].len == (int)haps[st.index].size();
            st.cigar = al.cigar;
            st.ref_pos = al.ref_begin;
        }
        st.pos_map = positions_map(st.cigar, haps[st.index].size());
    }
    // Smith-Waterman for the reads no haplotype took (:351-384).  The reference runs the banded traceback of every (read, haplotype)
    // pair and then uses one of them per read; here a pair whose score would replace the read's hit on that haplotype is only NOTED,
    // with the score the striped passes found - the traceback either confirms exactly that score or fails (then the hit stays as it
    // was) - and the traceback runs for the pair a read ends up picking (below).
    std::vector<int> noted(todo.empty() ? 0 : size_t(H) * size_t(n), -1);            // [haplotype][read] -> pair
    for (int r : todo)
        for (int hi = 0; hi < H; ++hi) {
            HapState& st = hs[size_t(hi)];
            if (st.score == 0) continue;
            const Ends& e = ends[k];
            if (e.score > 0 && e.score >= kSswThreshold && st.hits[r].score < e.score) noted[size_t(hi) * n + r] = int(k);
            ++k;
        }

    const std::vector<int> order = haplotype_order(hs);

    // every read onto the reference through its best haplotype (:386-433)
    out_cigar.assign(n, std::string());
    for (int r = 0; r < n; ++r) {
        out_pos[r] = positions[r];
        out_cigar[r] = cigars[r];
        int pick = -1;
        for (;;) {
            pick = pick_haplotype(order, [&](int h) { const int at = noted.empty() ? -1 : noted[size_t(h) * n + r];
                                                      return at >= 0 ? ends[size_t(at)].score : hs[size_t(h)].hits[r].score; },
                                  [&](int h) { return hs[size_t(h)].is_reference; });
            if (pick < 0 || noted.empty() || noted[size_t(pick) * n + r] < 0) break;
            // The pick is a noted pair: its traceback now.  Success installs the hit with the score it was noted with (the pick
            // stands); a failure leaves the read's hit on that haplotype as it was, which can only lower that entry - and lowering an
            // entry that was not picked never changes the pick, so the pairs not traced leave no trace - and the choice is made again.
            const size_t at = size_t(noted[size_t(pick) * n + r]);
            noted[size_t(pick) * n + r] = -1;
            const SwAlignment al = finish_pair(at);
            ReadHit& hit = hs[size_t(pick)].hits[r];
            if (al.score > 0 && al.score >= kSswThreshold && hit.score < al.score) {
                hit.score = al.score;
                hit.cigar = al.cigar;
                hit.position = al.ref_begin;
                hit.exact = false;
            }
        }
        if (pick < 0) continue;
        const HapState& st = hs[pick];
        const ReadHit& hit = st.hits[r];
        CTO_REQUIRE(hit.position >= 0 && hit.position < (int)st.pos_map.size(), CTO_EINVAL, "cto_realign_reads: read %d lies outside its haplotype", r);
        std::vector<Cop> ops;
        // an exact hit (fast pass) is the whole read in one '=' run; it is spelled out here, not stored per (haplotype, read)
        const std::vector<Cop> read_ops = hit.exact ? to_cops(std::vector<Op>{Op{'=', (int)reads[r].size()}}) : to_cops(hit.cigar);
        if (!compose(read_ops, to_cops(st.cigar), hit.position, (int)reads[r].size(), ops))
            CTO_REQUIRE(false, CTO_EINVAL, "cto_realign_reads: the haplotype alignment ends before read %d starts", r);
        if (!ops.empty()) {
            out_cigar[r].clear();
            append_cigar(out_cigar[r], ops);
            out_pos[r] = ref_start + st.ref_pos + hit.position + st.pos_map[hit.position];
        }
    }
    return CTO_OK;
}

}  // namespace cto_realign

namespace {

}  // namespace

extern "C" int cto_realign_reads(int n_reads, const char* const* seqs, const int32_t* positions, const char* const* cigars,
                                 const char* reference, const char* haplotypes, int32_t ref_start, int32_t ref_prefix,
                                 int32_t ref_suffix, int32_t* out_positions, char* cigar_buf, size_t cigar_cap, int64_t* cigar_off) try {
    CTO_REQUIRE(n_reads >= 0 && reference && haplotypes && out_positions && cigar_off && (cigar_buf || cigar_cap == 0), CTO_EINVAL,
                "cto_realign_reads: bad argument");
    cto_realign::Window w;
    int rc = w.init(n_reads, seqs, positions, cigars, reference, haplotypes, ref_start, ref_prefix, ref_suffix);
    if (rc != CTO_OK) return rc;
    w.fast_pass_host();
    w.collect_pairs();
    w.ends_host();
    std::vector<std::string> out;
    rc = w.finish(out_positions, out);
    if (rc != CTO_OK) return rc;
    return cto_realign_write_cigars(out, cigar_buf, cigar_cap, cigar_off);
}
CTO_CATCH("cto_realign_reads", int)

// Smith-Waterman alone (test hook and building block): query against ref, CIGAR text over S = X I D as SSW's C++ wrapper prints it.
// The evidence a read contributes to the window search of `reads_realignment` (src/realign_reads.py:306-352; the per-base loop of
// clairs_to_amd/realign_reads.py:RegionRealigner.feed, which is where that module's time went): every reference position the read
// contradicts - a mismatch with BQ >= min_bq on an A/C/G/T reference base; positions [p - n, p + n) around an insertion or soft clip
// of n bases none of which is below min_bq; the n positions of a deletion; the last two only inside [lo_ok, hi_ok] and next to an
// A/C/G/T reference base.  One entry of `out` per increment.  `bq` is the SAM text (phred + 33).  Indexing follows the Python it
// replaces (a negative reference index counts from the end); anything that would raise there - or more than `cap` increments -
// returns -1 and the caller runs the Python loop, so errors stay the reference's errors.
extern "C" int64_t cto_realign_read_evidence(const char* seq, int64_t seq_len, const char* bq, int64_t bq_len, const char* cigar, int64_t start,
                                             const char* ref, int64_t ref_len, int64_t ref0, int64_t lo_ok, int64_t hi_ok, int min_bq,
                                             int32_t* out, int64_t cap) {
    if (!seq || !bq || !cigar || !ref || !out) return -1;
    auto acgt = [](char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; };
    auto ref_at = [&](int64_t i, char& c) { if (i < 0) i += ref_len; if (i < 0 || i >= ref_len) return false; c = ref[i]; return true; };
    const int thr = min_bq + 33;
    int64_t rp = start, qp = 0, used = 0, n = 0;
    auto range = [&](int64_t a, int64_t b) {
        if (b - a > cap - used) return false;
        for (int64_t p = a; p < b; ++p) out[used++] = int32_t(p);
        return true;
    };
    for (const char* c = cigar; *c; ++c) {
        if (*c >= '0' && *c <= '9') { n = n * 10 + (*c - '0'); if (n > (int64_t(1) << 40)) return -1; continue; }
        const char op = *c;
        if (op == '=') { rp += n; qp += n; }
        else if (op == 'M' || op == 'X') {
            for (int64_t k = 0; k < n; ++k, ++rp, ++qp) {
                if (qp >= bq_len) return -1;
                if ((unsigned char)bq[qp] - 0 >= thr) {
                    char rb;
                    if (!ref_at(rp - ref0, rb)) return -1;
                    if (acgt(rb)) {
                        if (qp >= seq_len) return -1;
                        if (seq[qp] != rb) { if (used >= cap) return -1; out[used++] = int32_t(rp); }
                    }
                }
            }
        } else if (op == 'I' || op == 'S') {
            if (lo_ok <= rp && rp <= hi_ok) {
                char rb;
                if (!ref_at(rp - ref0 - 1, rb)) return -1;
                if (acgt(rb)) {
                    bool low = false;
                    for (int64_t k = qp; k < qp + n && k < bq_len && !low; ++k) low = (unsigned char)bq[k] < thr;
                    if (!low && !range(rp - n, rp + n)) return -1;
                }
            }
            qp += n;
        } else if (op == 'D') {
            if (lo_ok <= rp && rp <= hi_ok) {
                char rb;
                if (!ref_at(rp - ref0 - 1, rb)) return -1;
                if (acgt(rb) && !range(rp, rp + n)) return -1;
            }
            rp += n;
        }
        n = 0;
    }
    return used;
}

// Worker threads of the Smith-Waterman stage of cto_realign_reads (>= 1; also CTO_REALIGN_THREADS).  Results do not depend on it.
extern "C" int cto_set_realign_threads(int n) {
    CTO_REQUIRE(n >= 1 && n <= 1024, CTO_EINVAL, "cto_set_realign_threads: %d", n);
    g_threads = n;
    return CTO_OK;
}

// One striped pass alone (test hook: tests compare it with oracle/ssw_model.cpp): codes 0..4, out[4] = {score, ref_end, read_end, overflow}
extern "C" int cto_ssw_pass(const int8_t* ref, int ref_len, int reverse, const int8_t* read, int read_len, int lanes, int terminate, int32_t* out) try {
    CTO_REQUIRE(ref && read && out && ref_len >= 0 && read_len > 0 && (lanes == 16 || lanes == 8), CTO_EINVAL, "cto_ssw_pass: bad argument");
    const PassEnd e = striped_pass(ref, ref_len, reverse != 0, read, read_len, lanes, terminate);
    out[0] = e.score; out[1] = e.ref_end; out[2] = e.read_end; out[3] = e.overflow ? 1 : 0;
    return CTO_OK;
}
CTO_CATCH("cto_ssw_pass", int)

extern "C" int cto_ssw_align(const char* ref, const char* query, int32_t* score, int32_t* ref_begin, char* cigar_buf, size_t cigar_cap) try {
    CTO_REQUIRE(ref && query && score && ref_begin && cigar_buf && cigar_cap > 0, CTO_EINVAL, "cto_ssw_align: bad argument");
    const SwAlignment al = sw_align(Codes(ref, strlen(ref)).v, Codes(query, strlen(query)).v);
    std::string s;
    for (const Op& o : al.cigar) { s += std::to_string(o.len); s += o.op; }
    CTO_REQUIRE(s.size() + 1 <= cigar_cap, CTO_ENOMEM, "cto_ssw_align: cigar buffer too small");
    memcpy(cigar_buf, s.c_str(), s.size() + 1);
    *score = al.score;
    *ref_begin = al.ref_begin;
    return CTO_OK;
}
CTO_CATCH("cto_ssw_align", int)
