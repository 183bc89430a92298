// Illumina read realignment, every window of a run at once (SURVEY.md 8f #4b; the `realign_reads` leg of BASELINE configs[3]).
//
// The reference hands its native realigner ONE window per call (src/realign_reads.py:582-595 -> realign_reads(...),
// src/realign/realigner.cpp:782-857) from one Python process per low-QUAL call.  Two of that call's stages are data-parallel over
// (haplotype, read) pairs and carry nearly all of its arithmetic; cto_realign_windows runs them for ALL windows handed over in
// one launch each:
//   k_fast_pass   realigner.cpp:129-229 (FastPassAligner): a read is placed on a haplotype where one of its 32-mers matches
//                 exactly and the whole read has <= 2 mismatches.  The reference walks a hash index of the reads' k-mers; here
//                 one workgroup per (window, haplotype) tries every diagonal of every read against the haplotype held in LDS -
//                 brute force is O(L * span) byte compares per pair, a few microseconds of a CU - and reproduces the order-
//                 dependent parts of the original (which start a read keeps on a score tie, when a haplotype position counts as
//                 covered) from the time (haplotype position, read offset) each candidate would have been visited first.
//   k_sw_ends     ssw.c:118-529 (sw_sse2_byte / sw_sse2_word) as ssw_align runs them (:781-830): forward pass, word-mode rerun
//                 on overflow, backward pass.  One 16-lane DPP row is one SSE2 register: lane l holds the stripe positions
//                 q = l * seg + j of the query exactly as the striped layout of Farrar's kernel does, the byte shift
//                 _mm_slli_si128 is row_shr:1, the lazy-F loops and their exit tests are kept operation for operation because
//                 their corrections are not fed back into E - output CIGARs depend on it (csrc/realign.cpp header).  Four
//                 alignments per wavefront; H / E columns live in LDS.
// Everything after that - banded traceback between the end points, haplotype order, CIGAR composition - is strings and a few
// hundred cells per read and stays on the host (csrc/realign.cpp: Window::finish), so device results and host results meet in the
// same code and are held byte-equal by tests/test_gpu_realign.py against oracle/_ref (the reference's own realigner.cpp + SSW).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <numeric>
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
__device__ __forceinline__ int row_max16(int v) {
    v = max(v, dpp_i<0xB1>(v));      // quad_perm [1,0,3,2]
    v = max(v, dpp_i<0x4E>(v));      // quad_perm [2,3,0,1]
    v = max(v, dpp_i<0x141>(v));     // row_half_mirror
    v = max(v, dpp_i<0x140>(v));     // row_mirror
    return v;
}
__device__ __forceinline__ int row_min16(int v) { return -row_max16(-v); }
// _mm_slli_si128(x, one element): lane l takes lane l - 1 of its row, lane 0 takes 0
__device__ __forceinline__ int row_shl1(int v) { return dpp_i<0x111>(v); }

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
    __shared__ unsigned char s_hap[FP_LMAX], s_seed[FP_LMAX], s_read[FP_RMAX];
    __shared__ int s_cov[FP_LMAX];
    __shared__ unsigned long long s_best;
    __shared__ unsigned s_neg, s_d0_visit;
    __shared__ int s_d0_mism, s_dropped;
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
    for (int r = 0; r < n; ++r) {
        const int q0 = a.read_off[r0 + r], span = a.read_off[r0 + r + 1] - q0;
        __syncthreads();                  // the previous read's result has been taken
        if (span <= kKmer) {              // BuildIndex skips reads of <= 32 bases (:437-440)
            if (tid == 0) { a.hit_score[hb + r] = 0; a.hit_pos[hb + r] = -1; }
            continue;
        }
        for (int i = tid; i < span; i += FP_NT) s_read[i] = a.read_bytes[q0 + i];
        if (tid == 0) { s_best = 0ull; s_neg = 0xffffffffu; s_d0_visit = 0xffffffffu; s_d0_mism = 99; }
        __syncthreads();
        const int dmin = -(span - kKmer), dmax = L - kKmer;
        for (int d = dmin + tid; d <= dmax; d += FP_NT) {
            const int q_lo = d < 0 ? -d : 0, q_hi = min(span, L - d);
            const bool full = d >= 0 && d + span <= L;
            int run = 0, first = -1, mism = 0;
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
        if (tid == 0) {                   // start 0: k-mer hits of diagonal 0 and of every clipped diagonal
            const unsigned visit = min(s_d0_visit, s_neg);
            if (visit != 0xffffffffu && s_d0_mism <= kMaxMism) emit(0, span, s_d0_mism, int(visit >> 16), int(visit & 0xffffu));
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
// Striped Smith-Waterman end points.  ROWS alignments per workgroup of 16 * ROWS lanes.
struct SwDesc { int ref_off, R, q_off, Q; };
struct RowPass { int score, ref_end, read_end; bool overflow; };

constexpr int kBias = 6, kGapO = 8, kGapE = 2;

// query profile in striped order: entry (j, l) = code of query position l * seg + j, 7 = padding (scores 0 against everything).
// `rev_from` >= 0: the query is qraw[rev_from], qraw[rev_from - 1], ... (the reversed prefix of the backward pass)
__device__ __forceinline__ void build_profile(unsigned char* qprof, const signed char* qraw, int Q, int seg, int lanes, int l, int rev_from) {
    for (int j = 0; j < seg; ++j) {
        const int q = l * seg + j;
        unsigned char c = 7;
        if (l < lanes && q < Q) c = static_cast<unsigned char>(rev_from >= 0 ? qraw[rev_from - q] : qraw[q]);
        qprof[j * 16 + l] = c;
    }
}

template <bool BYTE>
__device__ RowPass row_pass(const signed char* refc, int r_begin, int r_end, int r_step, const unsigned char* qprof, int Q, int seg,
                            short* H0, short* H1, short* E, int terminate, int l, int rowshift) {
    constexpr int LANES = BYTE ? 16 : 8;
    const bool act = l < LANES;
    for (int j = 0; j < seg; ++j) { H0[j * 16 + l] = 0; H1[j * 16 + l] = 0; E[j * 16 + l] = 0; }
    short* store = H0;
    short* load = H1;
    int best = 0, ref_end = BYTE ? -1 : 0, best_q = 0x7fffffff;
    bool overflow = false;
    auto row_any = [&](bool p) { return ((__ballot(p) >> rowshift) & 0xffffull) != 0ull; };
    for (int i = r_begin; i != r_end; i += r_step) {
        const int rc = refc[i];
        int f = 0, colmax = 0;
        int h = row_shl1(int(store[(seg - 1) * 16 + l]));
        { short* t = store; store = load; load = t; }          // load = column i - 1 (final), store = column i
        for (int j = 0; j < seg; ++j) {
            const int qc = qprof[j * 16 + l];
            const int sc = qc == 7 ? 0 : ((qc == rc && rc < 4) ? 4 : -6);
            if (BYTE) h = max(min(h + sc + kBias, 255) - kBias, 0);
            else h = min(h + sc, 32767);
            const int e = E[j * 16 + l];
            h = max(h, max(e, f));
            if (!act) h = 0;
            colmax = max(colmax, h);
            store[j * 16 + l] = short(h);
            const int h2 = max(h - kGapO, 0);
            E[j * 16 + l] = short(max(max(e - kGapE, 0), h2));            // E never sees the lazy-F corrections below
            f = max(max(f - kGapE, 0), h2);
            h = load[j * 16 + l];
        }
        if (BYTE) {               // ssw.c:207-241: test, then correct; the chain wraps around the stripes
            f = row_shl1(f);
            int j = 0;
            while (row_any(f > max(int(store[j * 16 + l]) - kGapO, 0))) {
                const int hh = max(int(store[j * 16 + l]), f);
                colmax = max(colmax, hh);
                store[j * 16 + l] = short(hh);
                f = max(f - kGapE, 0);
                if (++j >= seg) { j = 0; f = row_shl1(f); }
            }
        } else {                  // ssw.c:446-459: correct, then test; at most `lanes` rounds
            bool done = false;
            for (int k = 0; k < LANES && !done; ++k) {
                f = row_shl1(f);
                if (!act) f = 0;
                for (int j = 0; j < seg; ++j) {
                    const int hh = max(int(store[j * 16 + l]), f);
                    colmax = max(colmax, hh);
                    store[j * 16 + l] = short(hh);
                    const int h2 = max(hh - kGapO, 0);
                    f = max(f - kGapE, 0);
                    if (!row_any(f > h2)) { done = true; break; }
                }
            }
        }
        colmax = row_max16(colmax);
        if (colmax > best) {
            best = colmax;
            if (BYTE && best + kBias >= 255) { overflow = true; break; }
            ref_end = i;
            int mq = 0x7fffffff;            // smallest linear query position that holds the new maximum
            for (int j = 0; j < seg; ++j)
                if (int(store[j * 16 + l]) == best) { mq = l * seg + j; break; }
            best_q = row_min16(mq);
        }
        if (colmax == terminate) break;
    }
    int read_end = Q - 1;
    if (best == 0) read_end = min(read_end, 0);          // the zeroed hmax matches a maximum of 0 at position 0
    else if (best_q < read_end) read_end = best_q;
    return RowPass{overflow ? 255 : best, ref_end, read_end, overflow};
}

template <int ROWS>
__global__ __launch_bounds__(16 * ROWS) void k_sw_ends(const signed char* pool, const SwDesc* desc, const int* order, int n, Ends* out,
                                                         int Rcap, int Qcap, int segcap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int row = threadIdx.x >> 4, l = threadIdx.x & 15;
    const int slot = blockIdx.x * ROWS + row;
    const bool live = slot < n;
    const int k = live ? order[slot] : 0;
    const size_t row_bytes = size_t(Rcap) + Qcap + size_t(segcap) * 16 + size_t(3) * segcap * 16 * sizeof(short);
    unsigned char* base = lds + size_t(row) * row_bytes;
    signed char* refc = reinterpret_cast<signed char*>(base);
    signed char* qraw = refc + Rcap;
    unsigned char* qprof = reinterpret_cast<unsigned char*>(qraw + Qcap);
    short* H0 = reinterpret_cast<short*>(qprof + size_t(segcap) * 16);
    short* H1 = H0 + size_t(segcap) * 16;
    short* E = H1 + size_t(segcap) * 16;
    const SwDesc d = live ? desc[k] : SwDesc{0, 0, 0, 0};
    for (int i = l; i < d.R; i += 16) refc[i] = pool[d.ref_off + i];
    for (int i = l; i < d.Q; i += 16) qraw[i] = pool[d.q_off + i];
    __syncthreads();
    if (!live) return;
    Ends e{0, 0, 0, 0, 0, 16};
    if (d.R > 0 && d.Q > 0) {
        const int rowshift = (threadIdx.x & 63) & ~15;
        int lanes = 16, seg = (d.Q + 15) / 16;
        build_profile(qprof, qraw, d.Q, seg, 16, l, -1);
        RowPass fw = row_pass<true>(refc, 0, d.R, 1, qprof, d.Q, seg, H0, H1, E, 255, l, rowshift);
        if (fw.overflow) {
            lanes = 8; seg = (d.Q + 7) / 8;
            build_profile(qprof, qraw, d.Q, seg, 8, l, -1);
            fw = row_pass<false>(refc, 0, d.R, 1, qprof, d.Q, seg, H0, H1, E, 65535, l, rowshift);
        }
        if (fw.score > 0) {
            const int Q2 = fw.read_end + 1, seg2 = (Q2 + lanes - 1) / lanes;
            build_profile(qprof, qraw, Q2, seg2, lanes, l, fw.read_end);
            const RowPass bw = lanes == 16 ? row_pass<true>(refc, fw.ref_end, -1, -1, qprof, Q2, seg2, H0, H1, E, fw.score, l, rowshift)
                                           : row_pass<false>(refc, fw.ref_end, -1, -1, qprof, Q2, seg2, H0, H1, E, fw.score, l, rowshift);
            e = Ends{fw.score, fw.ref_end, fw.read_end, bw.ref_end, bw.read_end, lanes};
        }
    }
    if (l == 0) out[k] = e;
}

// ------------------------------------------------------------------------------------------------------------------------
template <class T>
struct DevBuf {
    T* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { CTO_HIP(hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(n, 1) * sizeof(T))); return CTO_OK; }
    int put(const std::vector<T>& v, hipStream_t s) {
        int rc = alloc(v.size());
        if (rc != CTO_OK) return rc;
        if (!v.empty()) CTO_HIP(hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
        return CTO_OK;
    }
};

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class F>
void parallel_for(size_t n, int threads, F&& f) {
    std::atomic<size_t> next{0};
    auto work = [&]() { for (size_t i = next++; i < n; i = next++) f(i); };
    const int nt = int(std::min<size_t>(size_t(std::max(1, threads)), n));
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (std::thread& t : pool) t.join();
}

bool device_eligible(const Window& w) {
    if (w.reference.size() > size_t(FP_LMAX) || w.haps.empty()) return false;
    for (const std::string& h : w.haps) if (h.size() > size_t(FP_LMAX)) return false;
    for (const std::string& r : w.reads) if (r.size() > size_t(FP_RMAX)) return false;
    return true;
}

int fast_pass_device(std::vector<Window*>& ws, hipStream_t s, cto_realign_stats* st) {
    std::vector<unsigned char> hap_bytes, read_bytes, hap_isref;
    std::vector<int> hap_off{0}, hap_win, read_off{0}, win_read0{0}, win_prefix, win_suffix;
    std::vector<long long> hit_off;
    long long hits = 0;
    for (size_t wi = 0; wi < ws.size(); ++wi) {
        const Window& w = *ws[wi];
        for (const std::string& r : w.reads) { read_bytes.insert(read_bytes.end(), r.begin(), r.end()); read_off.push_back(int(read_bytes.size())); }
        win_read0.push_back(int(read_off.size()) - 1);
        win_prefix.push_back(w.ref_prefix);
        win_suffix.push_back(w.ref_suffix);
        for (const std::string& h : w.haps) {
            hap_bytes.insert(hap_bytes.end(), h.begin(), h.end());
            hap_off.push_back(int(hap_bytes.size()));
            hap_win.push_back(int(wi));
            hap_isref.push_back(h == w.reference ? 1 : 0);
            hit_off.push_back(hits);
            hits += w.n_reads();
        }
    }
    const int nh = int(hap_win.size());
    if (nh == 0) return CTO_OK;
    CTO_REQUIRE(hap_bytes.size() < (size_t(1) << 31) && read_bytes.size() < (size_t(1) << 31), CTO_EUNSUPPORTED,
                "cto_realign_windows: more than 2 GiB of sequence in one call; split it");
    DevBuf<unsigned char> d_hap, d_read, d_isref;
    DevBuf<int> d_hap_off, d_hap_win, d_read_off, d_win_read0, d_prefix, d_suffix, d_hit_score, d_hit_pos, d_hap_score;
    DevBuf<long long> d_hit_off;
    int rc;
    if ((rc = d_hap.put(hap_bytes, s)) || (rc = d_read.put(read_bytes, s)) || (rc = d_isref.put(hap_isref, s)) || (rc = d_hap_off.put(hap_off, s)) ||
        (rc = d_hap_win.put(hap_win, s)) || (rc = d_read_off.put(read_off, s)) || (rc = d_win_read0.put(win_read0, s)) ||
        (rc = d_prefix.put(win_prefix, s)) || (rc = d_suffix.put(win_suffix, s)) || (rc = d_hit_off.put(hit_off, s)) ||
        (rc = d_hit_score.alloc(size_t(hits))) || (rc = d_hit_pos.alloc(size_t(hits))) || (rc = d_hap_score.alloc(size_t(nh))))
        return rc;
    FpArgs a{d_hap.p, d_hap_off.p, d_hap_win.p, d_isref.p, d_hit_off.p, d_read.p, d_read_off.p, d_win_read0.p, d_prefix.p, d_suffix.p,
             d_hit_score.p, d_hit_pos.p, d_hap_score.p};
    hipEvent_t e0, e1;
    CTO_HIP(hipEventCreate(&e0)); CTO_HIP(hipEventCreate(&e1));
    CTO_HIP(hipEventRecord(e0, s));
    hipLaunchKernelGGL(k_fast_pass, dim3(unsigned(nh)), dim3(FP_NT), 0, s, a);
    CTO_HIP(hipGetLastError());
    CTO_HIP(hipEventRecord(e1, s));
    std::vector<int> hit_score(static_cast<size_t>(hits), 0), hit_pos(static_cast<size_t>(hits), 0), hap_score(static_cast<size_t>(nh), 0);
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
    size_t hcur = 0;
    for (Window* w : ws) {
        const size_t H = size_t(w->n_haps());
        w->set_fast_pass(hit_score.data() + hit_off[hcur], hit_pos.data() + hit_off[hcur], hap_score.data() + hcur);
        hcur += H;
    }
    return CTO_OK;
}

template <int ROWS>
int launch_sw(hipStream_t s, const signed char* pool, const SwDesc* desc, const int* order, int n, Ends* out, int Rcap, int Qcap) {
    if (n == 0) return CTO_OK;
    Rcap = (Rcap + 15) & ~15; Qcap = (Qcap + 15) & ~15;
    const int segcap = (Qcap + 7) / 8;
    const size_t row_bytes = size_t(Rcap) + Qcap + size_t(segcap) * 16 + size_t(3) * segcap * 16 * sizeof(short);
    const size_t smem = row_bytes * ROWS;
    CTO_REQUIRE(smem <= size_t(160) * 1024, CTO_EUNSUPPORTED, "cto_realign_windows: an alignment of %d x %d does not fit the LDS", Rcap, Qcap);
    CTO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sw_ends<ROWS>), hipFuncAttributeMaxDynamicSharedMemorySize, int(smem)));
    hipLaunchKernelGGL((k_sw_ends<ROWS>), dim3(unsigned((n + ROWS - 1) / ROWS)), dim3(16 * ROWS), smem, s, pool, desc, order, n, out, Rcap, Qcap, segcap);
    CTO_HIP(hipGetLastError());
    return CTO_OK;
}

int ends_device(std::vector<Window*>& ws, hipStream_t s, cto_realign_stats* st) {
    // code pool: per window the reference, its haplotypes, the reads that need Smith-Waterman - each once
    std::vector<signed char> pool;
    std::vector<SwDesc> desc;
    std::vector<size_t> first(ws.size() + 1, 0);
    for (size_t wi = 0; wi < ws.size(); ++wi) {
        Window& w = *ws[wi];
        first[wi] = desc.size();
        auto put = [&](const std::vector<int8_t>& v) { const int off = int(pool.size()); pool.insert(pool.end(), v.begin(), v.end()); return off; };
        const int ref_off = put(w.refc);
        std::vector<int> hap_at(w.hapc.size()), read_at(w.readc.size(), -1);
        for (size_t h = 0; h < w.hapc.size(); ++h) hap_at[h] = put(w.hapc[h]);
        for (int r : w.todo) read_at[r] = put(w.readc[r]);
        for (const cto_realign::SwPair& p : w.sw_pairs()) {
            // identify the operands by address (the pairs point into refc / hapc / readc)
            int roff = -1, qoff = -1;
            if (p.ref == w.refc.data()) roff = ref_off;
            else for (size_t h = 0; h < w.hapc.size(); ++h) if (p.ref == w.hapc[h].data()) { roff = hap_at[h]; break; }
            for (size_t h = 0; h < w.hapc.size() && qoff < 0; ++h) if (p.query == w.hapc[h].data()) qoff = hap_at[h];
            if (qoff < 0) for (int r : w.todo) if (p.query == w.readc[r].data()) { qoff = read_at[r]; break; }
            CTO_REQUIRE(roff >= 0 && qoff >= 0, CTO_EINVAL, "cto_realign_windows: internal: unknown operand");
            desc.push_back(SwDesc{roff, p.R, qoff, p.Q});
        }
        CTO_REQUIRE(pool.size() < (size_t(1) << 31), CTO_EUNSUPPORTED, "cto_realign_windows: more than 2 GiB of sequence in one call; split it");
    }
    first[ws.size()] = desc.size();
    const int n = int(desc.size());
    if (n == 0) return CTO_OK;
    // two classes: queries of read length (4 per wavefront) and haplotype-length queries (1 per wavefront); inside a class
    // by descending query length, so that the rows of a wavefront - and the waves of a round - run for about as long
    std::vector<int> small, large;
    int Rs = 0, Qs = 0, Rl = 0, Ql = 0;
    long long cells = 0;
    for (int k = 0; k < n; ++k) {
        cells += (long long)desc[k].R * desc[k].Q;
        if (desc[k].Q <= FP_RMAX) { small.push_back(k); Rs = std::max(Rs, desc[k].R); Qs = std::max(Qs, desc[k].Q); }
        else { large.push_back(k); Rl = std::max(Rl, desc[k].R); Ql = std::max(Ql, desc[k].Q); }
    }
    auto by_work = [&](int x, int y) { const long long a = (long long)desc[x].Q * desc[x].R, b = (long long)desc[y].Q * desc[y].R; return a != b ? a > b : x < y; };
    std::sort(small.begin(), small.end(), by_work);
    std::sort(large.begin(), large.end(), by_work);
    std::vector<int> order(small);
    order.insert(order.end(), large.begin(), large.end());
    DevBuf<signed char> d_pool;
    DevBuf<SwDesc> d_desc;
    DevBuf<int> d_order;
    DevBuf<Ends> d_out;
    int rc;
    if ((rc = d_pool.put(pool, s)) || (rc = d_desc.put(desc, s)) || (rc = d_order.put(order, s)) || (rc = d_out.alloc(size_t(n)))) return rc;
    hipEvent_t e0, e1;
    CTO_HIP(hipEventCreate(&e0)); CTO_HIP(hipEventCreate(&e1));
    CTO_HIP(hipEventRecord(e0, s));
    if ((rc = launch_sw<4>(s, d_pool.p, d_desc.p, d_order.p, int(small.size()), d_out.p, Rs, Qs))) return rc;
    if ((rc = launch_sw<1>(s, d_pool.p, d_desc.p, d_order.p + small.size(), int(large.size()), d_out.p, Rl, Ql))) return rc;
    CTO_HIP(hipEventRecord(e1, s));
    std::vector<Ends> ends(static_cast<size_t>(n), Ends{0, 0, 0, 0, 0, 16});
    CTO_HIP(hipMemcpyAsync(ends.data(), d_out.p, size_t(n) * sizeof(Ends), hipMemcpyDeviceToHost, s));
    CTO_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    CTO_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (st) { st->sw_ms += ms; st->sw_pairs += n; st->sw_cells += cells; }
    for (size_t wi = 0; wi < ws.size(); ++wi) ws[wi]->set_ends(ends.data() + first[wi]);
    return CTO_OK;
}

}  // namespace

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
        status[i] = ws[i].init(j.n_reads, j.seqs, j.positions, j.cigars, j.reference, j.haplotypes, j.ref_start, j.ref_prefix, j.ref_suffix);
        if (status[i] != CTO_OK) errors[i] = cto_last_error();
    });
    std::vector<Window*> dev, host;
    for (size_t i = 0; i < ws.size(); ++i) {
        if (status[i] != CTO_OK) continue;
        if (where == CTO_REALIGN_DEVICE && device_eligible(ws[i])) dev.push_back(&ws[i]);
        else host.push_back(&ws[i]);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!dev.empty()) {
        int rc = fast_pass_device(dev, s, stats);
        if (rc != CTO_OK) return rc;
        parallel_for(dev.size(), threads, [&](size_t i) { dev[i]->collect_pairs(); });
        rc = ends_device(dev, s, stats);
        if (rc != CTO_OK) return rc;
    }
    const double t1 = now_ms();
    parallel_for(host.size(), threads, [&](size_t i) { host[i]->fast_pass_host(); host[i]->collect_pairs(); host[i]->ends_host(); });
    parallel_for(ws.size(), threads, [&](size_t i) {
        if (status[i] != CTO_OK) return;
        cto_realign_job& j = jobs[i];
        std::vector<std::string> out;
        int rc = ws[i].finish(j.out_positions, out);
        if (rc == CTO_OK) rc = cto_realign_write_cigars(out, j.cigar_buf, j.cigar_cap, j.cigar_off);
        if (rc != CTO_OK) { status[i] = rc; errors[i] = cto_last_error(); }
    });
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
        stats->host_ms = now_ms() - t1;
    }
    if (first_bad >= 0) { cto::set_error("window %d: %s", first_bad, errors[size_t(first_bad)].c_str()); return status[size_t(first_bad)]; }
    return CTO_OK;
}
CTO_CATCH("cto_realign_windows", int)
