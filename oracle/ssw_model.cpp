// Scalar model of the striped Smith-Waterman pass of the Illumina realigner.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py):
// nothing under clairs_to_amd/ links or loads this file.
//
// One pass of the striped recurrence (ssw.c:118-311 for `lanes` = 16 unsigned bytes, :341-529 for 8 signed words of
// /root/reference/src/realign) stated lane by lane in plain integer code: stripes, the F chain that leaves stripe k and enters
// stripe k + 1, the two different lazy-F loops, E that never sees their corrections, the 8-bit overflow test with the bias of
// |mismatch|, the end position = first column that raises the maximum / smallest linear query position holding it.  Pinned: the
// product's SSE2 pass (clairs_to_amd/csrc/realign.cpp) built on it was equal to the compiled reference (oracle/_ref) on 640 golden
// and 10 000 fuzzed windows before it was vectorised; tests/test_realign.py now holds model, product and oracle/_ref together.
// Scores 4 / -6 / 8 / 2 as realigner.cpp:63-73 sets them.
#include <algorithm>
#include <cstdint>
#include <vector>

namespace {
constexpr int kMatch = 4, kMismatch = 6, kGapOpen = 8, kGapExt = 2;
inline int sub_score(int8_t a, int8_t b) { return (a == b && a < 4) ? kMatch : -kMismatch; }   // ssw_cpp.cpp:52-76

struct PassEnd { int score, ref_end, read_end; bool overflow; };

// `closed_form`: the two lazy-F loops replaced by what they compute when they run to the end - a max-plus scan of the stripes' outgoing
// F over the lanes, Fin[l] = max(F[l-1], Fin[l-1] - ext * seg), and ONE sweep H[l][j] = max(H[l][j], Fin[l] - ext * j) (all
// saturating at 0).  The loops' early exits are pure shortcuts (at an exit every lane's carried F is 0 or is at least gap_open below
// an H that a stronger chain - the main pass or an earlier round, already applied - put there), which is what
// tests/test_realign.py::test_lazy_f_closed_form_equals_the_loops pins column by column through `column_hash`; the device kernel
// (clairs_to_amd/csrc/realign_batch.hip) is built on the closed form.
PassEnd striped_pass(const int8_t* ref, int ref_len, bool reverse, const int8_t* read, int read_len, int lanes, int terminate,
                     bool closed_form = false, uint64_t* column_hash = nullptr) {
    const int seg = (read_len + lanes - 1) / lanes, P = seg * lanes;
    const bool byte_mode = lanes == 16;
    const int bias = kMismatch;                       // ssw_init: |most negative matrix entry|
    std::vector<int> prev(P, 0), cur(P, 0), E(P, 0), best_col(P, 0), F(lanes), Fl(lanes), hh(lanes);
    int best = 0, ref_end = byte_mode ? -1 : 0;
    bool overflow = false;
    const int begin = reverse ? ref_len - 1 : 0, end = reverse ? -1 : ref_len, step = reverse ? -1 : 1;
    for (int i = begin; i != end; i += step) {
        const int8_t rc = ref[i];
        prev.swap(cur);                               // prev = column i-1 (final), cur = scratch
        int colmax = 0;
        for (int lane = 0; lane < lanes; ++lane) {
            int f = 0;
            for (int j = 0; j < seg; ++j) {
                const int q = lane * seg + j;
                const int s = q < read_len ? sub_score(rc, read[q]) : 0;      // padding rows score 0 (profile = bias / 0)
                int h = (q > 0 ? prev[q - 1] : 0) + s;
                if (h < 0) h = 0;
                const int e = E[q];
                if (e > h) h = e;
                if (f > h) h = f;
                if (h > colmax) colmax = h;
                cur[q] = h;
                const int open = h > kGapOpen ? h - kGapOpen : 0;
                E[q] = std::max(e > kGapExt ? e - kGapExt : 0, open);       // E never sees the lazy-F corrections below
                f = std::max(f > kGapExt ? f - kGapExt : 0, open);
            }
            F[lane] = f;
        }
        // lazy F: the F chain that leaves stripe k enters stripe k+1
        auto shift = [&](std::vector<int>& v) { for (int l = lanes - 1; l > 0; --l) v[l] = v[l - 1]; v[0] = 0; };
        Fl = F;
        if (closed_form) {
            int fin = 0;
            for (int l = 1; l < lanes; ++l) {
                fin = std::max(F[l - 1], fin > kGapExt * seg ? fin - kGapExt * seg : 0);
                for (int j = 0; j < seg && fin > kGapExt * j; ++j) {
                    int& h = cur[l * seg + j];
                    if (fin - kGapExt * j > h) h = fin - kGapExt * j;
                    if (h > colmax) colmax = h;
                }
            }
        } else if (byte_mode) {                       // ssw.c:207-241
            shift(Fl);
            int j = 0;
            for (;;) {
                bool settled = true;
                for (int l = 0; l < lanes; ++l) {
                    const int h = cur[l * seg + j];
                    if (Fl[l] > (h > kGapOpen ? h - kGapOpen : 0)) { settled = false; break; }
                }
                if (settled) break;
                for (int l = 0; l < lanes; ++l) {
                    int& h = cur[l * seg + j];
                    if (Fl[l] > h) h = Fl[l];
                    if (h > colmax) colmax = h;
                    Fl[l] = Fl[l] > kGapExt ? Fl[l] - kGapExt : 0;
                }
                if (++j >= seg) { j = 0; shift(Fl); }
            }
        } else {                                      // ssw.c:446-459
            bool done = false;
            for (int k = 0; k < lanes && !done; ++k) {
                shift(Fl);
                for (int j = 0; j < seg; ++j) {
                    bool any = false;
                    for (int l = 0; l < lanes; ++l) {
                        int& h = cur[l * seg + j];
                        if (Fl[l] > h) h = Fl[l];
                        if (h > colmax) colmax = h;
                        hh[l] = h > kGapOpen ? h - kGapOpen : 0;
                        Fl[l] = Fl[l] > kGapExt ? Fl[l] - kGapExt : 0;
                        if (Fl[l] > hh[l]) any = true;
                    }
                    if (!any) { done = true; break; }
                }
            }
        }
        if (column_hash)
            for (int q = 0; q < P; ++q) *column_hash = (*column_hash ^ uint64_t(cur[q])) * 1099511628211ull;
        if (colmax > best) {
            best = colmax;
            if (byte_mode && best + bias >= 255) { overflow = true; break; }
            ref_end = i;
            best_col = cur;
        }
        if (colmax == terminate) break;
    }
    int read_end = read_len - 1;
    for (int q = 0; q < P; ++q)
        if (best_col[q] == best) { if (q < read_end) read_end = q; break; }
    return {overflow ? 255 : best, ref_end, read_end, overflow};
}

}  // namespace

// ref / read: base codes 0..4 (A C G T other).  out[4] = {score (255 on overflow), ref_end, read_end, overflow}
extern "C" void orc_ssw_pass(const int8_t* ref, int ref_len, int reverse, const int8_t* read, int read_len, int lanes, int terminate,
                             int* out) {
    out[0] = out[1] = out[2] = out[3] = 0;
    if (!ref || !read || ref_len < 0 || read_len <= 0 || (lanes != 16 && lanes != 8)) return;
    const PassEnd e = striped_pass(ref, ref_len, reverse != 0, read, read_len, lanes, terminate);
    out[0] = e.score; out[1] = e.ref_end; out[2] = e.read_end; out[3] = e.overflow ? 1 : 0;
}

// the same pass with the lazy-F loops (closed_form = 0) or their closed form (1); *column_hash = FNV-1a over every H column after its lazy-F step
extern "C" void orc_ssw_pass_ex(const int8_t* ref, int ref_len, int reverse, const int8_t* read, int read_len, int lanes, int terminate,
                                int closed_form, int* out, uint64_t* column_hash) {
    out[0] = out[1] = out[2] = out[3] = 0;
    *column_hash = 1469598103934665603ull;
    if (!ref || !read || ref_len < 0 || read_len <= 0 || (lanes != 16 && lanes != 8)) return;
    const PassEnd e = striped_pass(ref, ref_len, reverse != 0, read, read_len, lanes, terminate, closed_form != 0, column_hash);
    out[0] = e.score; out[1] = e.ref_end; out[2] = e.read_end; out[3] = e.overflow ? 1 : 0;
}
