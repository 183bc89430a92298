// mpileup TEXT -> column pack on the device (the tokeniser of decode_pileup_bases, and F9 / F10's row handling).
//
// The reference reads `samtools mpileup` text row by row in Python (src/create_tensor_pileup_calling.py:465-532 splits the row,
// :120-144 walks the base string character by character); csrc/pack.cpp does the same on host threads (one forward pass per row,
// 1.5 GB/s per thread) and was what a text-fed run waited for: 28 ms of CPU per 22 MB chunk against 1.9 ms of device time.  Here
// the text goes up as it is and the pack is born in HBM, as it is for BAM input (csrc/pileup.hip):
//   k_count_lines   one lane per 256-byte segment: rows that start in it                     -> scan -> row index of every segment
//   k_rows<COUNT>   one lane per segment, for each of its rows: the single forward pass of pack.cpp's fast_row (contig, position,
//                   reference base, depth, base string with ^x / $ / +n.. / -n.. , as many quality and mapping-quality characters
//                   as read-bases, '\n') - counts read-bases and DISTINCT indel keys (first-seen order, compared on the characters)
//                   -> scans -> col_off, key_off
//   k_rows<FILL>    one lane per row: the same pass again, now writing entries (code | kind << 4 | BQ << 6 | MQ << 13 | key id << 21),
//                   col_pos, col_ref, and per distinct key its meta byte, its merged candidate-extraction group and its alt_info
//                   string length                                                           -> scan -> key_str_off
//   k_key_strings   one lane per key: "I<ANCHOR><SEQ>" upper-cased / "D<reference slice>"
// A row is a chain of dependent byte reads, so a lane is slow - but there are 140 000 rows in a 4096-site chunk, and a wavefront's
// 64 rows are ~10 KB of consecutive text that stay in the vector L1 while its lanes walk them.
// Anything the single pass does not take - another field count, a short quality string, '\r', a byte outside the printable range,
// an indel or '^' running into the field's end, more than 32 indel-carrying read-bases in a row, an empty row, text that does not
// end in '\n', rows out of position order, a position outside the reference slice - sets a flag, the call returns *fallback = 1 and
// the caller runs cto_pack_from_mpileup, which defines the behaviour (and words the errors).  Held bit-equal to it, array for array and
// key string for key string, by tests/test_gpu_tokenise.py.
#include <unistd.h>
#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include "common.h"
#include "pack_internal.h"

using namespace cto;

namespace {

constexpr int SEG = 256;            // bytes of text per lane in the row-start search
constexpr int MAX_IND = 32;         // indel-carrying read-bases of one row this path interns in a lane's private memory

struct TokFlags {
    int slow;                       // a row (1 + its byte offset, clamped) the single pass declined
    int bad_order, oob;             // rows not in increasing position order; a position outside the reference slice
    int n_rows, n_keys;
    long long n_entries, key_str_bytes;
};

struct Buf {
    void* p = nullptr;
    size_t cap = 0;
    ~Buf() { if (p) (void)hipFree(p); }
    int ensure(size_t n) {
        if (n <= cap) return CTO_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = n + n / 4 + 4096;
        CTO_HIP(hipMalloc(&p, want));
        cap = want;
        return CTO_OK;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// ---- exclusive scans (int32 in, T out) over up to 2^30 elements: tile sums, one workgroup over the tile sums, apply ----
constexpr int SCAN_TILE = 4096;
template <class T>
__global__ __launch_bounds__(256) void k_tile_sum(const int* __restrict__ in, int n, T* __restrict__ tile_sum) {
    __shared__ T part[256];
    const int t0 = blockIdx.x * SCAN_TILE;
    T s = 0;
    for (int i = threadIdx.x; i < SCAN_TILE; i += 256) if (t0 + i < n) s += T(in[t0 + i]);
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) { if (int(threadIdx.x) < d) part[threadIdx.x] += part[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = part[0];
}
template <class T>
__global__ __launch_bounds__(1024) void k_tile_scan(T* __restrict__ tile_sum, int n_tiles, T* __restrict__ total) {
    __shared__ T part[1024];
    T carry = 0;
    for (int base = 0; base < n_tiles; base += 1024) {
        const int i = base + threadIdx.x;
        const T v = i < n_tiles ? tile_sum[i] : T(0);
        part[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const T a = int(threadIdx.x) >= d ? part[threadIdx.x - d] : T(0);
            __syncthreads();
            part[threadIdx.x] += a;
            __syncthreads();
        }
        if (i < n_tiles) tile_sum[i] = carry + part[threadIdx.x] - v;
        const T all = part[1023];
        __syncthreads();
        carry += all;
    }
    if (threadIdx.x == 0) *total = carry;
}
template <class T>
__global__ __launch_bounds__(256) void k_tile_apply(const int* __restrict__ in, int n, const T* __restrict__ tile_base, T* __restrict__ out) {
    __shared__ T part[256];
    const int t0 = blockIdx.x * SCAN_TILE;
    constexpr int PER = SCAN_TILE / 256;
    const int i0 = t0 + threadIdx.x * PER;
    T loc[PER];
    T s = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) { loc[k] = s; s += (i0 + k < n) ? T(in[i0 + k]) : T(0); }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const T a = int(threadIdx.x) >= d ? part[threadIdx.x - d] : T(0);
        __syncthreads();
        part[threadIdx.x] += a;
        __syncthreads();
    }
    const T base = tile_base[blockIdx.x] + part[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < PER; ++k) if (i0 + k < n) out[i0 + k] = base + loc[k];
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) out[n] = tile_base[blockIdx.x] + part[255];      // out has n + 1 elements
}
template <class T>
int scan_exclusive(hipStream_t s, const int* in, int n, T* out /* n + 1 */, T* tile_tmp, T* total) {
    if (n <= 0) { CTO_HIP(hipMemsetAsync(out, 0, sizeof(T), s)); CTO_HIP(hipMemsetAsync(total, 0, sizeof(T), s)); return CTO_OK; }
    const int tiles = int(cdiv(n, SCAN_TILE));
    hipLaunchKernelGGL(k_tile_sum<T>, dim3(unsigned(tiles)), dim3(256), 0, s, in, n, tile_tmp);
    hipLaunchKernelGGL(k_tile_scan<T>, dim3(1), dim3(1024), 0, s, tile_tmp, tiles, total);
    hipLaunchKernelGGL(k_tile_apply<T>, dim3(unsigned(tiles)), dim3(256), 0, s, in, n, tile_tmp, out);
    CTO_HIP(hipGetLastError());
    return CTO_OK;
}

// ---- character classes of pack.cpp: 0..11 read-base code, 12 indel sign, 13 '^', 14 skipped, 15 ends a field (byte <= 10) ----
__device__ __forceinline__ int char_class(unsigned c) {
    if (c <= 10u) return 15;
    switch (c) {
        case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3;
        case 'a': return 4; case 'c': return 5; case 'g': return 6; case 't': return 7;
        case '*': return 8; case '#': return 9; case 'N': return 10; case 'n': return 11;
        case '+': case '-': return 12;
        case '^': return 13;
        default: return 14;
    }
}
__device__ __forceinline__ unsigned char up_c(unsigned char c) { return (c >= 'a' && c <= 'z') ? static_cast<unsigned char>(c - 32) : c; }
__device__ __forceinline__ int ref_code_dev(unsigned char c) {
    switch (up_c(c)) { case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 0; }
}

__global__ __launch_bounds__(256) void k_count_lines(const unsigned char* __restrict__ text, long long len, int n_seg, int* __restrict__ cnt) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    const long long lo = (long long)s * SEG, hi = min(len, lo + SEG);
    int c = (s == 0 && len > 0) ? 1 : 0;                    // a row starts at byte 0 and behind every '\n' that is not the last byte
    for (long long p = lo; p < hi; ++p) c += (text[p] == '\n' && p + 1 < len) ? 1 : 0;
    // (a '\n' at p makes a row start at p + 1, which may lie in the next segment: it is counted where its '\n' is, and k_rows<COUNT>
    // walks the same '\n's, so the indices agree)
    cnt[s] = c;
}

struct RowArgs {
    const unsigned char* text; long long len;
    const unsigned char* ref; long long ref_start, ref_len;
    int max_indel_length;
    // COUNT: per segment
    const int* seg_base; int n_seg;
    long long* row_start; int* row_nt; int* row_nk; int* row_pos;
    // FILL: per row
    int n_rows;
    const long long* col_off; const int* key_off;
    unsigned* entries; int* col_pos; unsigned char* col_ref;
    unsigned char* key_meta; int* key_group; int* key_len; long long* key_seq; int* key_info;
    TokFlags* fl;
};

// One row.  FILL = false: counts; true: writes.  Returns false when the row is not one the single pass takes.
template <bool FILL>
__device__ bool one_row(const RowArgs& a, long long cur, int row) {
    const unsigned char* t = a.text;
    const long long len = a.len;
    long long q = cur;
    auto at = [&](long long p) -> unsigned { return p < len ? unsigned(t[p]) : 10u; };
    while (at(q) > 10u) ++q;                                           // contig
    if (at(q) != '\t' || q == cur) return false;                       // (an empty contig field: leave it to the host)
    ++q;
    const long long d0 = q;
    long long pos = 0;
    while (at(q) - '0' < 10u) { pos = pos * 10 + (at(q) - '0'); ++q; }
    if (q == d0 || q - d0 > 15 || at(q) != '\t') return false;
    ++q;
    while (at(q) > 10u) ++q;                                           // reference base
    if (at(q) != '\t') return false;
    ++q;
    while (at(q) > 10u) ++q;                                           // depth
    if (at(q) != '\t') return false;
    ++q;
    const long long b0 = q;
    int nt = 0, ni = 0;
    long long ind_seq[MAX_IND];
    int ind_len[MAX_IND], ind_at[MAX_IND];                             // ind_at = read-base index << 2 | kind
    for (;;) {
        const int cl = char_class(at(q));
        if (cl < 12) { ++nt; ++q; }
        else if (cl == 14) ++q;
        else if (cl == 13) { if (at(q + 1) <= 10u) return false; q += 2; }
        else if (cl == 12) {
            const int kind = at(q) == '+' ? 1 : 2;
            ++q;
            long long adv = 0;
            while (at(q) - '0' < 10u) { adv = adv * 10 + (at(q) - '0'); ++q; if (adv > (1 << 24)) return false; }
            if (nt == 0 || q + adv > len) return false;
            for (long long k = 0; k < adv; ++k) if (t[q + k] <= 10) return false;
            if (ni > 0 && (ind_at[ni - 1] >> 2) == nt - 1) --ni;      // a second annotation of the same read-base replaces the first
            if (ni >= MAX_IND) return false;
            ind_seq[ni] = q; ind_len[ni] = int(adv); ind_at[ni] = ((nt - 1) << 2) | kind;
            ++ni;
            q += adv;
        } else break;
    }
    if (at(q) != '\t' || nt > kMaxDepth) return false;
    const long long qs = q + 1, ms = qs + nt + 1, eol = ms + nt;
    if (eol >= len || t[qs + nt] != '\t' || t[eol] != '\n') return false;
    // distinct keys, first seen first: Counter key = read-base code + sign + sequence, case-sensitive (pack.cpp: intern_indel)
    int kid[MAX_IND], nk = 0;
    // read-base codes of the indel carriers need the base string again: walk it once more, only as far as needed
    int code_of[MAX_IND];
    if (ni > 0) {
        long long p = b0;
        int idx = 0, w = 0;
        while (w < ni) {
            const int cl = char_class(t[p]);
            if (cl < 12) { if (idx == (ind_at[w] >> 2)) { code_of[w] = cl; ++w; } ++idx; ++p; }
            else if (cl == 14) ++p;
            else if (cl == 13) p += 2;
            else {                                                     // an indel token: skip sign, digits and sequence
                ++p;
                long long adv = 0;
                while (unsigned(t[p]) - '0' < 10u) { adv = adv * 10 + (t[p] - '0'); ++p; }
                p += adv;
            }
        }
    }
    for (int i = 0; i < ni; ++i) {
        int found = -1;
        for (int j2 = 0; j2 < i && found < 0; ++j2) {
            if (ind_len[j2] != ind_len[i] || (ind_at[j2] & 3) != (ind_at[i] & 3) || code_of[j2] != code_of[i]) continue;
            bool eq = true;
            for (int k = 0; k < ind_len[i] && eq; ++k) eq = t[ind_seq[j2] + k] == t[ind_seq[i] + k];
            if (eq) found = kid[j2];
        }
        kid[i] = found >= 0 ? found : nk++;
    }
    if constexpr (!FILL) {
        a.row_start[row] = cur;
        a.row_nt[row] = nt;
        a.row_nk[row] = nk;
        a.row_pos[row] = int(min(pos, (long long)0x7fffffff));
        const long long ri = pos - a.ref_start;
        if (ri < 0 || ri >= a.ref_len || pos > 0x7fffffffLL) atomicMax(&a.fl->oob, 1);
        // quality characters: printable only (phred 0..94), as the host's single pass demands
        bool bad = false;
        for (int i = 0; i < nt; ++i) bad |= (unsigned(t[qs + i]) - 33u > 94u) | (unsigned(t[ms + i]) - 33u > 94u);
        return !bad;
    } else {
        const long long e0 = a.col_off[row];
        const int k0 = a.key_off[row];
        const long long ri = pos - a.ref_start;
        const unsigned char rb = a.ref[ri];
        const unsigned char ru = up_c(rb);
        a.col_pos[row] = int(pos);
        a.col_ref[row] = static_cast<unsigned char>(ref_code_dev(rb) | ((ru == 'A' || ru == 'C' || ru == 'G' || ru == 'T') ? 0 : 0x80));
        // entries: codes in a third walk of the base string, qualities beside them
        long long p = b0;
        int idx = 0, w = 0;
        while (idx < nt) {
            const int cl = char_class(t[p]);
            if (cl < 12) {
                unsigned e = unsigned(cl) | ((unsigned(t[qs + idx]) - 33u) << 6) | ((unsigned(t[ms + idx]) - 33u) << 13);
                while (w < ni && (ind_at[w] >> 2) < idx) ++w;
                if (w < ni && (ind_at[w] >> 2) == idx) {
                    const int tk = ind_at[w] & 3;
                    const int gate = tk == 1 ? ind_len[w] : ind_len[w] + 1;
                    e |= unsigned(gate > a.max_indel_length ? 3 : tk) << 4;
                    e |= unsigned(kid[w]) << 21;
                }
                a.entries[e0 + idx] = e;
                ++idx; ++p;
            } else if (cl == 14) ++p;
            else if (cl == 13) p += 2;
            else {
                ++p;
                long long adv = 0;
                while (unsigned(t[p]) - '0' < 10u) { adv = adv * 10 + (t[p] - '0'); ++p; }
                p += adv;
            }
        }
        // the row's distinct keys: meta byte, merged group (insertions by upper-cased anchor + sequence, deletions by length:
        // extract_candidates_calling.py:118-126), alt_info string length; sequence location for k_key_strings
        int grp[MAX_IND], ng = 0;
        for (int i = 0; i < ni; ++i) {
            bool first = true;
            for (int j2 = 0; j2 < i; ++j2) if (kid[j2] == kid[i]) { first = false; break; }
            if (!first) continue;
            const int tk = ind_at[i] & 3, code = code_of[i], sl = ind_len[i];
            const int gate = tk == 1 ? sl : sl + 1;
            const bool overlong = gate > a.max_indel_length;
            const bool fwd = code < 4 || code == 8 || code == 10;
            const unsigned char anchors[12] = {'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T', '*', '#', 'N', 'N'};
            const unsigned char anchor = tk == 1 ? anchors[code] : static_cast<unsigned char>('D');
            int g = -1;
            for (int j2 = 0; j2 < i && g < 0; ++j2) {                  // earlier FIRST occurrences only carry a group
                bool jfirst = true;
                for (int j3 = 0; j3 < j2; ++j3) if (kid[j3] == kid[j2]) { jfirst = false; break; }
                if (!jfirst) continue;
                const int tk2 = ind_at[j2] & 3;
                if (tk2 != tk || ind_len[j2] != sl) continue;
                if (tk == 2) { g = grp[j2]; break; }
                if (anchors[code_of[j2]] != anchor) continue;
                bool eq = true;
                for (int k = 0; k < sl && eq; ++k) eq = up_c(t[ind_seq[j2] + k]) == up_c(t[ind_seq[i] + k]);
                if (eq) g = grp[j2];
            }
            if (g < 0) g = ng++;
            grp[i] = g;
            const int k = k0 + kid[i];
            a.key_meta[k] = static_cast<unsigned char>(tk | (fwd ? 4 : 0) | (overlong ? 8 : 0));
            a.key_group[k] = g;
            long long take = min((long long)(sl + 1), (long long)a.max_indel_length);
            take = min(take, a.ref_len - ri);
            a.key_len[k] = tk == 1 ? 2 + sl : 1 + int(take);
            a.key_seq[k] = tk == 1 ? ind_seq[i] : ri;
            a.key_info[k] = (sl << 8) | (code << 4) | tk;
        }
        return true;
    }
}

__global__ __launch_bounds__(256) void k_rows_count(RowArgs a) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.n_seg) return;
    const long long lo = (long long)s * SEG, hi = min(a.len, lo + SEG);
    int row = a.seg_base[s];
    if (s == 0 && a.len > 0) { if (!one_row<false>(a, 0, row)) atomicMax(&a.fl->slow, 1); ++row; }
    for (long long p = lo; p < hi; ++p)
        if (a.text[p] == '\n' && p + 1 < a.len) { if (!one_row<false>(a, p + 1, row)) atomicMax(&a.fl->slow, int(min(p + 2, (long long)0x7fffffff))); ++row; }
}
__global__ __launch_bounds__(256) void k_rows_fill(RowArgs a) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_rows) return;
    if (r > 0 && a.row_pos[r] <= a.row_pos[r - 1]) atomicMax(&a.fl->bad_order, 1);
    one_row<true>(a, a.row_start[r], r);
}
__global__ __launch_bounds__(128) void k_key_strings(const unsigned char* __restrict__ text, const unsigned char* __restrict__ ref, int n_keys,
                                                     const long long* __restrict__ key_seq, const int* __restrict__ key_info,
                                                     const long long* __restrict__ str_off, char* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_keys) return;
    const int info = key_info[k], tk = info & 3, code = (info >> 4) & 15, sl = info >> 8;
    const long long o = str_off[k], n = str_off[k + 1] - o;
    char* dst = out + o;
    if (tk == 1) {
        const char anchors[12] = {'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T', '*', '#', 'N', 'N'};
        dst[0] = 'I';
        dst[1] = anchors[code];
        for (int i = 0; i < sl; ++i) dst[2 + i] = char(up_c(text[key_seq[k] + i]));
    } else {
        dst[0] = 'D';
        for (long long i = 0; i + 1 < n; ++i) dst[1 + i] = char(up_c(ref[key_seq[k] + i]));
    }
}

}  // namespace

struct cto_dev_tokeniser {
    Buf text, ref, seg_cnt, seg_base, row_start, row_nt, row_nk, row_pos, col_off, key_off, entries, col_pos, col_ref, key_meta, key_group,
        key_len, key_seq, key_info, str_off, key_str, tiles, flags;
    void* h_text = nullptr; size_t h_text_cap = 0;      // page-locked: the text on its way up
    void* h_stage = nullptr; size_t h_stage_cap = 0;    // page-locked: everything that comes back
    hipEvent_t ev = nullptr;
    ~cto_dev_tokeniser() {
        if (h_text) (void)hipHostFree(h_text);
        if (h_stage) (void)hipHostFree(h_stage);
        if (ev) (void)hipEventDestroy(ev);
    }
    int pin(void** p, size_t* cap, size_t n) {
        if (n <= *cap) return CTO_OK;
        if (*p) { (void)hipHostFree(*p); *p = nullptr; *cap = 0; }
        const size_t want = n + n / 4 + 65536;
        CTO_HIP(hipHostMalloc(p, want, hipHostMallocDefault));
        *cap = want;
        return CTO_OK;
    }
    int wait(hipStream_t s) {          // a sleeping wait: hipStreamSynchronize spins, and the producer threads share the cores
        CTO_HIP(hipEventRecord(ev, s));
        for (;;) {
            const hipError_t e = hipEventQuery(ev);
            if (e == hipSuccess) return CTO_OK;
            if (e != hipErrorNotReady) { set_error("cto_tokenise_device: %s", hipGetErrorString(e)); return CTO_EHIP; }
            usleep(50);
        }
    }
};

extern "C" int cto_dev_tokeniser_create(cto_dev_tokeniser** out) try {
    CTO_REQUIRE(out, CTO_EINVAL, "cto_dev_tokeniser_create: null argument");
    std::unique_ptr<cto_dev_tokeniser> c(new cto_dev_tokeniser());
    CTO_HIP(hipEventCreateWithFlags(&c->ev, hipEventDisableTiming));
    *out = c.release();
    return CTO_OK;
}
CTO_CATCH("cto_dev_tokeniser_create", int)

extern "C" void cto_dev_tokeniser_destroy(cto_dev_tokeniser* c) { delete c; }

// page-locked room for `len` bytes of text owned by the context: a caller that reads its file straight into it saves the staging copy
extern "C" char* cto_dev_tokeniser_buffer(cto_dev_tokeniser* c, size_t len) {
    if (!c || c->pin(&c->h_text, &c->h_text_cap, len + 16) != CTO_OK) return nullptr;
    return static_cast<char*>(c->h_text);
}

extern "C" int cto_tokenise_device(cto_dev_tokeniser* cx, const char* text, size_t len, const char* ref_seq, int64_t ref_start, size_t ref_len,
                                   int max_indel_length, void* stream, cto_pack_view* dev_view, cto_pack** host_lite, int* fallback) try {
    CTO_REQUIRE(cx && (text || len == 0) && ref_seq && dev_view && host_lite && fallback, CTO_EINVAL, "cto_tokenise_device: null argument");
    *fallback = 0;
    *host_lite = nullptr;
    hipStream_t s = static_cast<hipStream_t>(stream);
    memset(dev_view, 0, sizeof(*dev_view));
    // text that does not end in '\n' (or is empty, or would overflow the 32-bit row bookkeeping) is the host reader's
    if (len == 0 || text[len - 1] != '\n' || len >= (size_t(1) << 31)) { *fallback = 1; return CTO_OK; }
    int rc;
    if (text != cx->h_text) {
        if ((rc = cx->pin(&cx->h_text, &cx->h_text_cap, len + 16))) return rc;
        memcpy(cx->h_text, text, len);
    }
    const int n_seg = int(cdiv(int64_t(len), SEG));
    if ((rc = cx->text.ensure(len + 16)) || (rc = cx->ref.ensure(ref_len + 16)) || (rc = cx->seg_cnt.ensure(size_t(n_seg) * 4)) ||
        (rc = cx->seg_base.ensure(size_t(n_seg + 1) * 4)) || (rc = cx->tiles.ensure(size_t(cdiv(std::max<int64_t>(n_seg, int64_t(len / 8)), SCAN_TILE) + 2) * 8)) ||
        (rc = cx->flags.ensure(sizeof(TokFlags) + 64)) || (rc = cx->pin(&cx->h_stage, &cx->h_stage_cap, 4096)))
        return rc;
    CTO_HIP(hipMemcpyAsync(cx->text.p, cx->h_text, len, hipMemcpyHostToDevice, s));
    CTO_HIP(hipMemcpyAsync(cx->ref.p, ref_seq, ref_len, hipMemcpyHostToDevice, s));
    CTO_HIP(hipMemsetAsync(cx->flags.p, 0, sizeof(TokFlags), s));
    TokFlags* fl = cx->flags.as<TokFlags>();
    auto* hf = static_cast<TokFlags*>(cx->h_stage);
    auto fetch_flags = [&]() -> int {
        CTO_HIP(hipMemcpyAsync(hf, fl, sizeof(TokFlags), hipMemcpyDeviceToHost, s));
        return cx->wait(s);
    };
    const unsigned char* d_text = cx->text.as<unsigned char>();
    hipLaunchKernelGGL(k_count_lines, dim3(unsigned(cdiv(n_seg, 256))), dim3(256), 0, s, d_text, (long long)len, n_seg, cx->seg_cnt.as<int>());
    if ((rc = scan_exclusive<int>(s, cx->seg_cnt.as<int>(), n_seg, cx->seg_base.as<int>(), cx->tiles.as<int>(), &fl->n_rows))) return rc;
    if ((rc = fetch_flags())) return rc;
    const int n_rows = hf->n_rows;
    if (n_rows <= 0) { *fallback = 1; return CTO_OK; }
    if ((rc = cx->row_start.ensure(size_t(n_rows) * 8)) || (rc = cx->row_nt.ensure(size_t(n_rows) * 4)) || (rc = cx->row_nk.ensure(size_t(n_rows) * 4)) ||
        (rc = cx->row_pos.ensure(size_t(n_rows) * 4)) || (rc = cx->col_off.ensure(size_t(n_rows + 1) * 8)) || (rc = cx->key_off.ensure(size_t(n_rows + 1) * 4)) ||
        (rc = cx->col_pos.ensure(size_t(n_rows) * 4)) || (rc = cx->col_ref.ensure(size_t(n_rows) + 16)) ||
        (rc = cx->tiles.ensure(size_t(cdiv(std::max(n_rows, n_seg), SCAN_TILE) + 2) * 8)))
        return rc;
    RowArgs a{};
    a.text = d_text; a.len = (long long)len; a.ref = cx->ref.as<unsigned char>(); a.ref_start = ref_start; a.ref_len = (long long)ref_len;
    a.max_indel_length = max_indel_length; a.seg_base = cx->seg_base.as<int>(); a.n_seg = n_seg;
    a.row_start = cx->row_start.as<long long>(); a.row_nt = cx->row_nt.as<int>(); a.row_nk = cx->row_nk.as<int>(); a.row_pos = cx->row_pos.as<int>();
    a.n_rows = n_rows; a.fl = fl;
    hipLaunchKernelGGL(k_rows_count, dim3(unsigned(cdiv(n_seg, 256))), dim3(256), 0, s, a);
    CTO_HIP(hipGetLastError());
    if ((rc = scan_exclusive<long long>(s, cx->row_nt.as<int>(), n_rows, cx->col_off.as<long long>(), cx->tiles.as<long long>(), &fl->n_entries))) return rc;
    if ((rc = scan_exclusive<int>(s, cx->row_nk.as<int>(), n_rows, cx->key_off.as<int>(), cx->tiles.as<int>(), &fl->n_keys))) return rc;
    if ((rc = fetch_flags())) return rc;
    if (hf->slow || hf->oob) { *fallback = 1; return CTO_OK; }
    const long long n_entries = hf->n_entries;
    const int n_keys = hf->n_keys;
    if ((rc = cx->entries.ensure(size_t(std::max<long long>(n_entries, 1)) * 4)) || (rc = cx->key_meta.ensure(size_t(n_keys) + 16)) ||
        (rc = cx->key_group.ensure(size_t(n_keys + 1) * 4)) || (rc = cx->key_len.ensure(size_t(n_keys + 1) * 4)) ||
        (rc = cx->key_seq.ensure(size_t(n_keys + 1) * 8)) || (rc = cx->key_info.ensure(size_t(n_keys + 1) * 4)) ||
        (rc = cx->str_off.ensure(size_t(n_keys + 2) * 8)) || (rc = cx->tiles.ensure(size_t(cdiv(std::max(n_rows, n_keys), SCAN_TILE) + 2) * 8)))
        return rc;
    a.col_off = cx->col_off.as<long long>(); a.key_off = cx->key_off.as<int>(); a.entries = cx->entries.as<unsigned>(); a.col_pos = cx->col_pos.as<int>();
    a.col_ref = cx->col_ref.as<unsigned char>(); a.key_meta = cx->key_meta.as<unsigned char>(); a.key_group = cx->key_group.as<int>();
    a.key_len = cx->key_len.as<int>(); a.key_seq = cx->key_seq.as<long long>(); a.key_info = cx->key_info.as<int>();
    hipLaunchKernelGGL(k_rows_fill, dim3(unsigned(cdiv(n_rows, 256))), dim3(256), 0, s, a);
    CTO_HIP(hipGetLastError());
    if ((rc = scan_exclusive<long long>(s, cx->key_len.as<int>(), n_keys, cx->str_off.as<long long>(), cx->tiles.as<long long>(), &fl->key_str_bytes))) return rc;
    if ((rc = fetch_flags())) return rc;
    if (hf->bad_order) { *fallback = 1; return CTO_OK; }
    const long long sb = hf->key_str_bytes;
    if ((rc = cx->key_str.ensure(size_t(sb) + 16))) return rc;
    if (n_keys > 0) {
        hipLaunchKernelGGL(k_key_strings, dim3(unsigned(cdiv(n_keys, 128))), dim3(128), 0, s, d_text, cx->ref.as<unsigned char>(), n_keys,
                           cx->key_seq.as<long long>(), cx->key_info.as<int>(), cx->str_off.as<long long>(), cx->key_str.as<char>());
        CTO_HIP(hipGetLastError());
    }
    // the host's part of the pack (what cto_alt_info* read): positions, reference codes, key tables and strings - no entries
    std::unique_ptr<cto_pack> lite(new cto_pack());
    const size_t nc = size_t(n_rows), nk = size_t(n_keys);
    lite->col_pos.resize(nc);
    lite->col_ref.resize(nc);
    lite->key_off.resize(nc + 1);
    lite->col_off.assign(1, 0);
    lite->key_str_off.assign(nk + 1, 0);
    lite->key_str.resize(size_t(sb));
    lite->key_meta.resize(nk);
    lite->key_group.resize(nk);
    {
        const size_t bytes[7] = {nk ? (nk + 1) * 8 : 0, size_t(sb), nk, nk * 4, nc * 4, nc, (nc + 1) * 4};
        const void* src[7] = {cx->str_off.p, cx->key_str.p, cx->key_meta.p, cx->key_group.p, cx->col_pos.p, cx->col_ref.p, cx->key_off.p};
        void* dst[7] = {lite->key_str_off.data(), sb ? &lite->key_str[0] : nullptr, lite->key_meta.data(), lite->key_group.data(), lite->col_pos.data(),
                        lite->col_ref.data(), lite->key_off.data()};
        size_t off[7], total = 0;
        for (int i = 0; i < 7; ++i) { off[i] = total; total += (bytes[i] + 63) / 64 * 64; }
        if ((rc = cx->pin(&cx->h_stage, &cx->h_stage_cap, total + 64))) return rc;
        char* hs = static_cast<char*>(cx->h_stage);
        for (int i = 0; i < 7; ++i)
            if (bytes[i]) CTO_HIP(hipMemcpyAsync(hs + off[i], src[i], bytes[i], hipMemcpyDeviceToHost, s));
        if ((rc = cx->wait(s))) return rc;
        for (int i = 0; i < 7; ++i)
            if (bytes[i]) memcpy(dst[i], hs + off[i], bytes[i]);
    }
    dev_view->n_cols = n_rows;
    dev_view->n_entries = n_entries;
    dev_view->n_keys = n_keys;
    dev_view->col_pos = cx->col_pos.as<int32_t>();
    dev_view->col_ref = cx->col_ref.as<uint8_t>();
    dev_view->col_off = cx->col_off.as<int64_t>();
    dev_view->key_off = cx->key_off.as<int32_t>();
    dev_view->entries = cx->entries.as<uint32_t>();
    dev_view->key_meta = cx->key_meta.as<uint8_t>();
    dev_view->key_group = cx->key_group.as<int32_t>();
    *host_lite = lite.release();
    return CTO_OK;
}
CTO_CATCH("cto_tokenise_device", int)
