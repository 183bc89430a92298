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
#include <stdlib.h>
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
// SUMS: tile_base holds the tiles' SUMS (k_tile_sum's output, no k_tile_scan in between) and every workgroup adds up those in front of
// its own - for the few dozen tiles of a chunk's rows and keys that is cheaper than a launch; `total` is then written here
template <class T, bool SUMS>
__global__ __launch_bounds__(256) void k_tile_apply(const int* __restrict__ in, int n, const T* __restrict__ tile_base, T* __restrict__ out,
                                                    T* __restrict__ total) {
    __shared__ T part[256];
    __shared__ T s_base;
    if (SUMS) {
        T b = 0;
        for (int i = threadIdx.x; i < int(blockIdx.x); i += 256) b += tile_base[i];
        part[threadIdx.x] = b;
        __syncthreads();
        for (int d = 128; d > 0; d >>= 1) { if (int(threadIdx.x) < d) part[threadIdx.x] += part[threadIdx.x + d]; __syncthreads(); }
        if (threadIdx.x == 0) s_base = part[0];
        __syncthreads();
    }
    const T tile0 = SUMS ? s_base : tile_base[blockIdx.x];
    __syncthreads();
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
    const T base = tile0 + part[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < PER; ++k) if (i0 + k < n) out[i0 + k] = base + loc[k];
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) {
        out[n] = tile0 + part[255];      // out has n + 1 elements
        if (SUMS) *total = tile0 + part[255];
    }
}
template <class T>
int scan_exclusive(hipStream_t s, const int* in, int n, T* out /* n + 1 */, T* tile_tmp, T* total) {
    if (n <= 0) { CTO_HIP(hipMemsetAsync(out, 0, sizeof(T), s)); CTO_HIP(hipMemsetAsync(total, 0, sizeof(T), s)); return CTO_OK; }
    const int tiles = int(cdiv(n, SCAN_TILE));
    hipLaunchKernelGGL(k_tile_sum<T>, dim3(unsigned(tiles)), dim3(256), 0, s, in, n, tile_tmp);
    if (tiles <= 256) {                 // a chunk's rows / keys: two launches instead of three
        hipLaunchKernelGGL((k_tile_apply<T, true>), dim3(unsigned(tiles)), dim3(256), 0, s, in, n, tile_tmp, out, total);
    } else {
        hipLaunchKernelGGL(k_tile_scan<T>, dim3(1), dim3(1024), 0, s, tile_tmp, tiles, total);
        hipLaunchKernelGGL((k_tile_apply<T, false>), dim3(unsigned(tiles)), dim3(256), 0, s, in, n, tile_tmp, out, total);
    }
    CTO_HIP(hipGetLastError());
    return CTO_OK;
}

// ---- character classes of pack.cpp: 0..11 read-base code, 12 indel sign, 13 '^', 14 skipped, 15 ends a field (byte <= 10) ----
// as a table: a `switch` over the byte is a dozen compare-and-branch steps, each with its exec-mask bookkeeping, per byte and wavefront
__device__ const unsigned char kCharClass[256] = {
    15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14,
    14, 14, 14, 9, 14, 14, 14, 14, 14, 14, 8, 12, 14, 12, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14,
    14, 0, 14, 1, 14, 14, 14, 2, 14, 14, 14, 14, 14, 14, 10, 14, 14, 14, 14, 14, 3, 14, 14, 14, 14, 14, 14, 14, 14, 14, 13, 14,
    14, 4, 14, 5, 14, 14, 14, 6, 14, 14, 14, 14, 14, 14, 11, 14, 14, 14, 14, 14, 7, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14,
    14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14,
    14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14,
    14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14,
    14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14, 14};
__device__ __forceinline__ int char_class(unsigned c) { return kCharClass[c & 255u]; }
// the same from a workgroup's LDS copy of the table (k_rows_lanes: a byte's class is on the byte walk's dependent chain)
typedef const __attribute__((address_space(3))) unsigned char* lds_table;
__device__ __forceinline__ int char_class(lds_table cls, unsigned c) { return cls ? int(cls[c & 255u]) : int(kCharClass[c & 255u]); }
__device__ __forceinline__ unsigned char up_c(unsigned char c) { return (c >= 'a' && c <= 'z') ? static_cast<unsigned char>(c - 32) : c; }
__device__ __forceinline__ int ref_code_dev(unsigned char c) {
    switch (up_c(c)) { case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return 0; }
}

struct RowArgs {
    const unsigned char* text; long long len;
    const unsigned char* ref; long long ref_start, ref_len;
    int max_indel_length;
    int n_rows;
    const long long* row_start;
    // COUNT writes, FILL reads
    int* row_nt; int* row_nk; int* row_pos; int* row_b0; int* row_blen;
    // FILL
    const long long* col_off; const int* key_off;
    unsigned* entries; int* col_pos; unsigned char* col_ref;
    unsigned char* key_meta; int* key_group; int* key_len; long long* key_seq; int* key_info;
    TokFlags* fl;
};

// A forward reader of the text: the next <= 8 bytes sit in a register and a byte costs a shift, the following aligned 8 bytes are
// requested one refill ahead.  (A row parsed through one dependent global byte load per character - the first form of this file -
// is a chain of ~300 L2 round trips: 250 us per launch however many rows are in flight.)  The device copy of the text is padded, so
// the aligned word that holds the last byte may be read whole.
struct ByteStream {
    const unsigned char* t;
    long long pos, next;               // position of the byte peek() returns; offset of the word behind q2
    unsigned long long bits, q0, q1, q2;   // the word being consumed and the three behind it (24 bytes of look-ahead: an L2 round trip is
    int have;                              // ~700 cycles, eight bytes of parsing ~400)
    __device__ __forceinline__ static unsigned long long ld8(const unsigned char* t, long long a) {
        return *reinterpret_cast<const unsigned long long*>(t + a);
    }
    __device__ __forceinline__ void seek(const unsigned char* text, long long p) {
        t = text; pos = p;
        const long long a = p & ~7LL;
        const int o = int(p - a);
        bits = ld8(t, a) >> (8 * o);
        have = 8 - o;
        q0 = ld8(t, a + 8); q1 = ld8(t, a + 16); q2 = ld8(t, a + 24);
        next = a + 32;
    }
    __device__ __forceinline__ unsigned peek() const { return unsigned(bits & 0xffull); }
    __device__ __forceinline__ unsigned peek1() const { return have >= 2 ? unsigned((bits >> 8) & 0xffull) : unsigned(q0 & 0xffull); }
    __device__ __forceinline__ void step() {
        bits >>= 8; ++pos;
        if (--have == 0) { bits = q0; q0 = q1; q1 = q2; q2 = ld8(t, next); next += 8; have = 8; }
    }
    __device__ __forceinline__ unsigned take() { const unsigned c = peek(); step(); return c; }
    __device__ __forceinline__ void skip(long long n) { if (n < have) { bits >>= 8 * int(n); have -= int(n); pos += n; } else seek(t, pos + n); }
};

// The same reader over a wavefront's staged copy of its rows: a byte is one ds_read_u8 at a 32-bit index - no shift register, no
// refill branch, no 64-bit position.  PMC (tools/tokenise_pmc.sh), count pass, per wavefront of 64 rows: the register reader on the LDS
// copy 9.1 k vector + 17.8 k scalar + 3.4 k branch instructions, this reader 5.1 k + 16.7 k + 2.8 k (170 -> 156 us; fill 283 -> 265).
// A CU issues one instruction per cycle whatever its kind, and two thirds of them are SCALAR: the exec-mask bookkeeping of divergent
// loops and branches (8-13 s_* per trip: s_and_saveexec, s_or / s_andn2 on exec, s_cbranch) - 64 rows that sit at different places of
// their grammar.  (Most of those scalar instructions turned out to be char_class's `switch`: as a table, 33.8 M -> 9.1 M per launch and
// 156 -> 82 us.  A branch-free pass - one wave-uniform loop, states moved by selects - was written and measured before that: 209 us.)
typedef const __attribute__((address_space(3))) unsigned char* lds_bytes;
struct LdsStream {
    lds_bytes t;
    int pos;
    __device__ __forceinline__ void seek(const unsigned char* text, long long p) {
        t = (lds_bytes)text;           // `text` is the LDS copy (TextRef::t of a staged span)
        pos = int(p);
    }
    __device__ __forceinline__ unsigned peek() const { return t[pos]; }
    __device__ __forceinline__ unsigned peek1() const { return t[pos + 1]; }
    __device__ __forceinline__ void step() { ++pos; }
    __device__ __forceinline__ unsigned take() { return t[pos++]; }
    __device__ __forceinline__ void skip(long long n) { pos += int(n); }
};

// indel-carrying read-bases of the row being parsed (a lane's private memory; rows without indels never touch it)
struct RowIndels {
    long long seq[MAX_IND];
    int len[MAX_IND], at[MAX_IND], code[MAX_IND], kid[MAX_IND];          // at = read-base index << 2 | kind
    int n = 0, nk = 0;
    // distinct keys, first seen first: Counter key = read-base code + sign + sequence, case-sensitive (pack.cpp: intern_indel)
    __device__ void intern(const unsigned char* t) {
        nk = 0;
        for (int i = 0; i < n; ++i) {
            int found = -1;
            for (int j2 = 0; j2 < i && found < 0; ++j2) {
                if (len[j2] != len[i] || (at[j2] & 3) != (at[i] & 3) || code[j2] != code[i]) continue;
                bool eq = true;
                for (int k = 0; k < len[i] && eq; ++k) eq = t[seq[j2] + k] == t[seq[i] + k];
                if (eq) found = kid[j2];
            }
            kid[i] = found >= 0 ? found : nk++;
        }
    }
};

// The base string from `st` on: counts read-bases, collects the indel tokens.  Returns false when the single pass declines the row.
template <class Stream>
__device__ __forceinline__ bool walk_bases(Stream& st, const unsigned char* t, long long len, int& nt, RowIndels& ind, const unsigned char* cls_g = nullptr) {
    lds_table cls = (lds_table)cls_g;
    nt = 0;
    int last_code = 0;
    for (;;) {
        const unsigned c = st.peek();
        const int cl = char_class(cls, c);
        if (cl < 12) { last_code = cl; ++nt; st.step(); }
        else if (cl == 14) st.step();
        else if (cl == 13) { if (st.peek1() <= 10u) return false; st.step(); st.step(); }
        else if (cl == 12) {
            const int kind = c == '+' ? 1 : 2;
            st.step();
            long long adv = 0;
            while (st.peek() - '0' < 10u) { adv = adv * 10 + (st.peek() - '0'); st.step(); if (adv > (1 << 24)) return false; }
            if (nt == 0 || st.pos + adv > len) return false;
            for (long long k = 0; k < adv; ++k) if (t[st.pos + k] <= 10) return false;
            if (ind.n > 0 && (ind.at[ind.n - 1] >> 2) == nt - 1) --ind.n;      // a second annotation of the same read-base replaces the first
            if (ind.n >= MAX_IND) return false;
            ind.seq[ind.n] = st.pos; ind.len[ind.n] = int(adv); ind.at[ind.n] = ((nt - 1) << 2) | kind; ind.code[ind.n] = last_code;
            ++ind.n;
            st.skip(adv);
        } else break;
    }
    return true;
}

// Where a row's bytes are read: the text in HBM (base 0), or a wavefront's staged copy of its rows in LDS (t[0] = text[base]; offsets
// below are relative to t, `len` = the bytes that belong to rows - what lies behind is look-ahead padding)
struct TextRef { const unsigned char* t; long long base, len; const unsigned char* cls; };      // cls: the class table in LDS (or null)

// pass 1 of a row: the single forward pass of pack.cpp's fast_row; writes the row's counts, false = not a row this path takes
template <class Stream>
__device__ bool count_row(const RowArgs& a, const TextRef& T, long long cur, int row) {
    const unsigned char* t = T.t;
    const long long len = T.len;
    Stream st;
    st.seek(t, cur);
    if (st.peek() <= 10u) return false;                                 // an empty row / an empty contig field: the host's
    while (st.peek() > 10u) st.step();                                  // contig
    if (st.peek() != '\t') return false;
    st.step();
    long long pos = 0;
    int nd = 0;
    while (st.peek() - '0' < 10u) { pos = pos * 10 + (st.peek() - '0'); st.step(); ++nd; if (nd > 15) return false; }
    if (nd == 0 || st.peek() != '\t') return false;
    st.step();
    while (st.peek() > 10u) st.step();                                  // reference base
    if (st.peek() != '\t') return false;
    st.step();
    while (st.peek() > 10u) st.step();                                  // depth
    if (st.peek() != '\t') return false;
    st.step();
    const long long b0 = st.pos;
    int nt = 0;
    RowIndels ind;
    if (!walk_bases(st, t, len, nt, ind, T.cls)) return false;
    if (st.peek() != '\t' || nt > kMaxDepth) return false;
    const long long blen = st.pos - b0;
    st.step();
    // as many quality and mapping-quality characters as read-bases, printable (phred 0..94), then the end of the row
    if (st.pos + 2LL * nt + 1 >= len) return false;
    bool bad = false;
    for (int i = 0; i < nt; ++i) bad |= st.take() - 33u > 94u;
    if (st.take() != '\t') return false;
    for (int i = 0; i < nt; ++i) bad |= st.take() - 33u > 94u;
    if (bad || st.peek() != '\n') return false;
    if (ind.n > 0) ind.intern(t);
    a.row_nt[row] = nt;
    a.row_nk[row] = ind.nk;
    a.row_pos[row] = int(min(pos, (long long)0x7fffffff));
    a.row_b0[row] = int(b0 - cur);
    a.row_blen[row] = int(blen);
    const long long ri = pos - a.ref_start;
    if (ri < 0 || ri >= a.ref_len || pos > 0x7fffffffLL) atomicMax(&a.fl->oob, 1);
    return b0 - cur < (1LL << 30);
}

// pass 2 of a row: entries, column tables, the row's distinct keys
template <class Stream>
__device__ void fill_row(const RowArgs& a, const TextRef& T, long long cur, int row) {
    const unsigned char* t = T.t;
    const long long e0 = a.col_off[row];
    const int k0 = a.key_off[row];
    const int nt = a.row_nt[row];
    const long long pos = a.row_pos[row], ri = pos - a.ref_start;
    const unsigned char rb = a.ref[ri];
    const unsigned char ru = up_c(rb);
    a.col_pos[row] = int(pos);
    a.col_ref[row] = static_cast<unsigned char>(ref_code_dev(rb) | ((ru == 'A' || ru == 'C' || ru == 'G' || ru == 'T') ? 0 : 0x80));
    const long long b0 = cur + a.row_b0[row], qs = b0 + a.row_blen[row] + 1, ms = qs + nt + 1;
    RowIndels ind;
    if (a.row_nk[row] > 0) {                                             // the row's indel tokens and their key ids, as pass 1 saw them
        Stream sb;
        sb.seek(t, b0);
        int n2 = 0;
        (void)walk_bases(sb, t, T.len, n2, ind, T.cls);
        ind.intern(t);
    }
    Stream sb, sq, sm;
    sb.seek(t, b0); sq.seek(t, qs); sm.seek(t, ms);
    int idx = 0, w = 0;
    lds_table cls = (lds_table)T.cls;
    while (idx < nt) {
        const unsigned c = sb.peek();
        const int cl = char_class(cls, c);
        if (cl < 12) {
            unsigned e = unsigned(cl) | ((sq.take() - 33u) << 6) | ((sm.take() - 33u) << 13);
            if (w < ind.n && (ind.at[w] >> 2) == idx) {
                const int tk = ind.at[w] & 3;
                const int gate = tk == 1 ? ind.len[w] : ind.len[w] + 1;
                e |= unsigned(gate > a.max_indel_length ? 3 : tk) << 4;
                e |= unsigned(ind.kid[w]) << 21;
                ++w;
            }
            a.entries[e0 + idx] = e;
            ++idx;
            sb.step();
        } else if (cl == 14) sb.step();
        else if (cl == 13) { sb.step(); sb.step(); }
        else {                                                           // an indel token: sign, digits, sequence
            sb.step();
            long long adv = 0;
            while (sb.peek() - '0' < 10u) { adv = adv * 10 + (sb.peek() - '0'); sb.step(); }
            sb.skip(adv);
        }
    }
    // the row's distinct keys: meta byte, merged group (insertions by upper-cased anchor + sequence, deletions by length:
    // extract_candidates_calling.py:118-126), alt_info string length; where k_key_strings finds the sequence
    int grp[MAX_IND];
    int ng = 0;
    for (int i = 0; i < ind.n; ++i) {
        bool first = true;
        for (int j2 = 0; j2 < i; ++j2) if (ind.kid[j2] == ind.kid[i]) { first = false; break; }
        if (!first) continue;
        const int tk = ind.at[i] & 3, code = ind.code[i], sl = ind.len[i];
        const int gate = tk == 1 ? sl : sl + 1;
        const bool overlong = gate > a.max_indel_length;
        const bool fwd = code < 4 || code == 8 || code == 10;
        const unsigned char anchors[12] = {'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T', '*', '#', 'N', 'N'};
        const unsigned char anchor = tk == 1 ? anchors[code] : static_cast<unsigned char>('D');
        int g = -1;
        for (int j2 = 0; j2 < i && g < 0; ++j2) {                        // earlier FIRST occurrences only carry a group
            bool jfirst = true;
            for (int j3 = 0; j3 < j2; ++j3) if (ind.kid[j3] == ind.kid[j2]) { jfirst = false; break; }
            if (!jfirst) continue;
            if ((ind.at[j2] & 3) != tk || ind.len[j2] != sl) continue;
            if (tk == 2) { g = grp[j2]; break; }
            if (anchors[ind.code[j2]] != anchor) continue;
            bool eq = true;
            for (int k = 0; k < sl && eq; ++k) eq = up_c(t[ind.seq[j2] + k]) == up_c(t[ind.seq[i] + k]);
            if (eq) g = grp[j2];
        }
        if (g < 0) g = ng++;
        grp[i] = g;
        const int k = k0 + ind.kid[i];
        a.key_meta[k] = static_cast<unsigned char>(tk | (fwd ? 4 : 0) | (overlong ? 8 : 0));
        a.key_group[k] = g;
        long long take = min((long long)(sl + 1), (long long)a.max_indel_length);
        take = min(take, a.ref_len - ri);
        a.key_len[k] = tk == 1 ? 2 + sl : 1 + int(take);
        a.key_seq[k] = tk == 1 ? ind.seq[i] + T.base : ri;
        a.key_info[k] = (sl << 8) | (code << 4) | tk;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// A WAVEFRONT per row (CTO_TOK_WAVES=1; the lane-per-row kernels above are the default).  A row's ~250 bytes are classified 64 at a time
// with ballots instead of one after the other:
//   * tabs, control bytes                    -> field boundaries (exactly six tabs), declines
//   * '^x' pairs                             -> which bytes a live '^' consumes: runs of '^' alternate live / consumed; the carry-add trick
//                                               of simdjson's escaped-character scan resolves all runs of a 64-byte word at once
//   * '+n<seq>' / '-n<seq>' tokens           -> every sign that is not consumed is a token (a sequence holds no sign and no '^' in samtools'
//                                               output; a row where one does is declined): digits, n bytes masked
//   * read-bases                             -> class < 12, not consumed, not masked: their count is a popcount, their index a prefix popcount
// so the serial part of a row is its handful of indel tokens.  Rows longer than TOKW_CAP bytes are declined (the host's).
// Measured on MI355X (22 MB, 140 000 rows): count 0.27 ms, fill 0.39 ms against 0.21 / 0.32 ms lane per row.  It is not the idea that is slow
// but its granularity: a 170-byte row is 2.7 ballots' worth of bytes and every one of the ~15 ballot steps of a row is a dependent
// LDS-read -> compare -> scalar-mask chain, ~800 wave instructions per row, where 64 rows sharing a wavefront spend ~190 each.  The form
// that would win runs these masks over the TEXT (64 bytes of whatever rows they belong to, row boundaries as one more mask) - not built.
constexpr int TOKW_CAP = 4096, TOKW_WAVES = 4, TOKW_TOK = 32;
constexpr unsigned char F_CONS = 1, F_SKIP = 2;

struct WaveRow {                 // what one pass over a row's base string leaves behind (wave-uniform)
    int nt, ntok, nk;
    long long pos;
    int b0, b1, qs, ms, L;
};

template <bool FILL>
__global__ __launch_bounds__(64 * TOKW_WAVES) void k_rows_wave(RowArgs a) {
    __shared__ unsigned char s_buf[TOKW_WAVES][TOKW_CAP + 64];
    __shared__ unsigned char s_flg[TOKW_WAVES][TOKW_CAP + 64];
    __shared__ int s_tok[TOKW_WAVES][TOKW_TOK][6];        // sign position (relative to b0), sequence start (row offset), length, kind, carrier index, carrier code
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = blockIdx.x * TOKW_WAVES + wv;
    if (r >= a.n_rows) return;
    unsigned char* buf = s_buf[wv];
    unsigned char* flg = s_flg[wv];
    int (*tok)[6] = s_tok[wv];
    const unsigned long long below = (1ull << lane) - 1ull;
    const long long cur = a.row_start[r], nxt = r + 1 < a.n_rows ? a.row_start[r + 1] : a.len;
    const int L = int(min(nxt - cur - 1, (long long)TOKW_CAP + 1));           // without the '\n'
    auto decline = [&]() { if (lane == 0) atomicMax(&a.fl->slow, int(min(cur + 1, (long long)0x7fffffff))); };
    if (L <= 0 || L > TOKW_CAP) { decline(); return; }
    for (int i = lane; i < L; i += 64) buf[i] = a.text[cur + i];
    __builtin_amdgcn_wave_barrier();
    // ---- fields: exactly six tabs, no other byte <= 10 ----
    int tp[6] = {0, 0, 0, 0, 0, 0}, ntab = 0;
    bool ctrl = false;
    for (int base = 0; base < L; base += 64) {
        const unsigned c = base + lane < L ? buf[base + lane] : 'x';
        unsigned long long mt = __ballot(c == '\t');
        ctrl |= __ballot(c <= 10u && c != '\t') != 0ull;
        while (mt) {
            const int b = __ffsll((long long)mt) - 1;
            if (ntab < 6) tp[ntab] = base + b;
            ++ntab;
            mt &= mt - 1;
        }
    }
    if (ctrl || ntab != 6 || tp[0] == 0) { decline(); return; }
    long long pos = 0;
    {
        const int d0 = tp[0] + 1, d1 = tp[1];
        if (d1 == d0 || d1 - d0 > 15) { decline(); return; }
        bool okd = true;
        for (int i = d0; i < d1; ++i) { const unsigned d = unsigned(buf[i]) - '0'; okd &= d < 10u; pos = pos * 10 + (long long)d; }
        if (!okd) { decline(); return; }
    }
    const int b0 = tp[3] + 1, b1 = tp[4], qs = b1 + 1, t6 = tp[5], ms = t6 + 1, blen = b1 - b0;
    for (int i = lane; i < blen + 1; i += 64) flg[i] = 0;
    __builtin_amdgcn_wave_barrier();
    // ---- bytes consumed by a live '^' ----
    {
        unsigned long long prev = 0ull;
        bool tail = false;                                     // the field's last byte is a live '^': it would consume the tab
        for (int base = 0; base < blen; base += 64) {
            const unsigned c = base + lane < blen ? buf[b0 + base + lane] : 0u;
            unsigned long long bs = __ballot(c == '^');
            bs &= ~prev;
            const unsigned long long follows = (bs << 1) | prev, even = 0x5555555555555555ull;
            const unsigned long long odd_starts = bs & ~even & ~follows;
            const unsigned long long seq_even = odd_starts + bs;
            prev = seq_even < odd_starts ? 1ull : 0ull;
            const unsigned long long escaped = (even ^ (seq_even << 1)) & follows;
            if ((escaped >> lane) & 1ull) flg[base + lane] |= F_CONS;
            if (blen - base < 64) tail = ((escaped >> (blen - base)) & 1ull) != 0ull;
            else if (base + 64 >= blen) tail = prev != 0ull;
        }
        if (tail) { decline(); return; }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- indel tokens ----
    int ntok = 0;
    for (int base = 0; base < blen; base += 64) {
        const bool in = base + lane < blen;
        const unsigned c = in ? buf[b0 + base + lane] : 0u;
        unsigned long long m = __ballot(in && (c == '+' || c == '-') && !(flg[base + lane] & (F_CONS | F_SKIP)));
        while (m) {
            const int p = base + __ffsll((long long)m) - 1;
            m &= m - 1;
            long long adv = 0;
            int nd = 0;
            while (p + 1 + nd < blen && unsigned(buf[b0 + p + 1 + nd]) - '0' < 10u && nd < 9) { adv = adv * 10 + (buf[b0 + p + 1 + nd] - '0'); ++nd; }
            const int seq0 = p + 1 + nd;
            if (nd == 0 || nd >= 9 || adv == 0 || seq0 + adv > blen || ntok >= TOKW_TOK) { decline(); return; }
            bool bad = false;
            for (int i = lane; i < int(adv); i += 64) {
                const unsigned ch = buf[b0 + seq0 + i];
                bad |= ch == '+' || ch == '-' || ch == '^';
            }
            if (__ballot(bad)) { decline(); return; }
            for (int i = lane; i < nd + 1 + int(adv); i += 64) flg[p + i] |= F_SKIP;
            if (lane == 0) { tok[ntok][0] = p; tok[ntok][1] = b0 + seq0; tok[ntok][2] = int(adv); tok[ntok][3] = buf[b0 + p] == '+' ? 1 : 2; }
            ++ntok;
            __builtin_amdgcn_wave_barrier();
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- read-bases: count, and the carrier of every token ----
    int nt = 0, last_code = -1, tnext = 0;
    for (int base = 0; base < blen; base += 64) {
        const bool in = base + lane < blen;
        const unsigned c = in ? buf[b0 + base + lane] : 0u;
        const int cl = char_class(c);
        const unsigned long long mb = __ballot(in && cl < 12 && !(flg[base + lane] & (F_CONS | F_SKIP)));
        while (tnext < ntok && tok[tnext][0] < base + 64) {
            const int off = tok[tnext][0] - base;
            const unsigned long long mlow = off > 0 ? mb & ((off >= 64 ? ~0ull : (1ull << off)) - 1ull) : 0ull;
            const int idx = nt + __popcll(mlow) - 1;
            const int code = mlow ? char_class(buf[b0 + base + 63 - __clzll((long long)mlow)]) : last_code;
            if (lane == 0) { tok[tnext][4] = idx; tok[tnext][5] = code; }
            ++tnext;
        }
        if (mb) last_code = char_class(buf[b0 + base + 63 - __clzll((long long)mb)]);
        nt += __popcll(mb);
    }
    __builtin_amdgcn_wave_barrier();
    if (nt > kMaxDepth || t6 != qs + nt || L != ms + nt) { decline(); return; }
    {   // quality characters: printable (phred 0 .. 94)
        bool bad = false;
        for (int i = lane; i < nt; i += 64) bad |= (unsigned(buf[qs + i]) - 33u > 94u) | (unsigned(buf[ms + i]) - 33u > 94u);
        if (__ballot(bad)) { decline(); return; }
    }
    // a read-base annotated twice keeps its last annotation; a token in front of the first read-base is not this path's
    int kept[TOKW_TOK], nkept = 0;
    for (int t = 0; t < ntok; ++t) {
        if (tok[t][4] < 0) { decline(); return; }
        if (t + 1 < ntok && tok[t + 1][4] == tok[t][4]) continue;
        kept[nkept++] = t;
    }
    // distinct keys, first seen first (wave-uniform work on a handful of tokens)
    int kid[TOKW_TOK], nk = 0;
    for (int x = 0; x < nkept; ++x) {
        const int t = kept[x];
        int found = -1;
        for (int y = 0; y < x && found < 0; ++y) {
            const int u = kept[y];
            if (tok[u][2] != tok[t][2] || tok[u][3] != tok[t][3] || tok[u][5] != tok[t][5]) continue;
            bool eq = true;
            for (int k = 0; k < tok[t][2] && eq; ++k) eq = buf[tok[u][1] + k] == buf[tok[t][1] + k];
            if (eq) found = kid[y];
        }
        kid[x] = found >= 0 ? found : nk++;
    }
    const long long ri = pos - a.ref_start;
    if constexpr (!FILL) {
        if (lane == 0) {
            a.row_nt[r] = nt;
            a.row_nk[r] = nk;
            a.row_pos[r] = int(min(pos, (long long)0x7fffffff));
            if (ri < 0 || ri >= a.ref_len || pos > 0x7fffffffLL) atomicMax(&a.fl->oob, 1);
        }
        return;
    } else {
        const long long e0 = a.col_off[r];
        const int k0 = a.key_off[r];
        if (lane == 0) {
            if (r > 0 && a.row_pos[r] <= a.row_pos[r - 1]) atomicMax(&a.fl->bad_order, 1);
            const unsigned char rb = a.ref[ri];
            const unsigned char ru = up_c(rb);
            a.col_pos[r] = int(pos);
            a.col_ref[r] = static_cast<unsigned char>(ref_code_dev(rb) | ((ru == 'A' || ru == 'C' || ru == 'G' || ru == 'T') ? 0 : 0x80));
        }
        // entries: every read-base lane writes its own (index = prefix popcount), indel bits looked up in the kept tokens
        int n0 = 0;
        for (int base = 0; base < blen; base += 64) {
            const bool in = base + lane < blen;
            const unsigned c = in ? buf[b0 + base + lane] : 0u;
            const int cl = char_class(c);
            const bool isb = in && cl < 12 && !(flg[base + lane] & (F_CONS | F_SKIP));
            const unsigned long long mb = __ballot(isb);
            if (isb) {
                const int idx = n0 + __popcll(mb & below);
                unsigned e = unsigned(cl) | ((unsigned(buf[qs + idx]) - 33u) << 6) | ((unsigned(buf[ms + idx]) - 33u) << 13);
                for (int x = 0; x < nkept; ++x) {
                    const int t = kept[x];
                    if (tok[t][4] == idx) {
                        const int tk = tok[t][3];
                        const int gate = tk == 1 ? tok[t][2] : tok[t][2] + 1;
                        e |= unsigned(gate > a.max_indel_length ? 3 : tk) << 4;
                        e |= unsigned(kid[x]) << 21;
                    }
                }
                a.entries[e0 + idx] = e;
            }
            n0 += __popcll(mb);
        }
        // the row's distinct keys (first occurrences): meta, merged group, string length, where k_key_strings finds the sequence
        int grp[TOKW_TOK], ng = 0;
        for (int x = 0; x < nkept; ++x) {
            bool first = true;
            for (int y = 0; y < x; ++y) if (kid[y] == kid[x]) { first = false; break; }
            if (!first) continue;
            const int t = kept[x];
            const int tk = tok[t][3], code = tok[t][5], sl = tok[t][2];
            const int gate = tk == 1 ? sl : sl + 1;
            const bool overlong = gate > a.max_indel_length;
            const bool fwd = code < 4 || code == 8 || code == 10;
            const unsigned char anchors[12] = {'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T', '*', '#', 'N', 'N'};
            const unsigned char anchor = tk == 1 ? anchors[code] : static_cast<unsigned char>('D');
            int g = -1;
            for (int y = 0; y < x && g < 0; ++y) {
                bool yfirst = true;
                for (int z = 0; z < y; ++z) if (kid[z] == kid[y]) { yfirst = false; break; }
                if (!yfirst) continue;
                const int u = kept[y];
                if (tok[u][3] != tk || tok[u][2] != sl) continue;
                if (tk == 2) { g = grp[y]; break; }
                if (anchors[tok[u][5]] != anchor) continue;
                bool eq = true;
                for (int k = 0; k < sl && eq; ++k) eq = up_c(buf[tok[u][1] + k]) == up_c(buf[tok[t][1] + k]);
                if (eq) g = grp[y];
            }
            if (g < 0) g = ng++;
            grp[x] = g;
            if (lane == 0) {
                const int k = k0 + kid[x];
                a.key_meta[k] = static_cast<unsigned char>(tk | (fwd ? 4 : 0) | (overlong ? 8 : 0));
                a.key_group[k] = g;
                long long take = min((long long)(sl + 1), (long long)a.max_indel_length);
                take = min(take, a.ref_len - ri);
                a.key_len[k] = tk == 1 ? 2 + sl : 1 + int(take);
                a.key_seq[k] = tk == 1 ? cur + tok[t][1] : ri;
                a.key_info[k] = (sl << 8) | (code << 4) | tk;
            }
        }
    }
}

// zero-byte detector on eight bytes at once: bit 7 of every byte of the result that was '\n' in x
__device__ __forceinline__ unsigned long long newline_mask(unsigned long long x) {
    const unsigned long long y = x ^ 0x0a0a0a0a0a0a0a0aull, m = 0x7f7f7f7f7f7f7f7full;
    return ~(((y & m) + m) | y | m);                          // the exact form: no borrow from a zero byte into its neighbour
}

// The '\n's of a 16-byte piece that start a row (a '\n' at p makes a row start at p + 1; the text's last byte starts none): the two words'
// masks, bit 7 of a byte's place set.  A thread takes a piece, so a wavefront reads 1 KB of consecutive text (a thread per 256-byte
// segment read the same bytes sixteen cache lines apart: 18 + 38 us for the two kernels below, against 22 MB at HBM speed).
__device__ __forceinline__ void piece_masks(const unsigned char* __restrict__ text, long long len, long long q, unsigned long long& m0, unsigned long long& m1) {
    m0 = m1 = 0ull;
    if (q >= len) return;
    const uint4 v = *reinterpret_cast<const uint4*>(text + q);           // (the device copy is padded well past len)
    m0 = newline_mask((unsigned long long)v.x | ((unsigned long long)v.y << 32));
    m1 = newline_mask((unsigned long long)v.z | ((unsigned long long)v.w << 32));
    const long long ok = len - 1 - q;                                    // bytes of the piece whose '\n' would start a row
    if (ok < 16) {
        const int k0 = int(max(0LL, min(8LL, ok))), k1 = int(max(0LL, min(8LL, ok - 8)));
        m0 &= k0 >= 8 ? ~0ull : ((1ull << (8 * k0)) - 1ull);
        m1 &= k1 >= 8 ? ~0ull : ((1ull << (8 * k1)) - 1ull);
    }
}
constexpr int PIECES = SEG / 16;
static_assert(PIECES == 16, "a segment's pieces are a 16-lane DPP row");

// rows that start behind a '\n' of each segment (+ the row at byte 0)
__global__ __launch_bounds__(256) void k_count_lines(const unsigned char* __restrict__ text, long long len, int n_seg, int* __restrict__ cnt) {
    const long long piece = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long m0, m1;
    piece_masks(text, len, piece * 16, m0, m1);
    int c = __popcll(m0) + __popcll(m1) + ((piece == 0 && len > 0) ? 1 : 0);
#pragma unroll
    for (int o = 1; o < PIECES; o <<= 1) c += __shfl_xor(c, o);
    const long long s = piece / PIECES;
    if ((threadIdx.x & (PIECES - 1)) == 0 && s < n_seg) cnt[s] = c;
}

// row index -> byte offset of the row's first character: a piece's rows follow those of the pieces before it in its segment
__global__ __launch_bounds__(256) void k_row_starts(const unsigned char* __restrict__ text, long long len, int n_seg, const int* __restrict__ seg_base,
                                                    long long* __restrict__ row_start) {
    const long long piece = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long q = piece * 16, s = piece / PIECES;
    unsigned long long m0, m1;
    piece_masks(text, len, q, m0, m1);
    const int mine = __popcll(m0) + __popcll(m1) + ((piece == 0 && len > 0) ? 1 : 0);
    int before = mine;                                      // inclusive prefix over the segment's pieces
    const int l16 = threadIdx.x & (PIECES - 1);
#pragma unroll
    for (int o = 1; o < PIECES; o <<= 1) { const int t = __shfl_up(before, o); if (l16 >= o) before += t; }
    if (s >= n_seg || mine == 0) return;
    int row = seg_base[s] + before - mine;
    if (piece == 0) row_start[row++] = 0;
    while (m0) { const int b = __ffsll((long long)m0) - 1; row_start[row++] = q + (b >> 3) + 1; m0 &= m0 - 1; }
    while (m1) { const int b = __ffsll((long long)m1) - 1; row_start[row++] = q + 8 + (b >> 3) + 1; m1 &= m1 - 1; }
}
// One lane per row (a lane per segment that parsed "its" rows where it found them ran the parser once per '\n' position of the
// wavefront, one or two lanes at a time: 4.5 ms instead of 0.3).  The 64 rows of a wavefront are consecutive lines, i.e. ONE contiguous
// span of the text (~11 KB at 50x): the wave copies it into LDS with coalesced 16-byte loads and the lanes parse from there - a lane's
// byte stream then refills from LDS instead of waiting out an L2 round trip per 8 bytes, three streams at a time in the fill pass.
// Spans that do not fit (deep columns) are parsed from HBM as before; the decision is the wavefront's.
constexpr int TOKL_CAP = 16384;
template <bool FILL>
__global__ __launch_bounds__(64) void k_rows_lanes(RowArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char s_text[TOKL_CAP];
    const int lane = threadIdx.x;
    const int r0 = blockIdx.x * 64, r = r0 + lane;
    const int r1 = min(r0 + 64, a.n_rows);
    const long long span0 = a.row_start[r0], span1 = r1 < a.n_rows ? a.row_start[r1] : a.len;
    const long long a0 = span0 & ~15LL;
    const long long need = span1 - a0 + 48;                 // a stream looks 40 bytes past the byte it stands on
    __shared__ unsigned char s_cls[256];
    for (int i = lane; i < 256; i += 64) s_cls[i] = kCharClass[i];
    __syncthreads();
    TextRef T{a.text, 0, a.len, s_cls};
    if (need <= TOKL_CAP) {
        for (long long i = lane * 16LL; i < need; i += 64 * 16) *reinterpret_cast<uint4*>(s_text + i) = *reinterpret_cast<const uint4*>(a.text + a0 + i);
        __syncthreads();
        T = TextRef{s_text, a0, span1 - a0, s_cls};
    }
    if (r >= a.n_rows) return;
    const long long cur = a.row_start[r];
    const bool staged = T.t != a.text;                       // (the wavefront's decision)
    if (FILL) {
        if (r > 0 && a.row_pos[r] <= a.row_pos[r - 1]) atomicMax(&a.fl->bad_order, 1);
        if (staged) fill_row<LdsStream>(a, T, cur - T.base, r);
        else fill_row<ByteStream>(a, T, cur - T.base, r);
    } else {
        const bool ok = staged ? count_row<LdsStream>(a, T, cur - T.base, r) : count_row<ByteStream>(a, T, cur - T.base, r);
        if (!ok) atomicMax(&a.fl->slow, int(min(cur + 1, (long long)0x7fffffff)));
    }
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
    Buf text, ref, seg_cnt, seg_base, row_start, row_nt, row_nk, row_pos, row_b0, row_blen, col_off, key_off, entries, col_pos, col_ref, key_meta, key_group,
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
    if ((rc = cx->text.ensure(len + 128)) || (rc = cx->ref.ensure(ref_len + 16)) || (rc = cx->seg_cnt.ensure(size_t(n_seg) * 4)) ||
        (rc = cx->seg_base.ensure(size_t(n_seg + 1) * 4)) || (rc = cx->tiles.ensure(size_t(cdiv(std::max<int64_t>(n_seg, int64_t(len / 8)), SCAN_TILE) + 2) * 8)) ||
        (rc = cx->flags.ensure(sizeof(TokFlags) + 64)) || (rc = cx->pin(&cx->h_stage, &cx->h_stage_cap, 4096)))
        return rc;
    CTO_HIP(hipMemsetAsync(cx->text.as<char>() + (len & ~size_t(15)), '\n', 96 + (len & 15), s));     // the readers look up to five words past the end
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
    hipLaunchKernelGGL(k_count_lines, dim3(unsigned(cdiv(int64_t(n_seg) * PIECES, 256))), dim3(256), 0, s, d_text, (long long)len, n_seg, cx->seg_cnt.as<int>());
    if ((rc = scan_exclusive<int>(s, cx->seg_cnt.as<int>(), n_seg, cx->seg_base.as<int>(), cx->tiles.as<int>(), &fl->n_rows))) return rc;
    if ((rc = fetch_flags())) return rc;
    const int n_rows = hf->n_rows;
    if (n_rows <= 0) { *fallback = 1; return CTO_OK; }
    if ((rc = cx->row_start.ensure(size_t(n_rows) * 8)) || (rc = cx->row_nt.ensure(size_t(n_rows) * 4)) || (rc = cx->row_nk.ensure(size_t(n_rows) * 4)) ||
        (rc = cx->row_pos.ensure(size_t(n_rows) * 4)) || (rc = cx->row_b0.ensure(size_t(n_rows) * 4)) || (rc = cx->row_blen.ensure(size_t(n_rows) * 4)) || (rc = cx->col_off.ensure(size_t(n_rows + 1) * 8)) || (rc = cx->key_off.ensure(size_t(n_rows + 1) * 4)) ||
        (rc = cx->col_pos.ensure(size_t(n_rows) * 4)) || (rc = cx->col_ref.ensure(size_t(n_rows) + 16)) ||
        (rc = cx->tiles.ensure(size_t(cdiv(std::max(n_rows, n_seg), SCAN_TILE) + 2) * 8)))
        return rc;
    RowArgs a{};
    a.text = d_text; a.len = (long long)len; a.ref = cx->ref.as<unsigned char>(); a.ref_start = ref_start; a.ref_len = (long long)ref_len;
    a.max_indel_length = max_indel_length;
    a.row_start = cx->row_start.as<long long>(); a.row_nt = cx->row_nt.as<int>(); a.row_nk = cx->row_nk.as<int>(); a.row_pos = cx->row_pos.as<int>();
    a.row_b0 = cx->row_b0.as<int>(); a.row_blen = cx->row_blen.as<int>();
    a.n_rows = n_rows; a.fl = fl;
    CTO_HIP(hipMemsetAsync(cx->row_nt.p, 0, size_t(n_rows) * 4, s));        // rows the pass declines leave theirs unwritten
    CTO_HIP(hipMemsetAsync(cx->row_nk.p, 0, size_t(n_rows) * 4, s));
    CTO_HIP(hipMemsetAsync(cx->row_pos.p, 0, size_t(n_rows) * 4, s));
    hipLaunchKernelGGL(k_row_starts, dim3(unsigned(cdiv(int64_t(n_seg) * PIECES, 256))), dim3(256), 0, s, d_text, (long long)len, n_seg, cx->seg_base.as<int>(),
                       cx->row_start.as<long long>());
    // a lane per row (the default: 64 rows share a wavefront's instruction stream, ~190 instructions per row); CTO_TOK_WAVES=1: a wavefront
    // per row (ballots over 64 bytes at a time: ~800 instructions per row - measured slower, 0.67 against 0.52 ms for the two passes - kept as the
    // second statement of the pass the tests hold the first one to)
    const char* tw = getenv("CTO_TOK_WAVES");
    const bool by_lanes = !(tw && tw[0] == '1');
    if (by_lanes) hipLaunchKernelGGL(k_rows_lanes<false>, dim3(unsigned(cdiv(n_rows, 64))), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(k_rows_wave<false>, dim3(unsigned(cdiv(n_rows, TOKW_WAVES))), dim3(64 * TOKW_WAVES), 0, s, a);
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
    if (by_lanes) hipLaunchKernelGGL(k_rows_lanes<true>, dim3(unsigned(cdiv(n_rows, 64))), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(k_rows_wave<true>, dim3(unsigned(cdiv(n_rows, TOKW_WAVES))), dim3(64 * TOKW_WAVES), 0, s, a);
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
