// mpileup TEXT -> column pack on the device (the tokeniser of decode_pileup_bases, and F9 / F10's row handling).
//
// The reference reads `samtools mpileup` text row by row in Python (src/create_tensor_pileup_calling.py:465-532 splits the row,
// :120-144 walks the base string character by character); csrc/pack.cpp does the same on host threads (one forward pass per row,
// 1.5 GB/s per thread) and was what a text-fed run waited for: 28 ms of CPU per 22 MB chunk against 1.9 ms of device time.  Here
// the text goes up as it is and the pack is born in HBM, as it is for BAM input (csrc/pileup.hip):
//   k_count_lines   a thread per 16-byte piece: rows that start in each 256-byte segment         -> one scan launch -> k_row_starts
//   k_rows_walk     one lane per row, ONE forward pass (pack.cpp's fast_row: contig, position, reference base, depth, base string with
//                   ^x / $ / +n.. / -n..): the read-bases' codes (a byte each, in the text's own layout), the indel tokens as a chain of
//                   records in HBM, DISTINCT indel keys per row (first-seen order, compared on the characters), counts
//                   -> one scan launch (k_apply3: col_off, key_off, key-string offsets)
//   k_expand        a thread per read-base: entries (code | kind << 4 | BQ << 6 | MQ << 13 | key id << 21) from three coalesced byte runs,
//                   col_pos, col_ref; checks the quality strings and their separators
//   k_row_keys      one lane per row with keys: meta byte, merged candidate-extraction group, "I<ANCHOR><SEQ>" / "D<reference slice>"
// A row is a chain of dependent byte reads, so a lane is slow - but there are 140 000 rows in a 4096-site chunk, and a wavefront's
// 64 rows are ~10 KB of consecutive text that stay in the vector L1 while its lanes walk them.
// Anything the single pass does not take - another field count, a short quality string, '\r', a byte outside the printable range,
// an indel or '^' running into the field's end, more than 256 indel-carrying read-bases in a row, an empty row, text that does not
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
constexpr int MAX_IND = 256;        // indel-carrying read-bases of one row this path takes (interning compares every pair)

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
        // tile_base == nullptr: no tile sums at all - the workgroup adds up the ELEMENTS in front of its tile (a chunk's 86 000 segment counts:
        // the last of 21 workgroups reads 340 KB out of the L2; tile sums by atomics from the counting kernel cost it 0.8 ms - 5 400 atomics
        // on 21 addresses)
        T b = 0;
        if (tile_base == nullptr) {
            // 16-byte loads, eight in flight per thread (one element per trip waited out an L2 round trip per element: 45 us)
            const int4* in4 = reinterpret_cast<const int4*>(in);
            const int n4 = int(blockIdx.x) * (SCAN_TILE / 4);
            int i = threadIdx.x;
            for (; i + 7 * 256 < n4; i += 8 * 256) {
                int4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = in4[i + k * 256];
#pragma unroll
                for (int k = 0; k < 8; ++k) b += T(v[k].x) + T(v[k].y) + T(v[k].z) + T(v[k].w);
            }
            for (; i < n4; i += 256) { const int4 v = in4[i]; b += T(v.x) + T(v.y) + T(v.z) + T(v.w); }
        }
        else for (int i = threadIdx.x; i < int(blockIdx.x); i += 256) b += tile_base[i];
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

// An indel-carrying read-base of a row, in HBM: the records of a row form a chain in the order of the row's read-bases.  A record's slot is
// its sign's byte offset in the text / 4 - two tokens lie at least four bytes apart ("+1A" and the next read-base), so slots never collide,
// no allocation and no counter is needed, and a row may carry any number of them (the lane keeps the first / last slot in registers; the
// private arrays of the first form of this file - 784-928 bytes of scratch per lane - are gone).
struct TokRec {
    int len;        // bases inserted / deleted
    int info;       // read-base index << 2 | kind (1 insertion, 2 deletion) - 17 bits; code << 17 (4); sign offset & 3 << 21; digits << 23 (4)
    int next;       // slot of the row's next record, -1 at the end
    int kg;         // key id within the row (first seen first) | merged candidate-extraction group << 16 (k_row_keys)
};
__device__ __forceinline__ int tok_idx(int info) { return (info >> 2) & 0x7fff; }
__device__ __forceinline__ int tok_kind(int info) { return info & 3; }
__device__ __forceinline__ int tok_code(int info) { return (info >> 17) & 15; }
__device__ __forceinline__ long long tok_seq(int slot, int info) { return ((long long)slot << 2) + ((info >> 21) & 3) + 1 + ((info >> 23) & 15); }

struct RowArgs {
    const unsigned char* text; long long len;
    const unsigned char* ref; long long ref_start, ref_len;
    int max_indel_length;
    int n_rows;
    const long long* row_start;
    // the walk (k_rows_walk) writes, the later kernels read
    int* row_nt; int* row_nk; int* row_pos; int* row_b0; int* row_blen; int* row_tok; int* row_str;
    unsigned char* codes;            // the read-base codes of a row at [row start + b0 + i]: the walk's product, the text's own layout
    TokRec* tok;
    long long* tile_nt; int* tile_nk; long long* tile_str;      // sums per SCAN_TILE rows (atomics of the walk's wavefronts)
    // after the scans
    const long long* col_off; const int* key_off; const long long* row_str_off;
    unsigned* entries; int* col_pos; unsigned char* col_ref;
    unsigned char* key_meta; int* key_group; long long* str_off; char* key_str;
    TokFlags* fl;
};

// A forward reader of the text: the next <= 8 bytes sit in a register and a byte costs a shift, the following aligned 8 bytes are
// requested one refill ahead.  (A row parsed through one dependent global byte load per character - the first form of this file -
// is a chain of ~300 L2 round trips: 250 us per launch however many rows are in flight.)  The device copy of the text is padded, so
// the aligned word that holds the last byte may be read whole.  Used for the spans that do not fit the LDS (deep columns).
struct ByteStream {
    const unsigned char* t;
    unsigned char* codes;              // where put() stores: the codes buffer at this row's text offsets
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
    __device__ __forceinline__ void skip(long long n) { if (n < have) { bits >>= 8 * int(n); have -= int(n); pos += n; } else seek(t, pos + n); }
    __device__ __forceinline__ unsigned at(long long p) const { return t[p]; }
    __device__ __forceinline__ void put(long long p, int code) { codes[p] = static_cast<unsigned char>(code); }
};

// The same reader over a wavefront's staged copy of its rows: a byte is one ds_read_u8 at a 32-bit index - no shift register, no
// refill branch, no 64-bit position.  PMC (tools/tokenise_pmc.sh), count pass, per wavefront of 64 rows: the register reader on the LDS
// copy 9.1 k vector + 17.8 k scalar + 3.4 k branch instructions, this reader 5.1 k + 16.7 k + 2.8 k (170 -> 156 us; fill 283 -> 265).
// A CU issues one instruction per cycle whatever its kind, and two thirds of them are SCALAR: the exec-mask bookkeeping of divergent
// loops and branches (8-13 s_* per trip: s_and_saveexec, s_or / s_andn2 on exec, s_cbranch) - 64 rows that sit at different places of
// their grammar.  (Most of those scalar instructions turned out to be char_class's `switch`: as a table, 33.8 M -> 9.1 M per launch and
// 156 -> 82 us.  A branch-free pass - one wave-uniform loop, states moved by selects - was written and measured before that: 209 us.)
// put(): the code of a row's i-th read-base goes IN PLACE over the row's own base string - byte b0 + i lies at or before the byte being
// read, every read-base costs at least one byte - and the wavefront copies its span out to the codes buffer when its lanes are done.
typedef __attribute__((address_space(3))) unsigned char* lds_bytes;
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
    __device__ __forceinline__ void skip(long long n) { pos += int(n); }
    __device__ __forceinline__ unsigned at(long long p) const { return t[int(p)]; }
    __device__ __forceinline__ void put(long long p, int code) { t[int(p)] = static_cast<unsigned char>(code); }
};

// Where a row's bytes are read: the text in HBM (base 0), or a wavefront's staged copy of its rows in LDS (t[0] = text[base]; offsets
// below are relative to t, `len` = the bytes that belong to rows - what lies behind is look-ahead padding)
struct TextRef { const unsigned char* t; long long base, len; const unsigned char* cls; };      // cls: the class table in LDS (or null)

struct RowToks { int n, first, last, before_last, last_idx; };

// The base string from `st` on: counts read-bases, stores their codes, chains the indel tokens.  Returns false when the single pass declines the row.
template <class Stream>
__device__ __forceinline__ bool walk_bases(const RowArgs& a, Stream& st, const TextRef& T, int& nt, RowToks& tk) {
    lds_table cls = (lds_table)T.cls;
    nt = 0;
    tk.n = 0; tk.first = tk.last = tk.before_last = -1; tk.last_idx = -1;
    int last_code = 0;
    const long long b0 = st.pos;
    for (;;) {
        const unsigned c = st.peek();
        const int cl = char_class(cls, c);
        if (cl < 12) { last_code = cl; st.put(b0 + nt, cl); ++nt; st.step(); }
        else if (cl == 14) st.step();
        else if (cl == 13) { if (st.peek1() <= 10u) return false; st.step(); st.step(); }
        else if (cl == 12) {
            const int kind = c == '+' ? 1 : 2;
            const long long sign_abs = T.base + st.pos;
            st.step();
            long long adv = 0;
            int nd = 0;
            while (st.peek() - '0' < 10u) { adv = adv * 10 + (st.peek() - '0'); st.step(); ++nd; if (adv > (1 << 24) || nd > 15) return false; }
            if (nt == 0 || nt > kMaxDepth || st.pos + adv > T.len) return false;
            for (long long k = 0; k < adv; ++k) if (st.at(st.pos + k) <= 10u) return false;
            const int slot = int(sign_abs >> 2);
            const bool replace = tk.n > 0 && tk.last_idx == nt - 1;       // a second annotation of the same read-base replaces the first
            if (!replace && tk.n >= MAX_IND) return false;
            a.tok[slot] = TokRec{int(adv), ((nt - 1) << 2) | kind | (last_code << 17) | (int(sign_abs & 3) << 21) | (nd << 23), -1, 0};
            if (replace) {
                if (tk.n == 1) tk.first = slot; else a.tok[tk.before_last].next = slot;
            } else {
                if (tk.n == 0) tk.first = slot; else a.tok[tk.last].next = slot;
                tk.before_last = tk.last;
                ++tk.n;
            }
            tk.last = slot;
            tk.last_idx = nt - 1;
            st.skip(adv);
        } else break;
    }
    return true;
}

// distinct keys of a row, first seen first: Counter key = read-base code + sign + sequence, case-sensitive (pack.cpp: intern_indel); the
// sequences are compared on the text in HBM (the LDS copy's base strings are being overwritten with codes).  Returns the number of distinct
// keys; *str_bytes = the bytes of their alt_info strings
__device__ int intern_row(const RowArgs& a, const RowToks& tk, long long ri, int* str_bytes) {
    int nk = 0, sb = 0;
    int si = tk.first;
    for (int i = 0; i < tk.n; ++i) {
        const TokRec ri_ = a.tok[si];
        int found = -1;
        int sj = tk.first;
        for (int j = 0; j < i && found < 0; ++j) {
            const TokRec rj = a.tok[sj];
            if (rj.len == ri_.len && tok_kind(rj.info) == tok_kind(ri_.info) && tok_code(rj.info) == tok_code(ri_.info)) {
                const unsigned char* x = a.text + tok_seq(sj, rj.info);
                const unsigned char* y = a.text + tok_seq(si, ri_.info);
                bool eq = true;
                for (int k = 0; k < ri_.len && eq; ++k) eq = x[k] == y[k];
                if (eq) found = rj.kg & 0xffff;
            }
            sj = rj.next;
        }
        if (found < 0) {
            found = nk++;
            long long take = min((long long)(ri_.len + 1), (long long)a.max_indel_length);
            take = min(take, a.ref_len - ri);
            sb += tok_kind(ri_.info) == 1 ? 2 + ri_.len : 1 + int(take);
        }
        a.tok[si].kg = found;
        si = ri_.next;
    }
    *str_bytes = sb;
    return nk;
}

// pass 1 of a row: the single forward pass of pack.cpp's fast_row up to the end of the base string; writes the row's counts, the codes of
// its read-bases and its token chain; false = not a row this path takes.  (The quality and mapping-quality strings are not walked here:
// k_expand reads them, coalesced, and checks them and their separators on the text in HBM.)
template <class Stream>
__device__ bool walk_row(const RowArgs& a, const TextRef& T, Stream& st, long long cur, int row, int& nt_out, int& nk_out, int& sb_out) {
    const long long len = T.len;
    st.seek(T.t, cur);
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
    RowToks tk;
    if (!walk_bases(a, st, T, nt, tk)) return false;
    if (st.peek() != '\t' || nt > kMaxDepth) return false;
    const long long blen = st.pos - b0;
    // as many quality and mapping-quality characters as read-bases and the end of the row must lie inside the text (k_expand looks at them)
    if (st.pos + 1 + 2LL * nt + 1 >= len) return false;
    const long long ri = pos - a.ref_start;
    const bool oob = ri < 0 || ri >= a.ref_len || pos > 0x7fffffffLL;
    if (oob) atomicMax(&a.fl->oob, 1);
    int sb = 0;
    const int nk = (tk.n > 0 && !oob) ? intern_row(a, tk, ri, &sb) : 0;
    a.row_nt[row] = nt;
    a.row_nk[row] = nk;
    a.row_str[row] = sb;
    a.row_tok[row] = tk.first;
    a.row_pos[row] = int(min(pos, (long long)0x7fffffff));
    a.row_b0[row] = int(b0 - cur);
    a.row_blen[row] = int(blen);
    nt_out = nt; nk_out = nk; sb_out = sb;
    return b0 - cur < (1LL << 30);
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
// byte stream then reads LDS instead of waiting out an L2 round trip per 8 bytes.  Spans that do not fit (deep columns) are parsed from
// HBM; the decision is the wavefront's.  Round 6: ONE walk per row.  The walk leaves the read-bases' codes (a byte each, in the text's own
// layout), the row's token chain and its counts; entries are then written by k_expand - a thread per read-base, coalesced - and the key tables
// by k_row_keys.  (Rounds 4-5 walked every base string twice - counts first, entries after the scans - with the tokens of a row in a lane's
// private memory: 0.07 + 0.12 ms and 784-928 bytes of scratch per lane.)
constexpr int TOKL_CAP = 16384;
__global__ __launch_bounds__(64) void k_rows_walk(RowArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char s_text[TOKL_CAP];
    __shared__ unsigned char s_cls[256];
    const int lane = threadIdx.x;
    const int r0 = blockIdx.x * 64, r = r0 + lane;
    const int r1 = min(r0 + 64, a.n_rows);
    const long long span0 = a.row_start[r0], span1 = r1 < a.n_rows ? a.row_start[r1] : a.len;
    const long long a0 = span0 & ~15LL;
    const long long need = span1 - a0 + 48;                 // a reader looks a few bytes past the byte it stands on
    for (int i = lane; i < 256; i += 64) s_cls[i] = kCharClass[i];
    const bool staged = need <= TOKL_CAP;                    // (the wavefront's decision)
    if (staged)
        for (long long i = lane * 16LL; i < need; i += 64 * 16) *reinterpret_cast<uint4*>(s_text + i) = *reinterpret_cast<const uint4*>(a.text + a0 + i);
    __syncthreads();
    int nt = 0, nk = 0, sb = 0;
    if (r < a.n_rows) {
        const long long cur = a.row_start[r];
        bool ok;
        if (staged) {
            const TextRef T{s_text, a0, span1 - a0, s_cls};
            LdsStream st;
            ok = walk_row<LdsStream>(a, T, st, cur - a0, r, nt, nk, sb);
        } else {
            const TextRef T{a.text, 0, a.len, s_cls};
            ByteStream st;
            st.codes = a.codes;
            ok = walk_row<ByteStream>(a, T, st, cur, r, nt, nk, sb);
        }
        if (!ok) { nt = nk = sb = 0; atomicMax(&a.fl->slow, int(min(cur + 1, (long long)0x7fffffff))); }
    }
    __syncthreads();
    if (staged) {
        // the span's bytes - base strings now hold the codes - out to the codes buffer: exactly [span0, span1), the rows of this wavefront
        const long long lo = span0 - a0, hi = span1 - a0;
        for (long long i = lane * 16LL; i < hi; i += 64 * 16) {
            if (i >= lo && i + 16 <= hi) *reinterpret_cast<uint4*>(a.codes + a0 + i) = *reinterpret_cast<const uint4*>(s_text + i);
            else for (int k = 0; k < 16; ++k) if (i + k >= lo && i + k < hi) a.codes[a0 + i + k] = s_text[i + k];
        }
    }
    // the wavefront's sums into its tile's (SCAN_TILE is a multiple of 64: a wavefront's rows share a tile)
    long long snt = nt, ssb = sb;
    int snk = nk;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { snt += __shfl_xor(snt, o); ssb += __shfl_xor(ssb, o); snk += __shfl_xor(snk, o); }
    if (lane == 0) {
        const int tile = r0 / SCAN_TILE;
        if (snt) atomicAdd(reinterpret_cast<unsigned long long*>(a.tile_nt + tile), (unsigned long long)snt);
        if (snk) atomicAdd(a.tile_nk + tile, snk);
        if (ssb) atomicAdd(reinterpret_cast<unsigned long long*>(a.tile_str + tile), (unsigned long long)ssb);
    }
}

// The three exclusive scans a chunk needs after the walk, in ONE launch: read-bases -> col_off, distinct keys -> key_off, key-string bytes ->
// row_str_off.  A workgroup per SCAN_TILE rows; the tiles' sums came from the walk's atomics, every workgroup adds up those in front of its own.
__global__ __launch_bounds__(256) void k_apply3(const int* __restrict__ row_nt, const int* __restrict__ row_nk, const int* __restrict__ row_str, int n,
                                                const long long* __restrict__ tile_nt, const int* __restrict__ tile_nk, const long long* __restrict__ tile_str,
                                                long long* __restrict__ col_off, int* __restrict__ key_off, long long* __restrict__ str_off, TokFlags* __restrict__ fl) {
    __shared__ long long p_nt[256], p_sb[256];
    __shared__ int p_nk[256];
    long long b_nt = 0, b_sb = 0;
    int b_nk = 0;
    for (int i = threadIdx.x; i < int(blockIdx.x); i += 256) { b_nt += tile_nt[i]; b_nk += tile_nk[i]; b_sb += tile_str[i]; }
    p_nt[threadIdx.x] = b_nt; p_nk[threadIdx.x] = b_nk; p_sb[threadIdx.x] = b_sb;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (int(threadIdx.x) < d) { p_nt[threadIdx.x] += p_nt[threadIdx.x + d]; p_nk[threadIdx.x] += p_nk[threadIdx.x + d]; p_sb[threadIdx.x] += p_sb[threadIdx.x + d]; }
        __syncthreads();
    }
    const long long t_nt = p_nt[0], t_sb = p_sb[0];
    const int t_nk = p_nk[0];
    __syncthreads();
    constexpr int PER = SCAN_TILE / 256;
    const int i0 = blockIdx.x * SCAN_TILE + threadIdx.x * PER;
    long long l_nt[PER], l_sb[PER], s_nt = 0, s_sb = 0;
    int l_nk[PER], s_nk = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        l_nt[k] = s_nt; l_nk[k] = s_nk; l_sb[k] = s_sb;
        if (i0 + k < n) { s_nt += row_nt[i0 + k]; s_nk += row_nk[i0 + k]; s_sb += row_str[i0 + k]; }
    }
    p_nt[threadIdx.x] = s_nt; p_nk[threadIdx.x] = s_nk; p_sb[threadIdx.x] = s_sb;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const bool up = int(threadIdx.x) >= d;
        const long long x = up ? p_nt[threadIdx.x - d] : 0, z = up ? p_sb[threadIdx.x - d] : 0;
        const int y = up ? p_nk[threadIdx.x - d] : 0;
        __syncthreads();
        p_nt[threadIdx.x] += x; p_nk[threadIdx.x] += y; p_sb[threadIdx.x] += z;
        __syncthreads();
    }
    const long long o_nt = t_nt + p_nt[threadIdx.x] - s_nt, o_sb = t_sb + p_sb[threadIdx.x] - s_sb;
    const int o_nk = t_nk + p_nk[threadIdx.x] - s_nk;
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if (i0 + k < n) { col_off[i0 + k] = o_nt + l_nt[k]; key_off[i0 + k] = o_nk + l_nk[k]; str_off[i0 + k] = o_sb + l_sb[k]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) {
        col_off[n] = t_nt + p_nt[255]; key_off[n] = t_nk + p_nk[255]; str_off[n] = t_sb + p_sb[255];
        fl->n_entries = t_nt + p_nt[255]; fl->n_keys = t_nk + p_nk[255]; fl->key_str_bytes = t_sb + p_sb[255];
    }
}

// entries (code | kind << 4 | BQ << 6 | MQ << 13 | key id << 21), col_pos, col_ref: a workgroup per 64 rows, a thread per read-base - the codes,
// the quality and the mapping-quality characters of a row are three runs of consecutive bytes, the entries one run of words.  Also what the
// walk left unchecked: the two strings' separators and their characters' range (phred 0 .. 94), on the text in HBM; rows in position order.
__global__ __launch_bounds__(256) void k_expand(RowArgs a) {
    __shared__ long long s_off[65], s_b0[64];
    __shared__ int s_nt[64];
    const int r0 = blockIdx.x * 64, nr = min(64, a.n_rows - r0);
    const int t = threadIdx.x;
    if (t <= nr) s_off[t] = a.col_off[r0 + t];
    bool bad = false;
    if (t < nr) {
        const int r = r0 + t, nt = a.row_nt[r];
        const long long cur = a.row_start[r], b0 = cur + a.row_b0[r];
        s_b0[t] = b0;
        s_nt[t] = nt;
        const long long pos = a.row_pos[r], ri = pos - a.ref_start;
        if (a.row_b0[r] > 0) {                                 // a row the walk took (a declined one has set `slow` already)
            const long long qs = b0 + a.row_blen[r] + 1;
            bad = a.text[qs + nt] != '\t' || a.text[qs + nt + 1 + nt] != '\n';
            if (ri >= 0 && ri < a.ref_len) {
                const unsigned char rb = a.ref[ri];
                const unsigned char ru = up_c(rb);
                a.col_pos[r] = int(pos);
                a.col_ref[r] = static_cast<unsigned char>(ref_code_dev(rb) | ((ru == 'A' || ru == 'C' || ru == 'G' || ru == 'T') ? 0 : 0x80));
            }
            if (r > 0 && a.row_pos[r] <= a.row_pos[r - 1]) atomicMax(&a.fl->bad_order, 1);
        }
    }
    __syncthreads();
    const long long e0 = s_off[0], e1 = s_off[nr];
    for (long long e = e0 + t; e < e1; e += 256) {
        int lo = 0, hi = nr;                                   // the row of entry e: the last one whose offset is <= e
        while (hi - lo > 1) { const int m = (lo + hi) >> 1; if (s_off[m] <= e) lo = m; else hi = m; }
        const int i = int(e - s_off[lo]), nt = s_nt[lo];
        const long long b0 = s_b0[lo], qs = b0 + a.row_blen[r0 + lo] + 1;
        const unsigned q = unsigned(a.text[qs + i]) - 33u, m = unsigned(a.text[qs + nt + 1 + i]) - 33u;
        bad |= q > 94u || m > 94u;
        a.entries[e] = unsigned(a.codes[b0 + i]) | ((q & 127u) << 6) | ((m & 255u) << 13);
    }
    if (bad) atomicMax(&a.fl->slow, 1);
    __syncthreads();                                           // the entries of this workgroup's rows are written: the indel bits go on top
    if (t < nr) {
        int slot = a.row_tok[r0 + t];
        if (a.row_nk[r0 + t] > 0)
            while (slot >= 0) {
                const TokRec rec = a.tok[slot];
                const int tk = tok_kind(rec.info);
                const int gate = tk == 1 ? rec.len : rec.len + 1;
                a.entries[s_off[t] + tok_idx(rec.info)] |= (unsigned(gate > a.max_indel_length ? 3 : tk) << 4) | (unsigned(rec.kg & 0xffff) << 21);
                slot = rec.next;
            }
    }
}

// the distinct keys of every row that has some: meta byte, merged group (insertions by upper-cased anchor + sequence, deletions by length:
// extract_candidates_calling.py:118-126), the alt_info key string ("I<ANCHOR><SEQ>" upper-cased / "D<reference slice>") and its offset.
// A lane per row; a row's first occurrences are the records whose key id equals the number of distinct ids seen before them.
__global__ __launch_bounds__(64) void k_row_keys(RowArgs a) {
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= a.n_rows) return;
    if (r == a.n_rows - 1) a.str_off[a.key_off[a.n_rows]] = a.row_str_off[a.n_rows];
    if (a.row_nk[r] <= 0) return;
    const int k0 = a.key_off[r], first = a.row_tok[r];
    const long long ri = (long long)a.row_pos[r] - a.ref_start;
    const unsigned char anchors[12] = {'A', 'C', 'G', 'T', 'A', 'C', 'G', 'T', '*', '#', 'N', 'N'};
    long long so = a.row_str_off[r];
    int seen = 0, ng = 0;
    for (int si = first; si >= 0;) {
        const TokRec x = a.tok[si];
        const int kid = x.kg & 0xffff;
        if (kid == seen) {
            ++seen;
            const int tk = tok_kind(x.info), code = tok_code(x.info), sl = x.len;
            const unsigned char anchor = tk == 1 ? anchors[code] : static_cast<unsigned char>('D');
            int g = -1, seen_j = 0;
            for (int sj = first; sj != si && g < 0;) {              // earlier FIRST occurrences only carry a group
                const TokRec y = a.tok[sj];
                if ((y.kg & 0xffff) == seen_j) {
                    ++seen_j;
                    if (tok_kind(y.info) == tk && y.len == sl) {
                        if (tk == 2) g = (y.kg >> 16) & 0xffff;
                        else if (anchors[tok_code(y.info)] == anchor) {
                            const unsigned char* u = a.text + tok_seq(sj, y.info);
                            const unsigned char* v = a.text + tok_seq(si, x.info);
                            bool eq = true;
                            for (int k = 0; k < sl && eq; ++k) eq = up_c(u[k]) == up_c(v[k]);
                            if (eq) g = (y.kg >> 16) & 0xffff;
                        }
                    }
                }
                sj = y.next;
            }
            if (g < 0) g = ng++;
            a.tok[si].kg = kid | (g << 16);
            const int gate = tk == 1 ? sl : sl + 1;
            const bool fwd = code < 4 || code == 8 || code == 10;
            const int k = k0 + kid;
            a.key_meta[k] = static_cast<unsigned char>(tk | (fwd ? 4 : 0) | (gate > a.max_indel_length ? 8 : 0));
            a.key_group[k] = g;
            a.str_off[k] = so;
            char* dst = a.key_str + so;
            if (tk == 1) {
                const unsigned char* v = a.text + tok_seq(si, x.info);
                dst[0] = 'I';
                dst[1] = char(anchor);
                for (int i = 0; i < sl; ++i) dst[2 + i] = char(up_c(v[i]));
                so += 2 + sl;
            } else {
                long long take = min((long long)(sl + 1), (long long)a.max_indel_length);
                take = min(take, a.ref_len - ri);
                dst[0] = 'D';
                for (long long i = 0; i < take; ++i) dst[1 + i] = char(up_c(a.ref[ri + i]));
                so += 1 + take;
            }
        }
        si = x.next;
    }
}

}  // namespace

struct cto_dev_tokeniser {
    Buf text, ref, seg_cnt, seg_base, row_start, row_nt, row_nk, row_pos, row_b0, row_blen, row_tok, row_str, row_str_off, col_off, key_off, entries, col_pos, col_ref,
        key_meta, key_group, str_off, key_str, tiles, row_tiles, codes, tok, flags;
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
    // rows: newline counts per 256-byte segment, one launch for their scan, the row starts
    const int seg_tiles = int(cdiv(n_seg, SCAN_TILE));
    if ((rc = cx->tiles.ensure(size_t(seg_tiles + 2) * 8))) return rc;
    hipLaunchKernelGGL(k_count_lines, dim3(unsigned(cdiv(int64_t(n_seg) * PIECES, 256))), dim3(256), 0, s, d_text, (long long)len, n_seg, cx->seg_cnt.as<int>());
    hipLaunchKernelGGL((k_tile_apply<int, true>), dim3(unsigned(seg_tiles)), dim3(256), 0, s, cx->seg_cnt.as<int>(), n_seg, static_cast<const int*>(nullptr),
                       cx->seg_base.as<int>(), &fl->n_rows);
    CTO_HIP(hipGetLastError());
    if ((rc = fetch_flags())) return rc;
    const int n_rows = hf->n_rows;
    if (n_rows <= 0) { *fallback = 1; return CTO_OK; }
    const int row_tiles = int(cdiv(n_rows, SCAN_TILE));
    const size_t nr = size_t(n_rows);
    if ((rc = cx->row_start.ensure(nr * 8)) || (rc = cx->row_nt.ensure(nr * 4)) || (rc = cx->row_nk.ensure(nr * 4)) || (rc = cx->row_pos.ensure(nr * 4)) ||
        (rc = cx->row_b0.ensure(nr * 4)) || (rc = cx->row_blen.ensure(nr * 4)) || (rc = cx->row_tok.ensure(nr * 4)) || (rc = cx->row_str.ensure(nr * 4)) ||
        (rc = cx->col_off.ensure((nr + 1) * 8)) || (rc = cx->key_off.ensure((nr + 1) * 4)) || (rc = cx->row_str_off.ensure((nr + 1) * 8)) ||
        (rc = cx->col_pos.ensure(nr * 4)) || (rc = cx->col_ref.ensure(nr + 16)) || (rc = cx->codes.ensure(len + 128)) ||
        (rc = cx->tok.ensure((len / 4 + 64) * sizeof(TokRec))) || (rc = cx->row_tiles.ensure(size_t(row_tiles + 1) * 24)))
        return rc;
    RowArgs a{};
    a.text = d_text; a.len = (long long)len; a.ref = cx->ref.as<unsigned char>(); a.ref_start = ref_start; a.ref_len = (long long)ref_len;
    a.max_indel_length = max_indel_length;
    a.row_start = cx->row_start.as<long long>(); a.row_nt = cx->row_nt.as<int>(); a.row_nk = cx->row_nk.as<int>(); a.row_pos = cx->row_pos.as<int>();
    a.row_b0 = cx->row_b0.as<int>(); a.row_blen = cx->row_blen.as<int>(); a.row_tok = cx->row_tok.as<int>(); a.row_str = cx->row_str.as<int>();
    a.codes = cx->codes.as<unsigned char>(); a.tok = cx->tok.as<TokRec>();
    a.tile_nt = cx->row_tiles.as<long long>(); a.tile_str = a.tile_nt + (row_tiles + 1); a.tile_nk = reinterpret_cast<int*>(a.tile_str + (row_tiles + 1));
    a.n_rows = n_rows; a.fl = fl;
    CTO_HIP(hipMemsetAsync(cx->row_tiles.p, 0, size_t(row_tiles + 1) * 24, s));
    // rows the walk declines leave theirs unwritten (one memset over the six adjacent-in-meaning arrays would need one allocation: they are small)
    CTO_HIP(hipMemsetAsync(cx->row_nt.p, 0, nr * 4, s));
    CTO_HIP(hipMemsetAsync(cx->row_nk.p, 0, nr * 4, s));
    CTO_HIP(hipMemsetAsync(cx->row_pos.p, 0, nr * 4, s));
    CTO_HIP(hipMemsetAsync(cx->row_blen.p, 0, nr * 4, s));
    CTO_HIP(hipMemsetAsync(cx->row_b0.p, 0, nr * 4, s));
    CTO_HIP(hipMemsetAsync(cx->row_str.p, 0, nr * 4, s));
    CTO_HIP(hipMemsetAsync(cx->row_tok.p, 0xff, nr * 4, s));
    hipLaunchKernelGGL(k_row_starts, dim3(unsigned(cdiv(int64_t(n_seg) * PIECES, 256))), dim3(256), 0, s, d_text, (long long)len, n_seg, cx->seg_base.as<int>(),
                       cx->row_start.as<long long>());
    hipLaunchKernelGGL(k_rows_walk, dim3(unsigned(cdiv(n_rows, 64))), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_apply3, dim3(unsigned(row_tiles)), dim3(256), 0, s, a.row_nt, a.row_nk, a.row_str, n_rows, a.tile_nt, a.tile_nk, a.tile_str,
                       cx->col_off.as<long long>(), cx->key_off.as<int>(), cx->row_str_off.as<long long>(), fl);
    CTO_HIP(hipGetLastError());
    if ((rc = fetch_flags())) return rc;
    if (hf->slow || hf->oob) { *fallback = 1; return CTO_OK; }
    const long long n_entries = hf->n_entries;
    const int n_keys = hf->n_keys;
    const long long sb = hf->key_str_bytes;
    if ((rc = cx->entries.ensure(size_t(std::max<long long>(n_entries, 1)) * 4)) || (rc = cx->key_meta.ensure(size_t(n_keys) + 16)) ||
        (rc = cx->key_group.ensure(size_t(n_keys + 1) * 4)) || (rc = cx->str_off.ensure(size_t(n_keys + 2) * 8)) || (rc = cx->key_str.ensure(size_t(sb) + 16)))
        return rc;
    a.col_off = cx->col_off.as<long long>(); a.key_off = cx->key_off.as<int>(); a.row_str_off = cx->row_str_off.as<long long>();
    a.entries = cx->entries.as<unsigned>(); a.col_pos = cx->col_pos.as<int>(); a.col_ref = cx->col_ref.as<unsigned char>();
    a.key_meta = cx->key_meta.as<unsigned char>(); a.key_group = cx->key_group.as<int>(); a.str_off = cx->str_off.as<long long>(); a.key_str = cx->key_str.as<char>();
    hipLaunchKernelGGL(k_expand, dim3(unsigned(cdiv(n_rows, 64))), dim3(256), 0, s, a);
    hipLaunchKernelGGL(k_row_keys, dim3(unsigned(cdiv(n_rows, 64))), dim3(64), 0, s, a);
    CTO_HIP(hipGetLastError());
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
        size_t off[7], total = 256;                            // (the flags came down into the first bytes of the same buffer)
        for (int i = 0; i < 7; ++i) { off[i] = total; total += (bytes[i] + 63) / 64 * 64; }
        if ((rc = cx->pin(&cx->h_stage, &cx->h_stage_cap, total + 64))) return rc;         // (may move the buffer: nothing is in flight into it)
        char* hs = static_cast<char*>(cx->h_stage);
        const TokFlags* hf2 = reinterpret_cast<const TokFlags*>(hs);
        // slow (a quality character out of range, a separator out of place) and bad_order come down with the tables
        CTO_HIP(hipMemcpyAsync(hs, fl, sizeof(TokFlags), hipMemcpyDeviceToHost, s));
        for (int i = 0; i < 7; ++i)
            if (bytes[i]) CTO_HIP(hipMemcpyAsync(hs + off[i], src[i], bytes[i], hipMemcpyDeviceToHost, s));
        if ((rc = cx->wait(s))) return rc;
        if (hf2->slow || hf2->bad_order) { *fallback = 1; return CTO_OK; }
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
