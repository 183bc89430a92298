// Reads -> columns on the device (BASELINE.json north_star: "stages per-site read pileups ... on the GPU"):
// the column pack of cto_pack_from_bam, built in HBM from BGZF blocks that csrc/inflate.hip inflated there - the alignment
// records never travel back to the host and no pack travels up.
//
// PARITY UNPINNED against samtools (absent from both boxes), like the host reader it restates (csrc/bam.cpp: the pileup rules are
// listed in its header).  What IS held: every array of the pack equal to cto_pack_from_bam's, bit for bit, on every BAM the test
// suite writes (tests/test_gpu_pileup.py) - the host reader is the specification of this file.
//
// Pipeline (one chunk = one call, all kernels on the caller's stream):
//   k_linearise     the inflated blocks (256-byte aligned output slots) -> one contiguous record stream
//   k_chain         record boundaries: every virtual offset the .bai names (chunk starts, the linear index's 16 kb windows) is the
//                   head of a chain of `block_size` hops walked by one lane; counted, then written in file order
//   k_parse         one lane per record: header filters (reference id, excluded flags, MAPQ, orphan), CIGAR lengths (CG:B,I for
//                   > 65535 operations), the region test; the first record that ends the region's scan cuts the list
//   k_compact       accepted reads in file order
//   k_cover         +1 / -1 per read at the first / one-past-last REQUESTED position it covers (the BED intervals of the
//                   candidate windows, concatenated: a read covers a contiguous run of that space)
//   k_columns_*     running sum -> depth per requested position, rows only where depth > 0, entry offsets (tile sums, their scan
//                   by one workgroup, then every 4096-position tile on top of its base)
//   k_fill          one wave per read, one lane per CIGAR operation (prefix sums over 64 operations at a time give every lane
//                   its reference / query offsets): read-bases, deletion placeholders and the indel attached to the last base
//                   of an aligned run go straight to their place in their column - file order = the number of earlier reads
//                   that cover the position, counted against the ends of the reads still open at this read's start
//   k_order         one wave per column that holds an indel carrier: the distinct indel keys in first-seen order (candidates
//                   compared against the column's keys across lanes), merged candidate-extraction groups
//   k_keys_*        key tables and the alt_info key strings ("I<ANCHOR><SEQ>", "D<reference slice>")
// Two round trips to the host are needed (sizes for allocation): after k_columns and after the per-column key counts.
//
// Not done here - the call reports `fallback` and the caller uses the host reader: paired reads (mate-overlap quality edits are
// order dependent), reference skips (N), --max-depth or more reads open at some read's start (the cap is order dependent; k_live_marks), a column
// deeper than 2048, 2048 or more reads open at a read's start, or more than 64 distinct indel keys in a column.  The BGZF CRC-32 of every block is
// checked on the device (k_crc32_blocks), as htslib and the host reader check it.
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

struct DevRead {
    uint32_t off;                 // of the record (its block_size field) in the linear stream
    int32_t pos, end;             // 0-based, end exclusive
    uint32_t ops_off;             // CIGAR operations (the field or the CG tag)
    int32_t n_ops;
    uint32_t seq_off, qual_off;
    int32_t l_seq;
    uint8_t mapq, rev, no_qual, valid;
};

struct TmpEnt { uint32_t entry, rank, ind_q, ind; };      // ind = len << 2 | kind (0 none, 1 ins, 2 del)
struct KeyRec { uint32_t read, q, len; uint8_t code, kind, overlong, group; };

struct Flags {                    // written by the kernels, read by the host after each phase
    int stop_idx, err_idx, paired_idx, skip_idx;    // first record index with the condition (INT_MAX: none)
    int bad_chain, deep_col, many_keys, ref_oob;
    int bad_crc;                                    // 1 + index of a block whose inflated bytes fail the gzip trailer's CRC-32 (0: none)
    int n_rec, n_valid, n_cols, n_keys;
    long long n_entries, key_str_bytes;
    int max_live, max_len;                          // largest number of accepted reads still open at another accepted read's start;
                                                    // longest reference span of an accepted read
};

__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { return uint32_t(p[0]) | (uint32_t(p[1]) << 8); }
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { return uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24); }

__global__ void k_linearise(const uint8_t* __restrict__ src, const cto_bgzf_block* __restrict__ blocks, const int64_t* __restrict__ lin_off,
                            uint8_t* __restrict__ lin) {
    const cto_bgzf_block b = blocks[blockIdx.x];
    const uint8_t* s = src + b.out_off;
    uint8_t* d = lin + lin_off[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < b.isize; i += blockDim.x) d[i] = s[i];
}

// CRC-32 of every inflated block against its gzip trailer, as htslib (and the host reader) checks it: one wave per block, lane k
// runs the byte-table CRC over its own slice (the first slice is the short one, every other one 1024 bytes), lane 0 then chains the
// 64 partial registers: state after slice k = P_k xor Z(state after slice k-1), Z = "1024 zero bytes" as a 32 x 32 bit matrix
// (z1k[i] = image of bit i, from the host) - a CRC register is linear in its start value.
__global__ __launch_bounds__(64) void k_crc32_blocks(const uint8_t* __restrict__ src, const cto_bgzf_block* __restrict__ blocks, int n_blocks,
                                                     const uint32_t* __restrict__ z1k, Flags* fl) {
    __shared__ uint32_t table[256];
    __shared__ uint32_t part[64];
    __shared__ uint32_t zm[32];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) {
        uint32_t c = uint32_t(i);
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        table[i] = c;
    }
    if (lane < 32) zm[lane] = z1k[lane];
    __syncthreads();
    for (int b = blockIdx.x; b < n_blocks; b += gridDim.x) {
        const cto_bgzf_block bd = blocks[b];
        const int n = int(bd.isize);
        if (n == 0) continue;
        const int ns = (n + 1023) / 1024, r = n - 1024 * (ns - 1);
        const uint8_t* p = src + bd.out_off;
        uint32_t c = lane == 0 ? 0xFFFFFFFFu : 0u;
        if (lane < ns) {
            const int lo = lane == 0 ? 0 : r + 1024 * (lane - 1), len = lane == 0 ? r : 1024;
            for (int i = 0; i < len; ++i) c = table[(c ^ p[lo + i]) & 0xFFu] ^ (c >> 8);
        }
        __syncthreads();
        part[lane] = c;
        __syncthreads();
        if (lane == 0) {
            uint32_t st = part[0];
            for (int k = 1; k < ns; ++k) {
                uint32_t z = 0;
                for (int i = 0; i < 32; ++i) z ^= ((st >> i) & 1u) ? zm[i] : 0u;
                st = part[k] ^ z;
            }
            if ((st ^ 0xFFFFFFFFu) != bd.crc32) atomicCAS(&fl->bad_crc, 0, b + 1);
        }
    }
}

// chain k walks the records from starts[k] to starts[k + 1]; mode 0 counts, mode 1 writes their offsets at base[k]..
__global__ void k_chain(const uint8_t* __restrict__ lin, int64_t len, const int64_t* __restrict__ starts, int n_chains, int mode,
                        int* __restrict__ counts, const int* __restrict__ base, uint32_t* __restrict__ rec_off, Flags* fl) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_chains) return;
    int64_t o = starts[k];
    const int64_t limit = starts[k + 1];
    int n = 0;
    while (o < limit && o + 4 <= len) {
        const int64_t bsz = int64_t(int32_t(ld32(lin + o)));
        if (bsz < 32) { atomicExch(&fl->bad_chain, 2); break; }
        if (o + 4 + bsz > len) break;                         // the span ends inside a record the region does not need
        if (mode) rec_off[base[k] + n] = uint32_t(o);
        ++n;
        o += 4 + bsz;
    }
    if (o > limit) atomicExch(&fl->bad_chain, 1);              // a named offset that is not a record boundary
    if (!mode) counts[k] = n;
}

// Exclusive prefix sum across the 1024 threads of the one workgroup these scan kernels run as (wave shuffles, then the 16 wave
// totals through LDS); *total = the sum.  Two barriers.
__device__ __forceinline__ long long block_scan_excl(long long v, long long* total, long long* wsum /* [17] shared */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long u = __shfl_up(inc, d);
        if (lane >= d) inc += u;
    }
    __syncthreads();                                          // wsum may still be read from the previous call
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    long long before = 0, all = 0;
    for (int w = 0; w < 16; ++w) { const long long x = wsum[w]; if (w < wave) before += x; all += x; }
    *total = all;
    return before + inc - v;
}

// exclusive prefix sums of an int array by one workgroup of 1024 threads, 4096 elements per pass (coalesced)
template <typename Out>
__global__ __launch_bounds__(1024) void k_scan_small(const int* __restrict__ in, Out* __restrict__ out, int n, Out* total) {
    __shared__ long long wsum[17];
    const int t = threadIdx.x;
    long long carry = 0;
    for (int base = 0; base < n; base += 4096) {
        const int i0 = base + 4 * t;
        int v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = i0 + k < n ? in[i0 + k] : 0;
        long long tot;
        long long ex = carry + block_scan_excl((long long)v[0] + v[1] + v[2] + v[3], &tot, wsum);
#pragma unroll
        for (int k = 0; k < 4; ++k) { if (i0 + k < n) out[i0 + k] = Out(ex); ex += v[k]; }
        carry += tot;
    }
    if (t == 0) { out[n] = Out(carry); if (total) *total = Out(carry); }
}

// The same over arrays of any length, spread over the chip: block sums of 4096-element tiles, their scan by one workgroup, then
// every tile scans itself on top of its base (a region piled up at every position has a million columns: the one-workgroup
// form above took 0.7 ms for them, the three launches below ~15 us).
constexpr int SCAN_TILE = 4096;
__global__ __launch_bounds__(1024) void k_tile_sums(const int* __restrict__ in, int n, long long* __restrict__ tsum) {
    __shared__ long long wsum[17];
    const int i0 = blockIdx.x * SCAN_TILE + 4 * threadIdx.x;
    long long v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) v += i0 + k < n ? in[i0 + k] : 0;
    long long tot;
    (void)block_scan_excl(v, &tot, wsum);
    if (threadIdx.x == 0) tsum[blockIdx.x] = tot;
}
// exclusive scan in place of up to three interleaved arrays of tile sums (a[i * stride + j], j < stride) by one workgroup;
// totals[j] receives the sums
__global__ __launch_bounds__(1024) void k_scan_tiles(long long* __restrict__ a, int n, int stride, long long* __restrict__ totals) {
    __shared__ long long wsum[17];
    const int t = threadIdx.x;
    for (int j = 0; j < stride; ++j) {
        long long carry = 0;
        for (int base = 0; base < n; base += 1024) {
            const int i = base + t;
            const long long v = i < n ? a[size_t(i) * stride + j] : 0;
            long long tot;
            const long long ex = carry + block_scan_excl(v, &tot, wsum);
            if (i < n) a[size_t(i) * stride + j] = ex;
            carry += tot;
        }
        if (t == 0) totals[j] = carry;
        __syncthreads();
    }
}
template <typename Out>
__global__ __launch_bounds__(1024) void k_scan_apply(const int* __restrict__ in, Out* __restrict__ out, int n, const long long* __restrict__ tbase,
                                                     const long long* __restrict__ totals, Out* total) {
    __shared__ long long wsum[17];
    const int i0 = blockIdx.x * SCAN_TILE + 4 * threadIdx.x;
    int v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = i0 + k < n ? in[i0 + k] : 0;
    long long tot;
    long long ex = tbase[blockIdx.x] + block_scan_excl((long long)v[0] + v[1] + v[2] + v[3], &tot, wsum);
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (i0 + k < n) out[i0 + k] = Out(ex); ex += v[k]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[n] = Out(totals[0]); if (total) *total = Out(totals[0]); }
}

__global__ void k_parse(const uint8_t* __restrict__ lin, const uint32_t* __restrict__ rec_off, int n_rec, int tid, int beg0, int end0,
                        int excl_flags, int min_mq, DevRead* __restrict__ reads, Flags* fl) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    DevRead r{};
    r.off = rec_off[i];
    const uint8_t* b = lin + r.off + 4;
    const int64_t bsz = int64_t(int32_t(ld32(lin + r.off)));
    const int rtid = int(ld32(b)), pos = int(ld32(b + 4));
    const int l_name = b[8], mapq = b[9];
    const int n_cig = int(ld16(b + 12)), flag = int(ld16(b + 14));
    const int l_seq = int(ld32(b + 16));
    bool stop = false, ok = false;
    if (rtid != tid) stop = rtid > tid || rtid < 0;
    else if (pos >= end0) stop = true;
    else if (!((flag & excl_flags) || (flag & 4) || mapq < min_mq || n_cig == 0 || l_seq <= 0 || pos < 0 || ((flag & 1) && !(flag & 2)))) ok = true;
    if (stop) atomicMin(&fl->stop_idx, i);
    if (ok) {
        const int64_t need = 32 + int64_t(l_name) + int64_t(n_cig) * 4 + int64_t((l_seq + 1) / 2) + int64_t(l_seq);
        if (need > bsz) { atomicMin(&fl->err_idx, i); ok = false; }
    }
    if (ok) {
        const uint8_t* cg = b + 32 + l_name;
        const uint8_t* sq = cg + size_t(n_cig) * 4;
        const uint8_t* ql = sq + (l_seq + 1) / 2;
        int n_ops = n_cig;
        const uint8_t* ops = cg;
        if (n_cig == 2 && (ld32(cg) & 15) == 4 && int(ld32(cg) >> 4) == l_seq && (ld32(cg + 4) & 15) == 3) {     // CG:B,I holds the real CIGAR
            const uint8_t* aux = ql + l_seq;
            const uint8_t* aend = b + bsz;
            while (aux + 3 <= aend) {
                const char t0 = char(aux[0]), t1 = char(aux[1]), ty = char(aux[2]);
                aux += 3;
                size_t skip = 0;
                if (ty == 'A' || ty == 'c' || ty == 'C') skip = 1;
                else if (ty == 's' || ty == 'S') skip = 2;
                else if (ty == 'i' || ty == 'I' || ty == 'f') skip = 4;
                else if (ty == 'Z' || ty == 'H') { while (aux + skip < aend && aux[skip]) ++skip; ++skip; }
                else if (ty == 'B') {
                    if (aux + 5 > aend) break;
                    const char sub = char(aux[0]);
                    const uint32_t cnt = ld32(aux + 1);
                    const size_t esz = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                    if (t0 == 'C' && t1 == 'G' && sub == 'I' && aux + 5 + size_t(cnt) * 4 <= aend) { n_ops = int(cnt); ops = aux + 5; break; }
                    skip = 5 + size_t(cnt) * esz;
                } else break;
                aux += skip;
            }
        }
        long long rlen = 0, qlen = 0;
        bool has_skip = false;
        for (int k = 0; k < n_ops; ++k) {
            const uint32_t c = ld32(ops + size_t(k) * 4);
            const int opc = int(c & 15), len = int(c >> 4);
            if (opc == 0 || opc == 2 || opc == 3 || opc == 7 || opc == 8) rlen += len;
            if (opc == 0 || opc == 1 || opc == 4 || opc == 7 || opc == 8) qlen += len;
            has_skip |= opc == 3;
        }
        if (qlen != l_seq || rlen == 0) ok = false;
        else if (int64_t(pos) + rlen > 0x7fffffffLL) { atomicMin(&fl->err_idx, i); ok = false; }
        else if (pos + rlen <= beg0) ok = false;
        if (ok) {
            r.pos = pos;
            r.end = int32_t(pos + rlen);
            r.ops_off = uint32_t(ops - lin);
            r.n_ops = n_ops;
            r.seq_off = uint32_t(sq - lin);
            r.qual_off = uint32_t(ql - lin);
            r.l_seq = l_seq;
            r.mapq = uint8_t(mapq);
            r.rev = (flag & 16) != 0;
            r.no_qual = ql[0] == 0xff;
            r.valid = 1;
            if (flag & 1) atomicMin(&fl->paired_idx, i);
            if (has_skip) atomicMin(&fl->skip_idx, i);
        }
    }
    reads[i] = r;
}

// accepted reads (in front of the record that ended the scan) in file order: rid[j] = record index
__global__ void k_compact(const DevRead* __restrict__ reads, int n_rec, int* __restrict__ rid, Flags* fl) {
    __shared__ int part[1024];
    const int stop = fl->stop_idx;
    const int n = min(n_rec, stop), t = threadIdx.x, per = (n + blockDim.x - 1) / blockDim.x;
    const int lo = min(n, t * per), hi = min(n, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += reads[i].valid;
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < int(blockDim.x); d <<= 1) {
        const int v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s, longest = 0;
    for (int i = lo; i < hi; ++i) if (reads[i].valid) { rid[run++] = i; longest = max(longest, reads[i].end - reads[i].pos); }
    if (longest > 0) atomicMax(&fl->max_len, longest);
    if (t == int(blockDim.x) - 1) fl->n_valid = part[t];
}

// The --max-depth cap (host reader, csrc/bam.cpp: a read is dropped when max_depth accepted reads are still open at its start) can
// only bite where that many reads overlap a read's start.  Accepted reads are in file order = sorted by start, so read j is open at
// the start of exactly the reads j+1 .. nxt(j)-1, nxt(j) = the first later read starting at or behind j's end: +1 / -1 marks over the
// read index, a running sum, its maximum.  Below max_depth no read is dropped and the device pack is the host's; otherwise fallback.
__global__ void k_live_marks(const DevRead* __restrict__ reads, const int* __restrict__ rid, const Flags* fl, int* __restrict__ marks) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, n = fl->n_valid;
    if (j >= n) return;
    const int end = reads[rid[j]].end;
    int a = j + 1, b = n;
    while (a < b) { const int m = (a + b) >> 1; if (reads[rid[m]].pos >= end) b = m; else a = m + 1; }
    if (a > j + 1) { atomicAdd(&marks[j + 1], 1); atomicAdd(&marks[a], -1); }
}
__global__ void k_live_max(const int* __restrict__ marks, Flags* fl) {       // one workgroup
    __shared__ int part[1024], best[1024];
    const int n = fl->n_valid + 1, t = threadIdx.x, per = (n + blockDim.x - 1) / blockDim.x;
    const int lo = min(n, t * per), hi = min(n, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += marks[i];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < int(blockDim.x); d <<= 1) {
        const int v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s, m = 0;
    for (int i = lo; i < hi; ++i) { run += marks[i]; m = max(m, run); }
    best[t] = m;
    __syncthreads();
    for (int d = int(blockDim.x) >> 1; d > 0; d >>= 1) {
        if (t < d) best[t] = max(best[t], best[t + d]);
        __syncthreads();
    }
    if (t == 0) fl->max_live = best[0];
}

// Requested positions: n_iv sorted, disjoint 0-based intervals [lo, hi) clipped to the region; slot space = their concatenation
struct Ivs { const int* lo; const int* hi; const int* base; int n; int total; };
__device__ __forceinline__ int first_slot_ge(const Ivs& v, int p) {    // slot of the first requested position >= p (total: none)
    int a = 0, b = v.n;
    while (a < b) { const int m = (a + b) >> 1; if (v.hi[m] > p) b = m; else a = m + 1; }
    if (a >= v.n) return v.total;
    return v.base[a] + (p > v.lo[a] ? p - v.lo[a] : 0);
}

__global__ void k_cover(const DevRead* __restrict__ reads, const int* __restrict__ rid, const Flags* fl, Ivs iv, int* __restrict__ diff) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= fl->n_valid) return;
    const DevRead& r = reads[rid[j]];
    const int a = first_slot_ge(iv, r.pos), b = first_slot_ge(iv, r.end);
    if (b > a) { atomicAdd(&diff[a], 1); atomicAdd(&diff[b], -1); }
}

constexpr int ORD_DMAX_COLS = 2048;      // = ORD_DMAX of k_order
// One workgroup (1024 threads, 4096 slots per pass, coalesced): depth per slot (running sum of diff), rows where depth > 0
// (slot_col, col_slot), col_off.
// k_columns over the chip: (1) tile sums of the marks, scanned -> the depth in front of every tile; (2) per tile the number of
// positions with depth > 0 and the sum of their depths, scanned -> the first column / entry of every tile; (3) every tile writes
// its rows.  The one-workgroup form took 3.7 ms for the million positions of a region piled up without a BED.
__device__ __forceinline__ void tile_depths(const int* __restrict__ diff, int total, long long before, int i0, long long* wsum, int (&d)[4],
                                            long long (&dep)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = i0 + k < total ? diff[i0 + k] : 0;
    long long tot;
    long long depth = before + block_scan_excl((long long)d[0] + d[1] + d[2] + d[3], &tot, wsum);
#pragma unroll
    for (int k = 0; k < 4; ++k) { depth += d[k]; dep[k] = i0 + k < total ? depth : 0; }
}
__global__ __launch_bounds__(1024) void k_columns_count(const int* __restrict__ diff, int total, const long long* __restrict__ depth_base,
                                                        long long* __restrict__ tcnt /* [tiles][2] */) {
    __shared__ long long wsum[17];
    const int i0 = blockIdx.x * SCAN_TILE + 4 * threadIdx.x;
    int d[4];
    long long dep[4];
    tile_depths(diff, total, depth_base[blockIdx.x], i0, wsum, d, dep);
    long long nz = 0, ds = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { nz += dep[k] > 0; ds += dep[k]; }
    long long tn, td;
    (void)block_scan_excl(nz, &tn, wsum);
    (void)block_scan_excl(ds, &td, wsum);
    if (threadIdx.x == 0) { tcnt[size_t(blockIdx.x) * 2] = tn; tcnt[size_t(blockIdx.x) * 2 + 1] = td; }
}
__global__ __launch_bounds__(1024) void k_columns_write(const int* __restrict__ diff, int total, const long long* __restrict__ depth_base,
                                                        const long long* __restrict__ tcnt, const long long* __restrict__ totals,
                                                        int* __restrict__ slot_col, int* __restrict__ col_slot, long long* __restrict__ col_off, Flags* fl) {
    __shared__ long long wsum[17];
    const int i0 = blockIdx.x * SCAN_TILE + 4 * threadIdx.x;
    int d[4];
    long long dep[4];
    tile_depths(diff, total, depth_base[blockIdx.x], i0, wsum, d, dep);
    long long nz = 0, ds = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { nz += dep[k] > 0; ds += dep[k]; }
    long long tn, td;
    long long c = tcnt[size_t(blockIdx.x) * 2] + block_scan_excl(nz, &tn, wsum);
    long long o = tcnt[size_t(blockIdx.x) * 2 + 1] + block_scan_excl(ds, &td, wsum);
    bool deep = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i0 + k >= total) break;
        if (dep[k] > 0) { slot_col[i0 + k] = int(c); col_slot[c] = i0 + k; col_off[c] = o; ++c; o += dep[k]; deep |= dep[k] > ORD_DMAX_COLS; }
        else slot_col[i0 + k] = -1;
    }
    if (deep) atomicExch(&fl->deep_col, 1);
    if (blockIdx.x == 0 && threadIdx.x == 0) { fl->n_cols = int(totals[0]); fl->n_entries = totals[1]; col_off[totals[0]] = totals[1]; }
}

__global__ void k_col_meta(const int* __restrict__ col_slot, int n_cols, Ivs iv, const char* __restrict__ ref, long long ref_start, long long ref_len,
                           int32_t* __restrict__ col_pos, uint8_t* __restrict__ col_ref, Flags* fl) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cols) return;
    const int s = col_slot[c];
    int a = 0, b = iv.n;
    while (b - a > 1) { const int m = (a + b) >> 1; if (iv.base[m] <= s) a = m; else b = m; }
    const long long pos1 = (long long)iv.lo[a] + (s - iv.base[a]) + 1, ri = pos1 - ref_start;
    col_pos[c] = int32_t(pos1);
    if (ri < 0 || ri >= ref_len) { atomicExch(&fl->ref_oob, 1); col_ref[c] = 0; return; }
    char ch = ref[ri];
    if (ch >= 'a' && ch <= 'z') ch = char(ch - 32);
    const int code = ch == 'C' ? 1 : (ch == 'G' ? 2 : (ch == 'T' ? 3 : 0));
    const bool acgt = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T';
    col_ref[c] = uint8_t(code | (acgt ? 0 : 0x80));
}

__device__ __forceinline__ uint32_t dev_entry(int code, int bq, int mq) { return uint32_t(code) | (uint32_t(bq) << 6) | (uint32_t(mq) << 13); }
__constant__ int8_t kNib[16] = {-1, 0, 1, 10, 2, 10, 10, 10, 3, 10, 10, 10, 10, 10, 10, 10};

// One wave per accepted read, one lane per CIGAR operation (64 at a time).
// Where an entry goes (round 4): a column lists its reads in file order, and read j's place in the column of position p is the number
// of EARLIER reads that cover p - all of which started at or before j's start (the file is sorted), so they are the reads still open
// at j's start whose end lies behind p.  The wave collects the ends of those open reads once (the accepted reads in front of j, back
// to the first one that starts more than the longest read's span before j) and counts, per position, the ends behind it: no cursor
// atomics (device-scope atomics from eight XCDs on 51 M read-bases were most of this kernel's time), no rank to carry along, nothing
// to put back into order afterwards - the 4-byte entry lands in its final place.  Indel carriers (one read-base in a hundred) leave
// their details in `side` at the same index and a mark on their column for k_order.
constexpr int FILL_LIVE_MAX = 2048;          // reads open at a read's start the kernel can hold (more: the caller falls back)
__global__ __launch_bounds__(256) void k_fill(int live_cap, const uint8_t* __restrict__ lin, const DevRead* __restrict__ reads, const int* __restrict__ rid,
                                              Flags* fl, Ivs iv, const int* __restrict__ slot_col, const long long* __restrict__ col_off,
                                              int* __restrict__ marks, uint32_t* __restrict__ entries, TmpEnt* __restrict__ side,
                                              const char* __restrict__ ref, long long ref_start, long long ref_len) {
    // [wave][live_cap] ends as collected, then [wave][live_cap] sorted; live_cap = the launch's bound on reads open at a read's start
    // (k_live_max), so that ordinary depths leave the LDS - and with it the number of reads in flight per CU - alone
    extern __shared__ int s_fill[];
    int* const s_live_w = s_fill + (threadIdx.x >> 6) * live_cap;
    int* const s_sort_w = s_fill + (4 + (threadIdx.x >> 6)) * live_cap;
    // The four waves of a workgroup share ONE read (each walks the CIGAR for itself and takes every fourth 64-position slice of a long
    // run, every fourth short operation): the kernel lasts as long as its longest read's chain of dependent trips to memory
    // (position -> column -> entry offset -> store, ~3 us a slice), and a 30 kb read is 470 slices.
    // Workgroups are dealt to the 8 XCDs in turn: XCD x takes the x-th eighth of the (position-sorted) reads, so that the reads an XCD
    // works on at one time lie next to each other on the genome and their four-byte stores - every one to another cache line of the
    // column-major entries - meet again in that XCD's own L2 before the lines go to memory
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int per_xcd = (fl->n_valid + 7) >> 3;
    const int j = int(blockIdx.x & 7u) * per_xcd + int(blockIdx.x >> 3);
    if (int(blockIdx.x >> 3) >= per_xcd || j >= fl->n_valid) return;
    const DevRead r = reads[rid[j]];
    int n_live = 0;
    {
        const long long reach = (long long)r.pos - fl->max_len;       // a read that starts at or before it has ended by r.pos
        for (int i0 = j - 1; i0 >= 0; i0 -= 64) {
            const int i = i0 - lane;
            int ps = 0, en = 0;
            if (i >= 0) { const int ri = rid[i]; ps = reads[ri].pos; en = reads[ri].end; }
            const bool open = i >= 0 && en > r.pos;
            const unsigned long long m = __ballot(open);
            const int at = n_live + __popcll(m & ((1ull << lane) - 1ull));
            if (open && at < live_cap) s_live_w[at] = en;
            n_live += __popcll(m);
            if (__ballot(i >= 0 && (long long)ps <= reach)) break;
        }
        if (n_live > live_cap) { if (lane == 0) atomicExch(&fl->deep_col, 1); return; }      // cannot happen (live_cap > max_live); the caller falls back
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        // ascending (rank by counting; n_live is about the depth): a position's count is then a binary search, not a sweep
        for (int u0 = 0; u0 < n_live; u0 += 64) {
            const int idx = u0 + lane;
            const int v = idx < n_live ? s_live_w[idx] : 0x7fffffff;
            int rk = 0;
            for (int k = 0; k < n_live; ++k) { const int x = s_live_w[k]; rk += (x < v) || (x == v && k < idx); }
            if (idx < n_live) s_sort_w[rk] = v;
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
    }
    const int* const live = s_sort_w;
    const uint8_t* ops = lin + r.ops_off;
    const uint8_t* seq = lin + r.seq_off;
    const uint8_t* qual = lin + r.qual_off;
    const int mq = min(int(r.mapq), 93);
    auto base4 = [&](int q) { return (seq[q >> 1] >> ((~q & 1) << 2)) & 15; };
    auto bq = [&](int q) { return (r.no_qual || q >= r.l_seq) ? 0 : min(int(qual[q]), 93); };
    int ref_carry = r.pos, q_carry = 0;
    for (int k0 = 0; k0 < r.n_ops; k0 += 64) {
        const int k = k0 + lane;
        uint32_t c = 0;
        if (k < r.n_ops) c = ld32(ops + size_t(k) * 4);
        const int opc = int(c & 15), len = int(c >> 4);
        const bool cons_ref = k < r.n_ops && (opc == 0 || opc == 2 || opc == 3 || opc == 7 || opc == 8);
        const bool cons_q = k < r.n_ops && (opc == 0 || opc == 1 || opc == 4 || opc == 7 || opc == 8);
        int rs = cons_ref ? len : 0, qs = cons_q ? len : 0;
        const int rl = rs, ql = qs;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int a = __shfl_up(rs, d), b = __shfl_up(qs, d);
            if (lane >= d) { rs += a; qs += b; }
        }
        const int op_ref = ref_carry + rs - rl, op_q = q_carry + qs - ql;
        ref_carry += __shfl(rs, 63);
        q_carry += __shfl(qs, 63);
        const bool covers = k < r.n_ops && cons_ref && opc != 3;
        const bool aligned = opc != 2;
        // the indel that follows this aligned run (P operations skipped)
        uint32_t ind = 0, ind_q = 0;
        if (covers && aligned) {
            int nx = k + 1;
            while (nx < r.n_ops && (ld32(ops + size_t(nx) * 4) & 15) == 6) ++nx;
            if (nx < r.n_ops) {
                const uint32_t nc = ld32(ops + size_t(nx) * 4);
                const int nop = int(nc & 15), nlen = int(nc >> 4);
                if (nop == 1) { ind = (uint32_t(nlen) << 2) | 1u; ind_q = uint32_t(op_q + len); }
                else if (nop == 2) ind = (uint32_t(nlen) << 2) | 2u;
            }
        }
        // the read's contribution at reference position p of the operation (o_ref, o_q, o_len, ...)
        // the entry of reference position p and its column; the trip to the column's cursor is the caller's (several in flight)
        auto make = [&](int p, int kiv, int o_ref, int o_q, int o_len, bool o_aligned, uint32_t o_ind, uint32_t o_ind_q, TmpEnt* out) -> int {
            const int col = slot_col[iv.base[kiv] + (p - iv.lo[kiv])];
            TmpEnt e;
            e.rank = uint32_t(j);
            e.ind = 0;
            e.ind_q = 0;
            if (!o_aligned) {
                e.entry = dev_entry(r.rev ? 9 : 8, bq(o_q), mq);
            } else {
                const int q = o_q + (p - o_ref);
                int code = kNib[base4(q)];
                if (code < 0) {
                    const long long ri = (long long)p + 1 - ref_start;
                    char ch = (ri >= 0 && ri < ref_len) ? ref[ri] : 'N';
                    if (ch >= 'a' && ch <= 'z') ch = char(ch - 32);
                    code = ch == 'A' ? 0 : (ch == 'C' ? 1 : (ch == 'G' ? 2 : (ch == 'T' ? 3 : 10)));
                }
                if (r.rev) code += code < 4 ? 4 : 1;
                e.entry = dev_entry(code, bq(q), mq);
                if (p == o_ref + o_len - 1) { e.ind = o_ind; e.ind_q = o_ind_q; }
            }
            *out = e;
            return col;
        };
        auto emit = [&](int p, int kiv, int o_ref, int o_q, int o_len, bool o_aligned, uint32_t o_ind, uint32_t o_ind_q) {
            TmpEnt e;
            const int col = make(p, kiv, o_ref, o_q, o_len, o_aligned, o_ind, o_ind_q, &e);
            int lo = 0, n = n_live;                                             // first end behind p: the earlier reads that cover p
            while (n > 0) {
                const int half = n >> 1;
                if (live[lo + half] <= p) { lo += half + 1; n -= half + 1; } else n = half;
            }
            const long long at = col_off[col] + (n_live - lo);
            entries[at] = e.entry | ((e.ind & 3u) << 4);                        // carriers: provisional kind, k_order completes it
            if (e.ind & 3u) { side[at] = e; marks[col] = 1; }
        };
        auto first_iv = [&](int p) {
            int a = 0, b = iv.n;
            while (a < b) { const int m = (a + b) >> 1; if (iv.hi[m] > p) b = m; else a = m + 1; }
            return a;
        };
        // short operations (real long-read CIGARs: an indel every ~10-20 bases): the lane walks its own; long aligned runs
        // (hundreds of bases) would leave the other lanes idle, so the whole wave takes each of those in turn below
        constexpr int LONG_OP = 48;
        const bool is_long = covers && len > LONG_OP;
        if (covers && !is_long && (k & 3) == wv) {
            int p = op_ref, kiv = first_iv(op_ref);
            const int pend = op_ref + len;
            while (p < pend && kiv < iv.n) {
                if (p < iv.lo[kiv]) { p = iv.lo[kiv]; continue; }
                if (p >= iv.hi[kiv]) { ++kiv; continue; }
                const int stop = min(pend, iv.hi[kiv]);
                for (; p < stop; ++p) emit(p, kiv, op_ref, op_q, len, aligned, ind, ind_q);
            }
        }
        unsigned long long lm = __ballot(is_long);
        while (lm) {
            const int src = __ffsll((long long)lm) - 1;
            lm &= lm - 1;
            const int o_ref = __shfl(op_ref, src), o_q = __shfl(op_q, src), o_len = __shfl(len, src);
            const bool o_aligned = __shfl(int(aligned), src) != 0;
            const uint32_t o_ind = __shfl(ind, src), o_ind_q = __shfl(ind_q, src);
            const int pend = o_ref + o_len;
            for (int kiv = first_iv(o_ref); kiv < iv.n && iv.lo[kiv] < pend; ++kiv) {
                const int lo = max(o_ref, iv.lo[kiv]), hi = min(pend, iv.hi[kiv]);
                if (hi - lo > 64) {              // a long stretch of requested positions: every wave takes each fourth 64-position slice
                    for (int p = lo + lane + 64 * wv; p < hi; p += 256) emit(p, kiv, o_ref, o_q, o_len, o_aligned, o_ind, o_ind_q);
                } else if ((kiv & 3) == wv) {    // candidate windows (34 positions each): every wave takes each fourth window
                    const int p = lo + lane;
                    if (p < hi) emit(p, kiv, o_ref, o_q, o_len, o_aligned, o_ind, o_ind_q);
                }
            }
        }
    }
}

constexpr int ORD_KMAX = 64;

__device__ __forceinline__ int canon_nib(const uint8_t* seq, int q) {
    const int n = (seq[q >> 1] >> ((~q & 1) << 2)) & 15;
    return n == 0 ? 15 : n;                       // '=' prints as N, like nibble 15
}

struct IndEnt { uint32_t at, entry, rank, ind_q, ind; };     // at = the entry's place in the column once it is in file order

// The indel keys of the columns k_fill marked (round 4: the entries are in file order already): one wave per marked column walks its
// entries 64 at a time, and every indel carrier - in file order - is compared against the column's keys across lanes: the distinct keys
// in first-seen order, the merged candidate-extraction groups, the entry completed with kind and key id.  Columns without a carrier
// (six in ten at 50x) cost one coalesced load of 64 marks per wave trip.
__global__ __launch_bounds__(64) void k_order(const uint8_t* __restrict__ lin, const DevRead* __restrict__ reads, const int* __restrict__ rid,
                                              int n_cols, const long long* __restrict__ col_off, const int* __restrict__ marks,
                                              const TmpEnt* __restrict__ side, uint32_t* __restrict__ entries, int* __restrict__ n_keys_col,
                                              KeyRec* __restrict__ keyrec, int max_indel, Flags* fl) {
    __shared__ KeyRec keys[ORD_KMAX];
    __shared__ int grp_key[ORD_KMAX];             // group g is represented by the key that opened it
    const int lane = threadIdx.x;
    for (long long base = blockIdx.x * 64ll; base < n_cols; base += gridDim.x * 64ll) {
        const long long cc = base + lane;
        const bool marked = cc < n_cols && marks[cc] != 0;
        if (cc < n_cols && !marked) n_keys_col[cc] = 0;
        unsigned long long todo = __ballot(marked);
        while (todo) {
            const int c = int(base) + __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            const long long o = col_off[c];
            const int d = int(col_off[c + 1] - o);
            __syncthreads();
            int nk = 0, ng = 0;
            bool over = false;
            for (int i0 = 0; i0 < d && !over; i0 += 64) {
                const int i = i0 + lane;
                const uint32_t ent = i < d ? entries[o + i] : 0u;
                unsigned long long car = __ballot(i < d && ((ent >> 4) & 3u) != 0u);
                while (car && !over) {
                    const int src = __ffsll((long long)car) - 1;
                    car &= car - 1ull;
                    const int cat = i0 + src;
                    const TmpEnt cand = side[o + cat];                             // the same for every lane
                    const uint32_t centry = cand.entry;
                    const int kind = int(cand.ind & 3u), len = int(cand.ind >> 2), code = int(centry & 15u);
                    const uint8_t* cseq = lin + reads[rid[cand.rank]].seq_off;
                    bool hit = false;
                    if (lane < nk) {                                               // lane k compares the candidate with key k
                        const KeyRec kr = keys[lane];
                        if (int(kr.code) == code && int(kr.kind) == kind && int(kr.len) == len) {
                            hit = true;
                            if (kind == 1) {
                                const uint8_t* kseq = lin + reads[rid[kr.read]].seq_off;
                                for (int t = 0; t < len; ++t)
                                    if (canon_nib(kseq, int(kr.q) + t) != canon_nib(cseq, int(cand.ind_q) + t)) { hit = false; break; }
                            }
                        }
                    }
                    const unsigned long long hm = __ballot(hit);
                    int kid;
                    if (hm) {
                        kid = __ffsll((long long)hm) - 1;
                    } else {
                        if (nk >= ORD_KMAX) { over = true; break; }
                        kid = nk;
                        // merged group for candidate extraction: insertions by upper-cased anchor + sequence, deletions by length
                        const char anchor_c = "ACGTACGT*#NN"[code];
                        bool ghit = false;
                        if (lane < ng) {
                            const KeyRec gr = keys[grp_key[lane]];
                            if (int(gr.kind) == kind && int(gr.len) == len) {
                                if (kind == 2) ghit = true;
                                else if ("ACGTACGT*#NN"[gr.code] == anchor_c) {
                                    ghit = true;
                                    const uint8_t* gseq = lin + reads[rid[gr.read]].seq_off;
                                    for (int t = 0; t < len; ++t)
                                        if (canon_nib(gseq, int(gr.q) + t) != canon_nib(cseq, int(cand.ind_q) + t)) { ghit = false; break; }
                                }
                            }
                        }
                        const unsigned long long gm = __ballot(ghit);
                        const int g = gm ? __ffsll((long long)gm) - 1 : ng;
                        if (lane == 0) {
                            const int gate = kind == 1 ? len : len + 1;
                            keys[nk] = KeyRec{cand.rank, cand.ind_q, uint32_t(len), uint8_t(code), uint8_t(kind), uint8_t(gate > max_indel), uint8_t(g)};
                            if (!gm) grp_key[ng] = nk;
                        }
                        if (!gm) ++ng;
                        ++nk;
                        __syncthreads();
                    }
                    if (lane == 0) {
                        const int gate = kind == 1 ? len : len + 1;
                        entries[o + cat] = centry | (uint32_t(gate > max_indel ? 3 : kind) << 4) | (uint32_t(kid) << 21);
                    }
                }
            }
            if (over) { if (lane == 0) { atomicExch(&fl->many_keys, 1); n_keys_col[c] = 0; } continue; }
            __syncthreads();
            if (lane == 0) n_keys_col[c] = nk;
            for (int k = lane; k < nk; k += 64) keyrec[o + k] = keys[k];          // nk <= d: the column's own slots
        }
    }
}

// key tables in column order + the length of every alt_info key string
__global__ void k_keys_meta(int n_cols, const long long* __restrict__ col_off, const int* __restrict__ key_off, const KeyRec* __restrict__ keyrec,
                            const int32_t* __restrict__ col_pos, long long ref_start, long long ref_len, int max_indel, uint8_t* __restrict__ key_meta,
                            int32_t* __restrict__ key_group, KeyRec* __restrict__ key_final, int* __restrict__ key_col, int* __restrict__ key_len) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cols) return;
    const int k0 = key_off[c], nk = key_off[c + 1] - k0;
    for (int k = 0; k < nk; ++k) {
        const KeyRec kr = keyrec[col_off[c] + k];
        const bool fwd = kr.code < 4 || kr.code == 8 || kr.code == 10;
        key_meta[k0 + k] = uint8_t(kr.kind | (fwd ? 4 : 0) | (kr.overlong ? 8 : 0));
        key_group[k0 + k] = kr.group;
        key_final[k0 + k] = kr;
        key_col[k0 + k] = c;
        int sl;
        if (kr.kind == 1) sl = 2 + int(kr.len);
        else {
            const long long ri = (long long)col_pos[c] - ref_start;
            long long take = min((long long)kr.len + 1, (long long)max_indel);
            take = min(take, ref_len - ri);
            sl = 1 + int(take > 0 ? take : 0);
        }
        key_len[k0 + k] = sl;
    }
}

__global__ void k_keys_str(int n_keys, const KeyRec* __restrict__ key_final, const int* __restrict__ key_col, const long long* __restrict__ str_off,
                           const uint8_t* __restrict__ lin, const DevRead* __restrict__ reads, const int* __restrict__ rid,
                           const int32_t* __restrict__ col_pos, const char* __restrict__ ref, long long ref_start, char* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_keys) return;
    const KeyRec kr = key_final[k];
    char* o = out + str_off[k];
    const int n = int(str_off[k + 1] - str_off[k]);
    static const char kAnchor[] = "ACGTACGT*#NN";
    static const char kNt16[] = "NACMGRSVTWYHKDBN";          // upper case, '=' -> N
    if (kr.kind == 1) {
        o[0] = 'I';
        o[1] = kAnchor[kr.code];
        const uint8_t* seq = lin + reads[rid[kr.read]].seq_off;
        for (int t = 0; t + 2 < n; ++t) { const int q = int(kr.q) + t; o[2 + t] = kNt16[(seq[q >> 1] >> ((~q & 1) << 2)) & 15]; }
    } else {
        o[0] = 'D';
        const long long ri = (long long)col_pos[key_col[k]] - ref_start;
        for (int t = 0; t + 1 < n; ++t) { char ch = ref[ri + t]; if (ch >= 'a' && ch <= 'z') ch = char(ch - 32); o[1 + t] = ch; }
    }
}

struct Buf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return CTO_OK;
        if (p) { CTO_HIP(hipFree(p)); p = nullptr; cap = 0; }
        const size_t want = n + n / 4 + 4096;
        CTO_HIP(hipMalloc(&p, want));
        cap = want;
        return CTO_OK;
    }
    ~Buf() { if (p) (void)hipFree(p); }
    template <class T> T* as() { return static_cast<T*>(p); }
};

}  // namespace

struct cto_dev_pileup {
    Buf tile_a, tile_b, tile_tot;        // tile sums of the spread-out scans
    Buf lin, counts, base, rec_off, reads, rid, live, diff, slot_col, col_slot, col_off, col_pos, col_ref, cursor, tmp,
        entries, nkc, keyrec, key_off, key_meta, key_group, key_final, key_col, key_len, str_off, key_str, z1k;
    bool z1k_ready = false;
    Flags* h_flags = nullptr;            // page-locked mirror
    void* h_stage = nullptr;             // page-locked landing area of the small arrays that go back to the host (a copy to pageable
    size_t h_stage_cap = 0;              // memory blocks - and spins - until everything queued in front of it is done)
    // The chunk's small inputs (flags, block table, linear offsets, record starts, intervals, reference window) go up as ONE copy out
    // of page-locked memory: six hipMemcpyAsync calls from pageable vectors each pin their source on the fly, under a lock every
    // producer thread of the run shares.
    Buf up;
    void* h_up = nullptr;
    size_t h_up_cap = 0;
    int up_ensure(size_t n) {
        if (n <= h_up_cap) return CTO_OK;
        if (h_up) { CTO_HIP(hipHostFree(h_up)); h_up = nullptr; h_up_cap = 0; }
        const size_t want = n + n / 4 + 4096;
        CTO_HIP(hipHostMalloc(&h_up, want, hipHostMallocDefault));
        h_up_cap = want;
        return CTO_OK;
    }
    int stage_ensure(size_t n) {
        if (n <= h_stage_cap) return CTO_OK;
        if (h_stage) { CTO_HIP(hipHostFree(h_stage)); h_stage = nullptr; h_stage_cap = 0; }
        const size_t want = n + n / 4 + 4096;
        CTO_HIP(hipHostMalloc(&h_stage, want, hipHostMallocDefault));
        h_stage_cap = want;
        return CTO_OK;
    }
    hipEvent_t ev = nullptr;             // the driver's waits sleep on it: hipStreamSynchronize polls the completion signal from the calling
                                         // thread, and that thread shares sixteen host cores with everything else of a run
    ~cto_dev_pileup() { if (h_flags) (void)hipHostFree(h_flags); if (h_stage) (void)hipHostFree(h_stage); if (h_up) (void)hipHostFree(h_up); if (ev) (void)hipEventDestroy(ev); }
};

// waits for everything queued on `s` so far without occupying a core (pipeline.hip's wait_event)
static hipError_t sleepy_sync(cto_dev_pileup* cx, hipStream_t s) {
    hipError_t e = hipEventRecord(cx->ev, s);
    if (e != hipSuccess) return e;
    for (int spins = 0;; ++spins) {
        e = hipEventQuery(cx->ev);
        if (e != hipErrorNotReady) return e;
        if (spins >= 4) usleep(spins < 64 ? 50 : 200);
    }
}

extern "C" int cto_dev_pileup_create(cto_dev_pileup** out) try {
    CTO_REQUIRE(out, CTO_EINVAL, "cto_dev_pileup_create: null argument");
    std::unique_ptr<cto_dev_pileup> c(new cto_dev_pileup());
    CTO_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_flags), sizeof(Flags), hipHostMallocDefault));
    CTO_HIP(hipEventCreateWithFlags(&c->ev, hipEventDisableTiming));
    *out = c.release();
    return CTO_OK;
}
CTO_CATCH("cto_dev_pileup_create", int)

extern "C" void cto_dev_pileup_destroy(cto_dev_pileup* c) { delete c; }

// Record starts the index names inside the inflated span: chunk starts of the region's bins and the linear index's windows.
extern "C" int64_t cto_bam_record_starts(const char* bam_path, const char* bai_path, const char* ctg_name, int64_t start, int64_t end,
                                         int64_t file_begin, int64_t file_end, uint64_t* voffs, int64_t cap, int32_t* tid_out);

extern "C" int cto_pileup_device(cto_dev_pileup* cx, const void* d_inflated, const cto_bgzf_block* h_blocks, int64_t n_blocks,
                                 const uint64_t* rec_voffs, int64_t n_starts, int32_t tid, int64_t start, int64_t end, const int64_t* bed,
                                 int64_t n_bed, const char* ref_seq, int64_t ref_start, size_t ref_len, int excl_flags, int min_mq,
                                 int max_depth, int max_indel_length, void* stream, cto_pack_view* dev_view, cto_pack** host_lite,
                                 int* fallback) try {
    CTO_REQUIRE(cx && d_inflated && h_blocks && n_blocks > 0 && rec_voffs && n_starts > 0 && ref_seq && dev_view && host_lite && fallback,
                CTO_EINVAL, "cto_pileup_device: bad argument");
    CTO_REQUIRE(start >= 1 && end >= start && end < (int64_t(1) << 31), CTO_EINVAL, "cto_pileup_device: bad region");
    hipStream_t s = static_cast<hipStream_t>(stream);
    *fallback = 0;
    *host_lite = nullptr;
    // ---- host-side tables: block -> linear offset, record starts as linear offsets, requested intervals ----
    std::vector<int64_t> lin_off(size_t(n_blocks) + 1, 0);
    for (int64_t b = 0; b < n_blocks; ++b) lin_off[size_t(b) + 1] = lin_off[size_t(b)] + h_blocks[b].isize;
    const int64_t len = lin_off[size_t(n_blocks)];
    CTO_REQUIRE(len < (int64_t(1) << 32) - 65536, CTO_EUNSUPPORTED, "cto_pileup_device: more than 4 GiB of alignment records in one chunk");
    std::vector<int64_t> starts;
    for (int64_t i = 0; i < n_starts; ++i) {
        const int64_t coff = int64_t(rec_voffs[i] >> 16), uoff = int64_t(rec_voffs[i] & 0xffff);
        int64_t lo = 0, hi = n_blocks;
        while (lo < hi) { const int64_t m = (lo + hi) / 2; if (int64_t(h_blocks[m].file_off) < coff) lo = m + 1; else hi = m; }
        if (lo >= n_blocks || int64_t(h_blocks[lo].file_off) != coff || uoff > int64_t(h_blocks[lo].isize)) continue;     // outside the span
        starts.push_back(lin_off[size_t(lo)] + uoff);
    }
    CTO_REQUIRE(!starts.empty(), CTO_EINVAL, "cto_pileup_device: no record start inside the inflated span");
    std::sort(starts.begin(), starts.end());
    starts.erase(std::unique(starts.begin(), starts.end()), starts.end());
    const int n_chains = int(starts.size());
    starts.push_back(len);
    std::vector<int> ivs;                                        // lo[], hi[], base[] back to back
    {
        std::vector<std::pair<int64_t, int64_t>> v;
        if (bed) for (int64_t i = 0; i < n_bed; ++i) v.push_back({std::max<int64_t>(bed[2 * i], start - 1), std::min<int64_t>(bed[2 * i + 1], end)});
        else v.push_back({start - 1, end});
        std::vector<int> lo, hi, base;
        int64_t total = 0;
        for (auto& p : v)
            if (p.second > p.first) { lo.push_back(int(p.first)); hi.push_back(int(p.second)); base.push_back(int(total)); total += p.second - p.first; }
        CTO_REQUIRE(total < (int64_t(1) << 30), CTO_EUNSUPPORTED, "cto_pileup_device: too many requested positions");
        ivs = lo;
        ivs.insert(ivs.end(), hi.begin(), hi.end());
        ivs.insert(ivs.end(), base.begin(), base.end());
        ivs.push_back(int(total));
    }
    const int n_iv = int((ivs.size() - 1) / 3), total = ivs.back();
    Flags* hf = cx->h_flags;
    Flags* fl = nullptr;                 // the flags on the device (first part of the upload block, set below)
    auto fetch_flags = [&]() -> int {
        CTO_HIP(hipMemcpyAsync(hf, fl, sizeof(Flags), hipMemcpyDeviceToHost, s));
        CTO_HIP(sleepy_sync(cx, s));
        return CTO_OK;
    };
    auto empty_result = [&]() -> int {
        std::unique_ptr<cto_pack> p(new cto_pack());
        p->col_off.push_back(0);
        p->key_off.push_back(0);
        p->key_str_off.push_back(0);
        memset(dev_view, 0, sizeof(*dev_view));
        // an empty pack still has its two one-element offset arrays (col_off = key_off = {0}) and valid pointers everywhere
        int rc0;
        if ((rc0 = cx->col_off.ensure(64)) || (rc0 = cx->key_off.ensure(64)) || (rc0 = cx->col_pos.ensure(64)) || (rc0 = cx->col_ref.ensure(64)) ||
            (rc0 = cx->entries.ensure(64)) || (rc0 = cx->key_meta.ensure(64)) || (rc0 = cx->key_group.ensure(64)))
            return rc0;
        CTO_HIP(hipMemsetAsync(cx->col_off.p, 0, 64, s));
        CTO_HIP(hipMemsetAsync(cx->key_off.p, 0, 64, s));
        CTO_HIP(sleepy_sync(cx, s));
        dev_view->col_pos = cx->col_pos.as<int32_t>();
        dev_view->col_ref = cx->col_ref.as<uint8_t>();
        dev_view->col_off = cx->col_off.as<int64_t>();
        dev_view->key_off = cx->key_off.as<int32_t>();
        dev_view->entries = cx->entries.as<uint32_t>();
        dev_view->key_meta = cx->key_meta.as<uint8_t>();
        dev_view->key_group = cx->key_group.as<int32_t>();
        *host_lite = p.release();
        return CTO_OK;
    };
    if (total == 0) return empty_result();
    int rc;
    if ((rc = cx->lin.ensure(size_t(len) + 64)) || (rc = cx->counts.ensure(size_t(n_chains + 1) * 4)) || (rc = cx->base.ensure(size_t(n_chains + 2) * 4)) ||
        (rc = cx->diff.ensure(size_t(total + 1) * 4)) || (rc = cx->slot_col.ensure(size_t(total) * 4)) ||
        (rc = cx->col_slot.ensure(size_t(total) * 4)) || (rc = cx->col_off.ensure(size_t(total + 1) * 8)))
        return rc;
    Flags init{};
    init.stop_idx = init.err_idx = init.paired_idx = init.skip_idx = 0x7fffffff;
    *hf = init;
    // ---- one upload: flags | linear offsets | block table | record starts | intervals | reference window (256-byte aligned parts) ----
    const void* up_src[6] = {&init, lin_off.data(), h_blocks, starts.data(), ivs.data(), ref_seq};
    const size_t up_bytes[6] = {sizeof(Flags), lin_off.size() * 8, size_t(n_blocks) * sizeof(cto_bgzf_block), starts.size() * 8, ivs.size() * 4, ref_len};
    size_t up_off[6], up_total = 0;
    for (int i = 0; i < 6; ++i) { up_off[i] = up_total; up_total += (up_bytes[i] + 255) / 256 * 256 + 256; }
    if ((rc = cx->up.ensure(up_total)) || (rc = cx->up_ensure(up_total))) return rc;
    for (int i = 0; i < 6; ++i) memcpy(static_cast<char*>(cx->h_up) + up_off[i], up_src[i], up_bytes[i]);
    CTO_HIP(hipMemcpyAsync(cx->up.p, cx->h_up, up_total, hipMemcpyHostToDevice, s));
    char* const d_up = static_cast<char*>(cx->up.p);
    fl = reinterpret_cast<Flags*>(d_up + up_off[0]);
    const int64_t* d_lin_off = reinterpret_cast<const int64_t*>(d_up + up_off[1]);
    const cto_bgzf_block* d_blocks = reinterpret_cast<const cto_bgzf_block*>(d_up + up_off[2]);
    const int64_t* d_starts = reinterpret_cast<const int64_t*>(d_up + up_off[3]);
    int* d_ivs = reinterpret_cast<int*>(d_up + up_off[4]);
    CTO_HIP(hipMemsetAsync(cx->diff.p, 0, size_t(total + 1) * 4, s));
    const uint8_t* lin = cx->lin.as<uint8_t>();
    if (!cx->z1k_ready) {                                     // "1024 zero bytes" as a bit matrix, once per context
        uint32_t tbl[256], z[32];
        for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1; tbl[i] = c; }
        for (int i = 0; i < 32; ++i) { uint32_t c = 1u << i; for (int k = 0; k < 1024; ++k) c = tbl[c & 0xFFu] ^ (c >> 8); z[i] = c; }
        if ((rc = cx->z1k.ensure(sizeof(z)))) return rc;
        CTO_HIP(hipMemcpy(cx->z1k.p, z, sizeof(z), hipMemcpyHostToDevice));
        cx->z1k_ready = true;
    }
    hipLaunchKernelGGL(k_crc32_blocks, dim3(unsigned(std::min<int64_t>(n_blocks, 4096))), dim3(64), 0, s, static_cast<const uint8_t*>(d_inflated),
                       d_blocks, int(n_blocks), cx->z1k.as<uint32_t>(), fl);
    hipLaunchKernelGGL(k_linearise, dim3(unsigned(n_blocks)), dim3(256), 0, s, static_cast<const uint8_t*>(d_inflated), d_blocks,
                       d_lin_off, cx->lin.as<uint8_t>());
    // ---- record boundaries ----
    const unsigned cgrid = unsigned(cdiv(n_chains, 64));
    hipLaunchKernelGGL(k_chain, dim3(cgrid), dim3(64), 0, s, lin, len, d_starts, n_chains, 0, cx->counts.as<int>(), nullptr, nullptr, fl);
    hipLaunchKernelGGL(k_scan_small<int>, dim3(1), dim3(1024), 0, s, cx->counts.as<int>(), cx->base.as<int>(), n_chains, &fl->n_rec);
    CTO_HIP(hipGetLastError());
    if ((rc = fetch_flags())) return rc;
    if (hf->bad_crc) {
        set_error("cto_pileup_device: the BGZF block at file offset %llu fails its CRC-32", (unsigned long long)h_blocks[hf->bad_crc - 1].file_off);
        return CTO_EINVAL;
    }
    if (hf->bad_chain) {
        set_error(hf->bad_chain == 2 ? "cto_pileup_device: bad alignment block size" : "cto_pileup_device: an index offset is not a record boundary");
        return CTO_EINVAL;
    }
    const int n_rec = hf->n_rec;
    if (n_rec == 0) return empty_result();
    if ((rc = cx->rec_off.ensure(size_t(n_rec) * 4)) || (rc = cx->reads.ensure(size_t(n_rec) * sizeof(DevRead))) || (rc = cx->rid.ensure(size_t(n_rec) * 4)) ||
        (rc = cx->live.ensure(size_t(n_rec + 1) * 4)))
        return rc;
    CTO_HIP(hipMemsetAsync(cx->live.p, 0, size_t(n_rec + 1) * 4, s));
    hipLaunchKernelGGL(k_chain, dim3(cgrid), dim3(64), 0, s, lin, len, d_starts, n_chains, 1, cx->counts.as<int>(), cx->base.as<int>(),
                       cx->rec_off.as<uint32_t>(), fl);
    hipLaunchKernelGGL(k_parse, dim3(unsigned(cdiv(n_rec, 128))), dim3(128), 0, s, lin, cx->rec_off.as<uint32_t>(), n_rec, tid, int(start - 1), int(end),
                       excl_flags, min_mq, cx->reads.as<DevRead>(), fl);
    hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, s, cx->reads.as<DevRead>(), n_rec, cx->rid.as<int>(), fl);
    {
        hipLaunchKernelGGL(k_live_marks, dim3(unsigned(cdiv(n_rec, 128))), dim3(128), 0, s, cx->reads.as<DevRead>(), cx->rid.as<int>(), fl, cx->live.as<int>());
        hipLaunchKernelGGL(k_live_max, dim3(1), dim3(1024), 0, s, cx->live.as<int>(), fl);
    }
    Ivs iv{d_ivs, d_ivs + n_iv, d_ivs + 2 * n_iv, n_iv, total};
    hipLaunchKernelGGL(k_cover, dim3(unsigned(cdiv(n_rec, 128))), dim3(128), 0, s, cx->reads.as<DevRead>(), cx->rid.as<int>(), fl, iv, cx->diff.as<int>());
    {
        const int tiles = int(cdiv(total, SCAN_TILE));
        if ((rc = cx->tile_a.ensure(size_t(tiles + 1) * 8)) || (rc = cx->tile_b.ensure(size_t(tiles + 1) * 16)) || (rc = cx->tile_tot.ensure(64))) return rc;
        long long* ta = cx->tile_a.as<long long>();
        long long* tb = cx->tile_b.as<long long>();
        long long* tt = cx->tile_tot.as<long long>();
        hipLaunchKernelGGL(k_tile_sums, dim3(unsigned(tiles)), dim3(1024), 0, s, cx->diff.as<int>(), total, ta);
        hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, s, ta, tiles, 1, tt + 2);
        hipLaunchKernelGGL(k_columns_count, dim3(unsigned(tiles)), dim3(1024), 0, s, cx->diff.as<int>(), total, ta, tb);
        hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, s, tb, tiles, 2, tt);
        hipLaunchKernelGGL(k_columns_write, dim3(unsigned(tiles)), dim3(1024), 0, s, cx->diff.as<int>(), total, ta, tb, tt, cx->slot_col.as<int>(),
                           cx->col_slot.as<int>(), cx->col_off.as<long long>(), fl);
    }
    CTO_HIP(hipGetLastError());
    if ((rc = fetch_flags())) return rc;
    const int lim = hf->stop_idx;
    if (hf->err_idx < lim) { set_error("cto_pileup_device: alignment record shorter than its fields, or running past 2^31 - 1"); return CTO_EINVAL; }
    if (hf->paired_idx < lim || hf->skip_idx < lim || hf->deep_col || (max_depth > 0 && hf->max_live >= max_depth) || hf->max_live >= FILL_LIVE_MAX) {
        *fallback = 1;
        return CTO_OK;
    }
    const int n_cols = hf->n_cols;
    const long long n_entries = hf->n_entries;
    if (n_cols == 0) return empty_result();
    if ((rc = cx->col_pos.ensure(size_t(n_cols) * 4)) || (rc = cx->col_ref.ensure(size_t(n_cols))) || (rc = cx->cursor.ensure(size_t(n_cols) * 4)) ||
        (rc = cx->tmp.ensure(size_t(n_entries) * sizeof(TmpEnt))) || (rc = cx->entries.ensure(size_t(n_entries) * 4)) ||
        (rc = cx->nkc.ensure(size_t(n_cols + 1) * 4)) || (rc = cx->keyrec.ensure(size_t(n_entries) * sizeof(KeyRec))) || (rc = cx->key_off.ensure(size_t(n_cols + 1) * 4)))
        return rc;
    CTO_HIP(hipMemsetAsync(cx->cursor.p, 0, size_t(n_cols) * 4, s));
    const char* d_ref = d_up + up_off[5];
    hipLaunchKernelGGL(k_col_meta, dim3(unsigned(cdiv(n_cols, 256))), dim3(256), 0, s, cx->col_slot.as<int>(), n_cols, iv, d_ref, (long long)ref_start,
                       (long long)ref_len, cx->col_pos.as<int32_t>(), cx->col_ref.as<uint8_t>(), fl);
    int live_cap = 64;
    while (live_cap <= hf->max_live) live_cap <<= 1;
    hipLaunchKernelGGL(k_fill, dim3(unsigned(((hf->n_valid + 7) >> 3) << 3)), dim3(256), size_t(live_cap) * 8 * sizeof(int), s, live_cap, lin, cx->reads.as<DevRead>(), cx->rid.as<int>(), fl, iv,
                       cx->slot_col.as<int>(), cx->col_off.as<long long>(), cx->cursor.as<int>(), cx->entries.as<uint32_t>(), cx->tmp.as<TmpEnt>(), d_ref,
                       (long long)ref_start, (long long)ref_len);
    hipLaunchKernelGGL(k_order, dim3(unsigned(std::min<long long>(cdiv(n_cols, 64), 65536))), dim3(64), 0, s, lin, cx->reads.as<DevRead>(), cx->rid.as<int>(),
                       n_cols, cx->col_off.as<long long>(), cx->cursor.as<int>(), cx->tmp.as<TmpEnt>(), cx->entries.as<uint32_t>(), cx->nkc.as<int>(),
                       cx->keyrec.as<KeyRec>(), max_indel_length, fl);
    if (n_cols <= 4 * SCAN_TILE) {
        hipLaunchKernelGGL(k_scan_small<int>, dim3(1), dim3(1024), 0, s, cx->nkc.as<int>(), cx->key_off.as<int>(), n_cols, &fl->n_keys);
    } else {
        const int tiles = int(cdiv(n_cols, SCAN_TILE));
        if ((rc = cx->tile_a.ensure(size_t(tiles + 1) * 8)) || (rc = cx->tile_tot.ensure(64))) return rc;
        hipLaunchKernelGGL(k_tile_sums, dim3(unsigned(tiles)), dim3(1024), 0, s, cx->nkc.as<int>(), n_cols, cx->tile_a.as<long long>());
        hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, s, cx->tile_a.as<long long>(), tiles, 1, cx->tile_tot.as<long long>());
        hipLaunchKernelGGL(k_scan_apply<int>, dim3(unsigned(tiles)), dim3(1024), 0, s, cx->nkc.as<int>(), cx->key_off.as<int>(), n_cols,
                           cx->tile_a.as<long long>(), cx->tile_tot.as<long long>(), &fl->n_keys);
    }
    CTO_HIP(hipGetLastError());
    if ((rc = fetch_flags())) return rc;
    if (hf->ref_oob) { set_error("cto_pileup_device: a covered position lies outside the supplied reference"); return CTO_EINVAL; }
    if (hf->many_keys || hf->deep_col) { *fallback = 1; return CTO_OK; }          // deep_col here: more than FILL_LIVE reads open at a read's start
    const int n_keys = hf->n_keys;
    std::unique_ptr<cto_pack> lite(new cto_pack());
    lite->col_pos.resize(size_t(n_cols));
    lite->col_ref.resize(size_t(n_cols));
    lite->key_off.resize(size_t(n_cols) + 1);
    lite->col_off.assign(1, 0);
    lite->key_str_off.assign(size_t(n_keys) + 1, 0);
    if ((rc = cx->key_meta.ensure(16)) || (rc = cx->key_group.ensure(16))) return rc;      // valid pointers for a pack without keys
    if (n_keys > 0) {
        if ((rc = cx->key_meta.ensure(size_t(n_keys))) || (rc = cx->key_group.ensure(size_t(n_keys) * 4)) || (rc = cx->key_final.ensure(size_t(n_keys) * sizeof(KeyRec))) ||
            (rc = cx->key_col.ensure(size_t(n_keys) * 4)) || (rc = cx->key_len.ensure(size_t(n_keys + 1) * 4)) || (rc = cx->str_off.ensure(size_t(n_keys + 1) * 8)))
            return rc;
        hipLaunchKernelGGL(k_keys_meta, dim3(unsigned(cdiv(n_cols, 256))), dim3(256), 0, s, n_cols, cx->col_off.as<long long>(), cx->key_off.as<int>(),
                           cx->keyrec.as<KeyRec>(), cx->col_pos.as<int32_t>(), (long long)ref_start, (long long)ref_len, max_indel_length,
                           cx->key_meta.as<uint8_t>(), cx->key_group.as<int32_t>(), cx->key_final.as<KeyRec>(), cx->key_col.as<int>(), cx->key_len.as<int>());
        hipLaunchKernelGGL(k_scan_small<long long>, dim3(1), dim3(1024), 0, s, cx->key_len.as<int>(), cx->str_off.as<long long>(), n_keys, &fl->key_str_bytes);
        CTO_HIP(hipGetLastError());
        if ((rc = fetch_flags())) return rc;
        const long long sb = hf->key_str_bytes;
        if ((rc = cx->key_str.ensure(size_t(sb) + 16))) return rc;
        hipLaunchKernelGGL(k_keys_str, dim3(unsigned(cdiv(n_keys, 128))), dim3(128), 0, s, n_keys, cx->key_final.as<KeyRec>(), cx->key_col.as<int>(),
                           cx->str_off.as<long long>(), lin, cx->reads.as<DevRead>(), cx->rid.as<int>(), cx->col_pos.as<int32_t>(), d_ref,
                           (long long)ref_start, cx->key_str.as<char>());
        CTO_HIP(hipGetLastError());
        lite->key_str.resize(size_t(sb));
        lite->key_meta.resize(size_t(n_keys));
        lite->key_group.resize(size_t(n_keys));
        static_assert(sizeof(long long) == sizeof(int64_t), "");
    }
    {
        // everything the host keeps of the pack, through ONE page-locked landing area and one sleeping wait
        const size_t sb = lite->key_str.size(), nk = size_t(n_keys), nc = size_t(n_cols);
        const size_t bytes[7] = {nk ? (nk + 1) * 8 : 0, sb, nk, nk * 4, nc * 4, nc, (nc + 1) * 4};
        const void* src[7] = {cx->str_off.p, cx->key_str.p, cx->key_meta.p, cx->key_group.p, cx->col_pos.p, cx->col_ref.p, cx->key_off.p};
        void* dst[7] = {lite->key_str_off.data(), sb ? &lite->key_str[0] : nullptr, lite->key_meta.data(), lite->key_group.data(), lite->col_pos.data(),
                        lite->col_ref.data(), lite->key_off.data()};
        size_t off[7], total = 0;
        for (int i = 0; i < 7; ++i) { off[i] = total; total += (bytes[i] + 63) / 64 * 64; }
        if ((rc = cx->stage_ensure(total + 64))) return rc;
        char* hs = static_cast<char*>(cx->h_stage);
        for (int i = 0; i < 7; ++i)
            if (bytes[i]) CTO_HIP(hipMemcpyAsync(hs + off[i], src[i], bytes[i], hipMemcpyDeviceToHost, s));
        CTO_HIP(sleepy_sync(cx, s));
        for (int i = 0; i < 7; ++i)
            if (bytes[i]) memcpy(dst[i], hs + off[i], bytes[i]);
    }
    dev_view->n_cols = n_cols;
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
CTO_CATCH("cto_pileup_device", int)

// test / tool aid: n bytes of device memory to the host (synchronous)
extern "C" int cto_device_read(const void* d_src, void* h_dst, size_t n) {
    CTO_REQUIRE((d_src && h_dst) || n == 0, CTO_EINVAL, "cto_device_read: null argument");
    if (n) CTO_HIP(hipMemcpy(h_dst, d_src, n, hipMemcpyDeviceToHost));
    return CTO_OK;
}
