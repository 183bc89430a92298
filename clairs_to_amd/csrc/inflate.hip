// BGZF / DEFLATE (RFC 1951) decompression on the device: one wavefront per BGZF block.
//
// Why: the native BAM -> pack producer (bam.cpp) is bound by inflate on the host - 80 % of the 210 core-ms
// a 1 Mb x 50x chunk costs is file read + libdeflate + record parsing - while the device idles.  BGZF blocks
// are independent DEFLATE streams of at most 64 KiB, so a chunk's ~750 blocks inflate side by side.
//
// A DEFLATE stream is decoded serially, symbol by symbol, so the wave works as ONE decoder whose 64 lanes do the wide steps:
//   * bit reader: the compressed bytes are fetched 256 B at a time (one coalesced dword per lane, the next 256 B already in
//     flight) and handed to a 64-bit bit buffer dword by dword with v_readlane;
//   * Huffman decode: codes of up to 10 (literal / length) and 9 (distance) bits - nearly all of them - through a lookup table in
//     LDS indexed by the next bits of the stream, built by all lanes per DEFLATE block; longer codes without a table: the codes
//     are canonical, so lane L (1..15) holds first_code[L], count[L] and offset[L], bit-reverses the next L bits and tests
//     first <= code < first + count - exactly one lane hits; a ballot names the length, a readlane fetches offset + code - first,
//     the symbol comes out of the sorted symbol list held in 5 + 1 registers per lane;
//   * LZ77 window: the output buffer itself.  Literals are byte stores; matches are queued and resolved 256 at a time, one match
//     per lane, with loads that bypass the vector L1 (source index (k mod distance) for overlapping copies).
//     A 32 KiB ring in LDS was the first version: 4 wavefronts per CU, and a single decoder wave is a chain of dependent
//     scalar instructions, branches and one LDS round trip per symbol (500 cycles per symbol measured) - the chip inflated 36
//     chunks a second.  With 4 KB of LDS per wave, six waves per SIMD take turns on that chain.
// Measured on MI355X (tools/inflate_bench.py, a 1 Mb x 50x chunk: 77 MB of BAM in 1776 blocks -> 115 MB): 20 ms for one launch
// alone (15 ms with the literal stores taken out: the decode chain, not memory, is the cost), 8 ms per chunk with 4-8 launches in
// flight = 14 GB/s of inflated bytes (5.5 ms = 20.9 GB/s with the literal loop and the match queue below), about what 25 host
// cores of libdeflate deliver.  It pays beside the host cores, not instead
// of them: the chunk pipeline (pipeline.hip) sends some chunks through it on streams confined to part of the CUs - unconfined, the
// waves of a launch sit on every CU for tens of milliseconds and the networks' block kernels wait for them - and BAM -> VCF goes
// from 240 k sites/s (16 host cores) to 337-368 k.
// Where a block's 40 M cycles go (s_memtime stamps, a 65 000-byte block of BAM records = 21.6 k literals + 12.7 k matches of 3.4
// bytes on average): a link of the decode chain is ~80 scalar-unit and lane-0 instructions with a dozen taken branches, and a
// single wave issues such code at 10-16 cycles per instruction - 1.3 k cycles per symbol, where a host core needs ~10.  Tried on
// top and dropped: a 16 KiB LDS ring for near matches (the copy itself was not the cost; fewer waves per CU: 14-18 ms per chunk);
// literal runs decoded by all lanes at once (lane i looks up the symbol at bit offset i, a scalar walk follows the chain: correct,
// but BAM records break the run every 1.7 literals: 9.9 ms).  A device decoder that beats the host needs one block per LANE.
// (Round 6 built that form in two phases - tools/experiments/inflate_lanes_two_phase.hip - and measured it: correct, 7 x slower than this kernel;
// a wavefront whose lanes decode 64 different streams executes the union of their code paths at every step.)
// Later in the round, with 8 launches in flight (ms per chunk; 7.0 at that point): a literal loop of its own - table hit with a
// literal flag, all-lanes byte store, shift; the bit count derived from the reader's position instead of updated per symbol; the
// input bound checked where a dword is handed out - 6.2 (kept: ~27 instructions per literal instead of ~50).  8 instead of 7 waves
// per SIMD (63 VGPRs): no change - the launches are latency-, not occupancy-bound.  Skipping the store drain for matches that
// reach behind the last known-drained output position (two thirds reach more than 2 KB back): no change - waiting for the
// window load is a vmcnt(0), stores included.  Matches loaded into LDS slots with global_load_lds_ubyte and stored eight
// matches later under one wait (the window of 7 000 resident waves does not stay in the L2s; a source byte is a microsecond
// away and nothing downstream in the stream needs it): 8.7, slower, and not pursued to correctness.  The same idea without the
// LDS-DMA - matches queued in LDS (destination, source, length) and resolved 256 at a time by all lanes, one match per lane, in
// rounds that respect the dependencies (resolve_matches) - 5.8 (kept).  That it is worth 7 % and not a factor says where the
// bound is: the decoder is wave-uniform code, ~40 of its ~65 instructions per symbol run on the scalar unit, a CU has ONE scalar
// unit for its four SIMDs, and 256 CUs x 2.1 GHz / 40 = 13 G symbols/s = 17 GB/s - what is measured.  Waves per SIMD, memory
// latency and store drains are second-order once a CU holds enough waves to keep that unit busy.  Consequently the literal loop
// keeps its bit buffer in vector registers (13 scalar + 11 vector instructions per literal instead of 20 + 7): 5.5.
// Every loop is bounded by the block's compressed size (a symbol consumes at least one bit) or by constants; malformed input
// ends with a status code, never with a hang or an out-of-range access (the input buffer carries CTO_BGZF_PAD bytes of padding,
// every output slot is padded to 256 bytes).
#include <stdlib.h>
#include "common.h"

namespace {

#ifndef CTO_INF_TBL
#define CTO_INF_TBL 9
#endif
#ifndef CTO_INF_TBD
#define CTO_INF_TBD 8
#endif
#ifndef CTO_INF_TOK
#define CTO_INF_TOK 128
#endif
constexpr int TBL = CTO_INF_TBL, TBD = CTO_INF_TBD;       // index bits of the literal / length and the distance lookup tables

// base value and number of extra bits of length symbol ls = sym - 257 (0..28) and of distance symbol ds (0..29), RFC 1951 3.2.5, in
// closed form: the tables would be global-memory loads whose s_waitcnt also drains the byte stores in flight
__device__ __forceinline__ void len_code(int ls, int* base, int* extra) {
    const int e = ls < 8 ? 0 : (ls >> 2) - 1;
    *extra = ls == 28 ? 0 : e;
    *base = ls < 8 ? 3 + ls : (ls == 28 ? 258 : 3 + ((4 + (ls & 3)) << e));
}
__device__ __forceinline__ void dist_code(int ds, int* base, int* extra) {
    const int e = ds < 4 ? 0 : (ds >> 1) - 1;
    *extra = e;
    *base = ds < 4 ? 1 + ds : 1 + ((2 + (ds & 1)) << e);
}
__constant__ uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__device__ __forceinline__ uint32_t rl(uint32_t v, int lane) {      // value of `v` in lane `lane` (wave-uniform lane index)
    return uint32_t(__builtin_amdgcn_readlane(int(v), __builtin_amdgcn_readfirstlane(lane)));
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

typedef const __attribute__((address_space(1))) uint32_t* gdword_ptr;
struct Bits {                     // wave-uniform bit reader over dwords of global memory
    gdword_ptr src;               // dword-aligned base
    uint32_t win;                 // per lane: dword (wbase + lane).  The next 256 bytes are NOT kept in flight: a load pending across
                                  // loop iterations makes the compiler wait for it - and with it for every byte store in flight -
                                  // on each iteration; one exposed load per 256 bytes of input costs ~3 cycles per symbol
    int wbase, widx;              // dword index of win's lane 0; next dword to hand out
    uint64_t bb;                  // bit buffer, next bit = bit 0
    int cnt;                      // valid bits in bb
    int skip;                     // bits of the first dword that precede the stream
    int limit;                    // last dword index that may be handed out; beyond it the stream reads as zeros and `over` is set
    int over;
};
// bits consumed so far: what was handed out minus what is still buffered (kept out of bits_drop: two scalar instructions per symbol)
__device__ __forceinline__ long long bits_used(const Bits& b) { return (long long)b.widx * 32 - b.cnt - b.skip; }

__device__ __forceinline__ void bits_init(Bits& b, const uint8_t* p, long long in_bits, int lane) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    b.src = reinterpret_cast<gdword_ptr>(a & ~uintptr_t(3));
    b.wbase = 0;
    b.widx = 0;
    b.win = b.src[lane];
    b.bb = 0;
    b.cnt = 0;
    b.over = 0;
    const int skip = int(a & 3) * 8;       // bits of the first dword that precede the stream
    b.skip = skip;
    b.limit = int((skip + in_bits + 31) / 32) + 2;      // a decoder that runs ahead of a valid stream never needs more
    // first refill, then drop the leading bits
    b.bb = uint64_t(rl(b.win, 0)) | (uint64_t(rl(b.win, 1)) << 32);
    b.widx = 2;
    b.cnt = 64 - skip;
    b.bb >>= skip;
}
__device__ __forceinline__ void bits_refill(Bits& b, int lane) {
    if (b.cnt <= 32) {
        uint32_t d = 0;
        if (b.widx <= b.limit) {
            if (b.widx - b.wbase >= 64) {
                b.wbase += 64;
                b.win = b.src[b.wbase + lane];
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) HERE, once per 256 bytes: otherwise the compiler waits where the two
                                                     // paths meet, i.e. on every refill, and each wait also drains the byte stores
            }
            d = rl(b.win, b.widx - b.wbase);
        } else {
            b.over = 1;                              // malformed: the stream wants more than its payload holds; zeros from here on,
        }                                            // no load past the padding - every loop ends on its output bound
        b.bb |= uint64_t(d) << b.cnt;
        b.cnt += 32;
        ++b.widx;
    }
}
__device__ __forceinline__ uint32_t bits_peek(const Bits& b, int n) { return uint32_t(b.bb) & ((1u << n) - 1u); }
__device__ __forceinline__ void bits_drop(Bits& b, int n) { b.bb >>= n; b.cnt -= n; }

struct Huff {                     // per lane L: the codes of length L (lanes 1..15); sorted symbols: entry i in register i / 64, lane i % 64
    uint32_t first, count, offset;
    uint32_t sym[5];
};

// lens[0..n) (LDS) -> canonical decoder; scratch: sorted symbol list (LDS, n entries).  Returns false for an over-subscribed code.
template <int NREG>
__device__ bool huff_build(Huff& h, const uint8_t* lens, int n, uint16_t* sorted, int lane) {
    uint32_t cnt = 0;
    if (lane >= 1 && lane <= 15)
        for (int s = 0; s < n; ++s) cnt += (lens[s] == lane);
    h.count = cnt;
    uint32_t first = 0, offset = 0, code = 0, off = 0;
    int left = 1;
    bool ok = true;
#pragma unroll
    for (int L = 1; L <= 15; ++L) {
        const uint32_t c = rl(cnt, L);
        code <<= 1;                       // first code of length L
        left = (left << 1) - int(c);
        ok = ok && left >= 0;
        if (lane == L) { first = code; offset = off; }
        code += c;
        off += c;
    }
    h.first = first;
    h.offset = offset;
    if (lane >= 1 && lane <= 15) {        // lane L lists its symbols in increasing order behind offset[L]
        uint32_t k = offset;
        for (int s = 0; s < n; ++s)
            if (lens[s] == lane) sorted[k++] = uint16_t(s);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
        const int i = r * 64 + lane;
        h.sym[r] = i < n ? sorted[i] : 0u;
    }
    return ok;
}

// next symbol of the code (bits32 = the next 32 bits of the stream, bit 0 first); *len = its code length; -1: no such code
template <int NREG>
__device__ __forceinline__ int huff_decode(const Huff& h, uint32_t bits32, int lane, int* len) {
    const uint32_t rev = __brev(bits32);
    const bool live = lane >= 1 && lane <= 15;
    const uint32_t code = live ? rev >> (32 - lane) : 0u;
    const uint32_t rel = code - h.first;
    const bool hit = live && rel < h.count;
    const uint64_t m = __ballot(hit);
    if (m == 0) return -1;
    const int l = __ffsll((long long)m) - 1;
    const int idx = int(rl(h.offset + rel, l));
    *len = l;
    const int reg = idx >> 6, ln = idx & 63;
    uint32_t v = rl(h.sym[0], ln);
#pragma unroll
    for (int r = 1; r < NREG; ++r) {
        const uint32_t w = rl(h.sym[r], ln);
        v = reg == r ? w : v;
    }
    return int(v);
}

// Lookup table of the codes of up to TB bits: entry[next TB bits of the stream] = symbol | length << 9 (0: a longer code); with
// LITFLAG bit 15 marks the symbols below 256, so that the literal loop tests one bit.
// `fo`: scratch for first[16] | offset[16].  sorted[] / lens[] as left by huff_build.
template <int TB, bool LITFLAG = false>
__device__ void huff_table(const Huff& h, const uint8_t* lens, const uint16_t* sorted, uint32_t* fo, uint16_t* table, int lane) {
    if (lane < 16) { fo[lane] = h.first; fo[16 + lane] = h.offset; }
    for (int i = lane; i < (1 << TB); i += 64) table[i] = 0;
    const int total = int(rl(h.offset + h.count, 15));         // symbols that have a code
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < total; i += 64) {
        const uint32_t sym = sorted[i];
        const uint32_t len = lens[sym];
        if (len <= uint32_t(TB)) {
            const uint32_t code = fo[len] + (uint32_t(i) - fo[16 + len]);
            const uint32_t rev = __brev(code) >> (32 - len);
            const uint16_t e = uint16_t(sym | (len << 9) | ((LITFLAG && sym < 256) ? 0x8000u : 0u));
            for (uint32_t k = rev; k < (1u << TB); k += (1u << len)) table[k] = e;
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
}
template <int NREG, int TB>
__device__ __forceinline__ int huff_decode_fast(const Huff& h, const uint16_t* table, uint32_t bits32, int lane, int* len) {
    const uint32_t e = uint32_t(__builtin_amdgcn_readfirstlane(int(table[bits32 & ((1u << TB) - 1u)])));
    if (e != 0) { *len = int((e >> 9) & 15u); return int(e & 511u); }
    return huff_decode<NREG>(h, bits32, lane, len);
}

// ---- lookup tables of the symbol loop: 32-bit entries that carry everything a symbol needs --------------------------------------
// (libdeflate's idea: the loop never maps a symbol to its base value / extra-bit count - the table entry holds them)
//   literal / length table [next TBL bits]:  literal  E_LIT | code length << 8 | byte
//                                            end of block E_EOB | code length << 8
//                                            length   E_VAL | (code length + extra bits) << 24 | code length << 16 | extra bits << 12 | base (3..258)
//   distance table [next TBD bits]:          E_VAL | (code length + extra bits) << 24 | code length << 20 | extra bits << 16 | base (1..24577)
//   0: the code is longer than the index (canonical search, huff_decode);  E_BAD: a symbol RFC 1951 reserves (286, 287; 30, 31)
// The classes compare as unsigned numbers: literal > end of block > value > (E_BAD, 0).
constexpr uint32_t E_LIT = 0x80000000u, E_EOB = 0x40000000u, E_VAL = 0x20000000u, E_BAD = 1u;
__device__ __forceinline__ uint32_t litlen_entry(uint32_t sym, uint32_t len) {
    if (sym < 256u) return E_LIT | (len << 8) | sym;
    if (sym == 256u) return E_EOB | (len << 8);
    const int ls = int(sym) - 257;
    if (ls >= 29) return E_BAD;
    int base, extra;
    len_code(ls, &base, &extra);
    return E_VAL | ((len + uint32_t(extra)) << 24) | (len << 16) | (uint32_t(extra) << 12) | uint32_t(base);
}
__device__ __forceinline__ uint32_t dist_entry(uint32_t sym, uint32_t len) {
    if (sym >= 30u) return E_BAD;
    int base, extra;
    dist_code(int(sym), &base, &extra);
    return E_VAL | ((len + uint32_t(extra)) << 24) | (len << 20) | (uint32_t(extra) << 16) | uint32_t(base);
}
template <int TB, bool DIST>
__device__ void huff_table32(const Huff& h, const uint8_t* lens, const uint16_t* sorted, uint32_t* fo, uint32_t* table, int lane) {
    if (lane < 16) { fo[lane] = h.first; fo[16 + lane] = h.offset; }
    for (int i = lane; i < (1 << TB); i += 64) table[i] = 0;
    const int total = int(rl(h.offset + h.count, 15));         // symbols that have a code
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < total; i += 64) {
        const uint32_t sym = sorted[i];
        const uint32_t len = lens[sym];
        if (len <= uint32_t(TB)) {
            const uint32_t code = fo[len] + (uint32_t(i) - fo[16 + len]);
            const uint32_t rev = __brev(code) >> (32 - len);
            const uint32_t e = DIST ? dist_entry(sym, len) : litlen_entry(sym, len);
            for (uint32_t k = rev; k < (1u << TB); k += (1u << len)) table[k] = e;
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
}

enum { ST_OK = 0, ST_BAD_BTYPE = 1, ST_BAD_STORED = 2, ST_BAD_TABLE = 3, ST_BAD_CODE = 4, ST_BAD_DIST = 5, ST_OVERRUN_OUT = 6, ST_OVERRUN_IN = 7, ST_SHORT = 8, ST_BAD_SLOT = 9 };

}  // namespace

#ifdef CTO_INF_WAVES
#define CTO_INF_ATTR __attribute__((amdgpu_waves_per_eu(CTO_INF_WAVES, CTO_INF_WAVES)))
#else
#define CTO_INF_ATTR
#endif
extern "C" __global__ __launch_bounds__(64) CTO_INF_ATTR void k_bgzf_inflate(const uint8_t* __restrict__ comp, const cto_bgzf_block* __restrict__ blocks,
                                                                int n_blocks, uint8_t* __restrict__ out, int* __restrict__ status) {
    __shared__ uint8_t lens[320];
    __shared__ uint16_t sorted[320];
    __shared__ uint32_t fo[32];
    __shared__ uint32_t tab_l[1 << TBL];
    __shared__ uint32_t tab_d[1 << TBD];
    const int lane = threadIdx.x;
    const int blk = blockIdx.x;
    if (blk >= n_blocks) return;
    const cto_bgzf_block bd = blocks[blk];
    const int isize = int(bd.isize);
    // the literal path checks its output bound once per refill and may store up to 32 bytes behind isize: every slot must be followed by
    // CTO_BGZF_SLOT_PAD bytes that belong to nobody (cto_bgzf_scan lays them out so).  A table laid out by an older rule (isize + 4) is
    // refused here, block by block, instead of corrupting its neighbour's output; the pad behind the LAST slot is the caller's to provide
    // (cto_bgzf_scan's *out_bytes includes it).
    if (blk + 1 < n_blocks && blocks[blk + 1].out_off < bd.out_off + uint64_t(isize) + CTO_BGZF_SLOT_PAD && blocks[blk + 1].out_off >= bd.out_off) {
        if (threadIdx.x == 0) status[blk] = ST_BAD_SLOT;
        return;
    }
    const long long in_bits = (long long)bd.csize * 8;
    uint8_t* dst = out + bd.out_off;
    const uint8_t* win = dst;                // window reads are device-scope atomic loads: not through the vector L1 (it is not
                                             // coherent with this wave's own earlier stores)
    int st = ST_OK;
    int op = 0;                              // bytes produced
    // Matches are not copied where they are decoded: a source byte is a microsecond away (the windows of thousands of resident waves
    // do not stay in the L2s) and nothing that follows in the stream depends on it.  They are queued - destination, source, length -
    // and resolved TOK at a time by all lanes, one match per lane, so that the trip to memory is paid once per 64 matches instead
    // of once per match.  A match whose source ends inside the not-yet-resolved part of the queue waits for a later round
    // (resolve_matches); literals are stored as they are decoded and have landed before a round starts.
    constexpr int TOK = CTO_INF_TOK;
    __shared__ uint64_t tok[TOK];            // destination | source << 16 | length << 32
    int ntok = 0;
    auto resolve_matches = [&]() {
        if (ntok == 0) return;
        __builtin_amdgcn_s_waitcnt(0);       // every literal (and every earlier round's copy) has reached L2; the queue is in LDS
        __builtin_amdgcn_wave_barrier();
        int c = 0;
        uint64_t done = 0;                   // bit i: token c + i is resolved
        while (c < ntok) {
            const int t = c + lane;
            const bool mine = t < ntok && !((done >> lane) & 1);
            const uint64_t tk = mine ? tok[t] : 0;
            const int td = int(tk & 0xffffu), ts = int((tk >> 16) & 0xffffu), tn = int(tk >> 32);
            const int d = td - ts;
            const int first = int(tok[c] & 0xffffu);             // everything below the oldest unresolved destination is final
            const bool ready = mine && ts + (d < tn ? d : tn) <= first;
            if (ready) {
                // the (k mod d) form reads the repeating pattern of an overlapping copy from its first period: no byte of this copy
                // depends on a byte this copy writes
                if (d >= tn) {
                    for (int k = 0; k < tn; ++k) dst[td + k] = __hip_atomic_load(win + ts + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    for (int k = 0, j = 0; k < tn; ++k) {
                        dst[td + k] = __hip_atomic_load(win + ts + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        j = j + 1 == d ? 0 : j + 1;
                    }
                }
            }
            done |= __ballot(ready);
            const int adv = done == ~uint64_t(0) ? 64 : __ffsll((long long)~done) - 1;      // token c is always ready: adv >= 1
            c += adv;
            done = adv >= 64 ? 0 : done >> adv;
            __builtin_amdgcn_s_waitcnt(0);   // this round's copies have reached L2 before the next round reads them
            __builtin_amdgcn_wave_barrier();
        }
        ntok = 0;
    };

    if (isize > 0) {
        Bits b;
        bits_init(b, comp + bd.in_off, in_bits, lane);
        Huff hl, hd;
        bool final_block = false;
        while (!final_block && st == ST_OK) {
            bits_refill(b, lane);
            final_block = bits_peek(b, 1) != 0;
            const int btype = int(bits_peek(b, 3) >> 1);
            bits_drop(b, 3);
            if (btype == 0) {                                    // stored
                bits_drop(b, int((8 - (bits_used(b) & 7)) & 7));
                bits_refill(b, lane);
                const uint32_t len = bits_peek(b, 16);
                bits_drop(b, 16);
                bits_refill(b, lane);
                const uint32_t nlen = bits_peek(b, 16);
                bits_drop(b, 16);
                if ((len ^ nlen) != 0xffffu) { st = ST_BAD_STORED; break; }
                if (op + int(len) > isize) { st = ST_OVERRUN_OUT; break; }
                if (bits_used(b) + (long long)len * 8 > in_bits + 64) { st = ST_OVERRUN_IN; break; }     // before the copy loop reads on
                for (uint32_t i = 0; i < len; ++i) {
                    bits_refill(b, lane);
                    const uint32_t v = bits_peek(b, 8);
                    bits_drop(b, 8);
                    if (lane == 0) dst[op] = uint8_t(v);
                    ++op;
                }
                if (b.over || bits_used(b) > in_bits + 64) { st = ST_OVERRUN_IN; break; }
                continue;
            }
            if (btype == 3) { st = ST_BAD_BTYPE; break; }
            int nlit = 288, ndist = 30;
            if (btype == 1) {                                    // fixed codes
                for (int i = lane; i < 320; i += 64) lens[i] = uint8_t(i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : (i < 288 ? 8 : 5))));
            } else {                                             // dynamic codes
                bits_refill(b, lane);
                nlit = int(bits_peek(b, 5)) + 257;
                bits_drop(b, 5);
                ndist = int(bits_peek(b, 5)) + 1;
                bits_drop(b, 5);
                const int ncl = int(bits_peek(b, 4)) + 4;
                bits_drop(b, 4);
                if (nlit > 286 || ndist > 30) { st = ST_BAD_TABLE; break; }
                if (lane < 19) lens[lane] = 0;
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_wave_barrier();
                for (int i = 0; i < ncl; ++i) {
                    bits_refill(b, lane);
                    const uint32_t v = bits_peek(b, 3);
                    bits_drop(b, 3);
                    if (lane == 0) lens[kClOrder[i]] = uint8_t(v);
                }
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_wave_barrier();
                Huff hc;
                if (!huff_build<1>(hc, lens, 19, sorted, lane)) { st = ST_BAD_TABLE; break; }
                // the code lengths of both alphabets, run-length coded (the decoded lengths overwrite lens[] from 0 on:
                // hc is complete in registers by now)
                __builtin_amdgcn_wave_barrier();
                int i = 0, prev = 0;
                const int total = nlit + ndist;
                while (i < total && st == ST_OK) {
                    bits_refill(b, lane);
                    int l = 0;
                    const int s = huff_decode<1>(hc, uint32_t(b.bb), lane, &l);
                    if (s < 0) { st = ST_BAD_TABLE; break; }
                    bits_drop(b, l);
                    int rep = 1, val = s;
                    if (s == 16) { if (i == 0) { st = ST_BAD_TABLE; break; } rep = 3 + int(bits_peek(b, 2)); bits_drop(b, 2); val = prev; }
                    else if (s == 17) { rep = 3 + int(bits_peek(b, 3)); bits_drop(b, 3); val = 0; }
                    else if (s == 18) { rep = 11 + int(bits_peek(b, 7)); bits_drop(b, 7); val = 0; }
                    if (i + rep > total) { st = ST_BAD_TABLE; break; }
                    if (lane < rep) lens[i + lane] = uint8_t(val);
                    if (lane + 64 < rep) lens[i + lane + 64] = uint8_t(val);
                    if (lane + 128 < rep) lens[i + lane + 128] = uint8_t(val);
                    i += rep;
                    prev = val;
                    if (b.over || bits_used(b) > in_bits + 64) st = ST_OVERRUN_IN;
                }
                if (st != ST_OK) break;
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_wave_barrier();
                // the distance lengths follow the literal / length ones: move them to lens[288..)
                uint8_t dl = 0;
                if (lane < ndist) dl = lens[nlit + lane];
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_wave_barrier();
                if (lane < 32) lens[288 + lane] = lane < ndist ? dl : uint8_t(0);
                for (int k = nlit + lane; k < 288; k += 64) lens[k] = 0;
            }
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_wave_barrier();
            if (!huff_build<5>(hl, lens, 288, sorted, lane)) { st = ST_BAD_TABLE; break; }
            huff_table32<TBL, false>(hl, lens, sorted, fo, tab_l, lane);
            huff_build<1>(hd, lens + 288, 30, sorted, lane);     // an incomplete distance code is legal (one code, or none)
            huff_table32<TBD, true>(hd, lens + 288, sorted, fo, tab_d, lane);

            // ---- symbols ----
            // The decoder is wave-uniform code, and what bounds a chip full of these waves is plain instruction issue: a CU gets
            // through about ONE instruction per cycle of whatever kind (PMC with 8 launches in flight: SALU + VALU + LDS + VMEM
            // instructions of a launch / 256 CUs / the launch's share of the wall clock = 1.0 at 2.4 GHz - for the round-3 loop's 65
            // instructions per symbol and for a compiler-made vector-register loop's 43 alike).  The two paths nearly every symbol
            // takes - a literal, and a match whose two codes hit the tables - are therefore written out by hand for instruction
            // COUNT: 16 instructions per literal, 59 per match (the compiler's best: 26 / 75).  Bit buffer (two dwords, bits above
            // `cnt` zero) and output position in vector registers (every lane the same value), bit count and queue fill in scalar
            // ones; 32-bit table entries that carry base value, extra-bit count and their sum; the stored byte straight out of the
            // looked-up register; a literal run's output bound checked once per refill (<= 32 literals; the output slots carry
            // CTO_BGZF_SLOT_PAD bytes of padding), a match's exactly.  Everything else leaves the block with a reason code and is
            // handled in C++ below: a refill that needs the next 256 input bytes (or runs past the payload), codes longer than the
            // tables' index, end of block, a full match queue, malformed streams.
            {
                uint32_t vlo, vhi, vop;
                asm volatile("v_mov_b32 %0, %1" : "=v"(vlo) : "s"(uint32_t(b.bb)));
                asm volatile("v_mov_b32 %0, %1" : "=v"(vhi) : "s"(uint32_t(b.bb >> 32)));
                asm volatile("v_mov_b32 %0, %1" : "=v"(vop) : "s"(uint32_t(op)));
                int cnt = b.cnt;
                const uint32_t lds_tl = uint32_t(uintptr_t(tab_l)), lds_td = uint32_t(uintptr_t(tab_d)), lds_tok = uint32_t(uintptr_t(tok));
                enum { R_REFILL = 1, R_RARE = 2, R_EOB = 3, R_RARE_DIST = 4, R_OVERRUN = 5, R_BAD_DIST = 6, R_QUEUE = 7 };
                auto drop = [&](int nb) {
                    uint64_t v = (uint64_t(vhi) << 32) | vlo;
                    v >>= nb;
                    vlo = uint32_t(v);
                    vhi = uint32_t(v >> 32);
                    cnt -= nb;
                };
                auto refill = [&]() -> bool {                    // at least 32 valid bits afterwards; false: the literal run overran the output
                    if (cnt < 32) {
                        uint32_t d_ = 0;
                        if (b.widx <= b.limit) {
                            if (b.widx - b.wbase >= 64) {
                                b.wbase += 64;
                                b.win = b.src[b.wbase + lane];
                                __builtin_amdgcn_s_waitcnt(0x0F70);
                            }
                            d_ = rl(b.win, b.widx - b.wbase);
                        } else {
                            b.over = 1;                          // zeros from here on; the loop ends on its output bound
                        }
                        const uint64_t add = uint64_t(d_) << cnt;
                        vlo |= uint32_t(add);
                        vhi |= uint32_t(add >> 32);
                        cnt += 32;
                        ++b.widx;
                        if (uni(int(vop)) > isize) return false;
                    }
                    return true;
                };
                auto queue_match = [&](uint32_t n, uint32_t d) -> bool {
                    if (__ballot(d > vop || vop + n > uint32_t(isize))) { st = uni(int(d > vop)) ? ST_BAD_DIST : ST_OVERRUN_OUT; return false; }
                    tok[ntok] = uint64_t(vop | ((vop - d) << 16)) | (uint64_t(n) << 32);     // every lane, the same value
                    ++ntok;
                    vop += n;
                    if (ntok == TOK) resolve_matches();
                    return true;
                };
                auto slow_dist = [&](uint32_t n) -> bool {       // the distance of a match of length n, tables or canonical search
                    if (!refill()) { st = ST_OVERRUN_OUT; return false; }
                    uint32_t ed = uint32_t(uni(int(tab_d[vlo & ((1u << TBD) - 1u)])));
                    if (ed < E_VAL) {
                        if (ed != 0) { st = ST_BAD_DIST; return false; }
                        int dl0 = 0;
                        const int ds = huff_decode<1>(hd, vlo, lane, &dl0);
                        if (ds < 0) { st = ST_BAD_DIST; return false; }
                        ed = dist_entry(uint32_t(ds), uint32_t(dl0));
                        if (ed < E_VAL) { st = ST_BAD_DIST; return false; }
                    }
                    const int dl = int((ed >> 20) & 15u), dex = int((ed >> 16) & 15u);
                    const uint32_t d = (ed & 0x7fffu) + ((vlo >> dl) & ((1u << dex) - 1u));
                    drop(dl + dex);
                    return queue_match(n, d);
                };
                for (bool more = true; more && st == ST_OK;) {
                    int reason;
                    uint32_t e, vn, t0, t2;
                    int s0, s1, s2, s3;
                    uint32_t vtk = lds_tok + uint32_t(ntok) * 8u;     // LDS address of the next queue entry (a vector register inside)
                    asm volatile("v_mov_b32 %0, %0" : "+v"(vtk));
                    asm volatile(
                        "L_top%=:\n\t"
                        "s_cmp_lt_i32 %[cnt], 32\n\t"
                        "s_cbranch_scc1 L_refill%=\n"
                        "L_look%=:\n\t"
                        "v_and_b32 %[t0], %[mskl], %[lo]\n\t"
                        "v_lshl_add_u32 %[t0], %[t0], 2, %[tl]\n\t"
                        "ds_read_b32 %[t0], %[t0]\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "v_readfirstlane_b32 %[e], %[t0]\n\t"
                        "s_cmp_lt_i32 %[e], 0\n\t"
                        "s_cbranch_scc0 L_nonlit%=\n\t"
                        // ---- literal ----
                        "global_store_byte %[vop], %[t0], %[dst]\n\t"
                        "s_bfe_u32 %[s0], %[e], 0x40008\n\t"
                        "v_add_u32 %[vop], 1, %[vop]\n\t"
                        "v_alignbit_b32 %[lo], %[hi], %[lo], %[s0]\n\t"
                        "v_lshrrev_b32 %[hi], %[s0], %[hi]\n\t"
                        "s_sub_i32 %[cnt], %[cnt], %[s0]\n\t"
                        "s_branch L_top%=\n"
                        // ---- refill in front of a literal / length code ----
                        "L_refill%=:\n\t"
                        "s_cmp_gt_i32 %[widx], %[limit]\n\t"
                        "s_cbranch_scc1 L_x_refill%=\n\t"
                        "s_sub_i32 %[s0], %[widx], %[wbase]\n\t"
                        "s_cmp_lt_i32 %[s0], 64\n\t"
                        "s_cbranch_scc0 L_x_refill%=\n\t"
                        "v_readfirstlane_b32 %[s1], %[vop]\n\t"
                        "s_cmp_gt_i32 %[s1], %[isize]\n\t"
                        "s_cbranch_scc1 L_x_overrun%=\n\t"
                        "v_readlane_b32 %[s1], %[win], %[s0]\n\t"
                        "s_lshl_b32 %[s2], %[s1], %[cnt]\n\t"
                        "s_lshr_b32 %[s1], %[s1], 1\n\t"
                        "s_sub_i32 %[s3], 31, %[cnt]\n\t"
                        "s_lshr_b32 %[s1], %[s1], %[s3]\n\t"
                        "v_or_b32 %[lo], %[s2], %[lo]\n\t"
                        "v_or_b32 %[hi], %[s1], %[hi]\n\t"
                        "s_add_i32 %[cnt], %[cnt], 32\n\t"
                        "s_add_i32 %[widx], %[widx], 1\n\t"
                        "s_branch L_look%=\n"
                        // ---- not a literal ----
                        "L_nonlit%=:\n\t"
                        "s_cmp_lt_u32 %[e], 0x20000000\n\t"
                        "s_cbranch_scc1 L_x_rare%=\n\t"
                        "s_cmp_ge_u32 %[e], 0x40000000\n\t"
                        "s_cbranch_scc1 L_x_eob%=\n\t"
                        // length: base + extra bits
                        "s_bfe_u32 %[s0], %[e], 0x40010\n\t"
                        "s_bfe_u32 %[s1], %[e], 0x3000c\n\t"
                        "s_bfm_b32 %[s1], %[s1], 0\n\t"
                        "s_and_b32 %[s2], %[e], 0x1ff\n\t"
                        "v_lshrrev_b32 %[vn], %[s0], %[lo]\n\t"
                        "v_and_b32 %[vn], %[s1], %[vn]\n\t"
                        "v_add_u32 %[vn], %[s2], %[vn]\n\t"
                        "s_bfe_u32 %[s0], %[e], 0x50018\n\t"
                        "v_alignbit_b32 %[lo], %[hi], %[lo], %[s0]\n\t"
                        "v_lshrrev_b32 %[hi], %[s0], %[hi]\n\t"
                        "s_sub_i32 %[cnt], %[cnt], %[s0]\n\t"
                        "s_cmp_lt_i32 %[cnt], 32\n\t"
                        "s_cbranch_scc0 L_dist%=\n\t"
                        // refill in front of a distance code
                        "s_cmp_gt_i32 %[widx], %[limit]\n\t"
                        "s_cbranch_scc1 L_x_raredist%=\n\t"
                        "s_sub_i32 %[s0], %[widx], %[wbase]\n\t"
                        "s_cmp_lt_i32 %[s0], 64\n\t"
                        "s_cbranch_scc0 L_x_raredist%=\n\t"
                        "v_readlane_b32 %[s1], %[win], %[s0]\n\t"
                        "s_lshl_b32 %[s2], %[s1], %[cnt]\n\t"
                        "s_lshr_b32 %[s1], %[s1], 1\n\t"
                        "s_sub_i32 %[s3], 31, %[cnt]\n\t"
                        "s_lshr_b32 %[s1], %[s1], %[s3]\n\t"
                        "v_or_b32 %[lo], %[s2], %[lo]\n\t"
                        "v_or_b32 %[hi], %[s1], %[hi]\n\t"
                        "s_add_i32 %[cnt], %[cnt], 32\n\t"
                        "s_add_i32 %[widx], %[widx], 1\n"
                        "L_dist%=:\n\t"
                        "v_and_b32 %[t0], %[mskd], %[lo]\n\t"
                        "v_lshl_add_u32 %[t0], %[t0], 2, %[td]\n\t"
                        "ds_read_b32 %[t0], %[t0]\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "v_readfirstlane_b32 %[e], %[t0]\n\t"
                        "s_cmp_lt_u32 %[e], 0x20000000\n\t"
                        "s_cbranch_scc1 L_x_raredist%=\n\t"
                        "s_bfe_u32 %[s0], %[e], 0x40014\n\t"
                        "s_bfe_u32 %[s1], %[e], 0x40010\n\t"
                        "s_bfm_b32 %[s1], %[s1], 0\n\t"
                        "s_and_b32 %[s2], %[e], 0x7fff\n\t"
                        "v_lshrrev_b32 %[t0], %[s0], %[lo]\n\t"
                        "v_and_b32 %[t0], %[s1], %[t0]\n\t"
                        "v_add_u32 %[t0], %[s2], %[t0]\n\t"                      // t0 = distance
                        "s_bfe_u32 %[s0], %[e], 0x50018\n\t"
                        "v_alignbit_b32 %[lo], %[hi], %[lo], %[s0]\n\t"
                        "v_lshrrev_b32 %[hi], %[s0], %[hi]\n\t"
                        "s_sub_i32 %[cnt], %[cnt], %[s0]\n\t"
                        "v_cmp_gt_u32 vcc, %[t0], %[vop]\n\t"                  // distance beyond the output so far
                        "v_add_u32 %[t2], %[vop], %[vn]\n\t"                   // t2 = end of the match
                        "s_cbranch_vccnz L_x_baddist%=\n\t"
                        "v_cmp_lt_u32 vcc, %[isize], %[t2]\n\t"
                        "s_cbranch_vccnz L_x_overrun%=\n\t"
                        "v_sub_u32 %[t0], %[vop], %[t0]\n\t"                   // source
                        "v_lshl_or_b32 %[t0], %[t0], 16, %[vop]\n\t"           // destination | source << 16
                        "ds_write2_b32 %[vtk], %[t0], %[vn] offset1:1\n\t"
                        "v_add_u32 %[vtk], 8, %[vtk]\n\t"
                        "v_mov_b32 %[vop], %[t2]\n\t"
                        "s_add_i32 %[ntok], %[ntok], 1\n\t"
                        "s_cmp_eq_u32 %[ntok], %[tokcap]\n\t"
                        "s_cbranch_scc0 L_top%=\n\t"
                        "s_mov_b32 %[reason], 7\n\t"
                        "s_branch L_out%=\n"
                        "L_x_refill%=:\n\t"
                        "s_mov_b32 %[reason], 1\n\t"
                        "s_branch L_out%=\n"
                        "L_x_rare%=:\n\t"
                        "s_mov_b32 %[reason], 2\n\t"
                        "s_branch L_out%=\n"
                        "L_x_eob%=:\n\t"
                        "s_mov_b32 %[reason], 3\n\t"
                        "s_branch L_out%=\n"
                        "L_x_raredist%=:\n\t"
                        "s_mov_b32 %[reason], 4\n\t"
                        "s_branch L_out%=\n"
                        "L_x_overrun%=:\n\t"
                        "s_mov_b32 %[reason], 5\n\t"
                        "s_branch L_out%=\n"
                        "L_x_baddist%=:\n\t"
                        "s_mov_b32 %[reason], 6\n"
                        "L_out%=:\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        : [lo] "+v"(vlo), [hi] "+v"(vhi), [vop] "+v"(vop), [cnt] "+s"(cnt), [widx] "+s"(b.widx), [ntok] "+s"(ntok),
                          [vtk] "+v"(vtk), [reason] "=&s"(reason), [e] "=&s"(e), [vn] "=&v"(vn), [t0] "=&v"(t0), [t2] "=&v"(t2),
                          [s0] "=&s"(s0), [s1] "=&s"(s1), [s2] "=&s"(s2), [s3] "=&s"(s3)
                        : [win] "v"(b.win), [wbase] "s"(b.wbase), [limit] "s"(b.limit), [isize] "s"(isize), [dst] "s"(dst), [tl] "s"(lds_tl),
                          [td] "s"(lds_td), [mskl] "s"((1u << TBL) - 1u), [mskd] "s"((1u << TBD) - 1u), [tokcap] "s"(TOK)
                        : "vcc", "scc", "memory");
                    switch (reason) {
                    case R_QUEUE:
                        resolve_matches();
                        break;
                    case R_REFILL:                               // the next 256 input bytes, or the end of the payload
                        if (!refill()) st = ST_OVERRUN_OUT;
                        break;
                    case R_EOB:
                        drop(int((e >> 8) & 15u));
                        more = false;
                        break;
                    case R_OVERRUN:
                        st = ST_OVERRUN_OUT;
                        break;
                    case R_BAD_DIST:
                        st = ST_BAD_DIST;
                        break;
                    case R_RARE_DIST:                            // the length (vn) is consumed; the distance needs a refill beyond the
                        (void)slow_dist(vn);                     // window or a code longer than the table's index
                        break;
                    default: {                                   // R_RARE: a literal / length code longer than the table's index
                        if (e != 0) { st = ST_BAD_CODE; break; }
                        int l = 0;
                        const int sy = huff_decode<5>(hl, vlo, lane, &l);
                        if (sy < 0) { st = ST_BAD_CODE; break; }
                        const uint32_t e2 = litlen_entry(uint32_t(sy), uint32_t(l));
                        if (e2 & E_LIT) {
                            if (uni(int(vop)) >= isize) { st = ST_OVERRUN_OUT; break; }
                            dst[vop] = uint8_t(e2);
                            ++vop;
                            drop(l);
                        } else if (e2 >= E_EOB) {
                            drop(l);
                            more = false;
                        } else if (e2 >= E_VAL) {
                            const int ex = int((e2 >> 12) & 7u);
                            const uint32_t n = (e2 & 511u) + ((vlo >> l) & ((1u << ex) - 1u));
                            drop(l + ex);
                            (void)slow_dist(n);
                        } else {
                            st = ST_BAD_CODE;
                        }
                        break;
                    }
                    }
                }
                b.bb = uint64_t(uint32_t(uni(int(vlo)))) | (uint64_t(uint32_t(uni(int(vhi)))) << 32);
                b.cnt = cnt;
                op = uni(int(vop));
                if (st == ST_OK && op > isize) st = ST_OVERRUN_OUT;
                if (st == ST_OK && (b.over || bits_used(b) > in_bits + 64)) st = ST_OVERRUN_IN;
                if (st != ST_OK) break;
            }
            resolve_matches();                                   // end of this DEFLATE block: a stored block may follow
        }
        if (st == ST_OK && op != isize) st = ST_SHORT;
    }
    if (lane == 0) status[blk] = st;
}

extern "C" int cto_bgzf_inflate(const void* d_comp, const cto_bgzf_block* d_blocks, int n_blocks, void* d_out, int* d_status, void* stream) {
    using namespace cto;
    CTO_REQUIRE(n_blocks >= 0 && (n_blocks == 0 || (d_comp && d_blocks && d_out && d_status)), CTO_EINVAL, "cto_bgzf_inflate: null argument");
    if (n_blocks == 0) return CTO_OK;
    hipLaunchKernelGGL(k_bgzf_inflate, dim3(unsigned(n_blocks)), dim3(64), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint8_t*>(d_comp), d_blocks, n_blocks, static_cast<uint8_t*>(d_out), d_status);
    CTO_HIP(hipGetLastError());
    return CTO_OK;
}
