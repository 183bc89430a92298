// BGZF / DEFLATE decompression with one BGZF block per LANE, in two phases (round 6; CTO_INFLATE_LANES=1 routes cto_bgzf_inflate here).
//
// inflate.hip gives a block to a wavefront: one decoder per wave, wave-uniform code, bound by the CU's instruction issue (~31
// instructions per symbol, one symbol at a time).  The first block-per-lane experiment (round 2, tools/experiments/inflate_lanes.hip)
// was 5 x SLOWER than that: every step of a wave waited for input nobody had prefetched, for tables in private memory and - above all -
// for the LZ77 window: a match's source bytes are a trip to memory on the decoder's critical path.  This form takes the window out of
// the decoder altogether:
//   phase 1  k_lanes_decode   a lane per block: Huffman decoding only.  Literals are stored at their final place (the output position
//                             of every symbol is known without the window: lengths add up); a match becomes a RECORD (destination,
//                             distance, length) in the block's list.  Tables in LDS, lane-private (2.3 KB per lane: a 9-bit literal /
//                             length table, a 7-bit distance table, canonical count / symbol lists for longer codes), input through a
//                             double-buffered 16-byte register window per lane.
//   phase 2  k_lanes_matches  a wavefront per block: the records in order, 64 at a time, one per lane, in rounds that respect their
//                             dependences (a record is ready when its source lies below the oldest unresolved destination) - the
//                             wave-per-block kernel's resolve_matches on the whole list.
// Same interface, status codes and output as inflate.hip.
//
// MEASURED (round 6, one MI355X; tools/experiments/inflate_lanes_bench.py on a 1 Mb x 50x chunk: 53.8 MB in 1 238 blocks -> 80.8 MB; built into
// the library behind CTO_INFLATE_LANES=1 for the measurement, then taken out again): CORRECT on the first run - tests/test_gpu_inflate.py's twelve
// cases incl. the malformed streams, and the same SHA-256 of the chunk's output - and SLOW: 68 ms for one launch alone (20 wavefronts), 16.7 ms
// per chunk with 4 or more launches in flight, against 7.1 / 2.2 ms of the wavefront-per-block kernel.  Not memory this time: no scratch, no
// window reads, input prefetched - a step of a wavefront is ~4 200 cycles because its 64 lanes sit at 64 different places of the DEFLATE
// grammar and the wavefront executes the UNION of their paths every step: a literal (15 instructions), a match (60), the canonical search for
// a code longer than the 9-bit table (the compiler unrolls its 15 lengths: ~210, and with 5 % of the symbols needing it some lane of 64 nearly
// always does), the same for the distance code, the refill - ~500 instructions per step where the wave-uniform decoder spends 31 per symbol.
// Larger tables make the long-code path rare but do not fit: 2.3 KB per lane is already 147 KB per wavefront (one wavefront per CU); 11-bit
// tables would be 5.9 KB per lane.  To beat the wave-per-block kernel by 2 x a step would have to cost ~850 cycles.  Kept as a record.
#include <stdlib.h>
#include "common.h"

namespace {

enum { ST_OK = 0, ST_BAD_BTYPE = 1, ST_BAD_STORED = 2, ST_BAD_TABLE = 3, ST_BAD_CODE = 4, ST_BAD_DIST = 5, ST_OVERRUN_OUT = 6, ST_OVERRUN_IN = 7, ST_SHORT = 8, ST_BAD_SLOT = 9 };

constexpr int TL = 9, TD = 7;
// lane-private LDS region (bytes): literal / length table | distance table | code lengths | sorted literal / length symbols | sorted distance
// symbols | counts (literal / length, distance)
constexpr int O_TABL = 0, O_TABD = O_TABL + (2 << TL), O_LENS = O_TABD + (2 << TD), O_SYML = O_LENS + 320, O_SYMD = O_SYML + 576, O_CNTL = O_SYMD + 64,
              O_CNTD = O_CNTL + 32, REGION = O_CNTD + 32;
constexpr int STRIDE = REGION + 4;                  // an odd number of dwords: the lanes' same-index accesses fall on different banks
static_assert(REGION % 4 == 0 && ((STRIDE / 4) & 1) == 1, "lane stride must be an odd number of dwords");

typedef __attribute__((address_space(3))) unsigned char* lds_u8;
typedef __attribute__((address_space(3))) unsigned short* lds_u16;

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
__constant__ unsigned char kClOrder2[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// per-lane bit reader: 64-bit buffer, refilled a dword at a time out of a 16-byte register window with the next 16 bytes already requested
struct Rd {
    const uint4* src;            // 16-byte aligned base
    uint4 cur, nxt;
    int q;                       // index of the 16-byte piece in `nxt`
    int di;                      // next dword of `cur` (0..3)
    int limit;                   // last piece that may be loaded
    unsigned long long bb;
    int cnt;
    long long used;              // bits handed to the buffer so far
    bool over;
};
__device__ __forceinline__ uint4 ld16(const uint4* p) { return *p; }
__device__ __forceinline__ void rd_init(Rd& r, const unsigned char* p, long long in_bytes) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    r.src = reinterpret_cast<const uint4*>(a & ~uintptr_t(15));
    const int skip = int(a & 15);
    r.limit = int((skip + in_bytes + 15) / 16) + 1;     // (the input carries CTO_BGZF_PAD bytes of padding behind the payload)
    r.cur = ld16(r.src);
    r.nxt = ld16(r.src + 1);
    r.q = 1;
    r.di = 0;
    r.bb = 0; r.cnt = 0; r.used = 0; r.over = false;
    // drop the bytes in front of the stream: whole dwords first, then bits
    r.di = skip >> 2;
    const int sb = (skip & 3) * 8;
    // first fill
    for (int k = 0; k < 2; ++k) {
        const unsigned d = r.di == 0 ? r.cur.x : (r.di == 1 ? r.cur.y : (r.di == 2 ? r.cur.z : r.cur.w));
        r.bb |= (unsigned long long)d << r.cnt;
        r.cnt += 32;
        if (++r.di == 4) { r.cur = r.nxt; ++r.q; r.nxt = r.q <= r.limit ? ld16(r.src + r.q) : make_uint4(0u, 0u, 0u, 0u); r.di = 0; }
    }
    r.bb >>= sb;
    r.cnt -= sb;
    r.used = 64 - sb;
}
__device__ __forceinline__ void rd_fill(Rd& r) {
    if (r.cnt <= 32) {
        const unsigned d = r.di == 0 ? r.cur.x : (r.di == 1 ? r.cur.y : (r.di == 2 ? r.cur.z : r.cur.w));
        r.bb |= (unsigned long long)d << r.cnt;
        r.cnt += 32;
        r.used += 32;
        if (++r.di == 4) {
            r.cur = r.nxt;
            ++r.q;
            if (r.q <= r.limit) r.nxt = ld16(r.src + r.q);
            else { r.nxt = make_uint4(0u, 0u, 0u, 0u); r.over = r.q > r.limit + 1; }
            r.di = 0;
        }
    }
}
__device__ __forceinline__ unsigned rd_peek(const Rd& r, int n) { return unsigned(r.bb) & ((1u << n) - 1u); }
__device__ __forceinline__ void rd_drop(Rd& r, int n) { r.bb >>= n; r.cnt -= n; }
__device__ __forceinline__ long long rd_consumed(const Rd& r) { return r.used - r.cnt; }

// canonical decoding by code length (codes longer than a table's index; the code-length alphabet): counts[1..15], symbols sorted by code
__device__ int slow_decode(Rd& r, lds_u16 count, lds_u16 sym, int* len_out) {
    unsigned code = 0, first = 0, index = 0;
    const unsigned bits = unsigned(r.bb);
    for (int l = 1; l <= 15; ++l) {
        code |= (bits >> (l - 1)) & 1u;
        const unsigned c = count[l];
        if (code - first < c) { *len_out = l; return int(sym[index + (code - first)]); }
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

// lens[0..n) -> count[0..15] (count[0] = 0), symbols sorted by (length, value); false when over-subscribed
__device__ bool build_canon(lds_u8 lens, int n, lds_u16 count, lds_u16 sym) {
    for (int i = 0; i < 16; ++i) count[i] = 0;
    for (int i = 0; i < n; ++i) count[lens[i]] = static_cast<unsigned short>(count[lens[i]] + 1);
    count[0] = 0;
    int left = 1;
    unsigned short offs[16];
    offs[0] = 0; offs[1] = 0;
    bool ok = true;
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - int(count[l]);
        ok = ok && left >= 0;
        if (l < 15) offs[l + 1] = static_cast<unsigned short>(offs[l] + count[l]);
    }
    for (int i = 0; i < n; ++i) {
        const int l = lens[i];
        if (l) sym[offs[l]++] = static_cast<unsigned short>(i);
    }
    return ok;
}
// lookup table of the codes of up to TB bits: entry[next TB bits] = symbol | length << 9 (0: a longer code, or none)
template <int TB>
__device__ void build_table(lds_u16 count, lds_u16 sym, lds_u16 tab) {
    for (int i = 0; i < (1 << TB); ++i) tab[i] = 0;
    unsigned code = 0;
    int idx = 0;
    for (int l = 1; l <= TB; ++l) {
        const int c = count[l];
        for (int k = 0; k < c; ++k, ++idx, ++code) {
            const unsigned rev = __brev(code) >> (32 - l);
            const unsigned short e = static_cast<unsigned short>(sym[idx] | (l << 9));
            for (unsigned t = rev; t < (1u << TB); t += (1u << l)) tab[t] = e;
        }
        code <<= 1;
    }
}

struct MatchRec { unsigned dst_src; unsigned len; };     // destination | source << 16 ; length

__global__ __launch_bounds__(64) void k_lanes_decode(const unsigned char* __restrict__ comp, const cto_bgzf_block* __restrict__ blocks, int n_blocks,
                                                     unsigned char* __restrict__ out, int* __restrict__ status, MatchRec* __restrict__ mlist,
                                                     const long long* __restrict__ moff, int* __restrict__ mcount) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = threadIdx.x, blk = blockIdx.x * 64 + lane;
    if (blk >= n_blocks) return;
    const lds_u8 base = (lds_u8)lds_raw + lane * STRIDE;
    const lds_u16 tabl = (lds_u16)(base + O_TABL), tabd = (lds_u16)(base + O_TABD), syml = (lds_u16)(base + O_SYML), symd = (lds_u16)(base + O_SYMD),
                  cntl = (lds_u16)(base + O_CNTL), cntd = (lds_u16)(base + O_CNTD);
    const lds_u8 lens = base + O_LENS;
    const cto_bgzf_block bd = blocks[blk];
    const int isize = int(bd.isize);
    int st = ST_OK, op = 0, nm = 0;
    if (blk + 1 < n_blocks && blocks[blk + 1].out_off < bd.out_off + (unsigned long long)isize + CTO_BGZF_SLOT_PAD && blocks[blk + 1].out_off >= bd.out_off) {
        status[blk] = ST_BAD_SLOT;
        mcount[blk] = 0;
        return;
    }
    unsigned char* dst = out + bd.out_off;
    MatchRec* ml = mlist + moff[blk];
    const int mcap = int(moff[blk + 1] - moff[blk]);
    const long long in_bits = (long long)bd.csize * 8;
    if (isize > 0) {
        Rd r;
        rd_init(r, comp + bd.in_off, (long long)bd.csize);
        bool final_block = false;
        while (!final_block && st == ST_OK) {
            rd_fill(r);
            final_block = rd_peek(r, 1) != 0;
            const int btype = int(rd_peek(r, 3) >> 1);
            rd_drop(r, 3);
            if (btype == 0) {                                    // stored
                rd_drop(r, int((8 - (rd_consumed(r) & 7)) & 7));
                rd_fill(r);
                const unsigned len = rd_peek(r, 16);
                rd_drop(r, 16);
                rd_fill(r);
                const unsigned nlen = rd_peek(r, 16);
                rd_drop(r, 16);
                if ((len ^ nlen) != 0xffffu) { st = ST_BAD_STORED; break; }
                if (op + int(len) > isize) { st = ST_OVERRUN_OUT; break; }
                if (rd_consumed(r) + (long long)len * 8 > in_bits + 64) { st = ST_OVERRUN_IN; break; }
                for (unsigned i = 0; i < len; ++i) {
                    rd_fill(r);
                    dst[op++] = static_cast<unsigned char>(rd_peek(r, 8));
                    rd_drop(r, 8);
                }
                if (r.over || rd_consumed(r) > in_bits + 64) { st = ST_OVERRUN_IN; break; }
                continue;
            }
            if (btype == 3) { st = ST_BAD_BTYPE; break; }
            if (btype == 1) {
                for (int i = 0; i < 288; ++i) lens[i] = static_cast<unsigned char>(i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8)));
                for (int i = 0; i < 32; ++i) lens[288 + i] = static_cast<unsigned char>(i < 30 ? 5 : 0);
            } else {
                rd_fill(r);
                const int nlit = int(rd_peek(r, 5)) + 257;
                rd_drop(r, 5);
                const int ndist = int(rd_peek(r, 5)) + 1;
                rd_drop(r, 5);
                const int ncl = int(rd_peek(r, 4)) + 4;
                rd_drop(r, 4);
                if (nlit > 286 || ndist > 30) { st = ST_BAD_TABLE; break; }
                // the code-length code: lengths in lens[300..319), its canonical lists in the (not yet built) distance-symbol / count areas
                const lds_u8 cl = lens + 300;
                for (int i = 0; i < 19; ++i) cl[i] = 0;
                for (int i = 0; i < ncl; ++i) {
                    rd_fill(r);
                    cl[kClOrder2[i]] = static_cast<unsigned char>(rd_peek(r, 3));
                    rd_drop(r, 3);
                }
                if (!build_canon(cl, 19, cntd, symd)) { st = ST_BAD_TABLE; break; }
                int i = 0, prev = 0;
                const int total = nlit + ndist;
                // the run-length coded lengths of both alphabets go to the literal / length table's area first (the tables are built after)
                const lds_u8 tmp = base + O_TABL;
                while (i < total && st == ST_OK) {
                    rd_fill(r);
                    int l = 0;
                    const int s = slow_decode(r, cntd, symd, &l);
                    if (s < 0) { st = ST_BAD_TABLE; break; }
                    rd_drop(r, l);
                    int rep = 1, val = s;
                    if (s == 16) { if (i == 0) { st = ST_BAD_TABLE; break; } rep = 3 + int(rd_peek(r, 2)); rd_drop(r, 2); val = prev; }
                    else if (s == 17) { rep = 3 + int(rd_peek(r, 3)); rd_drop(r, 3); val = 0; }
                    else if (s == 18) { rep = 11 + int(rd_peek(r, 7)); rd_drop(r, 7); val = 0; }
                    if (i + rep > total) { st = ST_BAD_TABLE; break; }
                    for (int k = 0; k < rep; ++k) tmp[i + k] = static_cast<unsigned char>(val);
                    i += rep;
                    prev = val;
                    if (r.over || rd_consumed(r) > in_bits + 64) st = ST_OVERRUN_IN;
                }
                if (st != ST_OK) break;
                for (int k = 0; k < 288; ++k) lens[k] = k < nlit ? tmp[k] : static_cast<unsigned char>(0);
                for (int k = 0; k < 32; ++k) lens[288 + k] = k < ndist ? tmp[nlit + k] : static_cast<unsigned char>(0);
            }
            if (!build_canon(lens, 288, cntl, syml)) { st = ST_BAD_TABLE; break; }
            (void)build_canon(lens + 288, 30, cntd, symd);       // an incomplete distance code is legal (one code, or none)
            build_table<TL>(cntl, syml, tabl);
            build_table<TD>(cntd, symd, tabd);
            // ---- symbols ----
            for (;;) {
                rd_fill(r);
                unsigned e = tabl[unsigned(r.bb) & ((1u << TL) - 1u)];
                int sym, l;
                if (e != 0) { sym = int(e & 511u); l = int(e >> 9); }
                else { sym = slow_decode(r, cntl, syml, &l); if (sym < 0) { st = ST_BAD_CODE; break; } }
                rd_drop(r, l);
                if (sym < 256) {
                    if (op >= isize) { st = ST_OVERRUN_OUT; break; }
                    dst[op++] = static_cast<unsigned char>(sym);
                    continue;
                }
                if (sym == 256) break;
                const int ls = sym - 257;
                if (ls >= 29) { st = ST_BAD_CODE; break; }
                int lb, le;
                len_code(ls, &lb, &le);
                const int n = lb + int(rd_peek(r, le));
                rd_drop(r, le);
                rd_fill(r);
                e = tabd[unsigned(r.bb) & ((1u << TD) - 1u)];
                int ds;
                if (e != 0) { ds = int(e & 511u); l = int(e >> 9); }
                else { ds = slow_decode(r, cntd, symd, &l); if (ds < 0) { st = ST_BAD_DIST; break; } }
                rd_drop(r, l);
                if (ds >= 30) { st = ST_BAD_DIST; break; }
                int db, de;
                dist_code(ds, &db, &de);
                const int d = db + int(rd_peek(r, de));
                rd_drop(r, de);
                if (d > op) { st = ST_BAD_DIST; break; }
                if (op + n > isize) { st = ST_OVERRUN_OUT; break; }
                if (nm >= mcap) { st = ST_OVERRUN_OUT; break; }
                ml[nm++] = MatchRec{unsigned(op) | (unsigned(op - d) << 16), unsigned(n)};
                op += n;
                if (r.over || rd_consumed(r) > in_bits + 64) { st = ST_OVERRUN_IN; break; }
            }
            if (st == ST_OK && (r.over || rd_consumed(r) > in_bits + 64)) st = ST_OVERRUN_IN;
        }
        if (st == ST_OK && op != isize) st = ST_SHORT;
    }
    status[blk] = st;
    mcount[blk] = st == ST_OK ? nm : 0;
}

// the records of a block in order, 64 at a time, one per lane (inflate.hip's resolve_matches over the whole list)
__global__ __launch_bounds__(64) void k_lanes_matches(const cto_bgzf_block* __restrict__ blocks, int n_blocks, unsigned char* __restrict__ out,
                                                      const MatchRec* __restrict__ mlist, const long long* __restrict__ moff, const int* __restrict__ mcount) {
    const int blk = blockIdx.x, lane = threadIdx.x;
    if (blk >= n_blocks) return;
    const int ntok = mcount[blk];
    if (ntok == 0) return;
    unsigned char* dst = out + blocks[blk].out_off;
    const unsigned char* win = dst;
    const MatchRec* ml = mlist + moff[blk];
    int c = 0;
    unsigned long long done = 0;
    while (c < ntok) {
        const int t = c + lane;
        const bool mine = t < ntok && !((done >> lane) & 1);
        MatchRec tk{0u, 0u};
        if (mine) tk = ml[t];
        const int td = int(tk.dst_src & 0xffffu), ts = int(tk.dst_src >> 16), tn = int(tk.len);
        const int d = td - ts;
        const int first = int(__builtin_amdgcn_readfirstlane(int(ml[c].dst_src & 0xffffu)));     // everything below the oldest unresolved destination is final
        const bool ready = mine && ts + (d < tn ? d : tn) <= first;
        if (ready) {
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
        const int adv = done == ~0ull ? 64 : __ffsll((long long)~done) - 1;
        c += adv;
        done = adv >= 64 ? 0 : done >> adv;
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ void k_lanes_moff(const cto_bgzf_block* __restrict__ blocks, int n_blocks, long long* __restrict__ moff) {
    // one thread: offsets of the blocks' record lists (a record per three output bytes at most)
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        long long o = 0;
        for (int b = 0; b < n_blocks; ++b) { moff[b] = o; o += blocks[b].isize / 3 + 1; }
        moff[n_blocks] = o;
    }
}

}  // namespace

namespace cto {
// experimental entry (CTO_INFLATE_LANES=1): same contract as cto_bgzf_inflate; its scratch (record lists) lives for the call on the stream
int bgzf_inflate_lanes(const void* d_comp, const cto_bgzf_block* d_blocks, int n_blocks, void* d_out, int* d_status, void* stream, size_t out_bytes_hint) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lanes_decode), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * STRIDE);
    CTO_HIP(attr);
    long long* moff = nullptr;
    int* mcount = nullptr;
    MatchRec* ml = nullptr;
    const size_t recs = out_bytes_hint / 3 + size_t(n_blocks) * 2 + 64;
    CTO_HIP(hipMallocAsync(reinterpret_cast<void**>(&moff), size_t(n_blocks + 1) * 8, s));
    CTO_HIP(hipMallocAsync(reinterpret_cast<void**>(&mcount), size_t(n_blocks) * 4, s));
    CTO_HIP(hipMallocAsync(reinterpret_cast<void**>(&ml), recs * sizeof(MatchRec), s));
    hipLaunchKernelGGL(k_lanes_moff, dim3(1), dim3(64), 0, s, d_blocks, n_blocks, moff);
    hipLaunchKernelGGL(k_lanes_decode, dim3(unsigned((n_blocks + 63) / 64)), dim3(64), size_t(64) * STRIDE, s, static_cast<const unsigned char*>(d_comp), d_blocks,
                       n_blocks, static_cast<unsigned char*>(d_out), d_status, ml, moff, mcount);
    hipLaunchKernelGGL(k_lanes_matches, dim3(unsigned(n_blocks)), dim3(64), 0, s, d_blocks, n_blocks, static_cast<unsigned char*>(d_out), ml, moff, mcount);
    CTO_HIP(hipGetLastError());
    CTO_HIP(hipFreeAsync(ml, s)); CTO_HIP(hipFreeAsync(mcount, s)); CTO_HIP(hipFreeAsync(moff, s));
    return CTO_OK;
}
}  // namespace cto
