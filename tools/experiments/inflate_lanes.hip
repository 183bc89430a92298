// BGZF / DEFLATE decompression with one BGZF block per LANE (experimental: CTO_INFLATE_LANES=1 routes cto_bgzf_inflate here).
//
// inflate.hip gives a block to a wavefront and is bound by the scalar unit of a CU (one decoder per wave, ~40 scalar instructions
// per symbol).  Here every lane is a decoder of its own - plain per-thread code, 64 streams per wave instruction - with its
// tables in private memory: a 10-bit lookup table for the literal / length code and a 8-bit one for the distance code (entry =
// symbol | length << 9), longer codes through the canonical count / first-code walk.  Input bytes, window bytes and output bytes
// are per-lane byte accesses; the window is read with device-scope loads (a lane reads only its own block, but the vector L1 is not
// refreshed by the wave's stores).  Same interface and status codes as inflate.hip.
//
// State of the experiment (tools/inflate_bench.py concurrency, a chunk of 1 776 blocks = 28 waves): correct on the whole zlib parity
// suite (tests/test_gpu_inflate.py runs it), and SLOW - 113 ms for one launch alone, 29 ms per chunk with 4 launches in flight and no
// better with 8, against 23 / 5.8 ms of the wave-per-block kernel.  A symbol step of a wave costs ~7 000 cycles: the lanes of a wave
// take the literal, the match and the long-code path one after the other, and each path has its own dependent memory round trips
// (table look-ups in private memory, base / extra-bits tables in global memory, the input dword, the window bytes) with three waves
// per SIMD (146 VGPRs) to hide them.  Dword-wide input, plain loads for the window and the code-length counts in registers were
// each worth 0-10 %, and so were the lookup tables in LDS (1 156 bytes per lane, two waves per CU: 112 ms): what a step waits for is
// not the tables.  64 lanes x 1.5 bytes of input per step cross a cache line somewhere in the wave on every step, so every step
// contains a trip to memory for input that nobody prefetched; a match's window bytes are 3 dependent byte loads; and the three code
// paths run one after the other with one wave per SIMD to cover ~300-400 dependent instructions.  What it would take: a double-buffered dwordx4 input per lane, one wide load per match, tables in LDS with closed-form base / extra bits, no scratch
// (launches beyond four did not overlap - the scratch pool), and some twenty chunks in flight to fill the chip.
#include "common.h"

namespace {

enum { ST_OK = 0, ST_BAD_BTYPE = 1, ST_BAD_STORED = 2, ST_BAD_TABLE = 3, ST_BAD_CODE = 4, ST_BAD_DIST = 5, ST_OVERRUN_OUT = 6, ST_OVERRUN_IN = 7, ST_SHORT = 8 };
constexpr int LB = 10, DBITS = 8;

struct Rd {
    const uint8_t* p;
    uint32_t pos, end;            // next byte, bytes available (payload + padding the caller guarantees)
    uint64_t bb;
    int cnt;
    bool over;
};
__device__ __forceinline__ void fill(Rd& r) {
    if (r.cnt <= 32) {                      // four bytes at a time (an unaligned dword load; the input carries padding behind the payload)
        uint32_t v = 0;
        if (r.pos < r.end) {
            __builtin_memcpy(&v, r.p + r.pos, 4);
        } else {
            r.over = true;
        }
        r.bb |= uint64_t(v) << r.cnt;
        r.cnt += 32;
        r.pos += 4;
    }
}
__device__ __forceinline__ uint32_t take(Rd& r, int n) {
    const uint32_t v = uint32_t(r.bb) & ((1u << n) - 1u);
    r.bb >>= n;
    r.cnt -= n;
    return v;
}

// canonical code from lens[0..n): count[len], sorted symbols; false when over-subscribed
__device__ bool build(const uint8_t* lens, int n, uint16_t* count, uint16_t* sym) {
    for (int i = 0; i < 16; ++i) count[i] = 0;
    for (int i = 0; i < n; ++i) ++count[lens[i]];
    count[0] = 0;
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - int(count[l]);
        if (left < 0) return false;
    }
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = uint16_t(offs[l] + count[l]);
    for (int i = 0; i < n; ++i)
        if (lens[i]) sym[offs[lens[i]]++] = uint16_t(i);
    return true;
}
// lookup table of the codes of up to TB bits (0 = longer code or no code)
template <int TB>
__device__ void table(const uint16_t* count, const uint16_t* sym, uint16_t* tab) {
    for (int i = 0; i < (1 << TB); ++i) tab[i] = 0;
    uint32_t code = 0;
    int idx = 0;
    for (int l = 1; l <= TB; ++l) {
        for (int k = 0; k < int(count[l]); ++k, ++idx, ++code) {
            const uint32_t rev = __brev(code) >> (32 - l);
            const uint16_t e = uint16_t(sym[idx] | (l << 9));
            for (uint32_t j = rev; j < (1u << TB); j += (1u << l)) tab[j] = e;
        }
        code <<= 1;
    }
}
// the counts of a code in registers (constant indices after unrolling): the walk below then costs arithmetic, not a memory round trip
// per code length
struct Counts { uint16_t c[16]; };
__device__ __forceinline__ Counts in_registers(const uint16_t* count) {
    Counts k;
#pragma unroll
    for (int l = 0; l < 16; ++l) k.c[l] = count[l];
    return k;
}
__device__ __forceinline__ int walk_reg(Rd& r, const Counts& k, const uint16_t* sym) {
    int code = 0, first = 0, index = 0, found = -1, taken = 0;
#pragma unroll
    for (int l = 1; l <= 15; ++l) {
        if (found < 0) {
            code |= int((r.bb >> (l - 1)) & 1);
            const int c = k.c[l];
            if (code - c < first) { found = index + (code - first); taken = l; }
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
    }
    if (found < 0) return -1;
    r.bb >>= taken;
    r.cnt -= taken;
    return sym[found];
}
// slow path: walk the code lengths (bits arrive LSB first, codes are MSB first)
__device__ int walk(Rd& r, const uint16_t* count, const uint16_t* sym) {
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= 15; ++l) {
        code |= int(r.bb & 1);
        r.bb >>= 1;
        --r.cnt;
        const int c = count[l];
        if (code - c < first) return sym[index + (code - first)];
        index += c;
        first += c;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

__device__ const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ const uint8_t kOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

}  // namespace

extern "C" __global__ __launch_bounds__(64) void k_bgzf_inflate_lanes(const uint8_t* __restrict__ comp, const cto_bgzf_block* __restrict__ blocks,
                                                                      int n_blocks, uint8_t* __restrict__ out, int* __restrict__ status) {
    const int blk = blockIdx.x * 64 + threadIdx.x;
    if (blk >= n_blocks) return;
    const cto_bgzf_block bd = blocks[blk];
    const int isize = int(bd.isize);
    uint8_t* dst = out + bd.out_off;
    int st = ST_OK, op = 0;
    uint8_t lens[320];
    uint16_t lcount[16], dcount[16], lsym[288], dsym[32];
    uint16_t ltab[1 << LB], dtab[1 << DBITS];
    if (isize > 0) {
        Rd r{comp + bd.in_off, 0, bd.csize + 16u, 0, 0, false};
        bool last = false;
        while (!last && st == ST_OK) {
            fill(r);
            last = take(r, 1) != 0;
            const int type = int(take(r, 2));
            if (type == 0) {
                take(r, r.cnt & 7);
                fill(r);
                const uint32_t len = take(r, 16), nlen = take(r, 16);
                if ((len ^ nlen) != 0xffffu) { st = ST_BAD_STORED; break; }
                if (op + int(len) > isize) { st = ST_OVERRUN_OUT; break; }
                for (uint32_t i = 0; i < len && !r.over; ++i) {
                    fill(r);
                    dst[op++] = uint8_t(take(r, 8));
                }
                if (r.over) { st = ST_OVERRUN_IN; break; }
                continue;
            }
            if (type == 3) { st = ST_BAD_BTYPE; break; }
            int nlit = 288, ndist = 30;
            if (type == 1) {
                for (int i = 0; i < 288; ++i) lens[i] = uint8_t(i < 144 ? 8 : (i < 256 ? 9 : (i < 280 ? 7 : 8)));
                for (int i = 0; i < 30; ++i) lens[288 + i] = 5;
            } else {
                fill(r);
                nlit = int(take(r, 5)) + 257;
                ndist = int(take(r, 5)) + 1;
                const int ncl = int(take(r, 4)) + 4;
                if (nlit > 286 || ndist > 30) { st = ST_BAD_TABLE; break; }
                uint8_t cl[19];
                for (int i = 0; i < 19; ++i) cl[i] = 0;
                for (int i = 0; i < ncl; ++i) {
                    fill(r);
                    cl[kOrder[i]] = uint8_t(take(r, 3));
                }
                uint16_t ccount[16], csym[19];
                if (!build(cl, 19, ccount, csym)) { st = ST_BAD_TABLE; break; }
                int i = 0;
                const int total = nlit + ndist;
                while (i < total) {
                    fill(r);
                    const int s = walk(r, ccount, csym);
                    if (s < 0) { st = ST_BAD_TABLE; break; }
                    if (s < 16) { lens[i++] = uint8_t(s); continue; }
                    int rep, val = 0;
                    if (s == 16) { if (i == 0) { st = ST_BAD_TABLE; break; } val = lens[i - 1]; rep = 3 + int(take(r, 2)); }
                    else if (s == 17) rep = 3 + int(take(r, 3));
                    else rep = 11 + int(take(r, 7));
                    if (i + rep > total) { st = ST_BAD_TABLE; break; }
                    while (rep--) lens[i++] = uint8_t(val);
                    if (r.over) { st = ST_OVERRUN_IN; break; }
                }
                if (st != ST_OK) break;
                // distance lengths behind the literal / length ones -> lens[288..)
                uint8_t dl[32];
                for (int k = 0; k < 32; ++k) dl[k] = k < ndist ? lens[nlit + k] : uint8_t(0);
                for (int k = nlit; k < 288; ++k) lens[k] = 0;
                for (int k = 0; k < 32; ++k) lens[288 + k] = dl[k];
            }
            if (!build(lens, 288, lcount, lsym)) { st = ST_BAD_TABLE; break; }
            build(lens + 288, 30, dcount, dsym);             // an incomplete distance code is legal
            table<LB>(lcount, lsym, ltab);
            table<DBITS>(dcount, dsym, dtab);
            const Counts lk = in_registers(lcount), dk = in_registers(dcount);
            for (;;) {
                fill(r);
                int s;
                const uint16_t e = ltab[uint32_t(r.bb) & ((1u << LB) - 1u)];
                if (e) { s = e & 511; r.bb >>= (e >> 9); r.cnt -= (e >> 9); } else s = walk_reg(r, lk, lsym);
                if (s < 0) { st = ST_BAD_CODE; break; }
                if (r.over) { st = ST_OVERRUN_IN; break; }
                if (s < 256) {
                    if (op >= isize) { st = ST_OVERRUN_OUT; break; }
                    dst[op++] = uint8_t(s);
                    continue;
                }
                if (s == 256) break;
                const int ls = s - 257;
                if (ls >= 29) { st = ST_BAD_CODE; break; }
                const int n = int(kLenBase[ls]) + int(take(r, kLenExtra[ls]));
                fill(r);
                int ds;
                const uint16_t f = dtab[uint32_t(r.bb) & ((1u << DBITS) - 1u)];
                if (f) { ds = f & 511; r.bb >>= (f >> 9); r.cnt -= (f >> 9); } else ds = walk_reg(r, dk, dsym);
                if (ds < 0 || ds >= 30) { st = ST_BAD_DIST; break; }
                const int d = int(kDistBase[ds]) + int(take(r, kDistExtra[ds]));
                if (d > op) { st = ST_BAD_DIST; break; }
                if (op + n > isize) { st = ST_OVERRUN_OUT; break; }
                const uint8_t* src = dst + op - d;
                for (int k = 0; k < n; ++k) dst[op + k] = src[k];            // a lane reads only bytes it wrote itself
                op += n;
            }
        }
        if (st == ST_OK && op != isize) st = ST_SHORT;
    }
    status[blk] = st;
}

int launch_bgzf_inflate_lanes(const void* d_comp, const cto_bgzf_block* d_blocks, int n_blocks, void* d_out, int* d_status, hipStream_t stream) {
    hipLaunchKernelGGL(k_bgzf_inflate_lanes, dim3(unsigned((n_blocks + 63) / 64)), dim3(64), 0, stream, static_cast<const uint8_t*>(d_comp), d_blocks,
                       n_blocks, static_cast<uint8_t*>(d_out), d_status);
    CTO_HIP(hipGetLastError());
    return CTO_OK;
}
