// Candidate extraction on the column pack (reference: src/extract_candidates_calling.py).
//
// The reference scans `samtools mpileup --min-MQ 20 --min-BQ q` text of a 5 Mb chunk and keeps a position when
// its depth and some non-reference allele pass the AF / read-count gates (decode_pileup_bases :55-169, the
// candidate sets at :352-372).  Here the same gates run on the pack that tensor creation consumes, so a BAM is
// piled up ONCE for extraction, the AFF tensor and the NEG tensor.  HBM-bound integer work: one lane per column, the wave's
// 64 columns staged through LDS with coalesced loads, counters in registers (below).
#include <mutex>
#include "common.h"
#include "pack_internal.h"

namespace {

// One LANE per column, one wave per 64 consecutive columns (round 4; the round-1 kernel - one wave per 8 columns, the ~50 read-bases of a
// column on 50 lanes, counters in LDS - spent its time in LDS atomics on the one or two counters most of a column's read-bases hit:
// 1.24 TB/s on the 225 MB pack of a 1 Mb region; this one 1.8 TB/s, bound by the dependent-instruction latency of seven to ten waves
// per CU - 16 KB of LDS each - not by bandwidth).  The 64 columns' read-bases are one contiguous run of `entries`: the wave copies it into LDS with coalesced 256-byte loads
// (the only HBM traffic), then every lane walks its own column out of LDS and counts in its own registers - four 16-bit counters to
// a 64-bit word, no atomics, no cross-lane traffic.  Indel alleles (about one read-base in a hundred, indel mode only) are counted in a
// lane-private LDS row, with the global-atomic overflow path of before for a column with more than XG distinct alleles.
constexpr int XLANES = 64, XCAP = 4096, XG = 16;     // columns per wave, read-bases staged per wave (16 KB), allele groups per lane in LDS

struct XPack {
    int64_t n_cols;
    const uint8_t* col_ref;
    const int64_t* col_off;
    const int32_t* key_off;
    const uint32_t* entries;
    const uint8_t* key_meta;
    const int32_t* key_group;
};

template <bool select_indel>
__global__ __launch_bounds__(XLANES) void k_extract_candidates(
    XPack pk, int min_mq, int min_bq, double snv_min_af, double indel_min_af, double min_coverage, int alt_base_num,
    uint32_t* __restrict__ gscratch, uint8_t* __restrict__ flags, int32_t* __restrict__ depth_out) {
    __shared__ uint32_t s_ent[XCAP];
    __shared__ uint32_t s_grp[select_indel ? XLANES : 1][XG + 1];        // + 1: rows on different banks
    const int lane = threadIdx.x;
    const int64_t c0 = int64_t(blockIdx.x) * XLANES;
    const int64_t c = c0 + lane;
    const bool live = c < pk.n_cols;
    const int64_t my_off = live ? pk.col_off[c] : 0, my_end = live ? pk.col_off[c + 1] : 0;
    const int my_n = int(my_end - my_off);
    const int ncol = int(pk.n_cols - c0 < XLANES ? pk.n_cols - c0 : XLANES);
    const int64_t run0 = __shfl(my_off, 0), run1 = __shfl(my_end, ncol - 1);
    const int64_t run = run1 - run0;
    const bool staged = run <= XCAP;
    if (staged) {                                    // sixteen loads in flight per lane: a load per trip would cost a memory latency per 256 bytes
        const uint32_t* src = pk.entries + run0;
        for (int i0 = lane; i0 < int(run); i0 += XLANES * 16) {
            uint32_t v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int i = i0 + u * XLANES; v[u] = i < int(run) ? src[i] : 0u; }
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int i = i0 + u * XLANES; if (i < int(run)) s_ent[i] = v[u]; }
        }
    }
    int k0 = 0, nk = 0;
    if (select_indel && live) { k0 = pk.key_off[c]; nk = pk.key_off[c + 1] - k0; }
    const bool grp_lds = nk <= XG;
    if (select_indel) {
        if (grp_lds) { for (int g = 0; g < nk; ++g) s_grp[select_indel ? lane : 0][g] = 0u; }
        else { for (int g = 0; g < nk; ++g) gscratch[k0 + g] = 0u; }           // this column's own slice of the global counters
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    unsigned long long all4 = 0, pure4 = 0;          // 16-bit counters: A C G T (a column holds at most 32 767 read-bases)
    uint32_t stars = 0, n_row = 0, n_ind = 0;        // reads that pass --min-MQ (the row is printed at all); read-bases carrying an indel
    const int base = int(my_off - run0);
    int max_n = my_n;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) max_n = max(max_n, __shfl_xor(max_n, d));
    auto count = [&](uint32_t ent) {
        const uint32_t b = ent & 15u, kind = (ent >> 4) & 3u;
        const int bq = int((ent >> 6) & 127u), mq = int((ent >> 13) & 255u);
        n_row += mq >= min_mq ? 1u : 0u;
        if (mq >= min_mq && bq >= min_bq) {           // what samtools --min-MQ / --min-BQ leaves in the column
            const unsigned long long one = 1ull << (16u * (b & 3u));
            if (b < 8u) {                              // depth = bases + placeholders, summed at the end
                all4 += one;                           // pileup_dict[base] counts indel carriers too (:111-113)
                if (kind == 0u) pure4 += one;          // alt_dict single-base keys (:102)
            } else if (b == 8u || b == 9u) {
                ++stars;
            }
            if (select_indel && kind != 0u) {          // no length gate here, unlike tensor creation
                ++n_ind;
                const int g = pk.key_group[k0 + int(ent >> 21)];
                if (grp_lds) s_grp[select_indel ? lane : 0][g] += 1u;
                else atomicAdd(&gscratch[k0 + g], 1u);
            }
        }
    };
    constexpr int XU = 4;                             // read-bases per trip: XU LDS reads in flight
    for (int i = 0; i < max_n; i += XU) {
        uint32_t e[XU];
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            e[u] = 0u;
            if (i + u < my_n) e[u] = staged ? s_ent[base + i + u] : pk.entries[my_off + i + u];
        }
#pragma unroll
        for (int u = 0; u < XU; ++u)
            if (i + u < my_n) count(e[u]);
    }
    if (live) {
        uint32_t cn_all[4], cn_pure[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { cn_all[k] = uint32_t(all4 >> (16 * k)) & 0xffffu; cn_pure[k] = uint32_t(pure4 >> (16 * k)) & 0xffffu; }
        const int ref = pk.col_ref[c] & 3;
        const bool ref_ok = (pk.col_ref[c] & 0x80) == 0;     // rows whose reference base is not ACGT are skipped (:329-331)
        const int depth = int(stars + cn_all[0] + cn_all[1] + cn_all[2] + cn_all[3]);
        const double den = depth > 0 ? double(depth) : 1.0;
        bool pass_snv = false, has_alt_base = false, pass_indel = false;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (b == ref) continue;
            const int cnt = int(cn_all[b]);
            pass_snv = pass_snv || (double(cnt) / den >= snv_min_af && cnt >= alt_base_num);
            has_alt_base = has_alt_base || cn_pure[b] > 0u;
        }
        if (select_indel) {
            if (!grp_lds) __threadfence();
            for (int g = 0; g < nk; ++g) {           // groups are numbered from the column's first key; unused slots stay 0
                const int cnt = int(grp_lds ? s_grp[select_indel ? lane : 0][g] : __hip_atomic_load(&gscratch[k0 + g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                pass_indel = pass_indel || (double(cnt) / den >= indel_min_af && cnt >= alt_base_num);
            }
        }
        const bool pass_af = ref_ok && (pass_snv || pass_indel) && double(depth) > min_coverage;
        uint8_t f = 0;
        if (pass_af) {
            f |= 4;
            if (pass_snv && has_alt_base) f |= 1;
            if (select_indel && pass_indel) f |= 2;
        }
        if (ref_ok && n_row > 0u) {                 // what a hybrid / genotyping position needs to know of a row that fails the gates (:374-383)
            f |= 32;
            if (has_alt_base) f |= 8;
            if (select_indel && n_ind > 0u) f |= 16;
        }
        flags[c] = f;
        depth_out[c] = ref_ok ? depth : 0;   // skipped rows (reference base not ACGT) report nothing
    }
}


// candidate positions, in column order: the columns whose flag has `bit` set and whose position lies in [lo, hi] - and, for the
// SNV (bit 1) and indel (bit 2) lists, the marked columns (bit 6: a position of --hybrid_mode_vcf_fn / --genotyping_mode_vcf_fn) that
// fail the AF gates but show an alternative base / an indel at all (extract_candidates_calling.py:374-383)
__device__ __forceinline__ bool is_cand(const uint8_t* __restrict__ flags, const int32_t* __restrict__ col_pos, int64_t c, int64_t n, int bit,
                                        int32_t lo, int32_t hi) {
    if (c >= n) return false;
    const unsigned f = flags[c];
    const unsigned inject = bit == 1 ? 8u : bit == 2 ? 16u : 0u;
    if (!(f & unsigned(bit)) && !((f & 64u) && !(f & 4u) && (f & inject))) return false;
    const int32_t p = col_pos[c];
    return p >= lo && p <= hi;
}

// rows outside a BED: `samtools mpileup -l` prints a position p only inside a row  begin < p <= end  (0-based half-open rows; the
// intervals arrive sorted and merged).  clear = 0xff: the row does not exist (confident BED, :302); clear = 2 | 16: the position
// cannot be an indel candidate (--call_indels_only_in_these_regions, :437-446).
__global__ __launch_bounds__(256) void k_restrict(const int32_t* __restrict__ col_pos, int64_t n, const int32_t* __restrict__ iv, int n_iv,
                                                  unsigned clear, uint8_t* __restrict__ flags, int32_t* __restrict__ depth) {
    const int64_t c = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (c >= n) return;
    const int32_t q = col_pos[c] - 1;                 // 0-based
    int lo = 0, hi = n_iv;                            // first interval whose begin is beyond q
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (iv[2 * mid] <= q) lo = mid + 1; else hi = mid; }
    const bool inside = lo > 0 && q < iv[2 * lo - 1];
    if (!inside) {
        flags[c] = uint8_t(flags[c] & ~clear);
        if (clear == 0xffu && depth) depth[c] = 0;
    }
}
__global__ __launch_bounds__(256) void k_mark(const int32_t* __restrict__ col_pos, int64_t n, const int32_t* __restrict__ pos, int n_pos,
                                              unsigned set, uint8_t* __restrict__ flags) {
    const int64_t c = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (c >= n) return;
    const int32_t p = col_pos[c];
    int lo = 0, hi = n_pos;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (pos[mid] < p) lo = mid + 1; else hi = mid; }
    if (lo < n_pos && pos[lo] == p) flags[c] = uint8_t(flags[c] | set);
}

// The `tumor_alt_info` of a hybrid / genotyping position (extract_candidates_calling.py:352-354: depth + pileup_list, the allele counts
// of pileup_dict in decreasing order, ties in the order the dictionary met them): one LANE per listed position (a run has a few
// thousand of them at most) walks its column once and leaves counts and first-seen read indices; the host orders and prints them.
// rec[16]: column (-1: no row), depth, count A C G T I D, first-seen A C G T I D, the column's first key, the column's flags.
__global__ __launch_bounds__(64) void k_hybrid_info(XPack pk, const int32_t* __restrict__ col_pos, const int32_t* __restrict__ pos, int n_pos,
                                                    const uint8_t* __restrict__ flags, int min_mq, int min_bq, int select_indel,
                                                    int32_t* __restrict__ rec, uint32_t* __restrict__ gcnt, int32_t* __restrict__ gfirst) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_pos) return;
    int32_t* r = rec + int64_t(i) * 16;
    for (int k = 0; k < 16; ++k) r[k] = k >= 8 && k < 14 ? INT32_MAX : 0;
    r[0] = -1;
    const int32_t p = pos[i];
    int64_t lo = 0, hi = pk.n_cols;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (col_pos[mid] < p) lo = mid + 1; else hi = mid; }
    if (lo >= pk.n_cols || col_pos[lo] != p || !(flags[lo] & 32u)) return;
    const int64_t c = lo;
    const int k0 = pk.key_off[c], nk = pk.key_off[c + 1] - k0;
    if (select_indel)
        for (int g = 0; g < nk; ++g) { gcnt[k0 + g] = 0u; gfirst[k0 + g] = INT32_MAX; }
    int cnt[6] = {0, 0, 0, 0, 0, 0}, first[6] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MAX, INT32_MAX, INT32_MAX}, depth = 0;
    const int64_t e0 = pk.col_off[c], e1 = pk.col_off[c + 1];
    for (int64_t e = e0; e < e1; ++e) {
        const uint32_t ent = pk.entries[e];
        const uint32_t b = ent & 15u, kind = (ent >> 4) & 3u;
        const int bq = int((ent >> 6) & 127u), mq = int((ent >> 13) & 255u);
        if (mq < min_mq || bq < min_bq) continue;
        const int at = int(e - e0);
        if (b < 8u) { ++depth; ++cnt[b & 3u]; first[b & 3u] = min(first[b & 3u], at); }
        else if (b == 8u || b == 9u) ++depth;
        if (kind != 0u) {
            const int k = k0 + int(ent >> 21);
            if (select_indel) { const int g = k0 + pk.key_group[k]; ++gcnt[g]; gfirst[g] = min(gfirst[g], at); }
            else { const int w = (pk.key_meta[k] & 3u) == 1u ? 4 : 5; ++cnt[w]; first[w] = min(first[w], at); }
        }
    }
    r[0] = int32_t(c); r[1] = depth;
    for (int k = 0; k < 6; ++k) { r[2 + k] = cnt[k]; r[8 + k] = first[k]; }
    r[14] = k0; r[15] = int32_t(flags[c]);
}
__global__ __launch_bounds__(256) void k_cand_count(const uint8_t* __restrict__ flags, const int32_t* __restrict__ col_pos, int64_t n, int bit,
                                                    int32_t lo, int32_t hi, int32_t* __restrict__ block_cnt) {
    __shared__ int part[4];
    const int64_t c = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const unsigned long long m = __ballot(is_cand(flags, col_pos, c, n, bit, lo, hi));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ __launch_bounds__(1024) void k_cand_scan(int32_t* __restrict__ block_cnt, int n_blocks, int32_t* __restrict__ total) {   // one workgroup
    __shared__ int part[1024];
    const int t = threadIdx.x, per = (n_blocks + 1023) / 1024;
    const int lo = min(n_blocks, t * per), hi = min(n_blocks, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += block_cnt[i];
    part[t] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int i = lo; i < hi; ++i) { const int v = block_cnt[i]; block_cnt[i] = run; run += v; }       // exclusive
    if (t == 1023) *total = part[t];
}
__global__ __launch_bounds__(256) void k_cand_scatter(const uint8_t* __restrict__ flags, const int32_t* __restrict__ col_pos, int64_t n, int bit,
                                                      int32_t lo, int32_t hi, const int32_t* __restrict__ block_base, int32_t* __restrict__ out,
                                                      int64_t cap) {
    __shared__ int part[4];
    const int64_t c = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const bool take = is_cand(flags, col_pos, c, n, bit, lo, hi);
    const unsigned long long m = __ballot(take);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) part[w] = __popcll(m);
    __syncthreads();
    int base = block_base[blockIdx.x];
    for (int i = 0; i < w; ++i) base += part[i];
    const int64_t at = base + __popcll(m & ((1ull << lane) - 1ull));
    if (take && at < cap) out[at] = col_pos[c];
}

}  // namespace

int cto::extract_candidates_scratch(const cto_pack_view* dp, int min_mq, int min_bq, double snv_min_af, double indel_min_af, double min_coverage,
                                    int alt_base_num, int select_indel, uint32_t* scratch, uint8_t* flags, int32_t* depth, void* stream) {
    if (dp->n_cols == 0) return CTO_OK;
    XPack pk{dp->n_cols, dp->col_ref, dp->col_off, dp->key_off, dp->entries, dp->key_meta, dp->key_group};
    const unsigned grid = unsigned(cto::cdiv(dp->n_cols, XLANES));
    if (select_indel)
        hipLaunchKernelGGL(k_extract_candidates<true>, dim3(grid), dim3(XLANES), 0, static_cast<hipStream_t>(stream), pk, min_mq, min_bq, snv_min_af,
                           indel_min_af, min_coverage, alt_base_num, scratch, flags, depth);
    else
        hipLaunchKernelGGL(k_extract_candidates<false>, dim3(grid), dim3(XLANES), 0, static_cast<hipStream_t>(stream), pk, min_mq, min_bq, snv_min_af,
                           indel_min_af, min_coverage, alt_base_num, scratch, flags, depth);
    CTO_HIP(hipGetLastError());
    return CTO_OK;
}

extern "C" int cto_extract_candidates(const cto_pack_view* dp, int min_mq, int min_bq, double snv_min_af,
                                      double indel_min_af, double min_coverage, int alt_base_num, int select_indel,
                                      uint8_t* flags, int32_t* depth, void* stream) {
    CTO_REQUIRE(dp && flags && depth, CTO_EINVAL, "cto_extract_candidates: null argument");
    if (dp->n_cols == 0) return CTO_OK;
    // merged-allele counters of the overflow path (a column with more than XG distinct indel alleles), grown on demand.  One buffer
    // per calling THREAD and device: two threads never share counters, and a thread that changes streams has its next launch wait for
    // the last one (cto_run_chunks brings a buffer per chunk slot instead and needs neither)
    struct Scratch {
        uint32_t* p = nullptr;
        int64_t n = 0;
        int dev = -1;
        hipStream_t last = nullptr;
        hipEvent_t done = nullptr;
    };
    thread_local Scratch sc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int dev = -1;
    CTO_HIP(hipGetDevice(&dev));
    if (dev != sc.dev) { sc = Scratch{}; sc.dev = dev; }      // another device became current: start over there (the old buffer is that device's)
    if (!sc.done) CTO_HIP(hipEventCreateWithFlags(&sc.done, hipEventDisableTiming));
    if (dp->n_keys > sc.n) {
        if (sc.p) { CTO_HIP(hipEventSynchronize(sc.done)); (void)hipFree(sc.p); sc.p = nullptr; }
        sc.n = dp->n_keys + dp->n_keys / 4 + 1024;
        CTO_HIP(hipMalloc(reinterpret_cast<void**>(&sc.p), size_t(sc.n) * 4));
    } else if (sc.last != st && sc.p) {
        CTO_HIP(hipStreamWaitEvent(st, sc.done, 0));
    }
    const int rc = cto::extract_candidates_scratch(dp, min_mq, min_bq, snv_min_af, indel_min_af, min_coverage, alt_base_num, select_indel, sc.p, flags,
                                                   depth, stream);
    if (rc == CTO_OK) { CTO_HIP(hipEventRecord(sc.done, st)); sc.last = st; }
    return rc;
}

extern "C" int cto_candidate_positions(const cto_pack_view* dp, const uint8_t* flags, int bit, int32_t lo, int32_t hi, int32_t* out_pos,
                                       int64_t cap, int32_t* scratch, int32_t* n_out, void* stream) {
    CTO_REQUIRE(dp && (dp->n_cols == 0 || (flags && out_pos && scratch)) && n_out, CTO_EINVAL, "cto_candidate_positions: null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dp->n_cols == 0) { CTO_HIP(hipMemsetAsync(n_out, 0, 4, s)); return CTO_OK; }
    const int64_t nb = cto::cdiv(dp->n_cols, 256);
    CTO_REQUIRE(nb < (int64_t(1) << 30), CTO_EUNSUPPORTED, "cto_candidate_positions: too many columns");
    hipLaunchKernelGGL(k_cand_count, dim3(unsigned(nb)), dim3(256), 0, s, flags, dp->col_pos, dp->n_cols, bit, lo, hi, scratch);
    hipLaunchKernelGGL(k_cand_scan, dim3(1), dim3(1024), 0, s, scratch, int(nb), n_out);
    hipLaunchKernelGGL(k_cand_scatter, dim3(unsigned(nb)), dim3(256), 0, s, flags, dp->col_pos, dp->n_cols, bit, lo, hi, scratch, out_pos, cap);
    CTO_HIP(hipGetLastError());
    return CTO_OK;
}

extern "C" int cto_extract_restrict(const cto_pack_view* dp, uint8_t* flags, int32_t* depth, const int32_t* d_intervals, int n_intervals, int clear,
                                    void* stream) {
    CTO_REQUIRE(dp && (dp->n_cols == 0 || flags) && (n_intervals == 0 || d_intervals) && n_intervals >= 0, CTO_EINVAL, "cto_extract_restrict: bad argument");
    if (dp->n_cols == 0) return CTO_OK;
    hipLaunchKernelGGL(k_restrict, dim3(unsigned(cto::cdiv(dp->n_cols, 256))), dim3(256), 0, static_cast<hipStream_t>(stream), dp->col_pos, dp->n_cols,
                       d_intervals, n_intervals, unsigned(clear) & 0xffu, flags, depth);
    CTO_HIP(hipGetLastError());
    return CTO_OK;
}

extern "C" int cto_extract_mark(const cto_pack_view* dp, uint8_t* flags, const int32_t* d_pos, int n_pos, int set, void* stream) {
    CTO_REQUIRE(dp && (dp->n_cols == 0 || flags) && (n_pos == 0 || d_pos) && n_pos >= 0, CTO_EINVAL, "cto_extract_mark: bad argument");
    if (dp->n_cols == 0 || n_pos == 0) return CTO_OK;
    hipLaunchKernelGGL(k_mark, dim3(unsigned(cto::cdiv(dp->n_cols, 256))), dim3(256), 0, static_cast<hipStream_t>(stream), dp->col_pos, dp->n_cols, d_pos,
                       n_pos, unsigned(set) & 0xffu, flags);
    CTO_HIP(hipGetLastError());
    return CTO_OK;
}

extern "C" int cto_hybrid_info(const cto_pack_view* dp, const uint8_t* flags, const int32_t* d_pos, int n_pos, int min_mq, int min_bq, int select_indel,
                               int32_t* rec, uint32_t* gcnt, int32_t* gfirst, void* stream) {
    CTO_REQUIRE(dp && n_pos >= 0 && (n_pos == 0 || (d_pos && rec && (dp->n_cols == 0 || flags))) && (!select_indel || dp->n_keys == 0 || (gcnt && gfirst)),
                CTO_EINVAL, "cto_hybrid_info: bad argument");
    if (n_pos == 0) return CTO_OK;
    XPack pk{dp->n_cols, dp->col_ref, dp->col_off, dp->key_off, dp->entries, dp->key_meta, dp->key_group};
    hipLaunchKernelGGL(k_hybrid_info, dim3(unsigned(cto::cdiv(n_pos, 64))), dim3(64), 0, static_cast<hipStream_t>(stream), pk, dp->col_pos, d_pos, n_pos,
                       flags, min_mq, min_bq, select_indel ? 1 : 0, rec, gcnt, gfirst);
    CTO_HIP(hipGetLastError());
    return CTO_OK;
}
