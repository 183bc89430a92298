// Candidate extraction on the column pack (reference: src/extract_candidates_calling.py).
//
// The reference scans `samtools mpileup --min-MQ 20 --min-BQ q` text of a 5 Mb chunk and keeps a position when
// its depth and some non-reference allele pass the AF / read-count gates (decode_pileup_bases :55-169, the
// candidate sets at :352-372).  Here the same gates run on the pack that tensor creation consumes, so a BAM is
// piled up ONCE for extraction, the AFF tensor and the NEG tensor.  HBM-bound integer work, same shape as
// k_featurize_columns: one wave per 8 consecutive columns, counters in LDS, per-allele (merged key) counts in an
// LDS table with a global-atomic overflow path.
#include <mutex>
#include "common.h"
#include "pack_internal.h"

namespace {

constexpr int XCOLS = 8, XWAVES = 4, GCAP = 128, XCOPY = 2;    // same tiling as k_featurize_columns: short waves, two counter copies

struct XPack {
    int64_t n_cols;
    const uint8_t* col_ref;
    const int64_t* col_off;
    const int32_t* key_off;
    const uint32_t* entries;
    const uint8_t* key_meta;
    const int32_t* key_group;
};

__global__ __launch_bounds__(64 * XWAVES) void k_extract_candidates(
    XPack pk, int min_mq, int min_bq, double snv_min_af, double indel_min_af, double min_coverage, int alt_base_num,
    int select_indel, uint32_t* __restrict__ gscratch, uint8_t* __restrict__ flags, int32_t* __restrict__ depth_out) {
    __shared__ uint32_t s_cnt[XWAVES][XCOPY][XCOLS][12];     // '*' / '#' placeholders, all-base ACGT (4), pure-base ACGT (4)
    __shared__ int64_t s_off[XWAVES][XCOLS + 1];
    __shared__ int32_t s_koff[XWAVES][XCOLS + 1];
    __shared__ uint32_t s_g[XWAVES][GCAP];            // merged-allele counts, indexed like the wave's keys
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t c0 = (int64_t(blockIdx.x) * XWAVES + w) * XCOLS;
    int ncol = 0;
    if (c0 < pk.n_cols) ncol = int(pk.n_cols - c0 < XCOLS ? pk.n_cols - c0 : XCOLS);
    for (int i = lane; i < XCOPY * XCOLS * 12; i += 64) (&s_cnt[w][0][0][0])[i] = 0u;
    for (int i = lane; i < GCAP; i += 64) s_g[w][i] = 0u;
    if (lane <= XCOLS) {
        const int64_t ci = c0 + (lane < ncol ? lane : ncol);
        s_off[w][lane] = ncol > 0 ? pk.col_off[ci] : 0;
        s_koff[w][lane] = ncol > 0 ? pk.key_off[ci] : 0;
    }
    __syncthreads();
    const int kbase = s_koff[w][0];
    const int nkeys_w = s_koff[w][ncol] - kbase;
    const bool in_lds = nkeys_w <= GCAP;
    if (!in_lds) {      // rare: the wave owns this slice of the global scratch exclusively and zeroes it itself
        for (int k = lane; k < nkeys_w; k += 64) gscratch[kbase + k] = 0u;
        __threadfence();
    }
    if (ncol > 0) {
        const int64_t e_end = s_off[w][ncol];
        int cl = 0;
        int64_t e = s_off[w][0] + lane;
        uint32_t ent = e < e_end ? pk.entries[e] : 0u;
        while (e < e_end) {
            const int64_t en = e + 64;
            const uint32_t ent_next = en < e_end ? pk.entries[en] : 0u;
            while (e >= s_off[w][cl + 1]) ++cl;
            const uint32_t b = ent & 15u, kind = (ent >> 4) & 3u, kid = ent >> 21;
            const int bq = int((ent >> 6) & 127u), mq = int((ent >> 13) & 255u);
            if (mq >= min_mq && bq >= min_bq) {       // what samtools --min-MQ / --min-BQ leaves in the column
                uint32_t* c = s_cnt[w][lane & (XCOPY - 1)][cl];
                if (b < 8u) {                                        // depth = bases + placeholders, summed at the end
                    atomicAdd(&c[1 + (b & 3u)], 1u);                 // pileup_dict[base] counts indel carriers too (:111-113)
                    if (kind == 0u) atomicAdd(&c[5 + (b & 3u)], 1u); // alt_dict single-base keys (:102)
                } else if (b == 8u || b == 9u) {
                    atomicAdd(&c[0], 1u);
                }
                if (kind != 0u) {                                    // no length gate here, unlike tensor creation
                    const int kl = s_koff[w][cl] - kbase;            // first key of the column, wave-local
                    const int g = kl + pk.key_group[kbase + kl + int(kid)];
                    if (in_lds) atomicAdd(&s_g[w][g], 1u);
                    else atomicAdd(&gscratch[kbase + g], 1u);
                }
            }
            e = en;
            ent = ent_next;
        }
    }
    if (!in_lds) __threadfence();
    __syncthreads();
    if (lane < ncol) {
        const int64_t c = c0 + lane;
        uint32_t cn[12];
        for (int i = 0; i < 12; ++i) {
            cn[i] = 0u;
            for (int q = 0; q < XCOPY; ++q) cn[i] += s_cnt[w][q][lane][i];
        }
        const int ref = pk.col_ref[c] & 3;
        const bool ref_ok = (pk.col_ref[c] & 0x80) == 0;     // rows whose reference base is not ACGT are skipped (:329-331)
        const int depth = int(cn[0] + cn[1] + cn[2] + cn[3] + cn[4]);
        const double den = depth > 0 ? double(depth) : 1.0;
        bool pass_snv = false, has_alt_base = false, pass_indel = false;
        for (int b = 0; b < 4; ++b) {
            if (b == ref) continue;
            const int cnt = int(cn[1 + b]);
            pass_snv = pass_snv || (double(cnt) / den >= snv_min_af && cnt >= alt_base_num);
            has_alt_base = has_alt_base || cn[5 + b] > 0u;
        }
        if (select_indel) {
            const int k0 = s_koff[w][lane] - kbase, k1 = s_koff[w][lane + 1] - kbase;
            for (int g = k0; g < k1; ++g) {          // groups are numbered from the column's first key; unused slots stay 0
                const int cnt = int(in_lds ? s_g[w][g] : __hip_atomic_load(&gscratch[kbase + g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
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
        flags[c] = f;
        depth_out[c] = ref_ok ? depth : 0;   // skipped rows (reference base not ACGT) report nothing
    }
}

// scratch for the overflow path lives in the library (grown on demand, per process / device)
// merged-allele counters, grown on demand; one process drives one GPU, so one buffer per process
uint32_t* g_scratch = nullptr;
int64_t g_scratch_n = 0;
int g_scratch_dev = -1;

// candidate positions, in column order: the columns whose flag has `bit` set and whose position lies in [lo, hi]
__device__ __forceinline__ bool is_cand(const uint8_t* __restrict__ flags, const int32_t* __restrict__ col_pos, int64_t c, int64_t n, int bit,
                                        int32_t lo, int32_t hi) {
    if (c >= n || !(flags[c] & bit)) return false;
    const int32_t p = col_pos[c];
    return p >= lo && p <= hi;
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
    const unsigned grid = unsigned(cto::cdiv(dp->n_cols, XCOLS * XWAVES));
    hipLaunchKernelGGL(k_extract_candidates, dim3(grid), dim3(64 * XWAVES), 0, static_cast<hipStream_t>(stream), pk, min_mq, min_bq, snv_min_af,
                       indel_min_af, min_coverage, alt_base_num, select_indel, scratch, flags, depth);
    CTO_HIP(hipGetLastError());
    return CTO_OK;
}

extern "C" int cto_extract_candidates(const cto_pack_view* dp, int min_mq, int min_bq, double snv_min_af,
                                      double indel_min_af, double min_coverage, int alt_base_num, int select_indel,
                                      uint8_t* flags, int32_t* depth, void* stream) {
    CTO_REQUIRE(dp && flags && depth, CTO_EINVAL, "cto_extract_candidates: null argument");
    if (dp->n_cols == 0) return CTO_OK;
    // merged-allele counters of the overflow path (a wave whose 8 columns hold more than 128 distinct indel alleles): one buffer per
    // process, grown on demand - calls of this entry are serialised (cto_run_chunks brings a buffer per chunk slot instead)
    static std::mutex m;
    std::lock_guard<std::mutex> g(m);
    {
        int dev = -1;
        CTO_HIP(hipGetDevice(&dev));
        if (dev != g_scratch_dev) { g_scratch = nullptr; g_scratch_n = 0; g_scratch_dev = dev; }   // another device became current: start over there
    }
    if (dp->n_keys > g_scratch_n) {
        if (g_scratch) { CTO_HIP(hipDeviceSynchronize()); (void)hipFree(g_scratch); }
        g_scratch_n = dp->n_keys + dp->n_keys / 4 + 1024;
        CTO_HIP(hipMalloc(reinterpret_cast<void**>(&g_scratch), size_t(g_scratch_n) * 4));
    }
    return cto::extract_candidates_scratch(dp, min_mq, min_bq, snv_min_af, indel_min_af, min_coverage, alt_base_num, select_indel, g_scratch, flags,
                                           depth, stream);
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
