// Candidate extraction on the column pack (reference: src/extract_candidates_calling.py).
//
// The reference scans `samtools mpileup --min-MQ 20 --min-BQ q` text of a 5 Mb chunk and keeps a position when
// its depth and some non-reference allele pass the AF / read-count gates (decode_pileup_bases :55-169, the
// candidate sets at :352-372).  Here the same gates run on the pack that tensor creation consumes, so a BAM is
// piled up ONCE for extraction, the AFF tensor and the NEG tensor.  HBM-bound integer work, same shape as
// k_featurize_columns: one wave per 8 consecutive columns, counters in LDS, per-allele (merged key) counts in an
// LDS table with a global-atomic overflow path.
#include "common.h"

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

}  // namespace

extern "C" int cto_extract_candidates(const cto_pack_view* dp, int min_mq, int min_bq, double snv_min_af,
                                      double indel_min_af, double min_coverage, int alt_base_num, int select_indel,
                                      uint8_t* flags, int32_t* depth, void* stream) {
    CTO_REQUIRE(dp && flags && depth, CTO_EINVAL, "cto_extract_candidates: null argument");
    if (dp->n_cols == 0) return CTO_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    {
        int dev = -1;
        CTO_HIP(hipGetDevice(&dev));
        if (dev != g_scratch_dev) { g_scratch = nullptr; g_scratch_n = 0; g_scratch_dev = dev; }   // another device became current: start over there
    }
    if (dp->n_keys > g_scratch_n) {
        if (g_scratch) (void)hipFree(g_scratch);
        g_scratch_n = dp->n_keys + dp->n_keys / 4 + 1024;
        CTO_HIP(hipMalloc(reinterpret_cast<void**>(&g_scratch), size_t(g_scratch_n) * 4));
    }
    XPack pk{dp->n_cols, dp->col_ref, dp->col_off, dp->key_off, dp->entries, dp->key_meta, dp->key_group};
    const unsigned grid = unsigned(cto::cdiv(dp->n_cols, XCOLS * XWAVES));
    hipLaunchKernelGGL(k_extract_candidates, dim3(grid), dim3(64 * XWAVES), 0, s, pk, min_mq, min_bq, snv_min_af,
                       indel_min_af, min_coverage, alt_base_num, select_indel, g_scratch, flags, depth);
    CTO_HIP(hipGetLastError());
    return CTO_OK;
}
