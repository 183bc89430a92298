// Host-side text of the posterior / genotype / quality stage: VCF data rows for a whole chunk in one call.
//
// Counterpart of the per-site tail of clairs/call_variants.py (output_vcf_from_probability, lines 135-150 decode_alt_info,
// 306-365 allele ranking / ALT / REF, 367-380 drop rules, 401-415 AF / GT, 67-76 FILTER, 588-618 INFO / FORMAT) and of
// VcfWriter.write_row (shared/vcf.py:144-185).  The arg-max and QUAL come from the device epilogue (posterior.hip); what is
// left is string work, which at 16 ms per 4 000 sites in Python was a third of a chunk's wall time.
// clairs_to_amd/call_variants.py:vcf_row is the same logic one site at a time; tests hold the two equal on every fixture.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "common.h"

using namespace cto;

namespace {

struct Allele {
    const char* key;
    int klen;
    long long count;
};

// "<depth>-<k1> <c1> <k2> <c2> ...-": dict(zip(tokens[::2], tokens[1::2])) semantics - a repeated key keeps its first
// position and takes the last value; a dangling key without a count is dropped.  Returns false on a malformed number.
bool parse_alt_info(const char* s, int len, std::vector<Allele>& out, long long* depth) {
    out.clear();
    while (len > 0 && (s[len - 1] == ' ' || s[len - 1] == '\n' || s[len - 1] == '\r' || s[len - 1] == '\t')) --len;   // rstrip
    int p = 0;
    bool neg = false;
    if (p < len && (s[p] == '-' || s[p] == '+')) return false;     // int('-5') would parse, but '-' is the field separator
    long long d = 0;
    int nd = 0;
    while (p < len && s[p] >= '0' && s[p] <= '9') { d = d * 10 + (s[p] - '0'); ++p; ++nd; }
    if (nd == 0) return false;
    *depth = neg ? -d : d;
    if (p >= len) return true;                                      // "0" : no second field
    if (s[p] != '-') return false;
    ++p;
    int e = p;
    while (e < len && s[e] != '-') ++e;                              // second '-'-separated field
    // tokens split by single spaces (str.split(' ') keeps empty tokens)
    std::vector<std::pair<int, int>> tok;
    int b = p;
    for (int i = p; i <= e; ++i)
        if (i == e || s[i] == ' ') { tok.emplace_back(b, i - b); b = i + 1; }
    for (size_t t = 0; t + 1 < tok.size(); t += 2) {
        const char* k = s + tok[t].first;
        const int kl = tok[t].second;
        const char* v = s + tok[t + 1].first;
        const int vl = tok[t + 1].second;
        if (vl == 0) return false;
        long long c = 0;
        int i = 0;
        bool vneg = false;
        if (v[0] == '-' || v[0] == '+') { vneg = v[0] == '-'; i = 1; }
        if (i >= vl) return false;
        for (; i < vl; ++i) {
            if (v[i] < '0' || v[i] > '9') return false;
            c = c * 10 + (v[i] - '0');
        }
        if (vneg) c = -c;
        bool dup = false;
        for (Allele& a : out)
            if (a.klen == kl && memcmp(a.key, k, size_t(kl)) == 0) { a.count = c; dup = true; break; }
        if (!dup) out.push_back(Allele{k, kl, c});
    }
    if (*depth == 0 && out.size() == 1 && out[0].klen > 0 && (out[0].key[0] == 'D' || out[0].key[0] == 'I'))
        *depth = out[0].count;                                       // all-indel column (call_variants.py:143-148)
    return true;
}

}  // namespace

namespace {
// QUAL of one site as the reference's quality_score_from + round(q, 4) evaluate it on THIS host (clairs/call_variants.py:79-88:
// math.log is the C library's log).  The device epilogue flags the sites whose q * 1e4 lies within 1e-6 of a ...5 boundary
// (decision[.][1] bit 2) and leaves the winning posterior's bits in decision[.][2..3]: for those - and only for those - a last-bit
// difference between the device's log() and the host's could print a different 4th decimal, so they are re-evaluated here.
double host_qual(double p) {
#pragma clang fp contract(off)
    const double phred = -10.0 * (1.0 / 2.302585092994046);   // -10 * log(e, 10)
    double q = phred * std::log(((1.0 - p) + 1e-10) / (p + 1e-10)) + 2.0;
    q = q > 0.0 ? q : 0.0;
    if (!(q == q)) return q;
    // Python's round(q, 4): exact value of q to 4 decimals, ties to even (q * 1e4 = hi + lo exactly, one fma)
    const double hi = q * 1e4, lo = std::fma(q, 1e4, -hi);
    double r = std::nearbyint(hi);
    const double d = (hi - r) + lo;
    const bool odd = std::fmod(r, 2.0) != 0.0;
    if (d > 0.5 || (d == 0.5 && odd)) r += 1.0;
    else if (d < -0.5 || (d == -0.5 && odd)) r -= 1.0;
    return r / 1e4;
}
inline double flagged_posterior(const int32_t* dec) {
    const uint64_t bits = uint64_t(uint32_t(dec[2])) | (uint64_t(uint32_t(dec[3])) << 32);
    double p;
    memcpy(&p, &bits, 8);
    return p;
}
}  // namespace

// Host half of the QUAL evaluation: rewrites qual[i] of every site the device flagged as sitting on a rounding boundary with the host
// libm's value, clears the flag (decision[i][1] bit 2) and the posterior bits (decision[i][2..3]); afterwards decision is
// {argmax, flags (bits 0-1), 0, 0} and qual is exactly what clairs/call_variants.py:79-88 prints on this machine.  Idempotent.
// Returns the number of sites rewritten.  cto_vcf_rows_batch does the same on the fly, so a caller that only formats rows
// need not call this.
extern "C" int64_t cto_qual_finalize(int32_t* decision, double* qual, int64_t n) {
    if (n <= 0) return 0;
    CTO_REQUIRE(decision && qual, CTO_EINVAL, "cto_qual_finalize: null argument");
    int64_t fixed = 0;
    for (int64_t i = 0; i < n; ++i) {
        int32_t* dec = decision + i * 4;
        if (!(dec[1] & 4)) continue;
        qual[i] = host_qual(flagged_posterior(dec));
        dec[1] &= ~4;
        dec[2] = dec[3] = 0;
        ++fixed;
    }
    return fixed;
}

// One chunk of sites -> VCF data rows (each terminated by '\n') in `buf`.
//   chrom                 contig name
//   pos[n]                1-based positions
//   centre[n]             reference base of the row (column 3 of the probability rows, clairs/predict.py:415); the caller
//                         drops sites whose raw centre is not in "ACGT" (predict.py:219-228) by setting the skip bit below
//   alt_buf, alt_off[n+1] the AFF alt_info strings, as cto_alt_info_batch writes them
//   site_info[n][12]      cto_gather_windows output: [3] bit 0 = skip this site (no tensor / dropped by the caller), [4..8) forward A C G T,
//                         [8..12) reverse A C G T strand counts (clairs/predict.py:626-642)
//   decision[n][4], qual[n]  device epilogue outputs (cto_posterior): arg-max, flags; QUAL
//   K                     4 = SNV mode (--disable_indel_calling True), 6 = indel mode
//   show_ref              --show_ref;  qual_pass: --qual (FILTER LowQual below it), < 0 = no threshold
//   counts[4]             out: rows written, sites processed (skip bit clear), "low tumor coverage" events,
//                         sites whose probability hit the clamped bin (decision[.][1] != 0)
// Returns the bytes used, CTO_EINVAL for malformed input, or CTO_ENOMEM when cap is too small (call again with a larger
// buffer; 512 bytes per site + twice the allele strings is always enough).
extern "C" int64_t cto_vcf_rows_batch(const char* chrom, int64_t n, const int32_t* pos, const char* centre, const char* alt_buf,
                                      const int64_t* alt_off, const int32_t* site_info, const int32_t* decision, const double* qual,
                                      int K, int show_ref, double qual_pass, char* buf, size_t cap, int64_t* counts) try {
    CTO_REQUIRE(chrom && (n == 0 || (pos && centre && alt_buf && alt_off && site_info && decision && qual)) && buf && counts &&
                    (K == 4 || K == 6),
                CTO_EINVAL, "cto_vcf_rows_batch: bad argument");
    static const char ACGT[] = "ACGT";
    const bool snv_mode = K == 4;
    std::vector<Allele> d;
    std::string ref, alt;
    size_t used = 0;
    int64_t n_rows = 0, n_sites = 0, n_lowcov = 0, n_clamped = 0;
    const size_t chrom_len = strlen(chrom);
    for (int64_t i = 0; i < n; ++i) {
        const int32_t* info = site_info + i * 12;
        if (info[3] & 1) continue;
        const char cb = centre[i];
        ++n_sites;
        const int argmax = decision[i * 4];
        CTO_REQUIRE(argmax >= 0 && argmax < K, CTO_EINVAL, "cto_vcf_rows_batch: site %lld has arg-max %d outside [0, %d)", (long long)i,
                    argmax, K);
        if (decision[i * 4 + 1] & 3) ++n_clamped;
        const double q = (decision[i * 4 + 1] & 4) ? host_qual(flagged_posterior(decision + i * 4)) : qual[i];
        // 0/0 posterior (both heads of the winner printed as 0.00000000): the reference raises IndexError on this site
        // (call_variants.py:193); there is no posterior to format a row from - counted as clamped, reported by the caller
        if ((decision[i * 4 + 1] & 2) || std::isnan(q)) continue;
        const char* as = alt_buf + alt_off[i];
        const int alen = int(alt_off[i + 1] - alt_off[i]);
        long long depth = 0;
        CTO_REQUIRE(parse_alt_info(as, alen, d, &depth), CTO_EINVAL, "cto_vcf_rows_batch: malformed alt_info '%.*s' at %s:%d", alen, as, chrom,
                    pos[i]);
        bool is_variant = snv_mode ? (ACGT[argmax] != cb) : (argmax >= 4);
        bool is_reference = !is_variant;
        ref.assign(1, cb);
        alt.assign(1, cb);
        long long supported = 0;
        if (is_variant) {
            if (depth <= 0) { ++n_lowcov; continue; }
            // rank by count / depth, descending, stable (first-seen wins ties); 'R' and non-positive counts do not take part
            const Allele* best = nullptr;
            for (const Allele& a : d) {
                if (a.klen == 0) { continue; }
                if (a.key[0] == 'R' || a.count <= 0) continue;
                if (!best || a.count > best->count) best = &a;
            }
            if (!best) continue;
            supported = best->count;
            if (best->key[0] == 'X') {
                CTO_REQUIRE(best->klen >= 2, CTO_EINVAL, "cto_vcf_rows_batch: allele key '%.*s' too short", best->klen, best->key);
                alt.assign(1, best->key[1]);
                if (snv_mode) {
                    bool observed = false;
                    for (const Allele& a : d)
                        if (a.klen >= 2 && a.key[0] == 'X' && a.count > 0 && a.key[1] == ACGT[argmax]) observed = true;
                    if (!observed) { is_variant = false; is_reference = true; }   // the called base is not among the observed alleles
                }
            } else if (best->key[0] == 'I') {
                CTO_REQUIRE(best->klen >= 2, CTO_EINVAL, "cto_vcf_rows_batch: allele key '%.*s' too short", best->klen, best->key);
                if (best->key[1] != '#') alt.assign(best->key + 1, size_t(best->klen - 1));
                else { alt.assign(1, cb); alt.append(best->key + 2, size_t(best->klen - 2)); }
            } else if (best->key[0] == 'D') {
                alt.assign(1, cb);
                if (best->klen > 2) ref.append(best->key + 2, size_t(best->klen - 2));
            }
        }
        if ((!show_ref && is_reference) || (!is_reference && ref == alt)) continue;
        if (snv_mode && (ref.size() > 1 || alt.size() > 1)) continue;
        if (!snv_mode && ref.size() == 1 && alt.size() == 1 && !show_ref) continue;
        long long ref_num = 0;
        for (const Allele& a : d)
            if (a.klen > 0 && a.key[0] == 'R') ref_num = a.count;
        if (is_reference) { supported = ref_num; alt = "."; }
        double af = depth != 0 ? double(supported) / double(depth) : 0.0;
        if (af > 1.0) af = 1.0;
        const char* gt = is_reference ? "0/0" : (af < 1.0 ? "0/1" : "1/1");
        const char* flt = is_reference ? "RefCall" : ((qual_pass < 0 || q >= qual_pass) ? "PASS" : "LowQual");
        if (!show_ref && is_reference) continue;                    // VcfWriter.write_row skips 0/0 without show_ref_calls
        const int32_t* f = info + 4;
        const int32_t* r = info + 8;
        const size_t need = chrom_len + ref.size() + alt.size() + 400;
        if (used + need > cap) {
            set_error("cto_vcf_rows_batch: buffer too small (%zu bytes)", cap);
            return CTO_ENOMEM;
        }
        char ad[48];
        if (is_reference) snprintf(ad, sizeof ad, "%lld", supported);
        else snprintf(ad, sizeof ad, "%lld,%lld", ref_num, supported);
        used += size_t(snprintf(buf + used, cap - used,
                                "%s\t%d\t.\t%s\t%s\t%.4f\t%s\tFAU=%d;FCU=%d;FGU=%d;FTU=%d;RAU=%d;RCU=%d;RGU=%d;RTU=%d\t"
                                "GT:GQ:DP:AF:AD:AU:CU:GU:TU\t%s:%d:%lld:%.4f:%s:%d:%d:%d:%d\n",
                                chrom, pos[i], ref.c_str(), alt.c_str(), q, flt, f[0], f[1], f[2], f[3], r[0], r[1], r[2], r[3], gt,
                                int(q), depth, af, ad, f[0] + r[0], f[1] + r[1], f[2] + r[2], f[3] + r[3]));
        ++n_rows;
    }
    counts[0] = n_rows; counts[1] = n_sites; counts[2] = n_lowcov; counts[3] = n_clamped;
    return int64_t(used);
} CTO_CATCH("cto_vcf_rows_batch", int64_t)

// Candidate BED chunk file -> window centres (src/create_tensor_pileup_calling.py:347-370): rows `ctg <tab> x-17 <tab> x+17
// [<tab> type]` of contig `ctg`; position = start + 1, end = end + 1, centre = end - 18 when position < 1 (window clipped at
// the contig start) else position + (end - position) / 2 - 1.  Writes up to `cap` centres in file order (the caller sorts and
// de-duplicates: the reference keys a dict by them), span[0] / span[1] = min position / max end over the rows, and
// *has_types = 1 when some row carries the optional fourth column.  Returns the number of rows of the contig, or < 0.
extern "C" int64_t cto_bed_centres(const char* text, size_t len, const char* ctg, int32_t* out, int64_t cap, int64_t* span, int* has_types) try {
    CTO_REQUIRE(text && ctg && out && span && has_types, CTO_EINVAL, "cto_bed_centres: null argument");
    const size_t cl = strlen(ctg);
    int64_t n = 0, lo = INT64_MAX, hi = 0;
    *has_types = 0;
    for (const char* cur = text; cur < text + len;) {
        const char* eol = static_cast<const char*>(memchr(cur, '\n', size_t(text + len - cur)));
        if (!eol) eol = text + len;
        const char* e = eol;
        while (e > cur && (e[-1] == '\r' || e[-1] == ' ' || e[-1] == '\t')) --e;      // rstrip
        const char* t1 = static_cast<const char*>(memchr(cur, '\t', size_t(e - cur)));
        const char* t2 = t1 ? static_cast<const char*>(memchr(t1 + 1, '\t', size_t(e - t1 - 1))) : nullptr;
        if (t2 && size_t(t1 - cur) == cl && memcmp(cur, ctg, cl) == 0) {
            const char* t3 = static_cast<const char*>(memchr(t2 + 1, '\t', size_t(e - t2 - 1)));
            auto to_int = [](const char* b, const char* f, int64_t* v) {
                bool neg = false;
                if (b < f && (*b == '-' || *b == '+')) { neg = *b == '-'; ++b; }
                if (b >= f) return false;
                int64_t x = 0;
                for (; b < f; ++b) {
                    if (*b < '0' || *b > '9') return false;
                    x = x * 10 + (*b - '0');
                }
                *v = neg ? -x : x;
                return true;
            };
            int64_t a = 0, b = 0;
            CTO_REQUIRE(to_int(t1 + 1, t2, &a) && to_int(t2 + 1, t3 ? t3 : e, &b), CTO_EINVAL, "cto_bed_centres: malformed BED row '%.*s'",
                        int(std::min<ptrdiff_t>(e - cur, 80)), cur);
            if (t3 && !memchr(t3 + 1, '\t', size_t(e - t3 - 1))) *has_types = 1;        // exactly four columns
            const int64_t position = a + 1, end = b + 1;
            lo = std::min(lo, position);
            hi = std::max(hi, end);
            // Python's floor division for (end - position) // 2
            const int64_t d = end - position;
            const int64_t half = d >= 0 ? d / 2 : -((-d + 1) / 2);
            const int64_t centre = position < 1 ? end - 16 - 2 : position + half - 1;
            if (n < cap) out[n] = int32_t(centre);
            ++n;
        }
        cur = eol + 1;
    }
    span[0] = lo;
    span[1] = hi;
    return n;
} CTO_CATCH("cto_bed_centres", int64_t)
