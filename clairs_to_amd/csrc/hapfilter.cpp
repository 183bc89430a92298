// Long-read post-calling filters: the per-variant read-level evidence of src/haplotype_filtering.py
// (reference v0.4.4) - read start / end clustering, alternative-allele BQ / MQ, variant clusters, haplotype ancestry against
// phased germline variants, multi-haplotype support, the strand table for Fisher's test and the k-mer sequence entropy -
// computed for every call of one mpileup job in a single pass over the nine-column text
//     samtools mpileup --min-MQ q --min-BQ q --excl-flags 2316 [-l bed] -r ctg:lo-hi --output-MQ --output-QNAME --output-extra HP
// (haplotype_filtering.py:336-345 builds that command; :244-272 parses it; :344-707 turn it into the decisions).
//
// Host code by nature: a call looks at <= 201 columns x depth reads and there are thousands of calls, not millions; what the
// reference spends its time on is Python dict / set churn over read-name strings.  Here reads are interned to integers once per
// job, every column is an array of (read id, base, indel, BQ, MQ, HP) records, and each rule is a couple of loops over those
// arrays with per-call scratch indexed by read id.  The decisions are the reference's, quirk for quirk; the quirks that
// matter are spelled out where they are reproduced.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include "common.h"

using namespace cto;

namespace {

struct Rec {
    int read;            // interned "<qname>_<0|1>" (strand suffix from the base's case, :251-255)
    char base;           // upper-cased base character: A C G T N * #
    int indel_off, indel_len;   // sign + sequence as printed (raw case), empty when none
    int joined;          // id of upper(base + indel), the key of the column's Counter (:186)
    int bq, mq;
    char hp;             // first character of the HP field ('1', '2', anything else = untagged)
    bool hp_single;      // the HP field is exactly one character
};

struct Column {
    int64_t pos = 0;
    std::vector<Rec> recs;
    std::vector<int> rse;          // record indices flagged as read start / end (:176-185), -1 = "the last record" (Python index)
    bool present = false;
};

struct Job {
    std::string text_indels;       // backing store for indel strings
    std::vector<Column> cols;      // dense over [lo, hi]
    int64_t lo = 0, hi = -1;
    std::unordered_map<std::string, int> read_ids, joined_ids;
    std::vector<std::string> joined_names;
    std::vector<char> read_rev;    // per read id: the "_1" (reverse strand) suffix
    int intern(std::unordered_map<std::string, int>& m, const std::string& s, std::vector<std::string>* names = nullptr) {
        auto it = m.find(s);
        if (it != m.end()) return it->second;
        const int id = int(m.size());
        m.emplace(s, id);
        if (names) names->push_back(s);
        return id;
    }
    const Column* at(int64_t p) const {
        if (p < lo || p > hi) return nullptr;
        const Column& c = cols[size_t(p - lo)];
        return c.present ? &c : nullptr;
    }
};

inline char up(char c) { return (c >= 'a' && c <= 'z') ? char(c - 32) : c; }

// haplotype_filtering.py:157-187 (get_base_list) + :244-272 (_parse_mpileup_to_chunk_dict) for one row
int parse_row(const char* row, const char* end, Job& job, int64_t lo, int64_t hi) {
    const char* f[10];
    int nf = 0;
    f[nf++] = row;
    for (const char* q = row; q < end && nf < 10; ++q)
        if (*q == '\t') f[nf++] = q + 1;
    if (nf < 9) return CTO_OK;                          // fewer than nine columns: the reference skips the row (:248)
    auto fend = [&](int i) { return (i + 1 < nf) ? f[i + 1] - 1 : end; };
    int64_t pos = 0;
    for (const char* q = f[1]; q < fend(1); ++q) {
        CTO_REQUIRE(*q >= '0' && *q <= '9', CTO_EINVAL, "haplotype filter: bad position field in the mpileup text");
        pos = pos * 10 + (*q - '0');
    }
    if (pos < lo || pos > hi) return CTO_OK;
    Column& col = job.cols[size_t(pos - lo)];
    col.recs.clear();                                   // a repeated position replaces the earlier row (dict assignment)
    col.rse.clear();
    col.pos = pos;
    col.present = true;
    std::vector<int> starts, ends;
    const char* bs = f[4];
    const char* be = fend(4);
    for (const char* q = bs; q < be;) {
        const char c = *q;
        if (c == '+' || c == '-') {
            const char* p2 = q + 1;
            int64_t adv = 0;
            while (p2 < be && *p2 >= '0' && *p2 <= '9') { adv = adv * 10 + (*p2 - '0'); ++p2; }
            CTO_REQUIRE(!col.recs.empty(), CTO_EINVAL, "haplotype filter: indel before any base at position %lld", (long long)pos);
            const int64_t avail = std::min<int64_t>(adv, be - p2);
            Rec& r = col.recs.back();
            r.indel_off = int(job.text_indels.size());
            job.text_indels.push_back(c);
            job.text_indels.append(p2, size_t(avail));
            r.indel_len = int(avail) + 1;
            q = p2 + adv;
            continue;
        }
        if (c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N' || c == 'a' || c == 'c' || c == 'g' || c == 't' || c == 'n' ||
            c == '#' || c == '*') {
            Rec r;
            r.read = -1;
            r.base = c;                                  // case kept until the strand suffix is derived below
            r.indel_off = 0; r.indel_len = 0; r.joined = -1; r.bq = 0; r.mq = 0; r.hp = '*'; r.hp_single = true;
            col.recs.push_back(r);
        } else if (c == '^') {
            ++q;                                         // the mapping-quality character of a read start
            // the marker precedes its read's base, so this names the PREVIOUS record (-1 -> Python's "last element") - reproduced
            starts.push_back(int(col.recs.size()) - 1);
        }
        if (c == '$') ends.push_back(int(col.recs.size()) - 1);
        ++q;
    }
    // sets: duplicates collapse; the larger of the two sets is used, the end set on ties (:185)
    auto uniq = [](std::vector<int>& v) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); };
    uniq(starts);
    uniq(ends);
    col.rse = starts.size() > ends.size() ? starts : ends;
    const size_t n = col.recs.size();
    // BQ / MQ characters, read names, HP tags - one per record, same order
    const char* qs = f[5];
    const char* ms = f[6];
    CTO_REQUIRE(size_t(fend(5) - qs) >= n && size_t(fend(6) - ms) >= n, CTO_EINVAL,
                "haplotype filter: quality strings shorter than the base list at position %lld", (long long)pos);
    const char* nm = f[7];
    const char* nme = fend(7);
    const char* hp = f[8];
    const char* hpe = fend(8);
    while (hpe > hp && (hpe[-1] == '\n' || hpe[-1] == '\r')) --hpe;
    std::string key;
    for (size_t i = 0; i < n; ++i) {
        Rec& r = col.recs[i];
        r.bq = qs[i] - 33;
        r.mq = ms[i] - 33;
        CTO_REQUIRE(nm <= nme, CTO_EINVAL, "haplotype filter: fewer read names than bases at position %lld", (long long)pos);
        const char* ne = static_cast<const char*>(memchr(nm, ',', size_t(nme - nm)));
        if (!ne) ne = nme;
        const bool rev = r.base == '#' || (r.base >= 'a' && r.base <= 'z');
        key.assign(nm, size_t(ne - nm));
        key += rev ? "_1" : "_0";
        r.read = job.intern(job.read_ids, key);
        if (size_t(r.read) == job.read_rev.size()) job.read_rev.push_back(rev ? 1 : 0);
        nm = ne + 1;
        const char* he = static_cast<const char*>(memchr(hp, ',', size_t(hpe > hp ? hpe - hp : 0)));
        if (!he) he = hpe;
        r.hp = (he > hp) ? hp[0] : '\0';
        r.hp_single = (he - hp) == 1;
        hp = he + 1;
        r.base = up(r.base);
        key.assign(1, r.base);
        for (int k = 0; k < r.indel_len; ++k) key.push_back(up(job.text_indels[size_t(r.indel_off + k)]));
        r.joined = job.intern(job.joined_ids, key, &job.joined_names);
    }
    return CTO_OK;
}

// haplotype_filtering.py:99-152 with entropy_window = 33, kmer = 5 on the 33 reference characters around the call; the
// running sum is updated in the reference's order so that the result is the same double
double sequence_entropy(const char* seq, int len) {
    const int W = 33, K = 5;
    static const int num[26] = {/*A*/ 0, /*B*/ 1, /*C*/ 1, /*D*/ 0, 0, 0, /*G*/ 2, /*H*/ 0, 0, 0, /*K*/ 2, 0, /*M*/ 0, /*N*/ 0, 0, 0, 0,
                                /*R*/ 0, /*S*/ 1, /*T*/ 3, /*U*/ 3, /*V*/ 0, /*W*/ 0, 0, /*Y*/ 1, 0};
    double ent[W + 2];
    ent[0] = 0.0;
    for (int i = 1; i < W + 2; ++i) {
        const double e = 1.0 / W * i;
        ent[i] = e * std::log(e);
    }
    const double mul = -1 / std::log(double(W));
    std::vector<int> counts(size_t(1) << (2 * K), 0);
    const int mask = (1 << (2 * K)) - 1;
    int suffix = 0, prefix = 0;
    double sum = 0.0;
    for (int i = 0, i2 = -W; i2 < len; ++i, ++i2) {
        if (i < len) {
            const char c = up(seq[i]);
            suffix = ((suffix << 2) | ((c >= 'A' && c <= 'Z') ? num[c - 'A'] : 0)) & mask;
            sum -= ent[counts[size_t(suffix)]];
            counts[size_t(suffix)] += 1;
            sum += ent[counts[size_t(suffix)]];
        }
        if (i2 >= 0 && i < len) {
            const char c = up(seq[i2]);
            prefix = ((prefix << 2) | ((c >= 'A' && c <= 'Z') ? num[c - 'A'] : 0)) & mask;
            sum -= ent[counts[size_t(prefix)]];
            counts[size_t(prefix)] -= 1;
            sum += ent[counts[size_t(prefix)]];
        }
    }
    return sum * mul;
}

struct GermSite { int64_t pos; std::string alt; };

// "p1-ALT1,p2-ALT2" (:584-587); a set in the reference - duplicates collapse, order is irrelevant to the outcome
void parse_germline(const char* s, int len, std::vector<GermSite>& out) {
    out.clear();
    int b = 0;
    for (int i = 0; i <= len; ++i) {
        if (i == len || s[i] == ',') {
            if (i > b) {
                const char* dash = static_cast<const char*>(memchr(s + b, '-', size_t(i - b)));
                if (dash) {
                    GermSite g;
                    g.pos = atoll(std::string(s + b, size_t(dash - (s + b))).c_str());
                    g.alt.assign(dash + 1, size_t(s + i - dash - 1));
                    bool dup = false;
                    for (const GermSite& o : out) dup = dup || (o.pos == g.pos && o.alt == g.alt);
                    if (!dup) out.push_back(g);
                }
            }
            b = i + 1;
        }
    }
}

bool contains(const char* hay, int hlen, const char* needle, int nlen) {      // Python `needle in hay` ('' is in everything)
    if (nlen == 0) return true;
    for (int i = 0; i + nlen <= hlen; ++i)
        if (memcmp(hay + i, needle, size_t(nlen)) == 0) return true;
    return false;
}

}  // namespace

// flags per call, in this order
enum { F_PHASEABLE = 0, F_HETERO, F_HOMO, F_RSE, F_BQ, F_MQ, F_COEXIST, F_BOTH_SIDE, F_ENTROPY, F_NFLAGS };

// One mpileup job: nine-column text of one contig + the calls whose +-flanking windows it covers.
//   text/len            the rows (any positions outside [region_lo, region_hi] are ignored)
//   ref_seq             upper-cased reference of [region_lo, region_lo + ref_len) (samtools faidx ctg:lo-hi, :1096-1101)
//   n, pos[n]           the calls (1-based); fields / field_off[n + 1]: per call "REF\tALT\tHETERO_INFO\tHOMO_INFO" back to back
//   af[n]               AF of each call (1.0 when unknown, :1022)
//   flanking            100 (--flanking); max_co_exist_read_num = --min_alt_coverage (2); disable_rse = --disable_read_start_end_filtering
//   flags[n][9]         1 = pass / true, order of the enum above (phaseable, hetero, homo, read start-end, BQ, MQ, co-exist,
//                       hetero-both-side, sequence entropy)
//   strand[n][4]        a0, r0, a1, r1 of the 2x2 table for Fisher's exact test (:575-582; the caller computes the p-value in
//                       exact integer arithmetic as the reference does) - int64
extern "C" int cto_haplotype_filter(const char* text, size_t len, const char* ref_seq, int64_t region_lo, size_t ref_len, int64_t n,
                                    const int32_t* pos, const char* fields, const int64_t* field_off, const double* af, int flanking,
                                    int max_co_exist_read_num, int disable_rse, uint8_t* flags, int64_t* strand) try {
    CTO_REQUIRE(text && ref_seq && (n == 0 || (pos && fields && field_off && af && flags && strand)) && flanking > 0, CTO_EINVAL,
                "cto_haplotype_filter: bad argument");
    if (n == 0) return CTO_OK;
    Job job;
    job.lo = std::max<int64_t>(1, int64_t(*std::min_element(pos, pos + n)) - flanking);
    job.hi = int64_t(*std::max_element(pos, pos + n)) + flanking;
    CTO_REQUIRE(job.hi - job.lo < (int64_t(1) << 26), CTO_EUNSUPPORTED, "cto_haplotype_filter: job spans more than 64 Mb; split it");
    job.cols.resize(size_t(job.hi - job.lo + 1));
    for (const char* cur = text; cur < text + len;) {
        const char* eol = static_cast<const char*>(memchr(cur, '\n', size_t(text + len - cur)));
        if (!eol) eol = text + len;
        if (eol > cur) {
            const int rc = parse_row(cur, eol, job, job.lo, job.hi);
            if (rc != CTO_OK) return rc;
        }
        cur = eol + 1;
    }
    const int n_reads = int(job.read_ids.size());
    std::vector<int> hap(size_t(n_reads) + 1);           // hap_dict: 0 unless assigned (:608-612)
    std::vector<char> is_alt(size_t(n_reads) + 1), in_rse(size_t(n_reads) + 1), mark(size_t(n_reads) + 1);
    std::vector<GermSite> hetero, homo;
    std::vector<int> alt_reads, tmp_ids, cnt_ids, cnt_vals;
    for (int64_t v = 0; v < n; ++v) {
        uint8_t* fl = flags + v * F_NFLAGS;
        for (int k = 0; k < F_NFLAGS; ++k) fl[k] = 1;
        int64_t* st = strand + v * 4;
        const char* fs = fields + field_off[v];
        const char* fe = fields + field_off[v + 1];
        const char* tab[3] = {nullptr, nullptr, nullptr};
        int nt = 0;
        for (const char* q = fs; q < fe && nt < 3; ++q)
            if (*q == '\t') tab[nt++] = q;
        CTO_REQUIRE(nt == 3, CTO_EINVAL, "cto_haplotype_filter: call %lld needs REF, ALT, hetero and homo fields", (long long)v);
        const std::string ref_base(fs, size_t(tab[0] - fs)), alt_base(tab[0] + 1, size_t(tab[1] - tab[0] - 1));
        parse_germline(tab[1] + 1, int(tab[2] - tab[1] - 1), hetero);
        parse_germline(tab[2] + 1, int(fe - tab[2] - 1), homo);
        const int64_t p0 = pos[v];
        const bool is_snp = ref_base.size() == 1 && alt_base.size() == 1;
        const bool is_ins = ref_base.size() == 1 && alt_base.size() > 1;
        const bool is_del = ref_base.size() > 1 && alt_base.size() == 1;
        const int64_t ref_anchor = std::max<int64_t>(p0 - flanking, 1), ref_end = p0 + flanking + 1;
        // ref_seq_site = chunk_ref[ref_anchor - region_lo : ref_end - region_lo + 1], clipped like a Python slice (:596-601)
        const int64_t s0 = std::max<int64_t>(0, std::min<int64_t>(int64_t(ref_len), ref_anchor - region_lo));
        const int64_t s1 = std::max<int64_t>(s0, std::min<int64_t>(int64_t(ref_len), ref_end - region_lo + 1));
        const char* site = ref_seq + s0;
        const int site_len = int(s1 - s0);
        auto site_ref = [&](int64_t p, char* out) {      // rb = ref_seq_site[p - ref_anchor] when inside (:401-405)
            const int64_t ri = p - ref_anchor;
            if (ri < 0 || ri >= site_len) return false;
            *out = site[ri];
            return true;
        };
        const int64_t win_lo = std::max<int64_t>(p0 - flanking, 1), win_hi = p0 + flanking;
        std::fill(hap.begin(), hap.end(), 0);
        std::fill(is_alt.begin(), is_alt.end(), 0);
        std::fill(in_rse.begin(), in_rse.end(), 0);
        alt_reads.clear();
        int64_t all_hap[3] = {0, 0, 0}, alt_hap[3] = {0, 0, 0}, all_fwd[3] = {0, 0, 0}, all_rev[3] = {0, 0, 0}, alt_fwd[3] = {0, 0, 0},
                alt_rev[3] = {0, 0, 0};
        auto is_het_pos = [&](int64_t p) {
            for (const GermSite& g : hetero)
                if (g.pos == p) return true;
            return false;
        };
        auto rec_is_alt = [&](const Rec& r) {            // :631-648 / :665-676
            const char* ind = job.text_indels.data() + r.indel_off;
            if (is_snp) return r.indel_len == 0 && r.base == alt_base[0];
            if (is_ins) {
                if (r.indel_len == 0 || !memchr(ind, '+', size_t(r.indel_len))) return false;
                // upper(base + indel without '+') == alt
                size_t k = 0;
                if (alt_base.size() < 1 || alt_base[k++] != r.base) return false;
                for (int i = 0; i < r.indel_len; ++i) {
                    if (ind[i] == '+') continue;
                    if (k >= alt_base.size() || alt_base[k++] != up(ind[i])) return false;
                }
                return k == alt_base.size();
            }
            if (is_del) return int(ref_base.size()) == r.indel_len && r.indel_len > 0 && memchr(ind, '-', size_t(r.indel_len)) != nullptr;
            return false;
        };
        // ---- pass over the window: haplotypes, start / end reads, evidence at the call itself (:603-707) ----
        for (int64_t p = win_lo; p <= win_hi; ++p) {
            const Column* c = job.at(p);
            if (!c) continue;
            if (p == p0 || is_het_pos(p)) {
                // a read's haplotype is taken from the call position and from heterozygous germline positions only, first
                // assignment wins; HP values other than '1' / '2' leave it unphased.  (`hap in '12'` also accepts the empty
                // string and the two-character '12', on which int() / the list index then fail in the reference: unphased here.)
                for (const Rec& r : c->recs)
                    if (r.hp_single && (r.hp == '1' || r.hp == '2') && hap[size_t(r.read)] == 0) hap[size_t(r.read)] = r.hp - '0';
            }
            if (double(c->rse.size()) >= double(c->recs.size()) * 0.2) {     // eps_rse (:614-615)
                for (int idx : c->rse) {
                    const int k = idx < 0 ? int(c->recs.size()) + idx : idx;
                    if (k >= 0 && k < int(c->recs.size())) in_rse[size_t(c->recs[size_t(k)].read)] = 1;
                }
            }
            if (p != p0) continue;
            // pos_dict[p] = dict(zip(names, bases)): a read name printed twice keeps its LAST record; the evidence lists below
            // are built from zip(...), i.e. from every record
            double bq_sum = 0, mq_sum = 0;
            int64_t n_alt_rec = 0;
            for (const Rec& r : c->recs) {
                if (rec_is_alt(r)) { bq_sum += r.bq; mq_sum += r.mq; ++n_alt_rec; }
            }
            if (n_alt_rec > 0 && bq_sum / double(n_alt_rec) <= 20.0) fl[F_BQ] = 0;      // param.ont_min_bq (:651-652)
            if (n_alt_rec > 0 && mq_sum / double(n_alt_rec) <= 20.0) fl[F_MQ] = 0;      // param.min_mq (:654-655)
            for (const Rec& r : c->recs) {                                               // every record counts here (:657-663)
                const int h = hap[size_t(r.read)];
                all_hap[h] += 1;
                (job.read_rev[size_t(r.read)] ? all_rev : all_fwd)[h] += 1;
            }
            for (const Rec& r : c->recs)
                if (rec_is_alt(r) && !is_alt[size_t(r.read)]) { is_alt[size_t(r.read)] = 1; alt_reads.push_back(r.read); }
            for (int rd : alt_reads) {                                                   // a set of names (:678-683)
                const int h = hap[size_t(rd)];
                alt_hap[h] += 1;
                (job.read_rev[size_t(rd)] ? alt_rev : alt_fwd)[h] += 1;
            }
        }
        const int64_t n_alt = int64_t(alt_reads.size());
        // ---- read start / end (:363-367) ----
        if (!disable_rse && n_alt > 0) {
            int64_t hit = 0;
            for (int rd : alt_reads) hit += in_rse[size_t(rd)];
            if (double(hit) >= 0.3 * double(n_alt)) fl[F_RSE] = 0;
        }
        // ---- haplotypes of the supporting reads (:369-384) ----
        const int64_t hp1 = alt_hap[1], hp2 = alt_hap[2];
        const int64_t MAX = std::max(hp1, hp2), MIN = std::min(hp1, hp2);
        const double a = af[v];
        if ((is_snp && a < 0.1) || (!is_snp && a < 0.3)) {
            if (hp1 * hp2 > 0 && (MIN > max_co_exist_read_num || double(MAX) / double(MIN) <= 10)) fl[F_BOTH_SIDE] = 0;
        }
        const bool phasable = hp1 * hp2 == 0 || (double(MAX) / double(MIN) >= 5 && (hp1 > max_co_exist_read_num || hp2 > max_co_exist_read_num));
        const int hap_index = !phasable ? 0 : (hp1 > hp2 ? 1 : 2);
        // ---- variant cluster (:390-441) ----
        int64_t match_count = 0, ins_length = 0;
        for (int64_t p = win_lo; p <= win_hi; ++p) {
            const Column* c = job.at(p);
            char rb;
            if (!c || !site_ref(p, &rb) || p == p0) continue;
            // last record of every read name at this position
            tmp_ids.clear();
            for (size_t i = c->recs.size(); i-- > 0;) {
                const Rec& r = c->recs[i];
                if (mark[size_t(r.read)]) continue;
                mark[size_t(r.read)] = 1;
                tmp_ids.push_back(int(i));
            }
            for (int i : tmp_ids) mark[size_t(c->recs[size_t(i)].read)] = 0;
            cnt_ids.clear();
            cnt_vals.clear();
            int64_t n_alt_list = 0;
            for (size_t t = tmp_ids.size(); t-- > 0;) {          // pileup order
                const Rec& r = c->recs[size_t(tmp_ids[t])];
                if (r.indel_len > 3 && job.text_indels[size_t(r.indel_off)] == '+') ins_length += std::min<int64_t>(r.indel_len - 1, 2 * flanking);
                if (!is_alt[size_t(r.read)]) continue;
                const std::string& jn = job.joined_names[size_t(r.joined)];
                if (jn.size() == 1 && (jn[0] == rb || jn[0] == '#' || jn[0] == '*')) continue;      // base != rb and base not in '#*'
                ++n_alt_list;
                size_t k = 0;
                for (; k < cnt_ids.size(); ++k)
                    if (cnt_ids[k] == r.joined) break;
                if (k == cnt_ids.size()) { cnt_ids.push_back(r.joined); cnt_vals.push_back(0); }
                cnt_vals[k] += 1;
            }
            if (n_alt_list == 0) continue;
            // most common allele among the supporting reads; on a tie the reference takes whichever its set iteration yields
            // first (hash-order dependent) - here the one seen first in pileup order
            size_t top = 0;
            for (size_t k = 1; k < cnt_ids.size(); ++k)
                if (cnt_vals[k] > cnt_vals[top]) top = k;
            const double tc = double(cnt_vals[top]);
            if (tc >= double(n_alt) * 1.5 || tc <= double(n_alt) * 0.5) continue;
            // pos_counter_dict holds the column's Counter unless the column is pure reference (:698-705)
            bool pure_ref = true;
            int first_joined = -1;
            for (const Rec& r : c->recs) {
                if (first_joined < 0) first_joined = r.joined;
                if (r.joined != first_joined) { pure_ref = false; break; }
            }
            char centre = 0;
            const int64_t ci = p - region_lo;
            if (ci < 0 || ci >= int64_t(ref_len)) continue;       // no counter recorded for this position
            centre = ref_seq[ci];
            if (pure_ref && !c->recs.empty()) {
                const std::string& jn = job.joined_names[size_t(first_joined)];
                if (jn.size() == 1 && jn[0] == centre) continue;  // single key equal to the reference base: not recorded
            }
            int64_t col_cnt = 0;
            for (const Rec& r : c->recs) col_cnt += r.joined == cnt_ids[top];
            if (double(col_cnt) >= tc * 1.5) continue;
            ++match_count;
        }
        // ---- ancestry on heterozygous germline variants of the call's haplotype (:443-476) ----
        auto germ_match = [&](const Rec& r, const std::string& ab, bool first_two) {
            const char* ind = job.text_indels.data() + r.indel_off;
            if (ab.size() == 1) return r.indel_len == 0 && r.base == ab[0];                 // ''.join(value) == ab
            // insertion allele: ab[:2] (hetero) / ab[1:2] (homo) must occur in the printed indel sequence (sign dropped)
            if (r.indel_len <= 1) return false;
            return first_two ? contains(ind + 1, r.indel_len - 1, ab.data(), int(std::min<size_t>(2, ab.size())))
                             : contains(ind + 1, r.indel_len - 1, ab.data() + 1, int(std::min<size_t>(1, ab.size() - 1)));
        };
        auto last_records = [&](const Column* c) {                 // dict semantics: one (the last) record per read name
            tmp_ids.clear();
            for (size_t i = c->recs.size(); i-- > 0;) {
                const Rec& r = c->recs[i];
                if (mark[size_t(r.read)]) continue;
                mark[size_t(r.read)] = 1;
                tmp_ids.push_back(int(i));
            }
            for (int i : tmp_ids) mark[size_t(c->recs[size_t(i)].read)] = 0;
        };
        if (hap_index > 0) {
            for (const GermSite& g : hetero) {
                const Column* c = (g.pos >= win_lo && g.pos <= win_hi) ? job.at(g.pos) : nullptr;
                char rb;
                if (!c || !site_ref(g.pos, &rb) || g.alt.empty()) continue;
                last_records(c);
                int64_t overlap = 0, phased = 0, inter = 0;
                for (int i : tmp_ids) {
                    const Rec& r = c->recs[size_t(i)];
                    if (!germ_match(r, g.alt, true)) continue;
                    ++overlap;
                    if (hap[size_t(r.read)] == hap_index) {
                        ++phased;
                        if (is_alt[size_t(r.read)]) ++inter;
                    }
                }
                if (phased == 0 || double(phased) * 2 < double(overlap)) continue;
                if (inter == 0) { fl[F_HETERO] = 0; break; }
            }
        }
        // ---- ancestry on homozygous germline variants (:478-541) ----
        for (const GermSite& g : homo) {
            const Column* c = (g.pos >= win_lo && g.pos <= win_hi) ? job.at(g.pos) : nullptr;
            char rb;
            if (!c || !site_ref(g.pos, &rb) || g.alt.empty()) continue;
            last_records(c);
            int64_t hh[3] = {0, 0, 0}, ah[3] = {0, 0, 0}, n_inter = 0, n_overlap = 0;
            for (int i : tmp_ids) {
                const Rec& r = c->recs[size_t(i)];
                const int h = hap[size_t(r.read)];
                ah[h] += 1;
                const bool m = germ_match(r, g.alt, false);
                if (m) hh[h] += 1;
                if (is_alt[size_t(r.read)]) {
                    ++n_inter;
                    if (m) ++n_overlap;
                }
            }
            const int64_t all_n = ah[0] + ah[1] + ah[2];
            const double af_g = all_n > 0 ? double(hh[0] + hh[1] + hh[2]) / double(all_n) : 0.0;
            bool homo_phasable = true;
            if (ah[1] * ah[2] == 0) homo_phasable = false;
            else if (hh[1] * hh[2] > 0 && double(std::max(hh[1], hh[2])) / double(std::min(hh[1], hh[2])) <= 10) homo_phasable = false;
            if (af_g < 0.75 || homo_phasable) continue;
            if (n_inter == 0) continue;
            if (n_overlap == 0 || double(n_overlap) / double(n_inter) < 0.5) { fl[F_HOMO] = 0; break; }
        }
        // ---- co-existence, phaseable flag, strand table, entropy (:543-590) ----
        int64_t depth = all_hap[0] + all_hap[1] + all_hap[2];
        if (depth <= 0) depth = 1;
        if (match_count >= max_co_exist_read_num || double(ins_length) / double(depth) > 3) fl[F_COEXIST] = 0;
        fl[F_PHASEABLE] = (all_hap[1] * all_hap[2] > 0 && alt_hap[1] * alt_hap[2] == 0 &&
                           (alt_hap[1] > max_co_exist_read_num || alt_hap[2] > max_co_exist_read_num)) ? 1 : 0;
        const int64_t a0 = alt_fwd[0] + alt_fwd[1] + alt_fwd[2], a1 = alt_rev[0] + alt_rev[1] + alt_rev[2];
        st[0] = a0;
        st[1] = (all_fwd[0] + all_fwd[1] + all_fwd[2]) - a0;
        st[2] = a1;
        st[3] = (all_rev[0] + all_rev[1] + all_rev[2]) - a1;
        if (!is_snp) {
            // sqeuence_entropy_from: reference_sequence[100 - 16 : 100 + 17] with the module's own flanking = 100 (:155-160)
            const int b = std::min(site_len, 100 - 16), e = std::min(site_len, 100 + 17);
            if (sequence_entropy(site + b, e - b) < 0.9) fl[F_ENTROPY] = 0;
        }
    }
    return CTO_OK;
} CTO_CATCH("cto_haplotype_filter", int)
