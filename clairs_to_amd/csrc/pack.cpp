// Host side of the column pack: mpileup text -> binary columns, and the alt_info string builder.
//
// Behaviour follows src/create_tensor_pileup_calling.py of the reference:
//   tokeniser           :120-144   (decode_pileup_bases, first loop)
//   row handling        :472-497   (pos, bases, BQ, MQ columns; reference base via evc_base_from :82-92)
//   indel length gates  :173-176, :188-191 (deletion uses len(seq)+1: the off-by-one is kept)
//   alt_info            :158-209
// Written from the behaviour, not from the text, of those lines.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <string>
#include <unordered_map>
#include <vector>
#include "common.h"
#include "pack_internal.h"

namespace cto {

thread_local int tl_pack_threads = 0;
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace cto

extern "C" const char* cto_last_error(void) { return cto::g_err; }
extern "C" void cto_set_pack_threads(int n) { cto::tl_pack_threads = n > 0 ? n : 0; }
extern "C" int cto_version(void) { return 100; }

namespace cto {

void set_err(std::string* err, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    *err = buf;
}

void pack_begin(cto_pack* p, size_t entries_hint, size_t cols_hint) {
    p->entries.reserve(entries_hint);
    p->col_pos.reserve(cols_hint);
    p->col_ref.reserve(cols_hint);
    p->col_off.reserve(cols_hint + 1);
    p->key_off.reserve(cols_hint + 1);
    p->col_off.push_back(0);
    p->key_off.push_back(0);
    p->key_str_off.push_back(0);
}

// Distinct-key bookkeeping of one indel-carrying read-base: returns the entry's kind bits (1 ins, 2 del, 3 longer than
// max_indel_length) and the id of its Counter key within the column (first-seen order); a new key also gets its key_meta,
// its merged candidate-extraction group and its alt_info string.  `nkeys_col` counts the column's keys so far.
static int intern_indel(cto_pack* p, ColumnScratch& sc, int* nkeys_col, int code, int tkind, const char* seq, int seqlen, int64_t ri,
                        const char* ref_seq, size_t ref_len, int max_indel_length, uint32_t* kind_out, uint32_t* kid_out, std::string* err) {
    uint32_t kind = uint32_t(tkind), kid = 0;
    const int gate_len = (tkind == 1) ? seqlen : seqlen + 1;
    const bool overlong = gate_len > max_indel_length;
    if (overlong) kind = 3;
    // distinct Counter key: base char + sign + sequence, case-sensitive
    uint32_t h = 2166136261u ^ uint32_t(code * 4 + tkind);
    for (int j = 0; j < seqlen; ++j) h = (h ^ uint8_t(seq[j])) * 16777619u;
    const size_t nk = sc.keys.size();
    size_t at = nk;
    for (size_t i = 0; i < nk; ++i)
        if (sc.key_hash[i] == h) {
            const ColumnScratch::Key& k = sc.keys[i];
            if (k.len == seqlen && k.code == uint8_t(code) && k.kind == uint8_t(tkind) && memcmp(k.seq, seq, size_t(seqlen)) == 0) { at = i; break; }
        }
    if (at == nk) {
        if (*nkeys_col >= kMaxKeysPerCol) { set_err(err, "more than %d distinct indel keys in one column", kMaxKeysPerCol); return CTO_EUNSUPPORTED; }
        kid = uint32_t((*nkeys_col)++);
        sc.key_hash.push_back(h);
        sc.keys.push_back(ColumnScratch::Key{seq, seqlen, uint8_t(code), uint8_t(tkind)});
        const bool fwd = (code < 4) || code == 8 || code == 10;
        p->key_meta.push_back(uint8_t(tkind | (fwd ? 4 : 0) | (overlong ? 8 : 0)));
        // merged allele for candidate extraction: insertions by upper-cased anchor + sequence,
        // deletions by length (extract_candidates_calling.py:118-126)
        static const char kAnchor[] = "ACGTACGT*#NN";
        const char anchor = tkind == 1 ? kAnchor[code] : 'D';
        uint32_t gh = 2166136261u ^ uint32_t(uint8_t(anchor));
        if (tkind == 1) for (int j = 0; j < seqlen; ++j) gh = (gh ^ uint8_t(up(seq[j]))) * 16777619u;
        else gh ^= uint32_t(seqlen) * 2654435761u;
        const size_t ng = sc.groups.size();
        size_t g = ng;
        for (size_t i = 0; i < ng; ++i)
            if (sc.group_hash[i] == gh) {
                const ColumnScratch::Group& q = sc.groups[i];
                if (q.kind != uint8_t(tkind) || q.len != seqlen) continue;
                if (tkind == 2) { g = i; break; }
                if (q.anchor != anchor) continue;
                int j = 0;
                while (j < seqlen && up(q.seq[j]) == up(seq[j])) ++j;
                if (j == seqlen) { g = i; break; }
            }
        if (g == ng) {
            sc.group_hash.push_back(gh);
            sc.groups.push_back(ColumnScratch::Group{seq, seqlen, anchor, uint8_t(tkind)});
        }
        p->key_group.push_back(int32_t(g));
        // merged alt_info key
        if (tkind == 1) {
            p->key_str.push_back('I');
            p->key_str.push_back(kAnchor[code]);
            for (int j = 0; j < seqlen; ++j) p->key_str.push_back(up(seq[j]));
        } else {
            p->key_str.push_back('D');
            // chunk_ref_seq[:len+1] with chunk_ref_seq = ref[pos : pos+max_indel_length].upper()
            int64_t take = std::min<int64_t>(seqlen + 1, max_indel_length);
            take = std::min<int64_t>(take, int64_t(ref_len) - ri);
            for (int64_t j = 0; j < take; ++j) p->key_str.push_back(up(ref_seq[ri + j]));
        }
        p->key_str_off.push_back(int64_t(p->key_str.size()));
    } else {
        kid = uint32_t(at);
    }
    *kind_out = kind;
    *kid_out = kid;
    return CTO_OK;
}

static inline void column_scratch_reset(ColumnScratch& sc) {
    sc.keys.clear();
    sc.key_hash.clear();
    sc.groups.clear();
    sc.group_hash.clear();
}

static inline void column_end(cto_pack* p, int64_t pos, int64_t ri, const char* ref_seq) {
    p->col_pos.push_back(int32_t(pos));
    const char ru = up(ref_seq[ri]);
    const bool acgt = ru == 'A' || ru == 'C' || ru == 'G' || ru == 'T';
    p->col_ref.push_back(uint8_t(ref_code_of(ref_seq[ri]) | (acgt ? 0 : 0x80)));
    p->col_off.push_back(int64_t(p->entries.size()));
    p->key_off.push_back(int32_t(p->key_meta.size()));
}

int append_column(cto_pack* p, ColumnScratch& sc, int64_t pos, int64_t ri, const char* ref_seq, size_t ref_len,
                  int max_indel_length, const Tok* toks, int n, std::string* err) {
    if (n > kMaxDepth) { set_err(err, "column depth %d > %d unsupported", n, kMaxDepth); return CTO_EUNSUPPORTED; }
    column_scratch_reset(sc);
    int nkeys_col = 0;
    for (int i = 0; i < n; ++i) {
        const Tok& t = toks[i];
        const int bq = std::max(0, std::min(t.bq, 127));
        const int mq = std::max(0, std::min(t.mq, 255));
        uint32_t kind = 0, kid = 0;
        if (t.kind != 0) {
            const int rc = intern_indel(p, sc, &nkeys_col, t.code, t.kind, t.seq, t.seqlen, ri, ref_seq, ref_len, max_indel_length, &kind, &kid, err);
            if (rc != CTO_OK) return rc;
        }
        p->entries.push_back(uint32_t(t.code) | (kind << 4) | (uint32_t(bq) << 6) | (uint32_t(mq) << 13) | (kid << 21));
    }
    column_end(p, pos, ri, ref_seq);
    return CTO_OK;
}

int append_column_packed(cto_pack* p, ColumnScratch& sc, int64_t pos, int64_t ri, const char* ref_seq, size_t ref_len,
                         int max_indel_length, const uint32_t* ents, int n, const IndelAt* indels, int n_indel, std::string* err) {
    if (n > kMaxDepth) { set_err(err, "column depth %d > %d unsupported", n, kMaxDepth); return CTO_EUNSUPPORTED; }
    column_scratch_reset(sc);
    const size_t e0 = p->entries.size();
    p->entries.resize(e0 + size_t(n));
    memcpy(p->entries.data() + e0, ents, size_t(n) * sizeof(uint32_t));
    int nkeys_col = 0;
    for (int i = 0; i < n_indel; ++i) {
        const IndelAt& it = indels[i];
        uint32_t kind = 0, kid = 0;
        const int rc = intern_indel(p, sc, &nkeys_col, int(ents[it.idx] & 15u), it.kind, it.seq, it.seqlen, ri, ref_seq, ref_len,
                                    max_indel_length, &kind, &kid, err);
        if (rc != CTO_OK) return rc;
        p->entries[e0 + size_t(it.idx)] |= (kind << 4) | (kid << 21);
    }
    column_end(p, pos, ri, ref_seq);
    return CTO_OK;
}

// Concatenates per-thread packs of consecutive position ranges (offsets re-based); nullptr + *err when they overlap.
std::unique_ptr<cto_pack> merge_parts(std::vector<std::unique_ptr<cto_pack>>& parts, std::string* err, uint32_t* ext_entries,
                                      size_t ext_cap) {
    if (parts.size() > 2) {
        // Parallel form: the destination arrays are sized once (the entry array without a zero fill) and every part copies
        // itself to its offsets on its own thread - the serial concatenation below was half of a multi-threaded call's time.
        const size_t n = parts.size();
        std::vector<int64_t> e0(n + 1, 0), c0(n + 1, 0), k0(n + 1, 0), s0(n + 1, 0);
        int64_t last_pos = -1;
        for (size_t t = 0; t < n; ++t) {
            const cto_pack& q = *parts[t];
            if (!q.col_pos.empty()) {
                if (q.col_pos.front() <= last_pos) { *err = "pileup rows not in increasing position order"; return nullptr; }
                last_pos = q.col_pos.back();
            }
            e0[t + 1] = e0[t] + int64_t(q.entries.size());
            c0[t + 1] = c0[t] + int64_t(q.col_pos.size());
            k0[t + 1] = k0[t] + int64_t(q.key_meta.size());
            s0[t + 1] = s0[t] + int64_t(q.key_str.size());
        }
        std::unique_ptr<cto_pack> p(new cto_pack());
        uint32_t* ent_dst;
        if (ext_entries && size_t(e0[n]) <= ext_cap) {
            p->ext_entries = ent_dst = ext_entries;
            p->ext_n = size_t(e0[n]);
        } else {
            p->entries.resize(size_t(e0[n]));
            ent_dst = p->entries.data();
        }
        p->col_pos.resize(size_t(c0[n]));
        p->col_ref.resize(size_t(c0[n]));
        p->col_off.resize(size_t(c0[n]) + 1);
        p->key_off.resize(size_t(c0[n]) + 1);
        p->key_meta.resize(size_t(k0[n]));
        p->key_group.resize(size_t(k0[n]));
        p->key_str_off.resize(size_t(k0[n]) + 1);
        p->key_str.resize(size_t(s0[n]));
        p->col_off[0] = 0;
        p->key_off[0] = 0;
        p->key_str_off[0] = 0;
        auto copy_part = [&](size_t t) {
            const cto_pack& q = *parts[t];
            if (!q.entries.empty()) memcpy(ent_dst + e0[t], q.entries.data(), q.entries.size() * sizeof(uint32_t));
            if (!q.col_pos.empty()) {
                memcpy(p->col_pos.data() + c0[t], q.col_pos.data(), q.col_pos.size() * sizeof(int32_t));
                memcpy(p->col_ref.data() + c0[t], q.col_ref.data(), q.col_ref.size());
            }
            for (size_t i = 1; i < q.col_off.size(); ++i) p->col_off[size_t(c0[t]) + i] = q.col_off[i] + e0[t];
            for (size_t i = 1; i < q.key_off.size(); ++i) p->key_off[size_t(c0[t]) + i] = q.key_off[i] + int32_t(k0[t]);
            if (!q.key_meta.empty()) {
                memcpy(p->key_meta.data() + k0[t], q.key_meta.data(), q.key_meta.size());
                memcpy(p->key_group.data() + k0[t], q.key_group.data(), q.key_group.size() * sizeof(int32_t));
            }
            for (size_t i = 1; i < q.key_str_off.size(); ++i) p->key_str_off[size_t(k0[t]) + i] = q.key_str_off[i] + s0[t];
            if (!q.key_str.empty()) memcpy(&p->key_str[size_t(s0[t])], q.key_str.data(), q.key_str.size());
        };
        std::vector<std::thread> th;
        for (size_t t = 1; t < n; ++t) th.emplace_back(copy_part, t);
        copy_part(0);
        for (auto& x : th) x.join();
        for (auto& q : parts) q.reset();
        return p;
    }
    std::unique_ptr<cto_pack> p(parts[0].release());
    for (size_t t = 1; t < parts.size(); ++t) {
        const cto_pack& q = *parts[t];
        if (q.col_pos.empty()) continue;
        if (!p->col_pos.empty() && q.col_pos.front() <= p->col_pos.back()) {
            *err = "pileup rows not in increasing position order";
            return nullptr;
        }
        const int64_t e0 = p->col_off.back(), s0 = p->key_str_off.back();
        const int32_t k0 = p->key_off.back();
        p->col_pos.insert(p->col_pos.end(), q.col_pos.begin(), q.col_pos.end());
        p->col_ref.insert(p->col_ref.end(), q.col_ref.begin(), q.col_ref.end());
        for (size_t i = 1; i < q.col_off.size(); ++i) p->col_off.push_back(q.col_off[i] + e0);
        for (size_t i = 1; i < q.key_off.size(); ++i) p->key_off.push_back(q.key_off[i] + k0);
        p->entries.insert(p->entries.end(), q.entries.begin(), q.entries.end());
        p->key_meta.insert(p->key_meta.end(), q.key_meta.begin(), q.key_meta.end());
        p->key_group.insert(p->key_group.end(), q.key_group.begin(), q.key_group.end());
        for (size_t i = 1; i < q.key_str_off.size(); ++i) p->key_str_off.push_back(q.key_str_off[i] + s0);
        p->key_str += q.key_str;
        parts[t].reset();
    }
    if (ext_entries && p->entries.size() <= ext_cap) {
        if (!p->entries.empty()) memcpy(ext_entries, p->entries.data(), p->entries.size() * sizeof(uint32_t));
        p->ext_entries = ext_entries;
        p->ext_n = p->entries.size();
        decltype(p->entries)().swap(p->entries);
    }
    return p;
}

}  // namespace cto

using namespace cto;

namespace {

struct IndelTok { int idx, kind; const char* seq; int seqlen; };

// character classes of the mpileup base string: 0..11 = read-base with that pack code, 12 = indel sign, 13 = '^', 14 = skipped
struct CharClass {
    uint8_t t[256];
    constexpr CharClass() : t() {
        for (int i = 0; i < 256; ++i) t[i] = 14;
        t[int('A')] = 0; t[int('C')] = 1; t[int('G')] = 2; t[int('T')] = 3; t[int('a')] = 4; t[int('c')] = 5; t[int('g')] = 6; t[int('t')] = 7;
        t[int('*')] = 8; t[int('#')] = 9; t[int('N')] = 10; t[int('n')] = 11; t[int('+')] = 12; t[int('-')] = 12; t[int('^')] = 13;
    }
};
constexpr CharClass kCharClassTable{};
static const uint8_t* const kCharClass = kCharClassTable.t;

// The same with class 15 for the bytes that end a field (tab, newline, anything <= 10): the fast path below stops on it
struct FieldClass {
    uint8_t t[256];
    constexpr FieldClass() : t() {
        constexpr CharClass base{};
        for (int i = 0; i < 256; ++i) t[i] = i <= 10 ? 15 : base.t[i];
    }
};
constexpr FieldClass kFieldClassTable{};
static const uint8_t* const kFieldClass = kFieldClassTable.t;

// One row the way samtools writes it - seven fields, as many quality and mapping-quality characters as read-bases, '\n' at
// *eol - in a single forward pass: no field splitting first (nine memchr calls a row were a quarter of the tokeniser's time on
// 50x rows), the base codes go to `code`, indels to `indels`.  Anything unusual (other field counts, short quality strings,
// '\r', bytes outside the printable range, an indel or '^' running into the field's end) returns false with nothing
// committed, and the caller takes the general path below, which is the one that defines the behaviour.
static inline bool fast_row(const char* cur, const char* eol, uint8_t* code, std::vector<IndelTok>& indels, int64_t* pos_out, int* nt_out,
                            const char** qs_out, const char** ms_out) {
    const char* q = cur;
    while (uint8_t(*q) > 10) ++q;                                   // contig ('\n' at *eol stops every loop at the latest)
    if (*q != '\t') return false;
    ++q;
    const char* d0 = q;
    int64_t pos = 0;
    while (uint8_t(*q - '0') < 10) { pos = pos * 10 + (*q - '0'); ++q; }
    if (q == d0 || q - d0 > 15 || *q != '\t') return false;
    ++q;
    while (uint8_t(*q) > 10) ++q;                                   // reference base
    if (*q != '\t') return false;
    ++q;
    while (uint8_t(*q) > 10) ++q;                                   // depth
    if (*q != '\t') return false;
    ++q;
    int nt = 0;
    indels.clear();
    for (;;) {
        const uint8_t cl = kFieldClass[uint8_t(*q)];
        if (cl < 12) {
            code[nt++] = cl;
            ++q;
        } else if (cl == 14) {
            ++q;
        } else if (cl == 13) {
            if (uint8_t(q[1]) <= 10) return false;
            q += 2;
        } else if (cl == 12) {
            const char sign = *q++;
            int64_t adv = 0;
            while (uint8_t(*q - '0') < 10) {
                adv = adv * 10 + (*q - '0');
                ++q;
                if (adv > (1 << 24)) return false;
            }
            if (nt == 0 || adv > eol - q) return false;
            for (int64_t k = 0; k < adv; ++k)
                if (uint8_t(q[k]) <= 10) return false;
            if (!indels.empty() && indels.back().idx == nt - 1) indels.pop_back();
            indels.push_back(IndelTok{nt - 1, sign == '+' ? 1 : 2, q, int(adv)});
            q += adv;
        } else {
            break;
        }
    }
    if (*q != '\t' || nt > kMaxDepth) return false;
    const char* qs = q + 1;
    if (eol - qs != 2 * int64_t(nt) + 1 || qs[nt] != '\t') return false;
    *pos_out = pos;
    *nt_out = nt;
    *qs_out = qs;
    *ms_out = qs + nt + 1;
    return true;
}

// Parses the rows in [text, text + len) into `p` (offsets local to p).  Thread-safe: no shared state.
int parse_rows(const char* text, size_t len, const char* ref_seq, int64_t ref_start, size_t ref_len, int max_indel_length,
               cto_pack* p, std::string* err) {
    // one read-base is >= 3 characters of a row (base, BQ, MQ), one row >= ~40: reserve once instead of growing
    pack_begin(p, len / 3 + 16, len / 40 + 16);
    std::vector<IndelTok> indels;
    indels.reserve(64);
    std::vector<uint8_t> codes(4096);
    ColumnScratch sc;
    const char* cur = text;
    const char* end = text + len;
    int64_t last_pos = -1;
    while (cur < end) {
        const char* eol = static_cast<const char*>(memchr(cur, '\n', size_t(end - cur)));
        if (!eol) eol = end;
        const char* row_end = eol;
        while (row_end > cur && (row_end[-1] == '\r' || row_end[-1] == ' ')) --row_end;
        bool done = false;
        if (eol < end && row_end > cur) {                         // '\n' at *eol: the sentinel fast_row relies on
            if (codes.size() < size_t(eol - cur)) codes.resize(size_t(eol - cur) * 2);
            int64_t pos = 0;
            int nt = 0;
            const char *qs = nullptr, *ms = nullptr;
            if (fast_row(cur, eol, codes.data(), indels, &pos, &nt, &qs, &ms)) {
                const size_t e0 = p->entries.size();
                p->entries.resize(e0 + size_t(nt));
                uint32_t* ent = p->entries.data() + e0;
                const uint8_t* code = codes.data();
                uint32_t bad = 0;
                for (int i = 0; i < nt; ++i) {                    // printable characters only: phred 0..94, no clamp needed
                    const uint32_t b = uint32_t(uint8_t(qs[i])) - 33u, m = uint32_t(uint8_t(ms[i])) - 33u;
                    bad |= (b > 94u) | (m > 94u);
                    ent[i] = uint32_t(code[i]) | (b << 6) | (m << 13);
                }
                if (bad) {
                    p->entries.resize(e0);                        // a tab, a control or an 8-bit character in a quality string
                } else {
                    if (pos <= last_pos) { set_err(err, "mpileup rows not in increasing position order"); return CTO_EINVAL; }
                    last_pos = pos;
                    const int64_t ri = pos - ref_start;
                    if (ri < 0 || size_t(ri) >= ref_len) {
                        set_err(err, "position %lld outside the supplied reference [%lld, %lld)", (long long)pos,
                                       (long long)ref_start, (long long)(ref_start + int64_t(ref_len)));
                        return CTO_EINVAL;
                    }
                    column_scratch_reset(sc);
                    int nkeys_col = 0;
                    for (const IndelTok& it : indels) {
                        uint32_t kind = 0, kid = 0;
                        const int rc = intern_indel(p, sc, &nkeys_col, int(ent[it.idx] & 15u), it.kind, it.seq, it.seqlen, ri, ref_seq, ref_len,
                                                    max_indel_length, &kind, &kid, err);
                        if (rc != CTO_OK) return rc;
                        ent[it.idx] |= (kind << 4) | (kid << 21);
                    }
                    column_end(p, pos, ri, ref_seq);
                    done = true;
                }
            }
        }
        if (!done && row_end > cur) {
            // split the first seven tab-separated fields
            const char* f[8];
            int nf = 0;
            f[nf++] = cur;
            for (const char* q = cur; nf < 8;) {                  // memchr: the base / quality fields are thousands of characters long
                q = static_cast<const char*>(memchr(q, '\t', size_t(row_end - q)));
                if (!q) break;
                f[nf++] = ++q;
            }
            if (nf < 7) {
                set_err(err, "mpileup row has %d fields, need >= 7 (is --output-MQ on?)", nf);
                return CTO_EINVAL;
            }
            auto fend = [&](int i) { return (i + 1 < nf) ? f[i + 1] - 1 : row_end; };
            int64_t pos = 0;
            for (const char* q = f[1]; q < fend(1); ++q) {
                if (*q < '0' || *q > '9') { set_err(err, "bad position field"); return CTO_EINVAL; }
                pos = pos * 10 + (*q - '0');
            }
            if (pos <= last_pos) { set_err(err, "mpileup rows not in increasing position order"); return CTO_EINVAL; }
            last_pos = pos;
            int64_t ri = pos - ref_start;
            if (ri < 0 || size_t(ri) >= ref_len) {
                set_err(err, "position %lld outside the supplied reference [%lld, %lld)", (long long)pos,
                               (long long)ref_start, (long long)(ref_start + int64_t(ref_len)));
                return CTO_EINVAL;
            }
            const char* bs = f[4];
            const char* be = fend(4);
            const char* qs = f[5];
            const int nq = int(fend(5) - f[5]);
            const char* ms = f[6];
            const int nm = int(fend(6) - f[6]);
            // ---- tokenise the base string (reference :124-144), one pass straight into pack entries ----
            // read-bases become entries (code only) as they are met; an indel attaches to the entry before it and is remembered
            // in `indels`; the quality characters are merged in afterwards in one tight loop
            const size_t e0 = p->entries.size();
            p->entries.resize(e0 + size_t(be - bs));             // a base string never yields more read-bases than characters
            uint32_t* ent = p->entries.data() + e0;
            int nt = 0;
            indels.clear();
            for (const char* q = bs; q < be;) {
                const uint8_t cl = kCharClass[uint8_t(*q)];
                if (cl < 12) {                                    // a read-base
                    ent[nt++] = cl;
                    ++q;
                } else if (cl == 12) {                            // '+' / '-': <count><sequence>
                    const char sign = *q++;
                    int64_t adv = 0;
                    while (q < be && *q >= '0' && *q <= '9') { adv = adv * 10 + (*q - '0'); ++q; }
                    if (nt == 0) { set_err(err, "indel token before any base at pos %lld", (long long)pos); return CTO_EINVAL; }
                    const int avail = int(std::min<int64_t>(adv, be - q));
                    if (!indels.empty() && indels.back().idx == nt - 1) indels.pop_back();      // a second indel on one base replaces the first
                    indels.push_back(IndelTok{nt - 1, sign == '+' ? 1 : 2, q, avail});
                    q += adv;  // the reference advances by `adv` characters in total
                } else if (cl == 13) {                            // '^' + the mapping-quality character of a read start
                    q += 2;
                } else {
                    ++q;                                          // '$' and anything else
                }
            }
            // zip(base_list, mapping_quality) / zip(base_list, base_quality) truncate: entries without a
            // quality character contribute to no counter; they are dropped here (only malformed rows).
            const int n = int(std::min<size_t>(size_t(nt), size_t(std::min(nq, nm))));
            if (n > kMaxDepth) { set_err(err, "column depth %d > %d unsupported", n, kMaxDepth); return CTO_EUNSUPPORTED; }
            for (int i = 0; i < n; ++i) {
                const int bq = std::max(0, std::min(int(qs[i]) - 33, 127)), mq = std::max(0, std::min(int(ms[i]) - 33, 255));
                ent[i] |= (uint32_t(bq) << 6) | (uint32_t(mq) << 13);
            }
            column_scratch_reset(sc);
            int nkeys_col = 0;
            for (const IndelTok& it : indels) {
                if (it.idx >= n) break;
                uint32_t kind = 0, kid = 0;
                const int rc = intern_indel(p, sc, &nkeys_col, int(ent[it.idx] & 15u), it.kind, it.seq, it.seqlen, ri, ref_seq, ref_len,
                                            max_indel_length, &kind, &kid, err);
                if (rc != CTO_OK) return rc;
                ent[it.idx] |= (kind << 4) | (kid << 21);
            }
            p->entries.resize(e0 + size_t(n));
            column_end(p, pos, ri, ref_seq);
        }
        cur = eol + 1;
    }
    return CTO_OK;
}

}  // namespace

extern "C" int cto_pack_from_mpileup(const char* text, size_t len, const char* ref_seq, int64_t ref_start,
                                      size_t ref_len, int max_indel_length, cto_pack** out) {
    return cto::guarded("cto_pack_from_mpileup", [&] {
        return cto::pack_from_mpileup_impl(text, len, ref_seq, ref_start, ref_len, max_indel_length, nullptr, 0, out);
    });
}

int cto::pack_from_mpileup_impl(const char* text, size_t len, const char* ref_seq, int64_t ref_start, size_t ref_len, int max_indel_length,
                                uint32_t* ext_entries, size_t ext_cap, cto_pack** out) {
    CTO_REQUIRE(text && ref_seq && out, CTO_EINVAL, "cto_pack_from_mpileup: null argument");
    // rows are independent: split the text at line boundaries and tokenise the pieces on several host threads
    unsigned nt = std::thread::hardware_concurrency();
    nt = std::max(1u, std::min(nt, 32u));     // scales to ~3.9 GB/s of text at 32 threads (tools/tokenise_bench.py)
    nt = cto::pack_threads_or(nt);
    if (len < (size_t(1) << 22)) nt = 1;
    std::vector<size_t> cut(nt + 1, len);
    cut[0] = 0;
    for (unsigned t = 1; t < nt; ++t) {
        size_t pos = len / nt * t;
        const char* nl = static_cast<const char*>(memchr(text + pos, '\n', len - pos));
        cut[t] = nl ? size_t(nl - text) + 1 : len;
    }
    std::vector<std::unique_ptr<cto_pack>> parts(nt);
    std::vector<int> rcs(nt, CTO_OK);
    std::vector<std::string> errs(nt);
    auto work = [&](unsigned t) {
        rcs[t] = cto::guarded("cto_pack_from_mpileup", [&] {
            parts[t].reset(new cto_pack());
            return parse_rows(text + cut[t], cut[t + 1] - cut[t], ref_seq, ref_start, ref_len, max_indel_length, parts[t].get(), &errs[t]);
        });
        if (rcs[t] != CTO_OK && errs[t].empty()) errs[t] = cto_last_error();      // what guarded() caught, from this thread
    };
    const bool timing = getenv("CTO_PACK_TIMING") != nullptr;
    const auto T0 = std::chrono::steady_clock::now();
    if (nt == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t) th.emplace_back(work, t);
        for (auto& x : th) x.join();
    }
    const auto T1 = std::chrono::steady_clock::now();
    for (unsigned t = 0; t < nt; ++t)
        if (rcs[t] != CTO_OK) { cto::set_error("%s", errs[t].c_str()); return rcs[t]; }
    std::unique_ptr<cto_pack> p;
    {
        std::string merr;
        p = cto::merge_parts(parts, &merr, ext_entries, ext_cap);
        if (!p) { cto::set_error("%s", merr.c_str()); return CTO_EINVAL; }
    }
    if (timing)
        fprintf(stderr, "cto_pack_from_mpileup: %u threads, parse %.1f ms, merge %.1f ms\n", nt,
                std::chrono::duration<double, std::milli>(T1 - T0).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T1).count());
    *out = p.release();
    return CTO_OK;
}

extern "C" int cto_pack_from_arrays(const cto_pack_view* v, const int64_t* key_str_off, const char* key_str, cto_pack** out) try {
    CTO_REQUIRE(v && out, CTO_EINVAL, "cto_pack_from_arrays: null argument");
    CTO_REQUIRE(v->n_cols >= 0 && v->n_entries >= 0 && v->n_keys >= 0, CTO_EINVAL, "negative sizes");
    auto* p = new cto_pack();
    p->col_pos.assign(v->col_pos, v->col_pos + v->n_cols);
    p->col_ref.assign(v->col_ref, v->col_ref + v->n_cols);
    p->col_off.assign(v->col_off, v->col_off + v->n_cols + 1);
    p->key_off.assign(v->key_off, v->key_off + v->n_cols + 1);
    p->entries.assign(v->entries, v->entries + v->n_entries);
    p->key_meta.assign(v->key_meta, v->key_meta + v->n_keys);
    if (v->key_group) p->key_group.assign(v->key_group, v->key_group + v->n_keys);
    else p->key_group.assign(size_t(v->n_keys), 0);
    if (key_str_off && key_str) {
        p->key_str_off.assign(key_str_off, key_str_off + v->n_keys + 1);
        p->key_str.assign(key_str, size_t(key_str_off[v->n_keys]));
    } else {
        p->key_str_off.assign(size_t(v->n_keys) + 1, 0);
    }
    for (int64_t c = 0; c < v->n_cols; ++c) {
        if (p->col_off[size_t(c) + 1] - p->col_off[size_t(c)] > kMaxDepth) {
            delete p;
            cto::set_error("column depth > %d unsupported", kMaxDepth);
            return CTO_EUNSUPPORTED;
        }
        if (c > 0 && p->col_pos[size_t(c)] <= p->col_pos[size_t(c) - 1]) {
            delete p;
            cto::set_error("col_pos must be strictly increasing");
            return CTO_EINVAL;
        }
    }
    *out = p;
    return CTO_OK;
} CTO_CATCH("cto_pack_from_arrays", int)

extern "C" int cto_pack_view_of(const cto_pack* p, cto_pack_view* v) {
    CTO_REQUIRE(p && v, CTO_EINVAL, "cto_pack_view_of: null argument");
    v->n_cols = int64_t(p->col_pos.size());
    v->n_entries = int64_t(p->ext_entries ? p->ext_n : p->entries.size());
    v->n_keys = int64_t(p->key_meta.size());
    v->col_pos = p->col_pos.data();
    v->col_ref = p->col_ref.data();
    v->col_off = p->col_off.data();
    v->key_off = p->key_off.data();
    v->entries = p->ext_entries ? p->ext_entries : p->entries.data();
    v->key_meta = p->key_meta.data();
    v->key_group = p->key_group.data();
    return CTO_OK;
}

extern "C" int cto_pack_key_string(const cto_pack* p, int64_t k, const char** s) {
    CTO_REQUIRE(p && s && k >= 0 && size_t(k) + 1 < p->key_str_off.size(), CTO_EINVAL, "bad key index");
    *s = p->key_str.data() + p->key_str_off[size_t(k)];
    return int(p->key_str_off[size_t(k) + 1] - p->key_str_off[size_t(k)]);
}

extern "C" void cto_pack_free(cto_pack* p) { delete p; }

namespace {
// alt_info of one candidate column (create_tensor_pileup_calling.py:158-209); arguments as cto_alt_info, already validated
std::string alt_info_string(const cto_pack* p, int64_t col, int pass, const int16_t* cv, int32_t depth_aff,
                            const int32_t* colfirst_col, const uint32_t* keycnt, const int32_t* keyfirst) {
    cv += pass * 36;
    colfirst_col += pass * 4;
    struct Item { int64_t first; std::string key; int64_t count; };
    std::vector<Item> items;
    const int ref = p->col_ref[size_t(col)] & 3;
    static const char kB[] = "ACGT";
    // channel layout F0: A C G T at 0..3, a c g t at 9..12; the reference base's channel holds -(group sum)
    int64_t fwd[4], rev[4];
    int64_t sf = 0, sr = 0;
    for (int b = 0; b < 4; ++b) { fwd[b] = cv[b]; rev[b] = cv[9 + b]; if (b != ref) { sf += fwd[b]; sr += rev[b]; } }
    fwd[ref] = -fwd[ref] - sf;
    rev[ref] = -rev[ref] - sr;
    for (int b = 0; b < 4; ++b) {
        if (b == ref) continue;
        int64_t c = fwd[b] + rev[b];
        if (c > 0) items.push_back(Item{colfirst_col[b], std::string("X") + kB[b], c});
    }
    const int32_t k0 = p->key_off[size_t(col)], k1 = p->key_off[size_t(col) + 1];
    for (int32_t k = k0; k < k1; ++k) {
        int64_t c = pass == 0 ? (keycnt[k] & 0xffffu) : (keycnt[k] >> 16);
        if (c == 0) continue;
        std::string key(p->key_str.data() + p->key_str_off[size_t(k)], size_t(p->key_str_off[size_t(k) + 1] - p->key_str_off[size_t(k)]));
        bool merged = false;
        for (auto& it : items) {
            if (it.key == key) { it.count += c; it.first = std::min<int64_t>(it.first, keyfirst[2 * k + pass]); merged = true; break; }
        }
        if (!merged) items.push_back(Item{keyfirst[2 * k + pass], key, c});
    }
    std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.first < b.first; });
    std::string s = std::to_string(depth_aff) + "-";
    bool first = true;
    for (auto& it : items) {
        if (!first) s.push_back(' ');
        first = false;
        s += it.key; s.push_back(' '); s += std::to_string(it.count);
    }
    const int64_t refc = fwd[ref] + rev[ref];
    if (refc > 0) {
        if (!first) s.push_back(' ');
        s += std::string("R") + kB[ref] + " " + std::to_string(refc);
    }
    s.push_back('-');
    return s;
}
}  // namespace

extern "C" int cto_alt_info(const cto_pack* p, int64_t col, int pass, const int16_t* cv, int32_t depth_aff,
                            const int32_t* colfirst_col, const uint32_t* keycnt, const int32_t* keyfirst,
                            char* buf, size_t cap) try {
    CTO_REQUIRE(p && cv && colfirst_col && buf && cap > 0, CTO_EINVAL, "cto_alt_info: null argument");
    CTO_REQUIRE(col >= 0 && size_t(col) < p->col_pos.size(), CTO_EINVAL, "cto_alt_info: column out of range");
    CTO_REQUIRE(pass == 0 || pass == 1, CTO_EINVAL, "cto_alt_info: pass must be 0 (AFF) or 1 (NEG)");
    const std::string s = alt_info_string(p, col, pass, cv, depth_aff, colfirst_col, keycnt, keyfirst);
    CTO_REQUIRE(s.size() + 1 <= cap, CTO_EINVAL, "cto_alt_info: buffer too small (%zu needed)", s.size() + 1);
    memcpy(buf, s.c_str(), s.size() + 1);
    return int(s.size());
} CTO_CATCH("cto_alt_info", int)

// The rows of `<ctg>.<chunk>_hybrid_info` (extract_candidates_calling.py:352-354, 490-497) from cto_hybrid_info's records: per listed
// position with a row,  ctg \t pos \t REF \t <depth>-<allele count allele count ...>  - the alleles of pileup_dict (:108-126: a base counts
// its indel carriers too; insertions `I<ANCHOR><SEQ>` and deletions `D` + one N per deleted base when indel candidates are selected, bare
// `I` / `D` otherwise) in decreasing count, ties in the order the dictionary met them (a read's base before its indel).  For a row that
// passes the AF gates the reference prints, in place of each count, str(round(count / depth, 3)) - decode_pileup_bases returns its
// pileup_list re-made as fractions on that path (:149-153) and :353 prints whatever came back; reproduced: a correctly rounded "%.3f"
// without its trailing zeros is Python's str() of the rounded float.
extern "C" int64_t cto_hybrid_info_rows(const cto_pack* p, const char* ctg, int64_t n_pos, const int32_t* pos, const int32_t* rec, int select_indel,
                                        const uint32_t* gcnt, const int32_t* gfirst, char* buf, size_t cap) try {
    CTO_REQUIRE(p && ctg && n_pos >= 0 && (n_pos == 0 || (pos && rec)) && (buf || cap == 0), CTO_EINVAL, "cto_hybrid_info_rows: null argument");
    struct Item { int64_t count, when; std::string key; };
    std::string out;
    static const char kB[] = "ACGT";
    for (int64_t i = 0; i < n_pos; ++i) {
        const int32_t* r = rec + i * 16;
        const int64_t col = r[0];
        if (col < 0) continue;
        CTO_REQUIRE(size_t(col) < p->col_pos.size(), CTO_EINVAL, "cto_hybrid_info_rows: column out of range");
        std::vector<Item> items;
        for (int b = 0; b < 4; ++b)
            if (r[2 + b] > 0) items.push_back(Item{r[2 + b], int64_t(r[8 + b]) * 2, std::string(1, kB[b])});
        if (!select_indel) {
            if (r[6] > 0) items.push_back(Item{r[6], int64_t(r[12]) * 2 + 1, "I"});
            if (r[7] > 0) items.push_back(Item{r[7], int64_t(r[13]) * 2 + 1, "D"});
        } else {
            CTO_REQUIRE(gcnt && gfirst, CTO_EINVAL, "cto_hybrid_info_rows: the per-allele counts are missing");
            const int32_t k0 = p->key_off[size_t(col)], k1 = p->key_off[size_t(col) + 1];
            for (int32_t g = 0; g < k1 - k0; ++g) {
                if (gcnt[k0 + g] == 0) continue;
                int32_t k = k0;
                while (k < k1 && p->key_group[size_t(k)] != g) ++k;            // any key of the merged allele spells it
                CTO_REQUIRE(k < k1, CTO_EINVAL, "cto_hybrid_info_rows: allele group without a key");
                const char* ks = p->key_str.data() + p->key_str_off[size_t(k)];
                const size_t kl = size_t(p->key_str_off[size_t(k) + 1] - p->key_str_off[size_t(k)]);
                std::string key;
                if ((p->key_meta[size_t(k)] & 3) == 1) { key.assign(ks, kl); for (auto& ch : key) ch = cto::up(ch); }
                else {
                    // a deletion's key string is 'D' + the reference slice anchor .. last deleted base, cut at max_indel_length: its length
                    // gives the deleted length unless the deletion is over-long (key_meta bit 3) - then the pack does not say how long
                    CTO_REQUIRE(!(p->key_meta[size_t(k)] & 8), CTO_EUNSUPPORTED,
                                "cto_hybrid_info_rows: %s:%d carries a deletion longer than max_indel_length; its per-allele row cannot be printed",
                                ctg, int(pos[i]));
                    key = "D" + std::string(kl >= 2 ? kl - 2 : 0, 'N');
                }
                items.push_back(Item{int64_t(gcnt[k0 + g]), int64_t(gfirst[k0 + g]) * 2 + 1, key});
            }
        }
        std::sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.count != b.count ? a.count > b.count : a.when < b.when; });
        out += ctg; out.push_back('\t'); out += std::to_string(pos[i]); out.push_back('\t');
        out.push_back(kB[p->col_ref[size_t(col)] & 3]); out.push_back('\t');
        out += std::to_string(r[1]); out.push_back('-');
        const bool as_fraction = (r[15] & 4) != 0;
        const double den = r[1] > 0 ? double(r[1]) : 1.0;
        for (size_t j = 0; j < items.size(); ++j) {
            if (j) out.push_back(' ');
            out += items[j].key; out.push_back(' ');
            if (!as_fraction) { out += std::to_string(items[j].count); continue; }
            char num[64];
            int n = snprintf(num, sizeof num, "%.3f", double(items[j].count) / den);
            while (n > 0 && num[n - 1] == '0' && num[n - 2] != '.') --n;
            out.append(num, size_t(n));
        }
        out.push_back('\n');
    }
    if (out.size() <= cap && !out.empty()) memcpy(buf, out.data(), out.size());
    return int64_t(out.size());            // larger than cap: nothing was written, call again with that much room
} CTO_CATCH("cto_hybrid_info_rows", int64_t)

namespace {
// colvec: the column vectors of the whole pack (row = column index) or, with per_site, one row per candidate (row = site index)
int64_t alt_info_batch_impl(const cto_pack* p, int64_t n_sites, const int32_t* site_info, int pass, const int16_t* colvec, bool per_site,
                            const int32_t* sitefirst, const uint32_t* keycnt, const int32_t* keyfirst, char* buf, size_t cap,
                            int64_t* offsets) {
    CTO_REQUIRE(p && site_info && colvec && sitefirst && buf && offsets && n_sites >= 0, CTO_EINVAL, "cto_alt_info_batch: null argument");
    CTO_REQUIRE(pass == 0 || pass == 1, CTO_EINVAL, "cto_alt_info_batch: pass must be 0 (AFF) or 1 (NEG)");
    size_t used = 0;
    offsets[0] = 0;
    for (int64_t i = 0; i < n_sites; ++i) {
        const int64_t col = site_info[i * 12];
        if (col >= 0) {
            CTO_REQUIRE(size_t(col) < p->col_pos.size(), CTO_EINVAL, "cto_alt_info_batch: column out of range");
            const std::string s = alt_info_string(p, col, pass, colvec + (per_site ? i : col) * CTO_COLVEC_STRIDE,
                                                  site_info[i * 12 + 1 + pass], sitefirst + i * 8, keycnt, keyfirst);
            CTO_REQUIRE(used + s.size() <= cap, CTO_EINVAL, "cto_alt_info_batch: buffer too small");
            memcpy(buf + used, s.data(), s.size());
            used += s.size();
        }
        offsets[i + 1] = int64_t(used);
    }
    return int64_t(used);
}
}  // namespace

extern "C" int64_t cto_alt_info_batch(const cto_pack* p, int64_t n_sites, const int32_t* site_info, int pass, const int16_t* colvec,
                                      const int32_t* sitefirst, const uint32_t* keycnt, const int32_t* keyfirst, char* buf,
                                      size_t cap, int64_t* offsets) try {
    return alt_info_batch_impl(p, n_sites, site_info, pass, colvec, false, sitefirst, keycnt, keyfirst, buf, cap, offsets);
} CTO_CATCH("cto_alt_info_batch", int64_t)

extern "C" int64_t cto_alt_info_batch_sites(const cto_pack* p, int64_t n_sites, const int32_t* site_info, int pass,
                                            const int16_t* site_colvec, const int32_t* sitefirst, const uint32_t* keycnt,
                                            const int32_t* keyfirst, char* buf, size_t cap, int64_t* offsets) try {
    return alt_info_batch_impl(p, n_sites, site_info, pass, site_colvec, true, sitefirst, keycnt, keyfirst, buf, cap, offsets);
} CTO_CATCH("cto_alt_info_batch_sites", int64_t)
