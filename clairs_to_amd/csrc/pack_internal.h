// Internals shared by the two pack producers (mpileup text: pack.cpp, BAM: bam.cpp): the pack object and the per-column
// appender that turns a column's read-bases into entries + distinct indel keys.
#pragma once
#include <algorithm>
#include <memory>
#include <new>
#include <exception>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include "common.h"

// resize() without the zero fill: the entry array (tens of MB per chunk) is always overwritten right after it is grown - by
// the tokeniser's single pass and by the threads of the parallel merge
template <class T>
struct default_init_alloc : std::allocator<T> {
    template <class U> struct rebind { using other = default_init_alloc<U>; };
    template <class U, class... A>
    void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new (static_cast<void*>(p)) U;
        else ::new (static_cast<void*>(p)) U(std::forward<A>(a)...);
    }
};

struct cto_pack {
    std::vector<int32_t> col_pos;
    std::vector<uint8_t> col_ref;
    std::vector<int64_t> col_off;   // n_cols + 1
    std::vector<int32_t> key_off;   // n_cols + 1
    std::vector<uint32_t, default_init_alloc<uint32_t>> entries;
    // Entry array in memory the caller owns (the chunk pipeline's page-locked staging buffer): the merge of the tokeniser's
    // per-thread parts writes there directly and `entries` stays empty.  The caller keeps it alive as long as the pack.
    uint32_t* ext_entries = nullptr;
    size_t ext_n = 0;
    std::vector<uint8_t> key_meta;
    std::vector<int32_t> key_group;
    std::vector<int64_t> key_str_off;  // n_keys + 1
    std::string key_str;               // alt_info keys ("I<ANCHOR><SEQ>", "D<refslice>")
};


namespace cto {

inline char up(char c) { return (c >= 'a' && c <= 'z') ? char(c - 32) : c; }

inline int base_code(char c) {
    switch (c) {
        case 'A': return 0;  case 'C': return 1;  case 'G': return 2;  case 'T': return 3;
        case 'a': return 4;  case 'c': return 5;  case 'g': return 6;  case 't': return 7;
        case '*': return 8;  case '#': return 9;  case 'N': return 10; case 'n': return 11;
        default:  return -1;
    }
}

// evc_base_from(...).upper(): ACGT (any case) stay, everything else becomes 'A'.
inline int ref_code_of(char c) {
    switch (up(c)) {
        case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3;
        default:  return 0;
    }
}

struct Tok {
    int code;          // base code
    int kind;          // 0 none, 1 ins, 2 del
    const char* seq;   // indel sequence as mpileup prints it (not owned): inserted bases in the strand's case, N/n for deletions
    int seqlen;
    int bq, mq;        // phred values
};

constexpr int kMaxDepth = 32767;
constexpr int kMaxKeysPerCol = 2048;

// Distinct indel keys of the column being built.  A column has a handful of them, so the look-up is a scan over 32-bit hashes
// (a hit is confirmed on the characters); the sequences stay where the producer has them (mpileup text / BAM record) for the
// duration of the column.  (std::string keys in two hash maps cost ~150 ns per indel-carrying read-base - a third of the
// tokeniser's time on 50x long-read rows.)
struct ColumnScratch {
    struct Key   { const char* seq; int len; uint8_t code, kind; };     // Counter key: base character + sign + sequence, case-sensitive
    struct Group { const char* seq; int len; char anchor; uint8_t kind; };   // merged allele: anchor + upper-cased sequence / deletion length
    std::vector<uint32_t> key_hash, group_hash;
    std::vector<Key> keys;
    std::vector<Group> groups;
};

// Decoding threads of the pack producers for calls made from THIS thread (0: CTO_PACK_THREADS, else the producer's default): the chunk
// pipeline sets it in its producer threads, so that nobody has to edit the process environment around a call.
extern thread_local int tl_pack_threads;
inline unsigned pack_threads_or(unsigned dflt) {
    if (tl_pack_threads > 0) return unsigned(std::min(tl_pack_threads, 64));
    if (const char* e = getenv("CTO_PACK_THREADS")) return std::max(1u, std::min(unsigned(atoi(e)), 64u));
    return dflt;
}

// No C++ exception crosses the C ABI (or leaves a worker thread): allocation failures on hostile input come back as error codes.
template <class F>
int guarded(const char* what, F&& f) {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        set_error("%s: out of memory", what);
        return CTO_ENOMEM;
    } catch (const std::exception& e) {
        set_error("%s: %s", what, e.what());
        return CTO_EINVAL;
    } catch (...) {
        set_error("%s: unknown failure", what);
        return CTO_EINVAL;
    }
}

void set_err(std::string* err, const char* fmt, ...);
void pack_begin(cto_pack* p, size_t entries_hint, size_t cols_hint);
// Appends one column (position `pos`, reference index ri = pos - ref_start) made of toks[0..n)
int append_column(cto_pack* p, ColumnScratch& sc, int64_t pos, int64_t ri, const char* ref_seq, size_t ref_len,
                  int max_indel_length, const Tok* toks, int n, std::string* err);

// The same for a producer that packs the plain read-bases itself: ents[0..n) are finished entries (code | bq << 6 | mq << 13)
// except at the indices named by indels[0..n_indel), whose kind / key id are filled in here.
struct IndelAt {
    int idx;           // index into ents
    int kind;          // 1 ins, 2 del
    const char* seq;
    int seqlen;
};
inline uint32_t pack_entry(int code, int bq, int mq) { return uint32_t(code) | (uint32_t(bq) << 6) | (uint32_t(mq) << 13); }
int append_column_packed(cto_pack* p, ColumnScratch& sc, int64_t pos, int64_t ri, const char* ref_seq, size_t ref_len,
                         int max_indel_length, const uint32_t* ents, int n, const IndelAt* indels, int n_indel, std::string* err);

std::unique_ptr<cto_pack> merge_parts(std::vector<std::unique_ptr<cto_pack>>& parts, std::string* err, uint32_t* ext_entries = nullptr,
                                      size_t ext_cap = 0);
// cto_pack_from_mpileup with the entry array placed in ext_entries[0 .. ext_cap) when it fits (see cto_pack::ext_entries)
int pack_from_mpileup_impl(const char* text, size_t len, const char* ref_seq, int64_t ref_start, size_t ref_len, int max_indel_length,
                           uint32_t* ext_entries, size_t ext_cap, cto_pack** out);

// cto_extract_candidates with the caller's own overflow counters (scratch dev [n_keys] uint32): what concurrent callers on different
// streams use (csrc/pipeline.hip: one buffer per chunk slot); csrc/extract.hip
int extract_candidates_scratch(const cto_pack_view* dp, int min_mq, int min_bq, double snv_min_af, double indel_min_af, double min_coverage,
                               int alt_base_num, int select_indel, uint32_t* scratch, uint8_t* flags, int32_t* depth, void* stream);
}  // namespace cto
