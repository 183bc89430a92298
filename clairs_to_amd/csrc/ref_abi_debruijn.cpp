// The reference's de Bruijn consensus ABI, symbol for symbol, on top of libclairsto_amd.so - so that
//   dbg = ctypes.cdll.LoadLibrary(dbg_mod)                                                    (src/realign_reads.py:71)
//   dbg.get_consensus(c_ref, ",".join(reads), ",".join(" ".join(low-BQ positions)), n) -> POINTER(DBGPointer)   (:519-539)
// work unchanged when `dbg_mod` points at clairs_to_amd/realign/debruijn_graph.so.  Layout of the result = `struct_str_arr` of
// src/realign/debruijn_graph.h:41-45: int consensus_size; char* consensus[500] (the caller declares the first 200, :80-83).
// The parsing of the two joined strings follows src/realign/debruijn_graph.cpp:432-459 (split on ',', integers by `>>`).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/clairsto_amd.h"

namespace { constexpr int kMaxConsensus = 500; }
struct cto_ref_dbg_out { int consensus_size; char* consensus[kMaxConsensus]; };

static std::vector<std::string> split_commas(const char* s) {
    std::vector<std::string> v(1);
    for (; *s; ++s) { if (*s == ',') v.emplace_back(); else v.back() += *s; }
    return v;
}

extern "C" cto_ref_dbg_out* get_consensus(char* reference, char* c_reads, char* c_base_quality, int /*read_size*/) {
    cto_ref_dbg_out* out = static_cast<cto_ref_dbg_out*>(calloc(1, sizeof(cto_ref_dbg_out)));
    if (!out) return out;
    const std::vector<std::string> reads = split_commas(c_reads), bq = split_commas(c_base_quality);
    std::vector<const char*> rp;
    for (const std::string& r : reads) rp.push_back(r.c_str());
    std::vector<int32_t> low;
    std::vector<int64_t> off(reads.size() + 1, 0);
    for (size_t i = 0; i < reads.size(); ++i) {
        if (i < bq.size()) {
            const char* p = bq[i].c_str();
            for (;;) {                                   // `while (ss >> temp)`: stops at the first token that is not an integer
                char* e;
                const long v = strtol(p, &e, 10);
                if (e == p) break;
                low.push_back((int32_t)v);
                p = e;
            }
        }
        off[i + 1] = (int64_t)low.size();
    }
    size_t need = 0;
    int n = cto_dbg_consensus(reference, (int)rp.size(), rp.data(), low.data(), off.data(), nullptr, 0, &need);
    std::vector<char> buf(need + 1);
    n = cto_dbg_consensus(reference, (int)rp.size(), rp.data(), low.data(), off.data(), buf.data(), buf.size(), &need);
    if (n < 0) { fprintf(stderr, "[clairs_to_amd] get_consensus: %s\n", cto_last_error()); return out; }
    const char* p = buf.data();
    for (int i = 0; i < n && i < kMaxConsensus; ++i) { out->consensus[i] = strdup(p); p += strlen(p) + 1; }
    if (const char* log = getenv("CTO_DBG_LOG")) {       // fixture aid (tests/golden/gen_realign.py): what went in and what came out
        if (FILE* f = fopen(log, "a")) {
            fprintf(f, "%s\t", reference);
            for (int i = 0; i < n && i < kMaxConsensus; ++i) fprintf(f, "%s%s", i ? "," : "", out->consensus[i]);
            fputc('\n', f);
            fclose(f);
        }
    }
    out->consensus_size = n < kMaxConsensus ? n : kMaxConsensus;
    return out;
}

extern "C" void free_memory(cto_ref_dbg_out* p, int size) {
    if (!p) return;
    for (int i = 0; i < size && i < kMaxConsensus; ++i) free(p->consensus[i]);
    free(p);
}
