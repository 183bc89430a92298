// Window consensus by de Bruijn graph (SURVEY.md 8f #4b): the candidate haplotypes `reads_realignment` hands to the realigner
// (src/realign_reads.py:519-543 -> src/realign/debruijn_graph.cpp:208-232 `Build`, :387-428 `Prune`, :288-318 `CandidatePaths`).
//
// PARITY UNPINNED: the reference's implementation needs Boost.Graph, which this image does not have, so it cannot be compiled
// into oracle/_ref here.  This file is written from the reference's contract and is held to hand-derived vectors and properties
// (tests/test_realign.py): the result is the *sorted* list of haplotype strings, so vertex / edge iteration order - which in the
// reference is the address order of heap nodes - cannot show, with one exception that no implementation can reproduce: the
// "more than 256 partial paths -> give up" cut-off (:296-299) is evaluated before every pop of a breadth-first queue whose
// order within a level follows those addresses.  The count is monotone, so only graphs that end within a handful of paths of
// 256 can differ.
//
// Contract restated:
//   k from the smallest k in [10, min(101, |ref|-1)] for which the reference window has no repeated k-mer; the first k whose graph
//   (reference + reads) is acyclic is used, none -> no haplotypes.
//   Graph: vertices = k-mers; an edge per adjacent k-mer pair, weight = number of times seen, `is_ref` if the reference has it.
//   Reads contribute only stretches free of non-ACGT bases and of low-quality positions (BQ < 15, the caller's list).
//   Prune: drop non-reference edges seen once; keep vertices reachable from the source (first reference k-mer) and reaching the
//   sink (last reference k-mer).  Paths: all source -> sink / dead-end walks, breadth first, at most 256 open + closed.
#include <algorithm>
#include <cstring>
#include <deque>
#include <string>
#include <unordered_map>
#include <vector>
#include "common.h"

namespace {

struct Edge { int to; int weight; bool is_ref; };

struct Graph {
    int k;
    std::vector<std::string> kmers;                       // vertex id -> k-mer
    std::unordered_map<std::string, int> id_of;
    std::vector<std::vector<Edge>> out;
    int source = -1, sink = -1;

    int vertex(const std::string& s) {
        auto it = id_of.find(s);
        if (it != id_of.end()) return it->second;
        const int v = (int)kmers.size();
        id_of.emplace(s, v);
        kmers.push_back(s);
        out.emplace_back();
        return v;
    }
    void edge(int a, int b, bool is_ref) {
        for (Edge& e : out[a]) if (e.to == b) { ++e.weight; e.is_ref |= is_ref; return; }
        out[a].push_back({b, 1, is_ref});
    }
    // AddKmersAndEdges (:246-256): k-mers starting at start..end of `bases`
    void add_run(const std::string& bases, int start, int end, bool is_ref) {
        if (end <= 0) return;
        int prev = vertex(bases.substr(start, k));
        for (int i = start + 1; i <= end; ++i) {
            const int cur = vertex(bases.substr(i, k));
            edge(prev, cur, is_ref);
            prev = cur;
        }
    }
    void add_read(const std::string& read, const int32_t* bad, int64_t n_bad) {    // AddEdgesForRead (:263-286)
        const int len = (int)read.size(), stop = len - k;
        std::vector<uint8_t> is_bad(len, 0);
        for (int i = 0; i < len; ++i) { const char c = read[i]; is_bad[i] = !(c == 'A' || c == 'C' || c == 'G' || c == 'T'); }
        for (int64_t j = 0; j < n_bad; ++j) if (bad[j] >= 0 && bad[j] < len) is_bad[bad[j]] = 1;
        int i = 0;
        while (i < stop) {
            int nb = i;
            while (nb < len && !is_bad[nb]) ++nb;
            add_run(read, i, nb - k, false);
            i = nb + 1;
        }
    }
    bool has_cycle() const {                              // iterative three-colour DFS over every vertex (:159-165)
        const int n = (int)kmers.size();
        std::vector<uint8_t> colour(n, 0);
        std::vector<std::pair<int, size_t>> stack;
        for (int r = 0; r < n; ++r) {
            if (colour[r]) continue;
            colour[r] = 1;
            stack.push_back({r, 0});
            while (!stack.empty()) {
                auto& top = stack.back();
                if (top.second < out[top.first].size()) {
                    const int w = out[top.first][top.second++].to;
                    if (colour[w] == 1) return true;
                    if (colour[w] == 0) { colour[w] = 1; stack.push_back({w, 0}); }
                } else { colour[top.first] = 2; stack.pop_back(); }
            }
        }
        return false;
    }
    void prune() {                                        // Prune (:387-428)
        const int n = (int)kmers.size();
        for (auto& es : out) es.erase(std::remove_if(es.begin(), es.end(), [](const Edge& e) { return !e.is_ref && e.weight < 2; }), es.end());
        std::vector<std::vector<int>> in(n);
        for (int v = 0; v < n; ++v) for (const Edge& e : out[v]) in[e.to].push_back(v);
        auto reach = [&](int root, bool forward) {
            std::vector<uint8_t> seen(n, 0);
            std::vector<int> todo{root};
            seen[root] = 1;
            while (!todo.empty()) {
                const int v = todo.back();
                todo.pop_back();
                if (forward) { for (const Edge& e : out[v]) if (!seen[e.to]) { seen[e.to] = 1; todo.push_back(e.to); } }
                else { for (int w : in[v]) if (!seen[w]) { seen[w] = 1; todo.push_back(w); } }
            }
            return seen;
        };
        const std::vector<uint8_t> fwd = reach(source, true), bwd = reach(sink, false);
        for (int v = 0; v < n; ++v) {
            if (fwd[v] && bwd[v]) { auto& es = out[v]; es.erase(std::remove_if(es.begin(), es.end(), [&](const Edge& e) { return !(fwd[e.to] && bwd[e.to]); }), es.end()); }
            else out[v].clear();
        }
    }
    // CandidatePaths (:288-318) + HaplotypeForPath (:320-329).  false = the 256-path cut-off hit (no haplotypes at all)
    bool paths(std::vector<std::string>& haps) const {
        struct Path { std::string bases; int last; };       // first base of every k-mer on the way
        std::deque<Path> open;
        std::vector<Path> closed;
        open.push_back({std::string(1, kmers[source][0]), source});
        while (!open.empty()) {
            if (closed.size() + open.size() > 256) return false;
            Path p = std::move(open.front());
            open.pop_front();
            for (const Edge& e : out[p.last]) {
                Path q{p.bases + kmers[e.to][0], e.to};
                if (e.to == sink || out[e.to].empty()) closed.push_back(std::move(q)); else open.push_back(std::move(q));
            }
        }
        for (const Path& p : closed) haps.push_back(p.bases + kmers[p.last].substr(1));
        return true;
    }
};

int min_k_without_repeat(const std::string& ref, int max_k) {   // KMinMaxFromReference (:182-206)
    for (int k = 10; k <= max_k; ++k) {
        std::unordered_map<std::string, int> seen;
        bool repeat = false;
        for (size_t i = 0; i + k <= ref.size(); ++i)
            if (!seen.emplace(ref.substr(i, k), 1).second) { repeat = true; break; }
        if (!repeat) return k;
    }
    return -1;
}

std::vector<std::string> consensus(const std::string& ref, const std::vector<std::string>& reads, const int32_t* lowbq, const int64_t* lowbq_off) {
    std::vector<std::string> haps;
    const int max_k = std::min(101, (int)ref.size() - 1);
    const int min_k = min_k_without_repeat(ref, max_k);
    if (min_k < 0) return haps;
    for (int k = min_k; k <= max_k; ++k) {
        Graph g;
        g.k = k;
        g.add_run(ref, 0, (int)ref.size() - k, true);
        g.source = g.id_of.at(ref.substr(0, k));
        g.sink = g.id_of.at(ref.substr(ref.size() - k, k));
        for (size_t r = 0; r < reads.size(); ++r)
            g.add_read(reads[r], lowbq ? lowbq + lowbq_off[r] : nullptr, lowbq ? lowbq_off[r + 1] - lowbq_off[r] : 0);
        if (g.has_cycle()) continue;
        g.prune();
        if (!g.paths(haps)) haps.clear();
        std::sort(haps.begin(), haps.end());
        return haps;
    }
    return haps;
}

}  // namespace

// haplotypes come back '\0'-separated in buf (sorted); returns their number, or a negative error code
extern "C" int cto_dbg_consensus(const char* ref, int n_reads, const char* const* reads, const int32_t* lowbq, const int64_t* lowbq_off,
                                 char* buf, size_t cap, size_t* used) try {
    CTO_REQUIRE(ref && n_reads >= 0 && (n_reads == 0 || reads) && (buf || cap == 0) && (!lowbq || lowbq_off), CTO_EINVAL, "cto_dbg_consensus: bad argument");
    std::vector<std::string> rs(n_reads);
    for (int i = 0; i < n_reads; ++i) rs[i] = reads[i];
    const std::vector<std::string> haps = consensus(ref, rs, lowbq, lowbq_off);
    size_t need = 0;
    for (const std::string& h : haps) need += h.size() + 1;
    if (used) *used = need;
    CTO_REQUIRE(need <= cap, CTO_ENOMEM, "cto_dbg_consensus: buffer too small (%zu bytes needed)", need);
    size_t at = 0;
    for (const std::string& h : haps) { memcpy(buf + at, h.c_str(), h.size() + 1); at += h.size() + 1; }
    return (int)haps.size();
}
CTO_CATCH("cto_dbg_consensus", int)
