// Stages of one realignment window (csrc/realign.cpp), so that the batch driver (csrc/realign_batch.hip) can run the
// data-parallel ones on the device for every window of a run at once and leave the rest where it is:
//   1. fast pass       k-mer seeded, <= 2 mismatches            host: Window::fast_pass_host   device: k_fast_pass -> set_fast_pass
//   2. striped passes  forward + backward Smith-Waterman ends   host: Window::ends_host        device: k_sw        -> set_ends
//      for every haplotype against the reference and every read no haplotype took against every live haplotype
//   3. tracebacks      banded, between those ends               host: inside finish            device: plan_tracebacks -> k_banded -> set_traced
//   4. finish          haplotype order, picks, read -> reference composition (host; strings); runs the tracebacks it finds missing
// The reference runs all of it per window inside realign_reads(...) (src/realign/realigner.cpp:782-857).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace cto_realign {

// What the two striped passes of ssw_align (ssw.c:781-867) leave behind; score <= 0: nothing aligned.
struct Ends { int32_t score, ref_end, read_end, ref_begin, bw_read_end, lanes; };

struct Op { char op; int len; };
struct ReadHit { int position = -1, score = 0; bool exact = false; std::vector<Op> cigar; };   // ReadAlignment (realigner.h:103-127); exact: the fast pass's hit, CIGAR '=' x read length (not stored)
struct HapState {
    int index = 0, score = 0, ref_pos = 0;
    bool is_reference = false;
    std::vector<ReadHit> hits;
    std::vector<Op> cigar;
    std::vector<int> pos_map;
};

struct SwPair { const int8_t* ref; int R; const int8_t* query; int Q; };
struct SwAlignment { int score = 0, ref_begin = 0; std::vector<Op> cigar; };     // what ssw's Align returns of an alignment (ssw_cpp.cpp:78-215)
// The banded traceback of pair `pair` (ssw.c:531-741, as ssw_align calls it, :831-867): the sub-problem between the end points the
// striped passes found - ref[ref_begin .. +subR), query[read_begin .. +subQ), the score to reach, the first band.
struct TraceJob { int pair, ref_begin, read_begin, subR, subQ, score, band; };
constexpr int kTraceM = 0, kTraceI = 1, kTraceD = 2;                               // a run = len << 2 | op, in query order

class Window {
public:
    // returns CTO_OK or an error code (message set)
    int init(int n_reads, const char* const* seqs, const int32_t* positions, const char* const* cigars, const char* reference,
             const char* haplotypes, int ref_start, int ref_prefix, int ref_suffix);
    int n_reads() const { return int(reads.size()); }
    int n_haps() const { return int(haps.size()); }
    // ---- stage 1
    void fast_pass_host();
    // device results for this window: hit_score / hit_pos [H][n] (position -1 = no hit), hap_score [H]
    void set_fast_pass(const int32_t* hit_score, const int32_t* hit_pos, const int32_t* hap_score);
    // ---- stage 2: the pairs in the order finish() consumes them: H x (reference, haplotype), then todo reads x live haplotypes
    void collect_pairs();
    const std::vector<SwPair>& sw_pairs() const { return pairs; }
    void ends_host();
    void set_ends(const Ends* e) { ends.assign(e, e + pairs.size()); }
    // ---- stage 3
    // the tracebacks finish() is going to ask for if every one of them confirms its score (a prediction: finish() computes what it
    // finds missing itself): every haplotype against the reference, and per unplaced read the pair it would pick
    void plan_tracebacks(std::vector<TraceJob>& jobs) const;
    // a traceback done elsewhere (the device): ok = the banded pass reached the score and the walk stayed inside the band
    void set_traced(const TraceJob& job, bool ok, const int32_t* runs, int n_runs);
    int finish(int32_t* out_pos, std::vector<std::string>& out_cigar);

    std::vector<std::string> reads, haps;
    std::string reference;
    std::vector<int32_t> positions;
    std::vector<std::string> cigars;
    int ref_start = 0, ref_prefix = 0, ref_suffix = 0;
    std::vector<HapState> hs;
    std::vector<std::vector<int8_t>> hapc, readc;      // SSW base codes (readc only for the reads that need Smith-Waterman)
    std::vector<int8_t> refc;
    std::vector<int> todo;
    std::vector<SwPair> pairs;
    std::vector<Ends> ends;
    std::vector<int> traced_at;                        // pair -> index into traced, -1 = not done elsewhere
    std::vector<SwAlignment> traced;
};

int get_threads();
// the two striped passes of one alignment on the host (ssw.c:781-830): what Window::ends_host runs per pair
Ends ends_of_pair(const int8_t* ref, int R, const int8_t* query, int Q);
// one alignment outside a window (cto_ssw_align_batch): the traceback's sub-problem (false: the reference returns no alignment), the
// alignment with the traceback run here, and the alignment from runs delivered in set_traced's format
bool plan_pair(int R, int Q, const Ends& e, TraceJob& j);
SwAlignment alignment_of_pair(const int8_t* ref, int R, const int8_t* query, int Q, const Ends& e);
SwAlignment alignment_from_device_runs(const int8_t* ref, const int8_t* query, int Q, const Ends& e, const TraceJob& j, const int32_t* runs, int n_runs);
// a planned traceback on the host, in set_traced's format (what the device stage delivers); false = the banded pass fails
bool trace_runs_host(const Window& w, const TraceJob& job, std::vector<int32_t>& runs);

}  // namespace cto_realign

// cigars of one window into the caller's buffer: cigar_off[i] .. NUL-terminated, cigar_off[n] = bytes used (as cto_realign_reads)
int cto_realign_write_cigars(const std::vector<std::string>& out, char* cigar_buf, size_t cigar_cap, int64_t* cigar_off);
