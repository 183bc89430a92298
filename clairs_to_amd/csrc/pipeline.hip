// Native chunk pipeline: candidate chunk files + pileup source (mpileup text or BAM) -> p_<chunk>.vcf, the whole of what
// pileup_call.prepare_chunk / launch_chunk / finish_chunk do per chunk, as one C call (cto_run_chunks).
//
//   producers (N threads)  BED -> centres, reference slice (.fai), column pack (tokeniser / BAM reader; for some BAM chunks with the
//                          BGZF blocks inflated on the device, InflateCtx), one upload out of a page-locked staging buffer on the
//                          producer's own stream into the device buffers of a free slot
//   launcher (the caller)  waits for the upload event; featurisation, both networks, posterior; the candidates' column vectors
//                          gathered on the device; one copy of everything the writers need on the copy-back stream; an event
//   writers (M threads)    alt_info strings + every VCF record (two C calls), file write; the slot goes back to the pool
//
// The Python pipeline (call_chunks.run_pipeline) runs the same stages with the same C calls on thread pools.  Moving the loop here did
// not by itself change the rate (the interpreter was not the bound); what did was what the loop can own once it is native: the
// staging buffers the tokeniser merges into, buffers and contexts kept from chunk to chunk and from call to call, waits that sleep
// instead of spinning, CU-masked streams for the device inflate (DESIGN.md section 6 has the sequence of measurements).
// Outputs are byte-identical (tests/test_gpu_cli.py).
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <spawn.h>
#include <sys/wait.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "pack_internal.h"

using namespace cto;

extern char** environ;

namespace {

// Waits for an event without occupying a core: hipEventSynchronize / hipStreamSynchronize poll the completion signal from the calling
// thread (measured: every chunk waiting for the device inflate cost a second of CPU), and the producer and writer threads that wait
// here share the host with the threads that tokenise and inflate.
hipError_t wait_event(hipEvent_t ev) {
    for (int spins = 0;; ++spins) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        if (spins >= 4) usleep(spins < 64 ? 50 : 200);
    }
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
double cpu_s() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return double(ts.tv_sec) + double(ts.tv_nsec) * 1e-9; }   // this thread's CPU time

struct DevBuf {                          // a device allocation that only grows
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return CTO_OK;
        if (p) CTO_HIP(hipFree(p));
        p = nullptr;
        cap = 0;
        const size_t want = n + n / 4 + 256;
        CTO_HIP(hipMalloc(&p, want));
        cap = want;
        return CTO_OK;
    }
    ~DevBuf() { if (p) (void)hipFree(p); }
};
struct PinBuf {                          // page-locked host memory that only grows
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return CTO_OK;
        if (p) CTO_HIP(hipHostFree(p));
        p = nullptr;
        cap = 0;
        const size_t want = n + n / 4 + 256;
        CTO_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
        return CTO_OK;
    }
    int grow_keeping(size_t n, size_t keep) {              // ensure(n) that carries the first `keep` bytes over
        if (n <= cap) return CTO_OK;
        void* q = nullptr;
        const size_t want = n + n / 4 + 256;
        CTO_HIP(hipHostMalloc(&q, want, hipHostMallocDefault));
        if (p) {
            if (keep) memcpy(q, p, keep);
            CTO_HIP(hipHostFree(p));
        }
        p = q;
        cap = want;
        return CTO_OK;
    }
    ~PinBuf() { if (p) (void)hipHostFree(p); }
};

struct Slot {
    // pack on the device
    DevBuf pack_dev;                     // the pack arrays + the candidate positions, one allocation (256-byte aligned parts)
    PinBuf stage;                        // its page-locked source
    const int32_t* d_site_pos = nullptr;
    // featurisation / network / epilogue outputs
    DevBuf colvec, coldepth, x_aff, x_neg, la, ln, post;
    DevBuf dec_l, qual_l;                // decision / QUAL of one network launch, before they are dealt out to the chunks they belong to
    DevBuf xflags, xdepth, xscratch, cand, cand_scr;    // REGION jobs: candidate gates' outputs, overflow counters, candidate positions (+ count)
    PinBuf cand_host;
    DevBuf xmode;                        // REGION jobs with a confident BED / an indel BED / a hybrid list: intervals, positions, hybrid_info records
    PinBuf xmode_host;
    DevBuf res_dev;                      // site_info | candidate column vectors | sitefirst | decision | qual | keycnt | keyfirst
    PinBuf res_host;                     // the same bytes on the host, one copy per chunk
    size_t roff[7] = {0, 0, 0, 0, 0, 0, 0};
    hipEvent_t uploaded = nullptr, begin = nullptr, computed = nullptr, done = nullptr, kernels_end = nullptr;
    size_t res_total = 0;
    // host side of the chunk
    int64_t job = -1;
    int device = 0;
    cto_pack* pack = nullptr;
    cto_pack_view hv{}, dv{};
    std::vector<int32_t> sites;
    std::string ref;
    int64_t ref_start = 0;
    cto_dev_tokeniser* tok = nullptr;    // text input with cfg.device_tokenise: the slot's tokeniser context (text staging + row tables)
    ~Slot() {
        if (tok) cto_dev_tokeniser_destroy(tok);
        if (pack) cto_pack_free(pack);
        if (uploaded) (void)hipEventDestroy(uploaded);
        if (done) (void)hipEventDestroy(done);
        if (begin) (void)hipEventDestroy(begin);
        if (computed) (void)hipEventDestroy(computed);
        if (kernels_end) (void)hipEventDestroy(kernels_end);
    }
};

// One chunk's trip through the device inflate (csrc/inflate.hip): the BGZF byte range + block table in page-locked memory, their
// device copies, the inflated blocks on both sides, and a stream CONFINED to the first `cus` compute units
// (hipExtStreamCreateWithCUMask; tools/cumask_probe.hip: N leading bits = N / 8 CUs of every XCD).  A wave-per-block inflate launch
// occupies its CUs for tens of milliseconds; unconfined, the networks' block kernels - which need a CU's whole register file -
// wait for those waves to drain, confined they run on the other CUs.
struct InflateCtx {
    cto_dev_pileup* pile = nullptr;      // reads -> columns on the device (csrc/pileup.hip), created on first use
    int device = 0, cus = 0;
    hipStream_t stream = nullptr;
    hipEvent_t landed = nullptr;         // recorded behind the copy back; the producer thread sleeps on it (wait_event)
    PinBuf h_in, h_out, h_sites;         // h_sites: the chunk's candidate positions on their way up (page-locked like every copy source)
    DevBuf d_in, d_out;
    int open(int dev, int n_cus) {
        device = dev;
        cus = n_cus;
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < n_cus && i < 256; ++i) mask[i / 32] |= 1u << (i % 32);
        CTO_HIP(hipExtStreamCreateWithCUMask(&stream, 8, mask));
        CTO_HIP(hipEventCreateWithFlags(&landed, hipEventBlockingSync | hipEventDisableTiming));
        return CTO_OK;
    }
    ~InflateCtx() {
        if (pile) cto_dev_pileup_destroy(pile);
        if (landed) (void)hipEventDestroy(landed);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

template <class T>
struct Queue {                            // unbounded MPMC queue with a closed state
    std::mutex m;
    std::condition_variable cv;
    std::deque<T> q;
    bool closed = false;
    void push(T v) { { std::lock_guard<std::mutex> g(m); q.push_back(std::move(v)); } cv.notify_one(); }
    void close() { { std::lock_guard<std::mutex> g(m); closed = true; } cv.notify_all(); }
    bool try_pop(T* out) {
        std::lock_guard<std::mutex> g(m);
        if (q.empty()) return false;
        *out = std::move(q.front());
        q.pop_front();
        return true;
    }
    bool pop_for(T* out, int ms) {         // pop() that gives up after `ms` milliseconds
        std::unique_lock<std::mutex> g(m);
        if (!cv.wait_for(g, std::chrono::milliseconds(ms), [&] { return !q.empty() || closed; })) return false;
        if (q.empty()) return false;
        *out = std::move(q.front());
        q.pop_front();
        return true;
    }
    bool pop(T* out) {
        std::unique_lock<std::mutex> g(m);
        cv.wait(g, [&] { return !q.empty() || closed; });
        if (q.empty()) return false;
        *out = std::move(q.front());
        q.pop_front();
        return true;
    }
};

struct Mapped {                           // a file's bytes: mapped read-only, or - for a *.gz path - inflated into memory (zlib)
    const char* p = nullptr;
    size_t n = 0;
    std::vector<char> owned;
    // sniff = true: look at the first two bytes instead of the name (`gzip -fdc`, which the reference's bed_tree_from pipes every BED
    // through, shared/interval_tree.py:43, inflates what is gzip and passes on what is not)
    bool open(const char* path, std::string* err, bool sniff = false) {
        const size_t pl = strlen(path);
        bool gz = pl > 3 && strcmp(path + pl - 3, ".gz") == 0;        // the reference's readers gzip.open such files
        if (sniff && !gz) {
            const int fd0 = ::open(path, O_RDONLY | O_CLOEXEC);
            if (fd0 < 0) { *err = std::string("cannot open ") + path; return false; }
            unsigned char magic[2] = {0, 0};
            const ssize_t got = ::read(fd0, magic, 2);
            ::close(fd0);
            gz = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        }
        if (gz) {
            gzFile g = gzopen(path, "rb");
            if (!g) { *err = std::string("cannot open ") + path; return false; }
            (void)gzbuffer(g, 1 << 20);
            owned.resize(size_t(1) << 22);
            size_t got = 0;
            for (;;) {
                if (got == owned.size()) owned.resize(owned.size() * 2);
                const int r = gzread(g, owned.data() + got, unsigned(std::min<size_t>(owned.size() - got, size_t(1) << 30)));
                if (r < 0) { gzclose(g); *err = std::string("cannot inflate ") + path; return false; }
                if (r == 0) break;
                got += size_t(r);
            }
            gzclose(g);
            owned.resize(got);
            p = got ? owned.data() : nullptr;
            n = got;
            return true;
        }
        const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
        if (fd < 0) { *err = std::string("cannot open ") + path; return false; }
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); *err = std::string("cannot stat ") + path; return false; }
        n = size_t(st.st_size);
        if (n) {
            void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { ::close(fd); n = 0; *err = std::string("cannot map ") + path; return false; }
            p = static_cast<const char*>(m);
        }
        ::close(fd);
        return true;
    }
    ~Mapped() { if (p && n && owned.empty()) munmap(const_cast<char*>(p), n); }
};

// stdout of `argv` (a `samtools mpileup ...` command line) into `out`; false + *err when it cannot be started or exits non-zero
// (create_tensor_pileup_calling.py:426-446 pipes the same command; subprocess.run(check=True) in the Python mirror)
bool capture_stdout(const std::vector<std::string>& argv, std::vector<char>* out, std::string* err) {
    int fds[2];
    if (pipe2(fds, O_CLOEXEC) != 0) { *err = "pipe() failed"; return false; }
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_adddup2(&fa, fds[1], 1);
    std::vector<char*> av;
    for (const std::string& a : argv) av.push_back(const_cast<char*>(a.c_str()));
    av.push_back(nullptr);
    pid_t pid = 0;
    const int rc = posix_spawnp(&pid, av[0], &fa, nullptr, av.data(), environ);
    posix_spawn_file_actions_destroy(&fa);
    ::close(fds[1]);
    if (rc != 0) { ::close(fds[0]); *err = "cannot run " + argv[0] + ": " + strerror(rc); return false; }
    out->clear();
    out->resize(size_t(1) << 22);
    size_t got = 0;
    for (;;) {
        if (got == out->size()) out->resize(out->size() * 2);
        const ssize_t r = read(fds[0], out->data() + got, out->size() - got);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) break;
        got += size_t(r);
    }
    ::close(fds[0]);
    out->resize(got);
    int status = 0;
    while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {}
    if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) {
        *err = argv[0] + " mpileup failed (exit status " + std::to_string(WIFEXITED(status) ? WEXITSTATUS(status) : -1) + ")";
        return false;
    }
    return true;
}

struct FaiRec { int64_t length = 0, offset = 0, linebases = 0, linewidth = 0; bool ok = false; };

// <fasta>.fai (or <fasta without extension>.fai): the record of contig `ctg`  (fasta.py read_region)
bool fai_lookup(const std::string& fasta, const std::string& ctg, FaiRec* rec, std::string* err) {
    std::string fai = fasta + ".fai";
    FILE* f = fopen(fai.c_str(), "r");
    if (!f) {
        const size_t dot = fasta.rfind('.');
        if (dot != std::string::npos) { fai = fasta.substr(0, dot) + ".fai"; f = fopen(fai.c_str(), "r"); }
    }
    if (!f) { *err = "[ERROR] file " + fasta + ".fai not found"; return false; }
    char line[4096];
    while (fgets(line, sizeof(line), f)) {
        char* tab = strchr(line, '\t');
        if (!tab) continue;
        if (size_t(tab - line) == ctg.size() && memcmp(line, ctg.data(), ctg.size()) == 0) {
            long long a = 0, b = 0, c = 0, d = 0;
            if (sscanf(tab + 1, "%lld\t%lld\t%lld\t%lld", &a, &b, &c, &d) == 4 && c > 0 && d > 0) {
                rec->length = a; rec->offset = b; rec->linebases = c; rec->linewidth = d; rec->ok = true;
            }
            break;
        }
    }
    fclose(f);
    if (!rec->ok) { *err = "contig " + ctg + " not in " + fai; return false; }
    return true;
}

// 1-based inclusive [start, end] of the contig, upper-cased, clipped to the contig (fasta.py read_region)
bool read_region(const Mapped& fa, const FaiRec& r, int64_t start, int64_t end, std::string* out, std::string* err) {
    out->clear();
    if (fa.n >= 2 && (unsigned char)fa.p[0] == 0x1f && (unsigned char)fa.p[1] == 0x8b) {
        *err = "[ERROR] the reference is gzip / bgzip compressed: decompress it (and re-run samtools faidx) before use";
        return false;
    }
    start = std::max<int64_t>(1, start);
    end = std::min<int64_t>(r.length, end);
    if (end < start) return true;
    const int64_t s0 = start - 1, e0 = end;
    const int64_t b0 = r.offset + (s0 / r.linebases) * r.linewidth + s0 % r.linebases;
    const int64_t b1 = r.offset + ((e0 - 1) / r.linebases) * r.linewidth + (e0 - 1) % r.linebases + 1;
    if (b0 < 0 || b1 > int64_t(fa.n) || b1 < b0) { *err = "reference index points outside the FASTA file"; return false; }
    out->resize(size_t(end - start + 1));
    char* dst = &(*out)[0];
    size_t n = 0;
    for (const char* q = fa.p + b0; q < fa.p + b1;) {                 // line by line: memchr + one pass that folds the case
        const char* nl = static_cast<const char*>(memchr(q, '\n', size_t(fa.p + b1 - q)));
        const char* e = nl ? nl : fa.p + b1;
        for (const char* c = q; c < e; ++c)
            if (*c != '\r' && n < out->size()) dst[n++] = (*c >= 'a' && *c <= 'z') ? char(*c - 32) : *c;
        q = e + 1;
    }
    out->resize(n);
    return true;
}

// BED rows of `ctg` as 0-based [begin, end) intervals, sorted and merged (what `samtools mpileup -l` restricts positions to)
void bed_intervals(const char* text, size_t len, const std::string& ctg, std::vector<int64_t>* out) {
    std::vector<std::pair<int64_t, int64_t>> iv;
    size_t i = 0;
    while (i < len) {
        const char* nl = static_cast<const char*>(memchr(text + i, '\n', len - i));
        const size_t e = nl ? size_t(nl - text) : len;
        const char* row = text + i;
        const size_t rl = e - i;
        const char* t1 = static_cast<const char*>(memchr(row, '\t', rl));
        if (t1 && size_t(t1 - row) == ctg.size() && memcmp(row, ctg.data(), ctg.size()) == 0) {
            const char* t2 = static_cast<const char*>(memchr(t1 + 1, '\t', rl - size_t(t1 + 1 - row)));
            if (t2) {
                const long long a = atoll(std::string(t1 + 1, size_t(t2 - t1 - 1)).c_str());
                const char* t3 = static_cast<const char*>(memchr(t2 + 1, '\t', rl - size_t(t2 + 1 - row)));
                const size_t l3 = t3 ? size_t(t3 - t2 - 1) : rl - size_t(t2 + 1 - row);
                const long long b = atoll(std::string(t2 + 1, l3).c_str());
                iv.emplace_back(std::max<long long>(0, a), b);
            }
        }
        i = e + 1;
    }
    std::sort(iv.begin(), iv.end());
    out->clear();
    for (const auto& p : iv) {
        if (!out->empty() && p.first <= (*out)[out->size() - 1]) (*out)[out->size() - 1] = std::max((*out)[out->size() - 1], p.second);
        else { out->push_back(p.first); out->push_back(p.second); }
    }
}

__global__ void k_gather_rows(const int16_t* __restrict__ colvec, const int32_t* __restrict__ site_info, int64_t n, int16_t* __restrict__ out) {
    const int64_t i = blockIdx.x;
    if (i >= n) return;
    const int32_t c = site_info[i * 12];
    const int64_t col = c < 0 ? 0 : c;
    if (threadIdx.x < CTO_COLVEC_STRIDE) out[i * CTO_COLVEC_STRIDE + threadIdx.x] = colvec[col * CTO_COLVEC_STRIDE + threadIdx.x];
}

// never destroyed: at process exit the HIP runtime may already be gone when static destructors run
std::mutex& slot_cache_m() { static std::mutex* m = new std::mutex(); return *m; }
std::vector<std::unique_ptr<Slot>>& slot_cache() { static auto* v = new std::vector<std::unique_ptr<Slot>>(); return *v; }
std::vector<std::unique_ptr<InflateCtx>>& inflate_cache() { static auto* v = new std::vector<std::unique_ptr<InflateCtx>>(); return *v; }
struct SlotReturn {                      // hands a finished (or failed) call's slots back to the cache
    std::vector<std::unique_ptr<Slot>>* slots;
    ~SlotReturn() {
        std::lock_guard<std::mutex> g(slot_cache_m());
        for (auto& sl : *slots) {
            if (sl->pack) { cto_pack_free(sl->pack); sl->pack = nullptr; }
            slot_cache().push_back(std::move(sl));
        }
        slots->clear();
    }
};

struct CtxReturn {
    std::vector<std::unique_ptr<InflateCtx>>* ctx;
    ~CtxReturn() {
        std::lock_guard<std::mutex> g(slot_cache_m());
        for (auto& c : *ctx) inflate_cache().push_back(std::move(c));
        ctx->clear();
    }
};

constexpr int FLANK_POS = 33, EXPAND_REF = 1000;       // shared/param.py no_of_positions, expand_reference_region
constexpr int REGION_FLANK = 17;                       // flankingBaseNum + 1: the window columns of a candidate at the edge of a region

struct Run {
    const cto_run_cfg* cfg;
    const cto_chunk_job* jobs;
    int64_t n_jobs;
    std::atomic<int64_t> next_job{0};
    std::vector<std::unique_ptr<Slot>> slots;
    Queue<Slot*> free_slots, to_launch, to_write;
    std::mutex err_m;
    std::string first_error;
    std::atomic<bool> failed{false};
    std::atomic<int64_t> candidates{0}, sites{0}, rows{0}, low_cov{0}, clamped{0};
    std::mutex stat_m;
    double produce_s = 0, finish_s = 0, pack_s = 0, upload_s = 0, device_s = 0;
    Mapped fasta;
    std::vector<std::unique_ptr<InflateCtx>> inflate_ctx;
    Queue<InflateCtx*> free_ctx;
    std::atomic<int64_t> device_inflated{0}, device_piled{0}, device_tokenised{0};
    std::mutex fai_m;
    std::map<std::string, FaiRec> fai;
    // --call_indels_only_in_these_regions: per contig the rows as sorted, merged [begin, end) intervals (bed_tree_from of the reference)
    std::map<std::string, std::vector<int64_t>> indel_regions;
    bool load_indel_regions(std::string* err) {
        if (!cfg->indel_regions_bed || !cfg->indel_regions_bed[0]) return true;
        Mapped bed;
        if (!bed.open(cfg->indel_regions_bed, err, /*sniff=*/true)) return false;
        std::map<std::string, std::vector<std::pair<int64_t, int64_t>>> rows;
        size_t i = 0;
        int64_t row_id = 0;
        while (i < bed.n) {
            ++row_id;
            const char* nl = static_cast<const char*>(memchr(bed.p + i, '\n', bed.n - i));
            const size_t e = nl ? size_t(nl - bed.p) : bed.n;
            std::string row(bed.p + i, e - i);
            i = e + 1;
            if (row.empty() || row[0] == '#') continue;
            char name[256];
            long long a = 0, b = 0;
            if (row.find_first_not_of(" \t\r") == std::string::npos) continue;
            // a row the reference cannot split into name, start, end ends its run with an exception (interval_tree.py:47-55): an
            // unreadable BED must not turn into "no regions", which would let every indel candidate through
            if (sscanf(row.c_str(), "%255s %lld %lld", name, &a, &b) != 3) {
                *err = "[ERROR] Invalid bed input in " + std::to_string(row_id) + "-th row of " + std::string(cfg->indel_regions_bed) + ": " + row.substr(0, 80);
                return false;
            }
            if (b < a || a < 0 || b < 0) { *err = "[ERROR] Invalid bed input in " + std::string(cfg->indel_regions_bed) + ": " + row; return false; }
            if (a == b) ++b;
            rows[name].emplace_back(a, b);
        }
        for (auto& kv : rows) {
            std::sort(kv.second.begin(), kv.second.end());
            std::vector<int64_t>& out = indel_regions[kv.first];
            for (const auto& p : kv.second) {
                if (!out.empty() && p.first <= out.back()) out.back() = std::max(out.back(), p.second);
                else { out.push_back(p.first); out.push_back(p.second); }
            }
        }
        return true;
    }
    bool fai_of(const std::string& ctg, FaiRec* rec, std::string* err) {
        std::lock_guard<std::mutex> g(fai_m);
        auto it = fai.find(ctg);
        if (it != fai.end()) { *rec = it->second; return true; }
        if (!fai_lookup(cfg->ref_fa, ctg, rec, err)) return false;
        fai[ctg] = *rec;
        return true;
    }

    void fail(const std::string& msg) {
        std::lock_guard<std::mutex> g(err_m);
        if (first_error.empty()) first_error = msg;
        failed = true;
    }

    // cto_pack_from_bam with the chunk's BGZF blocks inflated on the device (the sequence of include/clairsto_amd.h: chunk span ->
    // scan -> inflate -> pile-up from memory).  *done = 0 with CTO_OK: nothing to send (no whole block in the span) - the caller reads
    // the chunk on the host.
    int pack_from_bam_device(const cto_chunk_job& j, const std::string& ctg, int64_t lo, int64_t hi, const std::vector<int64_t>& iv, Slot* s,
                             InflateCtx* c, int* done) {
        const bool timing = getenv("CTO_PIPE_TIMING") != nullptr;
        const double T0 = now_s(), C0 = cpu_s();
        int64_t fb = 0, fe = 0;
        int rc = cto_bam_chunk_span(j.bam_path, nullptr, ctg.c_str(), lo, hi, &fb, &fe);
        if (rc != CTO_OK) return rc;
        const size_t nbytes = fe > fb ? size_t(fe - fb) : 0;
        if (nbytes == 0) return CTO_OK;
        const double Cs = cpu_s();
        const size_t in_al = (nbytes + CTO_BGZF_PAD + 255) / 256 * 256;
        size_t cap = nbytes / 2048 + 64;
        if ((rc = c->h_in.ensure(in_al + cap * sizeof(cto_bgzf_block))) != CTO_OK) return rc;
        {
            const int fd = ::open(j.bam_path, O_RDONLY | O_CLOEXEC);
            CTO_REQUIRE(fd >= 0, CTO_EINVAL, "cannot open %s", j.bam_path);
            size_t got = 0;
            while (got < nbytes) {
                const ssize_t r = pread(fd, static_cast<char*>(c->h_in.p) + got, nbytes - got, off_t(fb) + off_t(got));
                if (r <= 0) break;
                got += size_t(r);
            }
            ::close(fd);
            CTO_REQUIRE(got == nbytes, CTO_EINVAL, "short read from %s", j.bam_path);
        }
        const double T1 = now_s(), C1 = cpu_s();
        int64_t n = 0, out_bytes = 0;
        for (;;) {
            memset(static_cast<char*>(c->h_in.p) + nbytes, 0, in_al - nbytes);
            n = cto_bgzf_scan(static_cast<const uint8_t*>(c->h_in.p), nbytes, fb, reinterpret_cast<cto_bgzf_block*>(static_cast<char*>(c->h_in.p) + in_al),
                              int64_t(cap), &out_bytes);
            if (n != CTO_ENOMEM || cap > (size_t(1) << 24)) break;
            cap *= 8;                                              // many tiny blocks
            if ((rc = c->h_in.grow_keeping(in_al + cap * sizeof(cto_bgzf_block), nbytes)) != CTO_OK) return rc;
        }
        if (n < 0) return int(n);
        if (n == 0) return CTO_OK;
        const auto* blocks = reinterpret_cast<const cto_bgzf_block*>(static_cast<char*>(c->h_in.p) + in_al);
        const size_t tbl = size_t(n) * sizeof(cto_bgzf_block), out_al = (size_t(std::max<int64_t>(out_bytes, 256)) + 255) / 256 * 256;
        if ((rc = c->d_in.ensure(in_al + tbl)) || (rc = c->d_out.ensure(out_al + size_t(n) * 4)) || (rc = c->h_out.ensure(out_al + size_t(n) * 4))) return rc;
        const double T2 = now_s(), C2 = cpu_s();
        CTO_HIP(hipMemcpyAsync(c->d_in.p, c->h_in.p, in_al + tbl, hipMemcpyHostToDevice, c->stream));
        if ((rc = cto_bgzf_inflate(c->d_in.p, reinterpret_cast<const cto_bgzf_block*>(static_cast<char*>(c->d_in.p) + in_al), int(n), c->d_out.p,
                                   reinterpret_cast<int*>(static_cast<char*>(c->d_out.p) + out_al), c->stream)))
            return rc;
        if (cfg->device_pileup) {
            // reads -> columns on the device: only the blocks' status words come back before the pile-up
            CTO_HIP(hipMemcpyAsync(static_cast<char*>(c->h_out.p) + out_al, static_cast<char*>(c->d_out.p) + out_al, size_t(n) * 4, hipMemcpyDeviceToHost, c->stream));
            CTO_HIP(hipEventRecord(c->landed, c->stream));
            CTO_HIP(wait_event(c->landed));
            const int* st0 = reinterpret_cast<const int*>(static_cast<char*>(c->h_out.p) + out_al);
            for (int64_t b = 0; b < n; ++b)
                CTO_REQUIRE(st0[b] == 0, CTO_EINVAL, "%s: the BGZF block at file offset %llu does not inflate (status %d)", j.bam_path,
                            (unsigned long long)blocks[b].file_off, st0[b]);
            if (!c->pile && (rc = cto_dev_pileup_create(&c->pile))) return rc;
            std::vector<uint64_t> voffs(size_t(4096 + ((hi - lo) >> 14) + 64));
            int32_t tid = -1;
            int64_t n_st = CTO_ENOMEM;
            for (int tries = 0; tries < 4 && n_st == CTO_ENOMEM; ++tries) {       // an index naming more offsets than expected: a larger table
                if (tries) voffs.resize(voffs.size() * 8);
                n_st = cto_bam_record_starts(j.bam_path, nullptr, ctg.c_str(), lo, hi, fb, fe, voffs.data(), int64_t(voffs.size()), &tid);
            }
            if (n_st < 0 && n_st != CTO_ENOMEM) return int(n_st);
            int fallback = n_st <= 0;                                               // still too many: the host reader takes the chunk
            cto_pack_view dvw{};
            cto_pack* lite = nullptr;
            if (!fallback) {
                rc = cto_pileup_device(c->pile, c->d_out.p, blocks, n, voffs.data(), n_st, tid, lo, hi, iv.empty() ? nullptr : iv.data(), int64_t(iv.size() / 2),
                                       s->ref.data(), s->ref_start, s->ref.size(), 2316, 0, cfg->max_depth, cfg->max_indel_length, c->stream, &dvw, &lite, &fallback);
                if (rc != CTO_OK) return rc;
            }
            if (!fallback) {
                // the pack's arrays move from the context (re-used by the next chunk) into the slot's one device allocation, laid out
                // as the upload path lays it out, + the candidate positions
                const size_t nc = size_t(dvw.n_cols), ne = size_t(dvw.n_entries), nk = size_t(dvw.n_keys), ns = s->sites.size();
                if ((rc = c->h_sites.ensure(s->sites.size() * 4 + 4)) != CTO_OK) { cto_pack_free(lite); return rc; }
                memcpy(c->h_sites.p, s->sites.data(), s->sites.size() * 4);
                const void* src[8] = {dvw.entries, dvw.col_pos, dvw.col_ref, dvw.col_off, dvw.key_off, dvw.key_meta, dvw.key_group, c->h_sites.p};
                const size_t bytes[8] = {ne * 4, nc * 4, nc, (nc + 1) * 8, (nc + 1) * 4, nk, nk * 4, ns * 4};
                size_t off[8], total = 0;
                for (int i = 0; i < 8; ++i) { off[i] = total; total += (bytes[i] + 255) / 256 * 256 + 256; }
                if ((rc = s->pack_dev.ensure(total)) != CTO_OK) { cto_pack_free(lite); return rc; }
                char* d = static_cast<char*>(s->pack_dev.p);
                for (int i = 0; i < 8; ++i)
                    if (bytes[i])
                        CTO_HIP(hipMemcpyAsync(d + off[i], src[i], bytes[i], i == 7 ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, c->stream));
                if (s->pack) cto_pack_free(s->pack);
                s->pack = lite;
                s->hv = cto_pack_view{};
                s->hv.n_cols = dvw.n_cols; s->hv.n_entries = dvw.n_entries; s->hv.n_keys = dvw.n_keys;
                s->dv = s->hv;
                s->dv.entries = reinterpret_cast<const uint32_t*>(d + off[0]);
                s->dv.col_pos = reinterpret_cast<const int32_t*>(d + off[1]);
                s->dv.col_ref = reinterpret_cast<const uint8_t*>(d + off[2]);
                s->dv.col_off = reinterpret_cast<const int64_t*>(d + off[3]);
                s->dv.key_off = reinterpret_cast<const int32_t*>(d + off[4]);
                s->dv.key_meta = reinterpret_cast<const uint8_t*>(d + off[5]);
                s->dv.key_group = reinterpret_cast<const int32_t*>(d + off[6]);
                s->d_site_pos = reinterpret_cast<const int32_t*>(d + off[7]);
                CTO_HIP(hipEventRecord(s->uploaded, c->stream));
                CTO_HIP(hipEventRecord(c->landed, c->stream));
                CTO_HIP(wait_event(c->landed));               // the context (and s->sites' bytes) are free for the next chunk
                if (timing)
                    fprintf(stderr, "device pile-up: %.1f MB in %lld blocks -> %lld columns, %lld entries: read %.1f ms, inflate + pile-up %.1f; thread CPU: span %.1f ms, read %.1f, scan %.1f, rest %.1f\n", nbytes / 1e6,
                            (long long)n, (long long)dvw.n_cols, (long long)dvw.n_entries, (T1 - T0) * 1e3, (now_s() - T2) * 1e3, (Cs - C0) * 1e3, (C1 - Cs) * 1e3, (C2 - C1) * 1e3, (cpu_s() - C2) * 1e3);
                *done = 2;
                ++device_inflated;
                ++device_piled;
                return CTO_OK;
            }
        }
        CTO_HIP(hipMemcpyAsync(c->h_out.p, c->d_out.p, out_al + size_t(n) * 4, hipMemcpyDeviceToHost, c->stream));
        CTO_HIP(hipEventRecord(c->landed, c->stream));
        const double T3 = now_s();
        CTO_HIP(wait_event(c->landed));
        const double T4 = now_s();
        const int* status = reinterpret_cast<const int*>(static_cast<char*>(c->h_out.p) + out_al);
        for (int64_t b = 0; b < n; ++b)
            CTO_REQUIRE(status[b] == 0, CTO_EINVAL, "%s: the BGZF block at file offset %llu does not inflate (status %d)", j.bam_path,
                        (unsigned long long)blocks[b].file_off, status[b]);
        rc = cto_pack_from_bam_inflated(j.bam_path, nullptr, ctg.c_str(), lo, hi, iv.empty() ? nullptr : iv.data(), int64_t(iv.size() / 2), s->ref.data(),
                                        s->ref_start, s->ref.size(), 2316, 0, cfg->max_depth, cfg->max_indel_length,
                                        static_cast<const uint8_t*>(c->h_out.p), size_t(out_bytes), blocks, n, &s->pack);
        if (timing)
            fprintf(stderr, "device inflate: %.1f MB in %lld blocks -> %.1f MB: read %.1f ms, scan + alloc %.1f, enqueue %.1f, on the device %.1f, pile-up %.1f\n",
                    nbytes / 1e6, (long long)n, out_bytes / 1e6, (T1 - T0) * 1e3, (T2 - T1) * 1e3, (T3 - T2) * 1e3, (T4 - T3) * 1e3, (now_s() - T4) * 1e3);
        if (rc == CTO_OK) { *done = 1; ++device_inflated; }
        return rc;
    }

    // mpileup text -> pack on the device (cto_tokenise_device): `text` is host memory (the slot's tokeniser buffer when the file was read
    // straight into it).  *done = 1: the pack sits in the slot's device buffers in the layout of the upload path, s->pack is the host's
    // lite part, s->uploaded is recorded; *done = 0: the text is one the single pass declines - the caller tokenises it on the host.
    int pack_from_text_device(Slot* s, const char* text, size_t len, hipStream_t stream, int* done) {
        *done = 0;
        int rc;
        if (!s->tok && (rc = cto_dev_tokeniser_create(&s->tok))) return rc;
        cto_pack_view dvw{};
        cto_pack* lite = nullptr;
        int fallback = 0;
        if ((rc = cto_tokenise_device(s->tok, text, len, s->ref.data(), s->ref_start, s->ref.size(), cfg->max_indel_length, stream, &dvw, &lite, &fallback)))
            return rc;
        if (fallback) return CTO_OK;
        const size_t nc = size_t(dvw.n_cols), ne = size_t(dvw.n_entries), nk = size_t(dvw.n_keys), ns = s->sites.size();
        if ((rc = s->stage.ensure(ns * 4 + 256)) != CTO_OK) { cto_pack_free(lite); return rc; }
        memcpy(s->stage.p, s->sites.data(), ns * 4);
        const void* src[8] = {dvw.entries, dvw.col_pos, dvw.col_ref, dvw.col_off, dvw.key_off, dvw.key_meta, dvw.key_group, s->stage.p};
        const size_t bytes[8] = {ne * 4, nc * 4, nc, (nc + 1) * 8, (nc + 1) * 4, nk, nk * 4, ns * 4};
        size_t off[8], total = 0;
        for (int i = 0; i < 8; ++i) { off[i] = total; total += (bytes[i] + 255) / 256 * 256 + 256; }
        if ((rc = s->pack_dev.ensure(total)) != CTO_OK) { cto_pack_free(lite); return rc; }
        char* d = static_cast<char*>(s->pack_dev.p);
        for (int i = 0; i < 8; ++i)
            if (bytes[i]) CTO_HIP(hipMemcpyAsync(d + off[i], src[i], bytes[i], i == 7 ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, stream));
        if (s->pack) cto_pack_free(s->pack);
        s->pack = lite;
        s->hv = cto_pack_view{};
        s->hv.n_cols = dvw.n_cols; s->hv.n_entries = dvw.n_entries; s->hv.n_keys = dvw.n_keys;
        s->dv = s->hv;
        s->dv.entries = reinterpret_cast<const uint32_t*>(d + off[0]);
        s->dv.col_pos = reinterpret_cast<const int32_t*>(d + off[1]);
        s->dv.col_ref = reinterpret_cast<const uint8_t*>(d + off[2]);
        s->dv.col_off = reinterpret_cast<const int64_t*>(d + off[3]);
        s->dv.key_off = reinterpret_cast<const int32_t*>(d + off[4]);
        s->dv.key_meta = reinterpret_cast<const uint8_t*>(d + off[5]);
        s->dv.key_group = reinterpret_cast<const int32_t*>(d + off[6]);
        s->d_site_pos = reinterpret_cast<const int32_t*>(d + off[7]);
        CTO_HIP(hipEventRecord(s->uploaded, stream));
        CTO_HIP(wait_event(s->uploaded));              // the site list left the staging buffer; the tokeniser's arrays are free for the next chunk
        *done = 1;
        ++device_tokenised;
        return CTO_OK;
    }

    // host half of a chunk + the upload; false = nothing to call in this chunk (no output) or an error (failed is set)
    bool produce(Slot* s, hipStream_t stream) {
        const cto_chunk_job& j = jobs[s->job];
        const std::string ctg = j.ctg_name;
        const bool region_job = j.bed_path == nullptr;           // candidates are extracted from the pile-up, not read from a BED
        std::string err;
        Mapped bed;
        int64_t ctg_start = 0, ctg_end = 0;
        int64_t cand_lo = 0, cand_hi = 0;                        // REGION job: rows of this range take part in the extraction
        if (region_job) {
            ctg_start = std::max<int64_t>(1, j.region_start);
            ctg_end = j.region_end;
            // extract_candidates_calling.py:289-292: reads (and therefore rows, and candidates) of ctg_start - 33 .. ctg_end + 33
            cand_lo = std::max<int64_t>(ctg_start - FLANK_POS, 1);
            cand_hi = ctg_end + FLANK_POS;
            s->sites.clear();
        } else {
            if (!bed.open(j.bed_path, &err)) { fail(err); return false; }
            std::vector<int32_t> centres(std::count(bed.p, bed.p + bed.n, '\n') + 2);
            int64_t span[2] = {0, 0};
            int has_types = 0;
            const int64_t n = cto_bed_centres(bed.p ? bed.p : "", bed.n, ctg.c_str(), centres.data(), int64_t(centres.size()), span, &has_types);
            if (n < 0) { fail(cto_last_error()); return false; }
            centres.resize(size_t(n));
            std::sort(centres.begin(), centres.end());
            centres.erase(std::unique(centres.begin(), centres.end()), centres.end());
            s->sites.swap(centres);
            candidates += int64_t(s->sites.size());
            if (s->sites.empty()) {
                if (cfg->verbose) fprintf(stderr, "[INFO] %s total processed positions: 0\n", j.ctg_name);
                return false;
            }
            ctg_start = span[0];
            ctg_end = span[1];
        }
        s->ref_start = std::max<int64_t>(1, ctg_start - EXPAND_REF);
        FaiRec fr;
        if (!fai_of(ctg, &fr, &err) || !read_region(fasta, fr, s->ref_start, ctg_end + EXPAND_REF, &s->ref, &err)) { fail(err); return false; }
        if (s->ref.empty()) { fail(std::string("[ERROR] Failed to load reference sequence from file (") + cfg->ref_fa + ")."); return false; }
        if (s->pack) { cto_pack_free(s->pack); s->pack = nullptr; }
        int rc = CTO_OK;
        bool piled_on_device = false;
        const double t_pack = now_s();
        if (j.mpileup_path) {
            const size_t pl = strlen(j.mpileup_path);
            const bool gz = pl > 3 && strcmp(j.mpileup_path + pl - 3, ".gz") == 0;
            int on_dev = 0;
            const char* tp = nullptr;
            size_t tn = 0;
            Mapped txt;
            if (cfg->device_tokenise && !gz) {
                // the file is read straight into the tokeniser's page-locked buffer: no mapping, no staging copy
                if (!s->tok && (rc = cto_dev_tokeniser_create(&s->tok)) != CTO_OK) { fail(cto_last_error()); return false; }
                const int fd = ::open(j.mpileup_path, O_RDONLY | O_CLOEXEC);
                if (fd < 0) { fail(std::string("cannot open ") + j.mpileup_path); return false; }
                struct stat st;
                if (fstat(fd, &st) != 0) { ::close(fd); fail(std::string("cannot stat ") + j.mpileup_path); return false; }
                tn = size_t(st.st_size);
                char* buf = cto_dev_tokeniser_buffer(s->tok, tn + 1);
                if (!buf) { ::close(fd); fail(cto_last_error()); return false; }
                size_t got = 0;
                while (got < tn) {
                    const ssize_t r = ::pread(fd, buf + got, tn - got, off_t(got));
                    if (r <= 0) break;
                    got += size_t(r);
                }
                ::close(fd);
                if (got != tn) { fail(std::string("short read of ") + j.mpileup_path); return false; }
                tp = buf;
                rc = pack_from_text_device(s, tp, tn, stream, &on_dev);
            } else {
                if (!txt.open(j.mpileup_path, &err)) { fail(err); return false; }
                tp = txt.p ? txt.p : "";
                tn = txt.n;
                if (cfg->device_tokenise) rc = pack_from_text_device(s, tp, tn, stream, &on_dev);
            }
            if (rc == CTO_OK && on_dev) piled_on_device = true;
            else if (rc == CTO_OK) {
                // the tokeniser's threads merge their entries straight into the staging buffer (a read-base is >= 3 characters of text)
                const size_t ecap = tn / 3 + 4096;
                if (s->stage.ensure(ecap * 4 + tn / 4 + (size_t(1) << 20)) != CTO_OK) { fail(cto_last_error()); return false; }
                rc = pack_from_mpileup_impl(tp, tn, s->ref.data(), s->ref_start, s->ref.size(), cfg->max_indel_length,
                                            static_cast<uint32_t*>(s->stage.p), ecap, &s->pack);
            }
        } else {
            std::vector<int64_t> iv;                               // REGION job: every position of the range (no -l)
            if (!region_job) bed_intervals(bed.p ? bed.p : "", bed.n, ctg, &iv);
            // REGION job: the candidate range + the flanks of the windows at its edges
            const int64_t lo = region_job ? std::max<int64_t>(1, cand_lo - REGION_FLANK) : std::max<int64_t>(1, ctg_start - FLANK_POS);
            const int64_t hi = region_job ? cand_hi + REGION_FLANK : ctg_end + FLANK_POS;
            if (cfg->samtools) {
                // the reference's own producer: `samtools mpileup` with --min-BQ 0 (one pileup serves both passes), its text tokenised
                std::vector<std::string> cmd = {cfg->samtools, "mpileup", "--reverse-del", "--output-MQ", "-r",
                                                ctg + ":" + std::to_string(lo) + "-" + std::to_string(hi), "--min-MQ", "0", "--min-BQ", "0"};
                if (!region_job) { cmd.push_back("-l"); cmd.push_back(j.bed_path); }      // the reference's order of options
                cmd.push_back("--excl-flags");
                cmd.push_back("2316");
                if (cfg->samtools_max_depth > 0) { cmd.push_back("--max-depth"); cmd.push_back(std::to_string(cfg->samtools_max_depth)); }
                cmd.push_back(j.bam_path);
                std::vector<char> text;
                if (!capture_stdout(cmd, &text, &err)) { fail(err); return false; }
                int on_dev = 0;
                if (cfg->device_tokenise && !text.empty()) rc = pack_from_text_device(s, text.data(), text.size(), stream, &on_dev);
                if (rc == CTO_OK && on_dev) piled_on_device = true;
                else if (rc == CTO_OK) {
                    const size_t ecap = text.size() / 3 + 4096;
                    if (s->stage.ensure(ecap * 4 + text.size() / 4 + (size_t(1) << 20)) != CTO_OK) { fail(cto_last_error()); return false; }
                    rc = pack_from_mpileup_impl(text.empty() ? "" : text.data(), text.size(), s->ref.data(), s->ref_start, s->ref.size(),
                                                cfg->max_indel_length, static_cast<uint32_t*>(s->stage.p), ecap, &s->pack);
                }
            } else {
                InflateCtx* c = nullptr;
                int done = 0;
                // a device-inflate context is free: this chunk's blocks go to the GPU.  A REGION job waits for one: piled up at every
                // position it is seven times a BED chunk's work, which the host reader needs most of a second of a core for
                // A BED chunk does not wait (CTO_CTX_WAIT_MS, default 0): measured with 96 chunks, waiting 0 / 10 / 20 / 40 ms for a context
                // sends 68 / 72 / 72 / 80 of them through the device and gives 667 / 654 / 646 / 612 k sites/s - for BED chunks the device
                // is the busier side, the cores take what it cannot
                static const int ctx_wait_ms = [] { const char* e = getenv("CTO_CTX_WAIT_MS"); return e ? atoi(e) : 0; }();
                if (region_job && cfg->device_pileup && !inflate_ctx.empty() ? free_ctx.pop(&c)
                                                                              : (inflate_ctx.empty() || ctx_wait_ms <= 0 ? free_ctx.try_pop(&c) : free_ctx.pop_for(&c, ctx_wait_ms))) {
                    rc = pack_from_bam_device(j, ctg, lo, hi, iv, s, c, &done);
                    free_ctx.push(c);
                    piled_on_device = rc == CTO_OK && done == 2;
                }
                if (!done && rc == CTO_OK)
                    rc = cto_pack_from_bam(j.bam_path, nullptr, ctg.c_str(), lo, hi, iv.empty() ? nullptr : iv.data(), int64_t(iv.size() / 2),
                                           s->ref.data(), s->ref_start, s->ref.size(), 2316, 0, cfg->max_depth, cfg->max_indel_length, &s->pack);
            }
        }
        if (rc != CTO_OK) { fail(cto_last_error()); return false; }
        if (piled_on_device) {               // the pack is in the slot's device buffers already (pack_from_bam_device), s->uploaded recorded
            { std::lock_guard<std::mutex> g(stat_m); pack_s += now_s() - t_pack; }
            return region_job ? extract_sites(s, j, cand_lo, cand_hi, stream) : true;
        }
        if (cto_pack_view_of(s->pack, &s->hv) != CTO_OK) { fail(cto_last_error()); return false; }
        // ---- upload ----
        const double t_up = now_s();
        // One copy per chunk out of a page-locked staging buffer.  (hipMemcpyAsync from the pack's pageable arrays goes through the
        // runtime's own staging buffer, which every producer thread shares: measured, 8 producers spent 2.4 ms per chunk in those
        // calls, 16 producers 6.6 ms, and the whole pipeline levelled off at ~8.5 GB/s of uploads = 1.4-1.6 M sites/s.)
        const cto_pack_view& h = s->hv;
        const size_t nc = size_t(h.n_cols), ne = size_t(h.n_entries), nk = size_t(h.n_keys), ns = s->sites.size();
        const void* src[8] = {h.entries, h.col_pos, h.col_ref, h.col_off, h.key_off, h.key_meta, h.key_group, s->sites.data()};
        const size_t bytes[8] = {ne * 4, nc * 4, nc, (nc + 1) * 8, (nc + 1) * 4, nk, nk * 4, ns * 4};
        size_t off[8], total = 0;
        for (int i = 0; i < 8; ++i) { off[i] = total; total += (bytes[i] + 255) / 256 * 256 + 256; }
        const bool in_place = h.entries == s->stage.p && ne > 0;          // entries first: already there for the text producer
        if (s->stage.grow_keeping(total, in_place ? bytes[0] : 0) != CTO_OK || s->pack_dev.ensure(total) != CTO_OK) { fail(cto_last_error()); return false; }
        char* hs = static_cast<char*>(s->stage.p);
        if (in_place) {                      // grow_keeping may have moved the staging buffer: nothing may keep pointing at the old one
            s->pack->ext_entries = reinterpret_cast<uint32_t*>(hs);
            s->hv.entries = reinterpret_cast<const uint32_t*>(hs);
        }
        for (int i = in_place ? 1 : 0; i < 8; ++i)
            if (bytes[i]) memcpy(hs + off[i], src[i], bytes[i]);
        if (hipMemcpyAsync(s->pack_dev.p, hs, total, hipMemcpyHostToDevice, stream) != hipSuccess) { fail("hipMemcpyAsync failed"); return false; }
        const char* d = static_cast<const char*>(s->pack_dev.p);
        s->dv = h;
        s->dv.entries = reinterpret_cast<const uint32_t*>(d + off[0]);
        s->dv.col_pos = reinterpret_cast<const int32_t*>(d + off[1]);
        s->dv.col_ref = reinterpret_cast<const uint8_t*>(d + off[2]);
        s->dv.col_off = reinterpret_cast<const int64_t*>(d + off[3]);
        s->dv.key_off = reinterpret_cast<const int32_t*>(d + off[4]);
        s->dv.key_meta = reinterpret_cast<const uint8_t*>(d + off[5]);
        s->dv.key_group = reinterpret_cast<const int32_t*>(d + off[6]);
        s->d_site_pos = reinterpret_cast<const int32_t*>(d + off[7]);
        if (hipEventRecord(s->uploaded, stream) != hipSuccess) { fail("hipEventRecord failed"); return false; }
        { std::lock_guard<std::mutex> g(stat_m); pack_s += t_up - t_pack; upload_s += now_s() - t_up; }
        return region_job ? extract_sites(s, j, cand_lo, cand_hi, stream) : true;
    }

    // REGION job: STEP 1 of the reference on the pack that is now in HBM (the gates of extract_candidates_calling.py:55-169 as
    // cto_extract_candidates runs them, the candidate list of :433-446 compacted on the device in position order).  The positions
    // stay in HBM as the chunk's site list and come to the host once (the writers print them); false = no candidate (no output).
    bool extract_sites(Slot* s, const cto_chunk_job& j, int64_t cand_lo, int64_t cand_hi, hipStream_t stream) {
        const double t0 = now_s();
        const int64_t nc = s->hv.n_cols;
        if (nc == 0) {
            if (j.candidates_path) { FILE* f = fopen(j.candidates_path, "w"); if (f) fclose(f); }
            if (j.hybrid_info_path) { FILE* f = fopen(j.hybrid_info_path, "w"); if (f) fclose(f); }
            if (cfg->verbose) fprintf(stderr, "[INFO] %s total processed positions: 0\n", j.ctg_name);
            return false;
        }
        const int64_t nb = cdiv(nc, 256);
        if (s->xflags.ensure(size_t(nc)) != CTO_OK || s->xdepth.ensure(size_t(nc) * 4) != CTO_OK ||
            s->xscratch.ensure(size_t(std::max<int64_t>(s->hv.n_keys, 1)) * 4) != CTO_OK || s->cand.ensure(size_t(nc) * 4 + 256) != CTO_OK ||
            s->cand_scr.ensure(size_t(nb + 2) * 4) != CTO_OK || s->cand_host.ensure(size_t(nc) * 4 + 256) != CTO_OK) {
            fail(cto_last_error());
            return false;
        }
        const bool indel = cfg->K == 6;
        int32_t* d_n = static_cast<int32_t*>(s->cand_scr.p) + nb + 1;
        int rc = extract_candidates_scratch(&s->dv, cfg->extract_min_mq, cfg->extract_min_bq, cfg->snv_min_af, indel ? cfg->indel_min_af : 1.0,
                                            cfg->min_coverage, cfg->alt_base_num, indel ? 1 : 0, static_cast<uint32_t*>(s->xscratch.p),
                                            static_cast<uint8_t*>(s->xflags.p), static_cast<int32_t*>(s->xdepth.p), stream);
        // the other modes of extract_candidates_calling, each on the flags in HBM and in the reference's order: rows outside the confident
        // BED do not exist (`samtools mpileup -l`, :302); indel candidates only inside --call_indels_only_in_these_regions (:437-446; the
        // user's own --bed_fn supersedes it, :438); positions of the hybrid / genotyping VCF are marked (:347-349, 370-383)
        const std::vector<int64_t>* indel_iv = nullptr;
        if (indel && !cfg->indel_bed_superseded) {
            const auto it = indel_regions.find(j.ctg_name);
            if (it != indel_regions.end()) indel_iv = &it->second;
        }
        const int64_t n_conf = j.restrict_to_confident ? std::max<int64_t>(j.n_confident_intervals, 0) : 0;
        const int64_t n_indel_iv = indel_iv ? int64_t(indel_iv->size() / 2) : 0;
        int64_t k_lo = 0, k_hi = 0;                    // the known positions inside the rows that take part
        if (j.known_pos && j.n_known_pos > 0) {
            k_lo = std::lower_bound(j.known_pos, j.known_pos + j.n_known_pos, int32_t(std::min<int64_t>(cand_lo, INT32_MAX))) - j.known_pos;
            k_hi = std::upper_bound(j.known_pos, j.known_pos + j.n_known_pos, int32_t(std::min<int64_t>(cand_hi, INT32_MAX))) - j.known_pos;
        }
        const int64_t n_known = k_hi - k_lo;
        const bool want_info = j.hybrid_info_path != nullptr;
        const size_t nk_pack = size_t(std::max<int64_t>(s->hv.n_keys, 1));
        // layout of xmode (int32 words): confident pairs | indel pairs | known positions | records [n_known][16] | gcnt [n_keys] | gfirst [n_keys]
        const size_t w_conf = 0, w_indel = w_conf + size_t(2 * n_conf), w_known = w_indel + size_t(2 * n_indel_iv), w_rec = w_known + size_t(n_known),
                     w_gcnt = w_rec + (want_info ? size_t(n_known) * 16 : 0), w_gfirst = w_gcnt + (want_info && indel ? nk_pack : 0),
                     w_end = w_gfirst + (want_info && indel ? nk_pack : 0);
        int32_t* xm = nullptr;
        int32_t* xh = nullptr;
        if (rc == CTO_OK && (j.restrict_to_confident || n_indel_iv > 0 || n_known > 0)) {
            if (s->xmode.ensure(w_end * 4 + 256) != CTO_OK || s->xmode_host.ensure(w_end * 4 + 256) != CTO_OK) { fail(cto_last_error()); return false; }
            xm = static_cast<int32_t*>(s->xmode.p);
            xh = static_cast<int32_t*>(s->xmode_host.p);
            if (n_conf > 0) memcpy(xh + w_conf, j.confident_intervals, size_t(2 * n_conf) * 4);
            for (int64_t i = 0; i < 2 * n_indel_iv; ++i) xh[w_indel + size_t(i)] = int32_t(std::min<int64_t>((*indel_iv)[size_t(i)], INT32_MAX));
            if (n_known > 0) memcpy(xh + w_known, j.known_pos + k_lo, size_t(n_known) * 4);
            if (w_rec > 0 && hipMemcpyAsync(xm, xh, w_rec * 4, hipMemcpyHostToDevice, stream) != hipSuccess) { fail("hipMemcpyAsync failed"); return false; }
            if (j.restrict_to_confident)
                rc = cto_extract_restrict(&s->dv, static_cast<uint8_t*>(s->xflags.p), static_cast<int32_t*>(s->xdepth.p), xm + w_conf, int(n_conf), 0xff, stream);
            if (rc == CTO_OK && n_indel_iv > 0)
                rc = cto_extract_restrict(&s->dv, static_cast<uint8_t*>(s->xflags.p), nullptr, xm + w_indel, int(n_indel_iv), 2 | 16, stream);
            if (rc == CTO_OK && n_known > 0)
                rc = cto_extract_mark(&s->dv, static_cast<uint8_t*>(s->xflags.p), xm + w_known, int(n_known), 64, stream);
            if (rc == CTO_OK && want_info && n_known > 0) {
                if (indel && hipMemsetAsync(xm + w_gcnt, 0, 2 * nk_pack * 4, stream) != hipSuccess) { fail("hipMemsetAsync failed"); return false; }
                rc = cto_hybrid_info(&s->dv, static_cast<const uint8_t*>(s->xflags.p), xm + w_known, int(n_known), cfg->extract_min_mq, cfg->extract_min_bq,
                                     indel ? 1 : 0, xm + w_rec, indel ? reinterpret_cast<uint32_t*>(xm + w_gcnt) : nullptr, indel ? xm + w_gfirst : nullptr, stream);
                if (rc == CTO_OK && hipMemcpyAsync(xh + w_rec, xm + w_rec, (w_end - w_rec) * 4, hipMemcpyDeviceToHost, stream) != hipSuccess) {
                    fail("hipMemcpyAsync failed");
                    return false;
                }
            }
        }
        if (rc == CTO_OK)
            rc = cto_candidate_positions(&s->dv, static_cast<const uint8_t*>(s->xflags.p), indel ? 2 : 1, int32_t(std::min<int64_t>(cand_lo, INT32_MAX)),
                                         int32_t(std::min<int64_t>(cand_hi, INT32_MAX)), static_cast<int32_t*>(s->cand.p), nc,
                                         static_cast<int32_t*>(s->cand_scr.p), d_n, stream);
        if (rc != CTO_OK) { fail(cto_last_error()); return false; }
        auto* h = static_cast<int32_t*>(s->cand_host.p);
        if (hipMemcpyAsync(h, d_n, 4, hipMemcpyDeviceToHost, stream) != hipSuccess || hipEventRecord(s->uploaded, stream) != hipSuccess ||
            wait_event(s->uploaded) != hipSuccess) { fail("candidate extraction failed on the device"); return false; }
        int64_t n = h[0];
        candidates += n;
        if (n > 0) {
            if (hipMemcpyAsync(h, s->cand.p, size_t(n) * 4, hipMemcpyDeviceToHost, stream) != hipSuccess || hipEventRecord(s->uploaded, stream) != hipSuccess ||
                wait_event(s->uploaded) != hipSuccess) { fail("candidate extraction failed on the device"); return false; }
            s->sites.assign(h, h + n);
        }
        if (want_info) {                         // `<ctg>.<chunk>_hybrid_info` (:352-354, 490-497): counts from the device, strings here
            FILE* f = fopen(j.hybrid_info_path, "w");
            if (!f) { fail(std::string("cannot write ") + j.hybrid_info_path); return false; }
            if (n_known > 0) {
                const uint32_t* gc = indel ? reinterpret_cast<const uint32_t*>(xh + w_gcnt) : nullptr;
                const int32_t* gf = indel ? xh + w_gfirst : nullptr;
                const int64_t need = cto_hybrid_info_rows(s->pack, j.ctg_name, n_known, xh + w_known, xh + w_rec, indel ? 1 : 0, gc, gf, nullptr, 0);
                if (need < 0) { fclose(f); fail(cto_last_error()); return false; }
                std::vector<char> text(size_t(need) + 1);
                if (need > 0 && cto_hybrid_info_rows(s->pack, j.ctg_name, n_known, xh + w_known, xh + w_rec, indel ? 1 : 0, gc, gf, text.data(), size_t(need)) != need) {
                    fclose(f); fail(cto_last_error()); return false;
                }
                if (need > 0) fwrite(text.data(), 1, size_t(need), f);
            }
            if (fclose(f) != 0) { fail(std::string("short write to ") + j.hybrid_info_path); return false; }
        }
        s->d_site_pos = static_cast<const int32_t*>(s->cand.p);
        if (j.candidates_path) {                 // the reference's BED chunk rows (extract_candidates_calling.py:450-488), one file per region
            FILE* f = fopen(j.candidates_path, "w");
            if (!f) { fail(std::string("cannot write ") + j.candidates_path); return false; }
            for (int64_t i = 0; i < n; ++i) fprintf(f, "%s\t%d\t%d\n", j.ctg_name, std::max(h[i] - 17, 1), h[i] + 17);
            if (fclose(f) != 0) { fail(std::string("short write to ") + j.candidates_path); return false; }
        }
        { std::lock_guard<std::mutex> g(stat_m); pack_s += now_s() - t0; }
        if (n == 0) {
            if (cfg->verbose) fprintf(stderr, "[INFO] %s total processed positions: 0\n", j.ctg_name);
            return false;
        }
        return true;
    }

    // Tensor creation of one chunk on `main`: the [33][34] inputs of both networks and what the writers need of the candidate columns.
    // One kernel (a workgroup per candidate, the column histograms never leave LDS); CTO_FUSED_FEATURIZE=0 selects the two-stage path
    // through the per-column vectors in HBM (kept for A/B runs - same results).
    bool fused_featurize = true;
    int tensors(Slot* s, int64_t n, float* x_aff, float* x_neg, int32_t* site_info, int16_t* site_colvec, int32_t* sitefirst, uint32_t* keycnt,
                int32_t* keyfirst, hipStream_t main) {
        int rc;
        // heavily overlapping windows (candidates a few bases apart) share most of their columns: the one-kernel path would histogram
        // them once per candidate, the two-stage path once (measured equal at ~8 columns per candidate; same results either way)
        if (fused_featurize && s->hv.n_cols >= 8 * n)
            return cto_featurize_sites(&s->dv, s->d_site_pos, n, cfg->min_bq, cfg->min_rescale_cov, x_aff, x_neg, nullptr, nullptr, site_info, site_colvec,
                                       sitefirst, keycnt, keyfirst, main);
        const size_t nc = size_t(std::max<int64_t>(s->hv.n_cols, 1));
        if ((rc = s->colvec.ensure(nc * CTO_COLVEC_STRIDE * 2)) || (rc = s->coldepth.ensure(nc * 8))) return rc;
        auto* colvec = static_cast<int16_t*>(s->colvec.p);
        if ((rc = cto_featurize_columns(&s->dv, cfg->min_bq, colvec, static_cast<int32_t*>(s->coldepth.p), keycnt, main))) return rc;
        if ((rc = cto_gather_windows(&s->dv, colvec, static_cast<int32_t*>(s->coldepth.p), s->d_site_pos, n, cfg->min_bq, cfg->min_rescale_cov, x_aff,
                                     x_neg, nullptr, nullptr, site_info, sitefirst, keyfirst, main)))
            return rc;
        hipLaunchKernelGGL(k_gather_rows, dim3(unsigned(n)), dim3(128), 0, main, colvec, site_info, n, site_colvec);
        CTO_HIP(hipGetLastError());
        return CTO_OK;
    }

    int launch(Slot* s, hipStream_t main, hipStream_t copy_back, cto_model* aff, cto_model* neg) {
        const int K = cfg->K;
        const int64_t n = int64_t(s->sites.size());
        const size_t nk = size_t(std::max<int64_t>(s->hv.n_keys, 1));
        int rc;
        // everything the writers need goes into ONE device buffer and comes back with ONE copy on the copy-back stream: seven copies
        // queued behind the kernels on the launch stream cost ~0.1 ms per chunk in which the next chunk's kernels could not start
        const size_t rbytes[7] = {size_t(n) * 48, size_t(n) * CTO_COLVEC_STRIDE * 2, size_t(n) * 32, size_t(n) * 16, size_t(n) * 8, nk * 4, nk * 8};
        size_t total = 0;
        for (int i = 0; i < 7; ++i) { s->roff[i] = total; total += (rbytes[i] + 255) / 256 * 256; }
        if ((rc = s->x_aff.ensure(size_t(n) * CTO_NPOS * CTO_NCHAN * 4)) || (rc = s->x_neg.ensure(size_t(n) * CTO_NPOS * CTO_NCHAN * 4)) ||
            (rc = s->la.ensure(size_t(K) * n * 8)) || (rc = s->ln.ensure(size_t(K) * n * 8)) || (rc = s->post.ensure(size_t(n) * K * 8)) ||
            (rc = s->res_dev.ensure(total)) || (rc = s->res_host.ensure(total)))
            return rc;
        char* rd = static_cast<char*>(s->res_dev.p);
        auto* site_info = reinterpret_cast<int32_t*>(rd + s->roff[0]);
        auto* site_colvec = reinterpret_cast<int16_t*>(rd + s->roff[1]);
        auto* sitefirst = reinterpret_cast<int32_t*>(rd + s->roff[2]);
        auto* decision = reinterpret_cast<int32_t*>(rd + s->roff[3]);
        auto* qual = reinterpret_cast<double*>(rd + s->roff[4]);
        auto* keycnt = reinterpret_cast<uint32_t*>(rd + s->roff[5]);
        auto* keyfirst = reinterpret_cast<int32_t*>(rd + s->roff[6]);
        CTO_HIP(hipStreamWaitEvent(main, s->uploaded, 0));
        CTO_HIP(hipEventRecord(s->begin, main));
        if ((rc = tensors(s, n, static_cast<float*>(s->x_aff.p), static_cast<float*>(s->x_neg.p), site_info, site_colvec, sitefirst, keycnt, keyfirst,
                          main)))
            return rc;
        const float* x_neg = cfg->neg_reads_aff ? static_cast<const float*>(s->x_aff.p) : static_cast<const float*>(s->x_neg.p);
        if ((rc = cto_model_forward(neg, x_neg, n, static_cast<float*>(s->ln.p), main))) return rc;
        if ((rc = cto_model_forward(aff, static_cast<const float*>(s->x_aff.p), n, static_cast<float*>(s->la.p), main))) return rc;
        if ((rc = cto_posterior(static_cast<const float*>(s->la.p), static_cast<const float*>(s->ln.p), K, n, cfg->d_lik, cfg->d_edges, nullptr,
                                static_cast<double*>(s->post.p), decision, qual, main)))
            return rc;
        CTO_HIP(hipEventRecord(s->kernels_end, main));
        CTO_HIP(hipEventRecord(s->computed, main));
        CTO_HIP(hipStreamWaitEvent(copy_back, s->computed, 0));
        CTO_HIP(hipMemcpyAsync(s->res_host.p, s->res_dev.p, total, hipMemcpyDeviceToHost, copy_back));
        CTO_HIP(hipEventRecord(s->done, copy_back));
        return CTO_OK;
    }

    // ---- the networks fed by a STREAM of sites instead of by chunks --------------------------------------------------------------
    // The recurrent kernels put 32 sites x one direction on a CU, so a launch is efficient when it is a whole number of rounds
    // (16 x CUs sites: 4096 on MI355X) and the reference's 10 000-site chunk files (shared/param.py:21) are 2.44 rounds of work in 2.83
    // rounds of time.  Sites are independent, so the launcher runs the networks on whole rounds only and carries the tail of a chunk
    // into the launch of the next one: the tail's input rows are copied to the front of the next chunk's input buffers, the
    // featurisation of that chunk appends behind them, and what the networks + the epilogue return is dealt back to the chunks it
    // belongs to (`pending`: chunk, first row, rows, in buffer order).  A chunk goes to the writers when its last row is back; what is
    // left at the end of the run (or when `max_pending` chunks are waiting - they hold slots the producers need) is launched as it is.
    struct Pending { Slot* s; int64_t row0, cnt; };
    std::vector<Pending> pending;
    Slot* carry_home = nullptr;              // whose x buffers hold the pending rows ...
    int64_t carry_at = 0;                    // ... starting at this row
    int64_t round_sites = 4096;
    size_t max_pending = 2;
    hipEvent_t flush_begin = nullptr, flush_end = nullptr;
    bool flush_timed = false;

    int64_t pending_rows() const { int64_t c = 0; for (const Pending& q : pending) c += q.cnt; return c; }

    // networks + epilogue over rows [at, at + m) of `home`'s input buffers; results dealt out to the first m pending rows
    int run_networks(Slot* home, int64_t at, int64_t m, hipStream_t main, hipStream_t copy_back, cto_model* aff, cto_model* neg,
                     std::vector<Slot*>* complete, hipEvent_t kernels_end) {
        const int K = cfg->K;
        int rc;
        const size_t row = size_t(CTO_NPOS) * CTO_NCHAN;
        if ((rc = home->la.ensure(size_t(K) * m * 8)) || (rc = home->ln.ensure(size_t(K) * m * 8)) || (rc = home->post.ensure(size_t(m) * K * 8)) ||
            (rc = home->dec_l.ensure(size_t(m) * 16)) || (rc = home->qual_l.ensure(size_t(m) * 8)))
            return rc;
        const float* xa = static_cast<const float*>(home->x_aff.p) + size_t(at) * row;
        const float* xn = cfg->neg_reads_aff ? xa : static_cast<const float*>(home->x_neg.p) + size_t(at) * row;
        if ((rc = cto_model_forward(neg, xn, m, static_cast<float*>(home->ln.p), main))) return rc;
        if ((rc = cto_model_forward(aff, xa, m, static_cast<float*>(home->la.p), main))) return rc;
        if ((rc = cto_posterior(static_cast<const float*>(home->la.p), static_cast<const float*>(home->ln.p), K, m, cfg->d_lik, cfg->d_edges, nullptr,
                                static_cast<double*>(home->post.p), static_cast<int32_t*>(home->dec_l.p), static_cast<double*>(home->qual_l.p), main)))
            return rc;
        if (kernels_end) CTO_HIP(hipEventRecord(kernels_end, main));   // before any chunk of this launch can reach a writer, which reads it
        int64_t o = 0;
        size_t used = 0;
        // chunks handed to *complete leave `pending` on EVERY way out of the loop (a failing HIP call returns from its middle): a
        // slot in both lists would be given back to the free list twice by the launcher's hand_over() + abandon_pending()
        struct ErasePrefix {
            std::vector<Pending>& v; size_t& n;
            ~ErasePrefix() { v.erase(v.begin(), v.begin() + long(n)); }
        } erase_prefix{pending, used};
        for (Pending& q : pending) {
            if (o >= m) break;
            const int64_t take = std::min(q.cnt, m - o);
            char* rd = static_cast<char*>(q.s->res_dev.p);
            CTO_HIP(hipMemcpyAsync(rd + q.s->roff[3] + size_t(q.row0) * 16, static_cast<char*>(home->dec_l.p) + size_t(o) * 16, size_t(take) * 16,
                                   hipMemcpyDeviceToDevice, main));
            CTO_HIP(hipMemcpyAsync(rd + q.s->roff[4] + size_t(q.row0) * 8, static_cast<char*>(home->qual_l.p) + size_t(o) * 8, size_t(take) * 8,
                                   hipMemcpyDeviceToDevice, main));
            q.row0 += take;
            q.cnt -= take;
            o += take;
            if (q.cnt == 0) {
                CTO_HIP(hipEventRecord(q.s->computed, main));
                CTO_HIP(hipStreamWaitEvent(copy_back, q.s->computed, 0));
                CTO_HIP(hipMemcpyAsync(q.s->res_host.p, q.s->res_dev.p, q.s->res_total, hipMemcpyDeviceToHost, copy_back));
                CTO_HIP(hipEventRecord(q.s->done, copy_back));
                complete->push_back(q.s);
                ++used;
            }
        }
        return CTO_OK;
    }

    // one chunk into the stream; chunks whose last row came back go to *complete (in chunk order)
    int launch_stream(Slot* s, hipStream_t main, hipStream_t copy_back, cto_model* aff, cto_model* neg, std::vector<Slot*>* complete) {
        const int64_t n = int64_t(s->sites.size()), c = pending_rows();
        const size_t nk = size_t(std::max<int64_t>(s->hv.n_keys, 1));
        const size_t row = size_t(CTO_NPOS) * CTO_NCHAN * 4;
        int rc;
        const size_t rbytes[7] = {size_t(n) * 48, size_t(n) * CTO_COLVEC_STRIDE * 2, size_t(n) * 32, size_t(n) * 16, size_t(n) * 8, nk * 4, nk * 8};
        size_t total = 0;
        for (int i = 0; i < 7; ++i) { s->roff[i] = total; total += (rbytes[i] + 255) / 256 * 256; }
        s->res_total = total;
        if ((rc = s->x_aff.ensure(size_t(c + n) * row)) || (rc = s->x_neg.ensure(size_t(c + n) * row)) ||
            (rc = s->res_dev.ensure(total)) || (rc = s->res_host.ensure(total)))
            return rc;
        char* rd = static_cast<char*>(s->res_dev.p);
        auto* site_info = reinterpret_cast<int32_t*>(rd + s->roff[0]);
        auto* site_colvec = reinterpret_cast<int16_t*>(rd + s->roff[1]);
        auto* sitefirst = reinterpret_cast<int32_t*>(rd + s->roff[2]);
        auto* keycnt = reinterpret_cast<uint32_t*>(rd + s->roff[5]);
        auto* keyfirst = reinterpret_cast<int32_t*>(rd + s->roff[6]);
        CTO_HIP(hipStreamWaitEvent(main, s->uploaded, 0));
        CTO_HIP(hipEventRecord(s->begin, main));
        if (c > 0) {                         // the rows still waiting move to the front of this chunk's input buffers
            CTO_HIP(hipMemcpyAsync(s->x_aff.p, static_cast<char*>(carry_home->x_aff.p) + size_t(carry_at) * row, size_t(c) * row, hipMemcpyDeviceToDevice, main));
            if (!cfg->neg_reads_aff)
                CTO_HIP(hipMemcpyAsync(s->x_neg.p, static_cast<char*>(carry_home->x_neg.p) + size_t(carry_at) * row, size_t(c) * row, hipMemcpyDeviceToDevice, main));
        }
        if ((rc = tensors(s, n, reinterpret_cast<float*>(static_cast<char*>(s->x_aff.p) + size_t(c) * row),
                          reinterpret_cast<float*>(static_cast<char*>(s->x_neg.p) + size_t(c) * row), site_info, site_colvec, sitefirst, keycnt, keyfirst,
                          main)))
            return rc;
        pending.push_back({s, 0, n});
        carry_home = s;
        carry_at = 0;
        const int64_t all = c + n;
        const int64_t m = pending.size() > max_pending ? all : all / round_sites * round_sites;
        if (m > 0) {
            if ((rc = run_networks(s, 0, m, main, copy_back, aff, neg, complete, s->kernels_end))) return rc;
            carry_at = m;
        } else {
            CTO_HIP(hipEventRecord(s->kernels_end, main));
        }
        return CTO_OK;
    }

    // the end of the run: what is still waiting is launched as it is
    int flush_stream(hipStream_t main, hipStream_t copy_back, cto_model* aff, cto_model* neg, std::vector<Slot*>* complete) {
        const int64_t c = pending_rows();
        if (c == 0) { for (const Pending& q : pending) complete->push_back(q.s); pending.clear(); return CTO_OK; }
        if (flush_begin) CTO_HIP(hipEventRecord(flush_begin, main));
        const int rc = run_networks(carry_home, carry_at, c, main, copy_back, aff, neg, complete, flush_end);
        if (rc == CTO_OK && flush_end) flush_timed = true;
        return rc;
    }

    // alt_info strings, VCF records, file
    bool finish(Slot* s) {
        const cto_chunk_job& j = jobs[s->job];
        if (wait_event(s->done) != hipSuccess) { fail("waiting for the chunk's results failed"); return false; }
        {
            float ms = 0;
            // the kernels this chunk's launch queued (with a tile stream, its own tail runs - and is counted - in the next chunk's launch)
            if (hipEventElapsedTime(&ms, s->begin, s->kernels_end) == hipSuccess) { std::lock_guard<std::mutex> g(stat_m); device_s += ms * 1e-3; }
        }
        const int64_t n = int64_t(s->sites.size());
        char* rh = static_cast<char*>(s->res_host.p);
        auto* info = reinterpret_cast<int32_t*>(rh + s->roff[0]);
        const auto* h_site_colvec = reinterpret_cast<const int16_t*>(rh + s->roff[1]);
        const auto* h_sitefirst = reinterpret_cast<const int32_t*>(rh + s->roff[2]);
        const auto* h_decision = reinterpret_cast<const int32_t*>(rh + s->roff[3]);
        const auto* h_qual = reinterpret_cast<const double*>(rh + s->roff[4]);
        static const uint32_t zero_k[1] = {0};
        static const int32_t zero_kf[2] = {0, 0};
        const uint32_t* keycnt = s->hv.n_keys ? reinterpret_cast<const uint32_t*>(rh + s->roff[5]) : zero_k;
        const int32_t* keyfirst = s->hv.n_keys ? reinterpret_cast<const int32_t*>(rh + s->roff[6]) : zero_kf;
        std::vector<int64_t> alt_off(size_t(n) + 1, 0);
        std::vector<char> alt(size_t(256 * n + (1 << 16)));
        int64_t used = -1;
        for (int tries = 0; tries < 8; ++tries) {
            used = cto_alt_info_batch_sites(s->pack, n, info, 0, h_site_colvec, h_sitefirst,
                                            keycnt, keyfirst, alt.data(), alt.size(), alt_off.data());
            if (used >= 0) break;
            if (!strstr(cto_last_error(), "buffer too small")) break;
            alt.resize(alt.size() * 4);
        }
        if (used < 0) { fail(cto_last_error()); return false; }
        std::vector<char> centre(static_cast<size_t>(n));
        for (int64_t i = 0; i < n; ++i) {
            const int64_t at = int64_t(s->sites[size_t(i)]) - s->ref_start;
            const char c = at >= 0 && at < int64_t(s->ref.size()) ? s->ref[size_t(at)] : 'N';
            centre[size_t(i)] = c;
            if (c != 'A' && c != 'C' && c != 'G' && c != 'T') info[i * 12 + 3] |= 1;      // predict.py:219-228: centre not in ACGT -> no row
        }
        std::vector<char> text(size_t(512 * std::max<int64_t>(n, 1) + 2 * used + 4096));
        int64_t counts[4] = {0, 0, 0, 0};
        int64_t tu = -1;
        for (int tries = 0; tries < 6; ++tries) {
            tu = cto_vcf_rows_batch(j.ctg_name, n, s->sites.data(), centre.data(), alt.data(), alt_off.data(), info, h_decision,
                                    h_qual, cfg->K, cfg->show_ref, cfg->qual_pass, text.data(), text.size(), counts);
            if (tu != CTO_ENOMEM) break;
            text.resize(text.size() * 4);
        }
        if (tu < 0) { fail(cto_last_error()); return false; }
        if (counts[0] > 0) {              // the reference removes VCFs without records (call_variants.py:859-867)
            FILE* f = fopen(j.vcf_path, "w");
            if (!f) { fail(std::string("cannot write ") + j.vcf_path); return false; }
            const size_t hl = strlen(cfg->vcf_header);
            const bool ok = fwrite(cfg->vcf_header, 1, hl, f) == hl && fwrite(text.data(), 1, size_t(tu), f) == size_t(tu);
            if (fclose(f) != 0 || !ok) { fail(std::string("short write to ") + j.vcf_path); return false; }
        } else {
            (void)unlink(j.vcf_path);
        }
        if (cfg->verbose) {
            for (int64_t i = 0; i < counts[2]; ++i) puts("low tumor coverage");             // call_variants.py:328, one line per such site
            if (counts[3]) {
                const int32_t* dec = h_decision;
                for (int64_t i = 0; i < n; ++i)
                    if (dec[i * 4 + 1] & 3)
                        fprintf(stderr, "[WARNING] %s:%d a probability printed as 1.00000000 / 0.00000000 falls outside the likelihood bins (the "
                                "reference raises IndexError here); %s\n", j.ctg_name, s->sites[size_t(i)],
                                (dec[i * 4 + 1] & 2) ? "no posterior, site skipped" : "bin clamped");
            }
            fprintf(stderr, "[INFO] %s total processed positions: %lld\n", j.ctg_name, (long long)counts[1]);
        }
        rows += counts[0];
        sites += counts[1];
        low_cov += counts[2];
        clamped += counts[3];
        return true;
    }
};

}  // namespace

static int run_chunks(const cto_run_cfg* cfg, const cto_chunk_job* jobs, int64_t n_jobs, void* stream, cto_run_stats* stats) {
    CTO_REQUIRE(cfg && (jobs || n_jobs == 0) && cfg->aff && cfg->neg && cfg->d_lik && cfg->d_edges && cfg->ref_fa && cfg->vcf_header, CTO_EINVAL,
                "cto_run_chunks: null argument");
    CTO_REQUIRE(cfg->K == 4 || cfg->K == 6, CTO_EINVAL, "cto_run_chunks: K must be 4 or 6");
    for (int64_t i = 0; i < n_jobs; ++i)
        CTO_REQUIRE(jobs[i].ctg_name && jobs[i].vcf_path && (jobs[i].mpileup_path || jobs[i].bam_path) &&
                        (jobs[i].bed_path || (jobs[i].region_start >= 0 && jobs[i].region_end >= std::max<int64_t>(jobs[i].region_start, 1))),   // a first chunk of --chunk_id starts at 0 (:262)
                    CTO_EINVAL, "cto_run_chunks: job %lld is incomplete", (long long)i);
    if (stats) memset(stats, 0, sizeof(*stats));
    if (n_jobs == 0) return CTO_OK;
    const int producers = std::max(1, cfg->producers), writers = std::max(1, cfg->writers);
    const int depth = cfg->depth > 0 ? cfg->depth : producers + writers + 2;
    Run run;
    run.cfg = cfg;
    run.jobs = jobs;
    run.n_jobs = n_jobs;
    {
        std::string err;
        CTO_REQUIRE(run.fasta.open(cfg->ref_fa, &err), CTO_EINVAL, "cto_run_chunks: %s", err.c_str());
        CTO_REQUIRE(run.load_indel_regions(&err), CTO_EINVAL, "cto_run_chunks: %s", err.c_str());
    }
    int dev = 0;
    CTO_HIP(hipGetDevice(&dev));
    // slots (device + page-locked buffers, events) outlive the call: allocating and freeing ~100 MB of them per slot costs tens of
    // milliseconds, which a short chunk list would pay on every call; cto_run_release() frees them
    SlotReturn slots_back{&run.slots};
    {
        std::lock_guard<std::mutex> g(slot_cache_m());
        auto& cache = slot_cache();
        for (size_t i = 0; i < cache.size() && int(run.slots.size()) < depth;)
            if (cache[i]->device == dev) {
                run.slots.push_back(std::move(cache[i]));
                cache.erase(cache.begin() + long(i));
            } else {
                ++i;
            }
    }
    while (int(run.slots.size()) < depth) {
        run.slots.emplace_back(new Slot());
        run.slots.back()->device = dev;
        CTO_HIP(hipEventCreateWithFlags(&run.slots.back()->uploaded, hipEventDisableTiming));
        CTO_HIP(hipEventCreate(&run.slots.back()->begin));
        CTO_HIP(hipEventCreateWithFlags(&run.slots.back()->computed, hipEventDisableTiming));
        CTO_HIP(hipEventCreate(&run.slots.back()->kernels_end));
        CTO_HIP(hipEventCreateWithFlags(&run.slots.back()->done, hipEventBlockingSync));      // writers sleep, not spin, until their chunk is back
    }
    for (auto& sl : run.slots) run.free_slots.push(sl.get());
    // device-inflate contexts (BAM jobs only), kept across calls like the slots
    CtxReturn ctx_back{&run.inflate_ctx};
    bool any_bam = false;
    for (int64_t i = 0; i < n_jobs; ++i) any_bam = any_bam || !jobs[i].mpileup_path;
    if (any_bam && cfg->inflate_cus > 0 && cfg->inflate_jobs > 0) {
        int n_cu = 256;
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
        const int cus = std::max(8, std::min(cfg->inflate_cus, n_cu * 3 / 4));      // the networks keep at least a quarter of the chip
        {
            std::lock_guard<std::mutex> g(slot_cache_m());
            auto& cache = inflate_cache();
            for (size_t i = 0; i < cache.size() && int(run.inflate_ctx.size()) < cfg->inflate_jobs;)
                if (cache[i]->device == dev && cache[i]->cus == cus) {
                    run.inflate_ctx.push_back(std::move(cache[i]));
                    cache.erase(cache.begin() + long(i));
                } else {
                    ++i;
                }
        }
        while (int(run.inflate_ctx.size()) < cfg->inflate_jobs) {
            run.inflate_ctx.emplace_back(new InflateCtx());
            if (run.inflate_ctx.back()->open(dev, cus) != CTO_OK) {           // no CU-masked streams on this runtime: host inflate only
                if (cfg->verbose) fprintf(stderr, "[WARNING] device inflate disabled: %s\n", cto_last_error());
                run.inflate_ctx.clear();
                break;
            }
        }
        for (auto& c : run.inflate_ctx) run.free_ctx.push(c.get());
    }
    hipStream_t main = static_cast<hipStream_t>(stream), own_main = nullptr;
    if (!main) {
        // The legacy default stream synchronises with every blocking stream - the CU-masked inflate streams are such - and the networks
        // and the inflate launches would exclude each other in time (measured: BAM -> VCF 350 k instead of 400-490 k sites/s).  The
        // kernels get a non-blocking stream of this call, ordered behind what the caller has queued on the default stream so far.
        hipEvent_t before = nullptr;
        CTO_HIP(hipStreamCreateWithFlags(&own_main, hipStreamNonBlocking));
        CTO_HIP(hipEventCreateWithFlags(&before, hipEventDisableTiming));
        CTO_HIP(hipEventRecord(before, nullptr));
        CTO_HIP(hipStreamWaitEvent(own_main, before, 0));
        CTO_HIP(hipEventDestroy(before));
        main = own_main;
    }
    hipStream_t copy_back = nullptr;
    CTO_HIP(hipStreamCreateWithFlags(&copy_back, hipStreamNonBlocking));
    // a second compute stream for every other chunk, when the caller brought a second pair of handles (cto_run_cfg.aff2)
    hipStream_t second = nullptr;
    if (cfg->aff2 && cfg->neg2) {
        hipEvent_t before = nullptr;
        CTO_HIP(hipStreamCreateWithFlags(&second, hipStreamNonBlocking));
        CTO_HIP(hipEventCreateWithFlags(&before, hipEventDisableTiming));
        CTO_HIP(hipEventRecord(before, main));
        CTO_HIP(hipStreamWaitEvent(second, before, 0));
        CTO_HIP(hipEventDestroy(before));
    }
    const double t_begin = now_s();
    std::atomic<int> producers_left{producers};

    // The producers share TWO copy streams (CTO_COPY_STREAMS=n; 0: a stream each, as until the end of round 6).  The runtime maps a process's
    // streams onto four hardware queues; with a stream per producer the stream the networks run on shares its queue with one or two copy
    // streams, whose waits (a copy's completion) then stand in front of the networks' launches.  Two interleaved A/B runs of every file-to-file
    // leg: text -> VCF 1.97-1.98 -> 2.04-2.06 M sites/s, with the device tokeniser 1.85 -> 1.93-1.96 M, all-device BAM on two cores 0.56-0.58
    // -> 0.59-0.60 M, BAM -> VCF, REGION jobs and 10 000-site chunks unchanged.  A producer only queues on its stream and waits on its own
    // events, so what another producer queues in between costs it little.
    static const int shared_n = [] { const char* e = getenv("CTO_COPY_STREAMS"); return e ? atoi(e) : 2; }();
    struct CopyStreams {                     // (declared before the threads: destroyed after they are joined)
        std::vector<hipStream_t> v;
        ~CopyStreams() { for (hipStream_t c : v) { (void)hipStreamSynchronize(c); (void)hipStreamDestroy(c); } }
    } copy_streams;
    for (int i = 0; i < shared_n; ++i) {
        hipStream_t c = nullptr;
        CTO_HIP(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
        copy_streams.v.push_back(c);
    }
    const std::vector<hipStream_t>& shared_copy = copy_streams.v;
    std::vector<std::thread> threads;
    threads.reserve(size_t(producers + writers));
    struct JoinAll {                         // whatever happens below, a started thread is joined (queues closed first: they all wake up)
        Run& run;
        std::vector<std::thread>& th;
        ~JoinAll() {
            run.to_launch.close();
            run.to_write.close();
            run.free_slots.close();
            for (auto& t : th)
                if (t.joinable()) t.join();
        }
    } join_all{run, threads};
    auto start = [&](auto&& body) -> bool {
        try {
            threads.emplace_back(std::forward<decltype(body)>(body));
            return true;
        } catch (const std::exception& e) {      // the system is out of threads
            run.fail(std::string("cannot start a thread: ") + e.what());
            return false;
        }
    };
    for (int t = 0; t < producers; ++t)
        if (!start([&run, &producers_left, dev, t, &shared_copy] {
            hipStream_t copy = nullptr;
            const bool shared = !shared_copy.empty();
            tl_pack_threads = run.cfg->pack_threads;                 // the tokeniser's / BAM decoder's own threads per call
            if (shared) { (void)hipSetDevice(dev); copy = shared_copy[size_t(t) % shared_copy.size()]; }
            else if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&copy, hipStreamNonBlocking) != hipSuccess) run.fail("producer: no HIP stream");
            for (;;) {
                if (run.failed) break;
                const int64_t j = run.next_job.fetch_add(1);
                if (j >= run.n_jobs) break;
                Slot* s = nullptr;
                if (!run.free_slots.pop(&s)) break;
                if (run.failed) {                                    // the run failed while this thread waited for a slot: what the slot's
                    run.free_slots.push(s);                          // last chunk queued on the device may still be running - leave it alone
                    break;
                }
                s->job = j;
                const double t0 = now_s();
                bool ok = false;
                try {
                    ok = copy && run.produce(s, copy);
                } catch (const std::exception& e) {                  // bad_alloc on a huge chunk: an error of the run, not of the process
                    run.fail(std::string("producer: ") + e.what());
                }
                { std::lock_guard<std::mutex> g(run.stat_m); run.produce_s += now_s() - t0; }
                if (ok) run.to_launch.push(s);
                else run.free_slots.push(s);                         // nothing to call here (or an error: `failed` is set)
            }
            if (copy) { (void)hipStreamSynchronize(copy); if (!shared) (void)hipStreamDestroy(copy); }
            if (--producers_left == 0) run.to_launch.close();
        })) {
            if (--producers_left == 0) run.to_launch.close();        // this one never ran
        }
    for (int t = 0; t < writers; ++t)
        (void)start([&run, dev] {
            (void)hipSetDevice(dev);
            Slot* s = nullptr;
            while (run.to_write.pop(&s)) {
                const double t0 = now_s();
                try {
                    if (!run.failed) run.finish(s);
                } catch (const std::exception& e) {
                    run.fail(std::string("writer: ") + e.what());
                }
                { std::lock_guard<std::mutex> g(run.stat_m); run.finish_s += now_s() - t0; }
                run.free_slots.push(s);
            }
        });
    if (run.failed) run.free_slots.close();  // a thread did not start: nobody may wait for a slot that no writer will hand back
    // ---- launcher: this thread ----
    double launch_s = 0, wait_s = 0;
    int rc = CTO_OK;
    int64_t launched = 0;
    // the networks take a stream of sites, not chunks (Run::launch_stream), unless two compute streams take the chunks in turn or the
    // caller turns it off (CTO_TILE_STREAM=0: every chunk is its own launch, as before)
    static const bool stream_off = [] { const char* e = getenv("CTO_TILE_STREAM"); return e && e[0] == '0'; }();
    const bool tile_stream = !second && !stream_off;
    static const bool fused_off = [] { const char* e = getenv("CTO_FUSED_FEATURIZE"); return e && e[0] == '0'; }();
    run.fused_featurize = !fused_off;
    if (tile_stream) {
        int n_cu = 256;
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
        run.round_sites = int64_t(16) * std::max(n_cu, 1);
        // chunks that may wait for rows.  Each holds a slot, and a chunk only stops waiting when a LATER chunk is launched: with every
        // slot waiting no producer could ever deliver that chunk, so at most depth - 1 wait (the launch that would make it `depth`
        // takes everything that is pending instead)
        run.max_pending = size_t(std::max(0, std::min(4, depth - 1)));
        (void)hipEventCreate(&run.flush_begin);
        (void)hipEventCreate(&run.flush_end);
    }
    std::vector<Slot*> complete;
    auto hand_over = [&] {
        for (Slot* c : complete) {
            if (run.failed) run.free_slots.push(c); else run.to_write.push(c);
        }
        complete.clear();
    };
    auto abandon_pending = [&] {             // a failed run: the chunks still waiting for rows give their slots back
        if (!run.pending.empty()) (void)hipStreamSynchronize(main);
        for (const Run::Pending& q : run.pending) run.free_slots.push(q.s);
        run.pending.clear();
    };
    for (;;) {
        Slot* s = nullptr;
        const double t0 = now_s();
        if (!run.to_launch.pop(&s)) break;
        const double t1 = now_s();
        wait_s += t1 - t0;
        bool queued = false;
        if (!run.failed) {
            int r;
            if (tile_stream) {
                r = run.launch_stream(s, main, copy_back, cfg->aff, cfg->neg, &complete);
                queued = r == CTO_OK || std::any_of(run.pending.begin(), run.pending.end(), [s](const Run::Pending& q) { return q.s == s; });
            } else {
                const bool odd = second && (launched++ & 1);
                r = run.launch(s, odd ? second : main, copy_back, odd ? cfg->aff2 : cfg->aff, odd ? cfg->neg2 : cfg->neg);
                if (r == CTO_OK) { complete.push_back(s); queued = true; }
            }
            if (r != CTO_OK) { run.fail(cto_last_error()); rc = r; }
        }
        launch_s += now_s() - t1;
        if (!queued) {
            (void)hipStreamSynchronize(main);                         // its buffers may be in use by what was queued before the failure
            run.free_slots.push(s);
        }
        hand_over();
        if (run.failed) abandon_pending();
    }
    if (tile_stream && !run.failed) {
        const double t1 = now_s();
        const int r = run.flush_stream(main, copy_back, cfg->aff, cfg->neg, &complete);
        if (r != CTO_OK) { run.fail(cto_last_error()); rc = r; }
        launch_s += now_s() - t1;
        hand_over();
    }
    if (run.failed) abandon_pending();
    run.to_write.close();
    run.free_slots.close();
    for (auto& th : threads) th.join();
    (void)hipStreamSynchronize(main);
    if (second) { (void)hipStreamSynchronize(second); (void)hipStreamDestroy(second); }
    (void)hipStreamSynchronize(copy_back);
    (void)hipStreamDestroy(copy_back);
    if (run.flush_timed) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, run.flush_begin, run.flush_end) == hipSuccess) run.device_s += ms * 1e-3;
    }
    if (run.flush_begin) (void)hipEventDestroy(run.flush_begin);
    if (run.flush_end) (void)hipEventDestroy(run.flush_end);
    if (own_main) (void)hipStreamDestroy(own_main);
    if (stats) {
        stats->candidates = run.candidates;
        stats->sites = run.sites;
        stats->rows = run.rows;
        stats->low_coverage = run.low_cov;
        stats->clamped = run.clamped;
        stats->seconds = now_s() - t_begin;
        stats->produce_s = run.produce_s;
        stats->pack_s = run.pack_s;
        stats->upload_s = run.upload_s;
        stats->device_s = run.device_s;
        stats->device_inflated = run.device_inflated;
        stats->device_piled = run.device_piled;
        stats->device_tokenised = run.device_tokenised;
        stats->launch_s = launch_s;
        stats->launcher_wait_s = wait_s;
        stats->finish_s = run.finish_s;
    }
    if (run.failed) {
        set_error("cto_run_chunks: %s", run.first_error.c_str());
        return rc != CTO_OK ? rc : CTO_EINVAL;
    }
    return CTO_OK;
}

// no C++ exception crosses the C ABI: what the set-up or the launcher thread throws (allocation failure) becomes an error code; the
// worker threads catch their own
extern "C" int cto_run_chunks(const cto_run_cfg* cfg, const cto_chunk_job* jobs, int64_t n_jobs, void* stream, cto_run_stats* stats) {
    try {
        return run_chunks(cfg, jobs, n_jobs, stream, stats);
    } catch (const std::bad_alloc&) {
        set_error("cto_run_chunks: out of memory");
        return CTO_ENOMEM;
    } catch (const std::exception& e) {
        set_error("cto_run_chunks: %s", e.what());
        return CTO_EINVAL;
    }
}

extern "C" int cto_run_release(void) {
    std::vector<std::unique_ptr<Slot>> drop;
    std::vector<std::unique_ptr<InflateCtx>> drop_ctx;
    {
        std::lock_guard<std::mutex> g(slot_cache_m());
        drop.swap(slot_cache());
        drop_ctx.swap(inflate_cache());
    }
    int cur = 0;
    CTO_HIP(hipGetDevice(&cur));
    for (auto& sl : drop) {
        CTO_HIP(hipSetDevice(sl->device));
        sl.reset();
    }
    for (auto& c : drop_ctx) {
        CTO_HIP(hipSetDevice(c->device));
        c.reset();
    }
    CTO_HIP(hipSetDevice(cur));
    return CTO_OK;
}
