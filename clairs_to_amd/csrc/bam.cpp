// BAM -> column pack producer: what `samtools mpileup --reverse-del --output-MQ -r ctg:s-e --min-MQ 0
// --min-BQ 0 -l <bed> --excl-flags 2316 [--max-depth N]` followed by the text tokeniser would yield, without the text.
//
// PARITY UNPINNED: neither samtools nor htslib exists on the build or GPU boxes, so this reader is validated only against
// an independent restatement of the pileup rules below on BAM files written by the test-suite itself
// (tests/bamutil.py, tests/test_bam_reader.py).  The drivers keep `samtools mpileup` as their default producer.
//
// Pileup rules implemented (SAM/BAM specification v1 sections 4.2, 5.1-5.3 for the formats; samtools-mpileup(1) and
// SURVEY.md Appendix B for the column semantics):
//   * a record is used if it is mapped to the region's reference, (flag & excl_flags) == 0, MAPQ >= min_mq, has a CIGAR
//     and SEQ, and - as mpileup does without -A - is not an "orphan" (PAIRED set without PROPER_PAIR);
//   * reads enter a column in file order (= coordinate order, ties by file position);
//   * M / = / X : one read-base per reference position, base letter upper case on the forward strand, lower case on the
//     reverse strand ('=' resolves to the reference base), BQ = QUAL at that query position;
//   * D : placeholder '*' (forward) / '#' (reverse, --reverse-del) at every deleted position, carrying the BQ of the
//     query base that follows the deletion (0 past the end of the read);
//   * the last aligned base before an I (or D) carries the indel: inserted bases in the strand's case, or len x 'N'/'n'
//     for a deletion (no -f: deleted bases print as N); an insertion that is not preceded by an aligned base of the
//     same read (start of read, after a clip / skip / deletion) is not reported;
//   * N (reference skip) contributes nothing (samtools prints '>' / '<', which the reference's decoder ignores);
//   * S / H / P consume as the specification says and contribute nothing;
//   * MQ and BQ are capped at 93, the largest value mpileup's phred+33 characters can carry;
//   * a read is dropped when `max_depth` reads are already active at its start (htslib's per-file maxcnt); the outcome does
//     not depend on the number of decoding threads (a call whose ranges hit the cap is redone unsplit);
//   * rows exist only for positions covered by >= 1 read-base or placeholder, inside [start, end] and inside the BED
//     intervals when given.
//   * read-pair overlaps (mpileup without -x, as the reference runs it): where both mates of a pair have an aligned base at a
//     position, the first mate's base keeps min(200, qa + qb) when they agree and the better base keeps 0.8 x its quality
//     when they differ; the other base's quality becomes 0 (soften_overlap below).  Paired-end short reads only.  WHICH mate
//     keeps the base when they agree is a second unpinned point (advisor, round 2): recent htslib releases may pick it per read
//     name instead of always favouring the first - with no htslib on either box this cannot be settled here, so for paired-end
//     input `samtools` stays the reference producer and `--bam_reader native|gpu` is offered for the long-read platforms.
// Not implemented (documented deviations): BAQ (needs -f, which the reference does not pass), CRAM, multi-file input.
//
// I/O: the file is mapped; every BGZF block is inflated (libdeflate when the runtime library is present, else zlib) and checked
// against the CRC-32 of its gzip trailer, as htslib does.  Blocks a caller inflated elsewhere (cto_pack_from_bam_inflated: on the
// device, csrc/inflate.hip) are looked up by file offset and CRC-checked the same way.  Columns are built in runs of requested
// positions, read by read (Producer below).
#include <dlfcn.h>
#include <zlib.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <memory>
#include <thread>

#include "pack_internal.h"

using namespace cto;

namespace {

// ------------------------------------------------------------------------------------------------ BGZF
// libdeflate inflates BGZF blocks 2-3x faster than zlib.  Its headers are not installed here, only the runtime library, so
// the three entry points of its stable v1 ABI are resolved with dlopen; zlib remains the fallback.
struct LibDeflate {
    void* h = nullptr;
    void* (*alloc)() = nullptr;
    int (*inflate)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    uint32_t (*crc)(uint32_t, const void*, size_t) = nullptr;
    void (*release)(void*) = nullptr;
    LibDeflate() {
        if (getenv("CTO_NO_LIBDEFLATE")) return;
        for (const char* name : {"libdeflate.so.0", "libdeflate.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) return;
        alloc = reinterpret_cast<void* (*)()>(dlsym(h, "libdeflate_alloc_decompressor"));
        inflate = reinterpret_cast<int (*)(void*, const void*, size_t, void*, size_t, size_t*)>(dlsym(h, "libdeflate_deflate_decompress"));
        release = reinterpret_cast<void (*)(void*)>(dlsym(h, "libdeflate_free_decompressor"));
        crc = reinterpret_cast<uint32_t (*)(uint32_t, const void*, size_t)>(dlsym(h, "libdeflate_crc32"));
        if (!alloc || !inflate || !release) { alloc = nullptr; inflate = nullptr; release = nullptr; }
    }
    bool ok() const { return inflate != nullptr; }
};
const LibDeflate& libdeflate();
// CRC-32 of a BGZF block's inflated bytes (the gzip trailer holds the expected value; htslib checks it too)
uint32_t block_crc(const uint8_t* p, size_t n);

const LibDeflate& libdeflate() {
    static const LibDeflate ld;
    return ld;
}

uint32_t block_crc(const uint8_t* p, size_t n) {
    if (libdeflate().crc) return libdeflate().crc(0, p, n);
    return uint32_t(crc32(crc32(0L, Z_NULL, 0), p, uInt(n)));
}

// BGZF blocks that were inflated elsewhere (on the device: cto_bgzf_inflate), looked up by their file offset
struct PreInflated {
    const uint8_t* data = nullptr;
    const cto_bgzf_block* blocks = nullptr;     // sorted by file_off
    int64_t n = 0;
    const cto_bgzf_block* find(int64_t coff) const {
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) / 2;
            if (int64_t(blocks[mid].file_off) < coff) lo = mid + 1; else hi = mid;
        }
        return (lo < n && int64_t(blocks[lo].file_off) == coff) ? blocks + lo : nullptr;
    }
};

struct Bgzf {
    const uint8_t* map = nullptr;       // the BAM file, mapped: blocks are inflated straight out of the page cache
    int64_t fsize = 0;
    std::vector<uint8_t> block;         // inflated current block (when it was inflated here)
    const uint8_t* bptr = nullptr;      // the current block's inflated bytes: block.data() or a slot of `pre`
    size_t blen = 0;
    PreInflated pre;
    int64_t block_coffset = -1;         // file offset of the current block
    int64_t next_coffset = 0;           // file offset of the block after it
    size_t upos = 0;                    // read position inside `block`
    z_stream zs;
    bool zs_init = false;
    void* ld = nullptr;                 // libdeflate decompressor when available
    std::string err;

    ~Bgzf() {
        if (zs_init) inflateEnd(&zs);
        if (ld) libdeflate().release(ld);
        if (map && fsize > 0) munmap(const_cast<uint8_t*>(map), size_t(fsize));
    }
    bool open(const char* path) {
        const int fd = ::open(path, O_RDONLY | O_CLOEXEC);
        if (fd < 0) { err = std::string("cannot open ") + path; return false; }
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd); err = std::string(path) + " is not a regular file"; return false; }
        fsize = int64_t(st.st_size);
        if (fsize > 0) {
            void* m = mmap(nullptr, size_t(fsize), PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { ::close(fd); fsize = 0; err = std::string("cannot map ") + path; return false; }
            map = static_cast<const uint8_t*>(m);
        }
        ::close(fd);
        memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, -15) != Z_OK) { err = "inflateInit2 failed"; return false; }
        zs_init = true;
        if (libdeflate().ok()) ld = libdeflate().alloc();
        return true;
    }
    // loads the block that starts at file offset `coff`; false at EOF (err stays empty) or on error
    bool load(int64_t coff) {
        if (const cto_bgzf_block* pb = pre.find(coff)) {        // already inflated: a view, no file access
            bptr = pre.data + pb->out_off;
            blen = pb->isize;
            if (blen && block_crc(bptr, blen) != pb->crc32) { err = "BGZF block fails its CRC-32"; return false; }
            block_coffset = coff;
            next_coffset = coff + int64_t(pb->bsize);
            upos = 0;
            return true;
        }
        if (coff < 0 || coff > fsize) { err = "seek failed"; return false; }
        if (coff == fsize) return false;   // clean EOF
        const uint8_t* h = map + coff;
        const int64_t left = fsize - coff;
        if (left < 18 || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { err = "not a BGZF block header"; return false; }
        const int xlen = h[10] | (h[11] << 8);
        if (left < 12 + int64_t(xlen)) { err = "truncated BGZF extra field"; return false; }
        // the BC subfield is normally first; scan the extra field in general
        const uint8_t* extra = h + 12;
        int bsize = -1;
        for (int i = 0; i + 4 <= xlen;) {
            const int slen = extra[size_t(i) + 2] | (extra[size_t(i) + 3] << 8);
            if (extra[size_t(i)] == 'B' && extra[size_t(i) + 1] == 'C' && slen == 2 && i + 6 <= xlen)
                bsize = (extra[size_t(i) + 4] | (extra[size_t(i) + 5] << 8)) + 1;
            i += 4 + slen;
        }
        if (bsize < 0) { err = "BGZF block without BC subfield"; return false; }
        const int cdata = bsize - xlen - 12 - 8;     // deflate payload; then CRC32 + ISIZE
        if (cdata < 0) { err = "bad BGZF block size"; return false; }
        if (left < int64_t(bsize)) { err = "truncated BGZF block"; return false; }
        const uint8_t* payload = h + 12 + xlen;
        const uint8_t* tail = payload + cdata;
        const uint32_t isize = uint32_t(tail[4]) | (uint32_t(tail[5]) << 8) | (uint32_t(tail[6]) << 16) | (uint32_t(tail[7]) << 24);
        if (isize > 65536) { err = "BGZF block claims more than 64 KiB of data"; return false; }     // the format's limit
        block.resize(isize);
        if (isize && ld) {
            size_t got_out = 0;
            if (libdeflate().inflate(ld, payload, size_t(cdata), block.data(), isize, &got_out) != 0 || got_out != isize) {
                err = "inflate failed";
                return false;
            }
        } else if (isize) {
            inflateReset(&zs);
            zs.next_in = const_cast<uint8_t*>(payload);
            zs.avail_in = uInt(cdata);
            zs.next_out = block.data();
            zs.avail_out = uInt(isize);
            const int rc = inflate(&zs, Z_FINISH);
            if (rc != Z_STREAM_END || zs.avail_out != 0) { err = "inflate failed"; return false; }
        }
        if (isize) {
            const uint32_t want = uint32_t(tail[0]) | (uint32_t(tail[1]) << 8) | (uint32_t(tail[2]) << 16) | (uint32_t(tail[3]) << 24);
            if (block_crc(block.data(), isize) != want) { err = "BGZF block fails its CRC-32"; return false; }
        }
        bptr = block.data();
        blen = block.size();
        block_coffset = coff;
        next_coffset = coff + bsize;
        upos = 0;
        return true;
    }
    bool seek(uint64_t voff) {
        const int64_t coff = int64_t(voff >> 16);
        if (coff != block_coffset && !load(coff)) return false;
        upos = size_t(voff & 0xffff);
        return upos <= blen;
    }
    // virtual offset of the next byte; the end of a block is reported as the start of the next one, as index chunks do
    uint64_t tell() const {
        if (block_coffset >= 0 && upos >= blen) return uint64_t(next_coffset) << 16;
        return (uint64_t(block_coffset) << 16) | uint64_t(upos);
    }
    // reads exactly n bytes across block boundaries; false at EOF / error
    bool read(void* dst, size_t n) {
        uint8_t* d = static_cast<uint8_t*>(dst);
        while (n > 0) {
            if (block_coffset < 0 || upos >= blen) {
                if (!load(block_coffset < 0 ? 0 : next_coffset)) return false;
                if (blen == 0) continue;              // empty blocks (e.g. the EOF marker) are skipped
            }
            const size_t take = std::min(n, blen - upos);
            memcpy(d, bptr + upos, take);
            upos += take;
            d += take;
            n -= take;
        }
        return true;
    }
};

inline int32_t le32(const uint8_t* p) { return int32_t(uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24)); }
inline uint64_t le64(const uint8_t* p) { return uint64_t(uint32_t(le32(p))) | (uint64_t(uint32_t(le32(p + 4))) << 32); }

// ------------------------------------------------------------------------------------------------ BAI
struct Chunk { uint64_t beg, end; };

// bins that may hold alignments overlapping [beg, end) (0-based), SAM specification section 5.3
void reg2bins(int64_t beg, int64_t end, std::vector<uint32_t>* bins) {
    --end;
    bins->push_back(0);
    for (int k = 1 + int(beg >> 26); k <= 1 + int(end >> 26); ++k) bins->push_back(uint32_t(k));
    for (int k = 9 + int(beg >> 23); k <= 9 + int(end >> 23); ++k) bins->push_back(uint32_t(k));
    for (int k = 73 + int(beg >> 20); k <= 73 + int(end >> 20); ++k) bins->push_back(uint32_t(k));
    for (int k = 585 + int(beg >> 17); k <= 585 + int(end >> 17); ++k) bins->push_back(uint32_t(k));
    for (int k = 4681 + int(beg >> 14); k <= 4681 + int(end >> 14); ++k) bins->push_back(uint32_t(k));
}

// chunks of reference `tid` that may overlap [beg, end), merged and sorted; false on a malformed index
bool bai_query(const char* path, int tid, int64_t beg, int64_t end, std::vector<Chunk>* out, std::string* err,
               std::vector<uint64_t>* linear = nullptr) {
    FILE* f = fopen(path, "rb");
    if (!f) { *err = std::string("cannot open index ") + path; return false; }
    std::vector<uint8_t> buf;
    off_t sz = -1;
    if (fseeko(f, 0, SEEK_END) == 0) sz = ftello(f);
    if (sz < 0 || sz > (off_t(1) << 32) || fseeko(f, 0, SEEK_SET) != 0) {       // not seekable (a pipe, a directory) or absurdly large
        fclose(f);
        *err = std::string("cannot read index ") + path;
        return false;
    }
    buf.resize(size_t(sz));
    const bool ok = fread(buf.data(), 1, buf.size(), f) == buf.size();
    fclose(f);
    if (!ok || buf.size() < 8 || memcmp(buf.data(), "BAI\1", 4) != 0) { *err = "not a BAI index"; return false; }
    size_t o = 4;
    auto need = [&](size_t n) { return o + n <= buf.size(); };
    const int n_ref = le32(buf.data() + o); o += 4;
    if (tid < 0 || tid >= n_ref) { *err = "reference not in the index"; return false; }
    std::vector<uint32_t> want;
    reg2bins(beg, end, &want);
    std::sort(want.begin(), want.end());
    std::vector<Chunk> chunks;
    uint64_t min_off = 0;
    for (int r = 0; r <= tid; ++r) {
        if (!need(4)) { *err = "truncated BAI"; return false; }
        const int n_bin = le32(buf.data() + o); o += 4;
        if (n_bin < 0) { *err = "malformed BAI"; return false; }
        for (int b = 0; b < n_bin; ++b) {
            if (!need(8)) { *err = "truncated BAI"; return false; }
            const uint32_t bin = uint32_t(le32(buf.data() + o));
            const int n_chunk = le32(buf.data() + o + 4);
            o += 8;
            if (n_chunk < 0) { *err = "malformed BAI"; return false; }
            if (!need(size_t(n_chunk) * 16)) { *err = "truncated BAI"; return false; }
            if (r == tid && bin != 37450 && std::binary_search(want.begin(), want.end(), bin))
                for (int c = 0; c < n_chunk; ++c) chunks.push_back(Chunk{le64(buf.data() + o + size_t(c) * 16), le64(buf.data() + o + size_t(c) * 16 + 8)});
            o += size_t(n_chunk) * 16;
        }
        if (!need(4)) { *err = "truncated BAI"; return false; }
        const int n_intv = le32(buf.data() + o); o += 4;
        if (n_intv < 0) { *err = "malformed BAI"; return false; }
        if (!need(size_t(n_intv) * 8)) { *err = "truncated BAI"; return false; }
        if (r == tid && n_intv > 0) {
            const int64_t w = std::min<int64_t>(beg >> 14, n_intv - 1);
            min_off = le64(buf.data() + o + size_t(w) * 8);
            if (linear) {
                linear->resize(size_t(n_intv));
                for (int i = 0; i < n_intv; ++i) (*linear)[size_t(i)] = le64(buf.data() + o + size_t(i) * 8);
            }
        }
        o += size_t(n_intv) * 8;
    }
    std::sort(chunks.begin(), chunks.end(), [](const Chunk& a, const Chunk& b) { return a.beg < b.beg; });
    for (const Chunk& c : chunks) {
        if (c.end <= min_off) continue;                       // entirely before the first alignment that can overlap
        Chunk d{std::max(c.beg, min_off), c.end};
        if (!out->empty() && d.beg <= out->back().end) out->back().end = std::max(out->back().end, d.end);
        else out->push_back(d);
    }
    return true;
}

// ------------------------------------------------------------------------------------------------ records + pileup
struct Read {
    int32_t pos = 0;            // 0-based leftmost
    int32_t end = 0;            // 0-based exclusive reference end
    uint8_t mapq = 0;
    bool rev = false;
    std::vector<uint32_t> cigar;
    std::vector<uint8_t> raw;   // the alignment record as read; packed SEQ (4 bits per base) and QUAL are decoded on demand, since a
                                // BED-restricted pileup touches a small part of a long read
    int32_t l_seq = 0;
    bool no_qual = false;
    std::string mate_key;       // QNAME of a paired read (empty otherwise): mates find each other through it
    size_t seq_off = 0, qual_off = 0;   // where SEQ and QUAL start in `raw` (the whole record is kept: no second copy)
    int base4(int q) const { return (raw[seq_off + size_t(q >> 1)] >> ((~q & 1) << 2)) & 15; }
    int bq(int q) const { return (no_qual || q >= l_seq) ? 0 : std::min(int(raw[qual_off + size_t(q)]), 93); }
    uint8_t* qual_at(int q) { return &raw[qual_off + size_t(q)]; }
    // query index of the aligned base (M / = / X) at 0-based reference position rpos, or -1 (deletion, skip, outside the read)
    int query_at(int32_t rpos) const {
        int32_t rp = pos, qp = 0;
        for (uint32_t c : cigar) {
            const int len = int(c >> 4), opc = int(c & 15);
            const bool cons_ref = opc == 0 || opc == 2 || opc == 3 || opc == 7 || opc == 8;
            const bool cons_q = opc == 0 || opc == 1 || opc == 4 || opc == 7 || opc == 8;
            if (cons_ref && rpos < rp + len) return (cons_q && rpos >= rp) ? qp + (rpos - rp) : -1;
            if (cons_ref) rp += len;
            if (cons_q) qp += len;
        }
        return -1;
    }
    // cursor: CIGAR op index, offset inside it, reference / query positions at the start of the op
    size_t op = 0;
    int32_t op_ref = 0, op_q = 0;
    // reference positions op_ref <= rpos < fast_end lie inside the current op, which is an aligned run (M / = / X), and are not
    // its last base (after which an indel may follow): the column loop packs them without looking at the CIGAR
    int32_t fast_end = INT32_MIN;
};

// entry code of a 4-bit BAM base on the forward strand: A C G T -> 0..3, '=' -> -1 (take the reference base), IUPAC codes -> N
const int8_t kNibCode[16] = {-1, 0, 1, 10, 2, 10, 10, 10, 3, 10, 10, 10, 10, 10, 10, 10};

const char kNt16[] = "=ACMGRSVTWYHKDBN";

// Read-pair overlap handling of `samtools mpileup` (on unless -x / --ignore-overlaps-removal; the reference never passes -x,
// SURVEY.md App. B): where the two mates of a pair cover the same reference position with an aligned base each, one base is
// kept and the other nullified by a base quality of 0 (samtools-mpileup(1), "--ignore-overlaps-removal"); the kept base gets
// the sum of both qualities (capped at 200) when the mates agree, 0.8 x the larger quality when they differ (htslib's
// tweak_overlap_quality).  `a` is the mate that entered the pileup first.  A quality-0 base still prints under --min-BQ 0 - it
// is the AFF pass's --min_bq and the LBQ channels that see the difference.  Long reads are unpaired: nothing happens.
void soften_overlap(Read& a, Read& b) {
    if (a.no_qual || b.no_qual) return;
    const int32_t lo = std::max(a.pos, b.pos), hi = std::min(a.end, b.end);
    for (int32_t rp = lo; rp < hi; ++rp) {
        const int qa = a.query_at(rp), qb = b.query_at(rp);
        if (qa < 0 || qb < 0) continue;
        uint8_t* pa = a.qual_at(qa);
        uint8_t* pb = b.qual_at(qb);
        if (a.base4(qa) == b.base4(qb)) {
            const int s = int(*pa) + int(*pb);
            *pa = uint8_t(s > 200 ? 200 : s);
            *pb = 0;
        } else if (*pa >= *pb) {
            *pa = uint8_t(0.8 * *pa);
            *pb = 0;
        } else {
            *pb = uint8_t(0.8 * *pb);
            *pa = 0;
        }
    }
}

struct Producer {
    cto_pack* p;
    ColumnScratch sc;
    const char* ref_seq;
    int64_t ref_start;
    size_t ref_len;
    int max_indel;
    // Columns are built in runs of up to RUN consecutive requested positions, read by read: a read's bases inside an aligned
    // CIGAR run go to their columns in one tight loop (sequence, qualities and cursor stay in cache), and the order of the
    // entries inside a column is still the order of the reads in `active` (= the order mpileup prints them in).
    static constexpr int RUN = 64;
    std::vector<uint32_t> ents[RUN];    // per column of the current run: packed entries, indel fields still empty
    std::vector<IndelAt> indels[RUN];   // ... and which of them carry an indel
    std::deque<std::string> arena;      // inserted sequences of the current run (IndelAt::seq points into these)
    std::string nbuf_up, nbuf_lo;       // runs of 'N' / 'n' for deletion keys
    std::string err;

    void begin_run(int ncol) {
        for (int c = 0; c < ncol; ++c) { ents[c].clear(); indels[c].clear(); }
        if (!arena.empty()) arena.clear();
    }
    // deletion keys are runs of 'N' / 'n' out of two shared buffers that emit_one() grows on demand
    void seal_column(int c) {
        for (IndelAt& it : indels[c])
            if (it.kind == 2) {
                const uint32_t code = ents[c][size_t(it.idx)] & 15u;
                const bool rev = (code >= 4 && code <= 7) || code == 9 || code == 11;
                it.seq = (rev ? nbuf_lo : nbuf_up).data();
            }
    }
    // the read's contribution to the columns of reference positions lo .. hi (0-based, inside the read); run0 = position of column 0
    void emit_run(Read& r, int32_t lo, int32_t hi, int32_t run0) {
        const int mq = std::min(int(r.mapq), 93);              // mpileup prints min(MAPQ, 93) + 33, likewise for BQ
        int32_t rp = lo;
        while (rp <= hi) {
            if (rp < r.fast_end) {                             // inside an aligned run, not its last base: no CIGAR work
                const int32_t stop = std::min(hi + 1, r.fast_end);
                int q = r.op_q + (rp - r.op_ref);
                for (; rp < stop; ++rp, ++q) ents[rp - run0].push_back(pack_entry(code_at(r, q, rp), r.bq(q), mq));
            } else {
                emit_one(r, rp, rp - run0, mq);
                ++rp;
            }
        }
    }
    int code_at(const Read& r, int q, int32_t rpos) const {
        int code = kNibCode[r.base4(q)];
        if (code < 0) {                                        // '=': the reference base (mpileup prints IUPAC codes; the decoder
            const int64_t ri = int64_t(rpos) + 1 - ref_start;   // ignores all but ACGTN)
            const char b = (ri >= 0 && size_t(ri) < ref_len) ? up(ref_seq[ri]) : 'N';
            code = b == 'A' ? 0 : (b == 'C' ? 1 : (b == 'G' ? 2 : (b == 'T' ? 3 : 10)));
        }
        return r.rev ? code + ((code < 4) ? 4 : 1) : code;     // A..T -> a..t, N -> n
    }
    // advances the read's cursor to reference position `rpos` (0-based) and appends its contribution to column `col`
    void emit_one(Read& r, int32_t rpos, int col, int mq) {
        std::vector<uint32_t>& ents = this->ents[col];
        std::vector<IndelAt>& indels = this->indels[col];
        while (r.op < r.cigar.size()) {
            const uint32_t c = r.cigar[r.op];
            const int len = int(c >> 4), opc = int(c & 15);
            const bool cons_ref = opc == 0 || opc == 2 || opc == 3 || opc == 7 || opc == 8;
            const bool cons_q = opc == 0 || opc == 1 || opc == 4 || opc == 7 || opc == 8;
            if (cons_ref && rpos < r.op_ref + len) break;
            if (cons_ref) r.op_ref += len;
            if (cons_q) r.op_q += len;
            ++r.op;
        }
        r.fast_end = INT32_MIN;
        if (r.op >= r.cigar.size()) return;
        const uint32_t c = r.cigar[r.op];
        const int len = int(c >> 4), opc = int(c & 15);
        const int off = rpos - r.op_ref;
        if (opc == 3) return;                                  // N: reference skip, nothing in the pack
        if (opc == 2) {                                        // D: placeholder
            ents.push_back(pack_entry(r.rev ? 9 : 8, r.bq(r.op_q), mq));
            return;
        }
        // aligned base
        r.fast_end = r.op_ref + len - 1;
        const int q = r.op_q + off;
        ents.push_back(pack_entry(code_at(r, q, rpos), r.bq(q), mq));
        if (off == len - 1) {                                  // last base of the op: does an indel follow?
            size_t nx = r.op + 1;
            while (nx < r.cigar.size() && (r.cigar[nx] & 15) == 6) ++nx;    // P
            if (nx < r.cigar.size()) {
                const int nop = int(r.cigar[nx] & 15), nlen = int(r.cigar[nx] >> 4);
                if (nop == 1) {
                    arena.emplace_back();
                    std::string& s = arena.back();
                    for (int i = 0; i < nlen; ++i) {
                        char ib = kNt16[r.base4(q + 1 + i)];
                        if (ib == '=') ib = 'N';
                        s.push_back(r.rev ? char(ib | 0x20) : ib);
                    }
                    indels.push_back(IndelAt{int(ents.size()) - 1, 1, s.data(), nlen});
                } else if (nop == 2) {
                    std::string& nb = r.rev ? nbuf_lo : nbuf_up;      // may grow again within this column: the pointer is set
                    if (int(nb.size()) < nlen) nb.assign(size_t(nlen), r.rev ? 'n' : 'N');      // by seal_column()
                    indels.push_back(IndelAt{int(ents.size()) - 1, 2, nullptr, nlen});
                }
            }
        }
    }
};

bool in_bed(const int64_t* bed, int64_t n_bed, int64_t pos1, int64_t* cursor) {
    if (!bed) return true;
    const int64_t p0 = pos1 - 1;
    while (*cursor < n_bed && bed[2 * *cursor + 1] <= p0) ++*cursor;     // intervals sorted by start, merged by the caller
    return *cursor < n_bed && bed[2 * *cursor] <= p0;
}

}  // namespace

namespace {

// One position range [start, end] on one thread: own file handle, own index query.  Errors go through set_error (thread-local).
// tid of `ctg_name`: reads the BAM header through `bz` (positioned at the start of the file)
int read_header_tid(Bgzf& bz, const char* bam_path, const char* ctg_name, int* tid_out) {
    uint8_t h4[4];
    CTO_REQUIRE(bz.read(h4, 4) && memcmp(h4, "BAM\1", 4) == 0, CTO_EINVAL, "cto_pack_from_bam: %s is not a BAM file%s%s", bam_path,
                bz.err.empty() ? "" : ": ", bz.err.c_str());
    CTO_REQUIRE(bz.read(h4, 4), CTO_EINVAL, "cto_pack_from_bam: truncated header");
    {
        CTO_REQUIRE(le32(h4) >= 0 && le32(h4) <= (1 << 28), CTO_EINVAL, "cto_pack_from_bam: bad header text length");
        std::vector<uint8_t> text(size_t(le32(h4)));
        CTO_REQUIRE(text.empty() || bz.read(text.data(), text.size()), CTO_EINVAL, "cto_pack_from_bam: truncated header text");
    }
    CTO_REQUIRE(bz.read(h4, 4), CTO_EINVAL, "cto_pack_from_bam: truncated header");
    const int n_ref = le32(h4);
    CTO_REQUIRE(n_ref >= 0, CTO_EINVAL, "cto_pack_from_bam: bad reference count");
    int tid = -1;
    for (int r = 0; r < n_ref; ++r) {
        CTO_REQUIRE(bz.read(h4, 4), CTO_EINVAL, "cto_pack_from_bam: truncated reference list");
        CTO_REQUIRE(le32(h4) > 0 && le32(h4) <= 65536, CTO_EINVAL, "cto_pack_from_bam: bad reference name length");
        std::vector<char> name(size_t(le32(h4)));
        CTO_REQUIRE(bz.read(name.data(), name.size()) && bz.read(h4, 4), CTO_EINVAL, "cto_pack_from_bam: truncated reference list");
        name.back() = 0;
        if (tid < 0 && strcmp(name.data(), ctg_name) == 0) tid = r;
    }
    CTO_REQUIRE(tid >= 0, CTO_EINVAL, "cto_pack_from_bam: contig %s not in the BAM header", ctg_name);
    *tid_out = tid;
    return CTO_OK;
}

// CIGARs with more than 65535 operations live in the CG:B,I tag; the CIGAR field then holds the placeholder <l_seq>S<ref_len>N
// (SAM specification, section 4.2.2).  *ops / *n_ops are redirected to the tag's array when the record is of that form.
static void resolve_cg_tag(const uint8_t* cg, int n_cig, int l_seq, const uint8_t* aux, const uint8_t* aend, const uint8_t** ops, int* n_ops) {
    if (!(n_cig == 2 && (le32(cg) & 15) == 4 && int(uint32_t(le32(cg)) >> 4) == l_seq && (le32(cg + 4) & 15) == 3)) return;
    while (aux + 3 <= aend) {
        const char t0 = char(aux[0]), t1 = char(aux[1]), ty = char(aux[2]);
        aux += 3;
        size_t skip = 0;
        if (ty == 'A' || ty == 'c' || ty == 'C') skip = 1;
        else if (ty == 's' || ty == 'S') skip = 2;
        else if (ty == 'i' || ty == 'I' || ty == 'f') skip = 4;
        else if (ty == 'Z' || ty == 'H') { while (aux + skip < aend && aux[skip]) ++skip; ++skip; }
        else if (ty == 'B') {
            if (aux + 5 > aend) break;
            const char sub = char(aux[0]);
            const uint32_t cnt = uint32_t(le32(aux + 1));
            const size_t esz = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
            if (t0 == 'C' && t1 == 'G' && sub == 'I' && aux + 5 + size_t(cnt) * 4 <= aend) {
                *n_ops = int(cnt);
                *ops = aux + 5;
                return;
            }
            skip = 5 + size_t(cnt) * esz;
        } else break;                                   // unknown type: stop scanning
        aux += skip;
    }
}

int pack_from_bam_range(const char* bam_path, const char* bai_path, const char* ctg_name, int64_t start, int64_t end,
                        const int64_t* bed, int64_t n_bed, const char* ref_seq, int64_t ref_start, size_t ref_len,
                        int excl_flags, int min_mq, int max_depth, int max_indel_length, const PreInflated& pre, cto_pack** out,
                        bool* cap_hit = nullptr) {
    Bgzf bz;
    CTO_REQUIRE(bz.open(bam_path), CTO_EINVAL, "cto_pack_from_bam: %s", bz.err.c_str());
    bz.pre = pre;
    uint8_t h4[4];
    int tid = -1;
    {
        const int rch = read_header_tid(bz, bam_path, ctg_name, &tid);
        if (rch != CTO_OK) return rch;
    }
    // ---- index ----
    std::vector<Chunk> chunks;
    {
        std::string err, idx = bai_path ? std::string(bai_path) : std::string(bam_path) + ".bai";
        CTO_REQUIRE(bai_query(idx.c_str(), tid, start - 1, end, &chunks, &err), CTO_EINVAL, "cto_pack_from_bam: %s", err.c_str());
    }
    std::unique_ptr<cto_pack> pk(new cto_pack());
    pack_begin(pk.get(), 1 << 16, 1 << 10);
    Producer pr;
    pr.p = pk.get();
    pr.ref_seq = ref_seq;
    pr.ref_start = ref_start;
    pr.ref_len = ref_len;
    pr.max_indel = max_indel_length;

    std::deque<Read> active;
    int64_t prev_start = 0;              // 0-based start of the last record that reached this point (see the flush below)
    int64_t next_col = start;            // next 1-based position to emit
    int64_t bed_cursor = 0;
    const int64_t beg0 = start - 1, end0 = end;   // 0-based half-open region
    auto flush_until = [&](int64_t limit1) -> int {   // emit columns next_col .. limit1 (1-based, inclusive)
        while (next_col <= limit1) {
            while (!active.empty() && active.front().end <= next_col - 1) active.pop_front();
            if (active.empty()) { next_col = limit1 + 1; break; }       // nothing can cover the positions up to the limit
            int64_t last = std::min<int64_t>(limit1, next_col + Producer::RUN - 1);     // run of requested positions next_col .. last
            if (bed) {
                if (!in_bed(bed, n_bed, next_col, &bed_cursor)) {
                    next_col = bed_cursor < n_bed ? std::max<int64_t>(next_col + 1, bed[2 * bed_cursor] + 1) : limit1 + 1;
                    continue;
                }
                last = std::min<int64_t>(last, bed[2 * bed_cursor + 1]);      // 1-based inclusive end of the interval
            }
            const int ncol = int(last - next_col + 1);
            pr.begin_run(ncol);
            const int32_t lo0 = int32_t(next_col - 1), hi0 = int32_t(last - 1);
            size_t dead = 0;
            for (Read& r : active) {
                const int32_t lo = std::max(lo0, r.pos), hi = std::min(hi0, r.end - 1);
                if (lo <= hi) pr.emit_run(r, lo, hi, lo0);
                else dead += r.end <= lo0;
            }
            if (dead > 32 && dead * 2 > active.size())       // finished reads parked behind a long one: compact, keeping file order
                active.erase(std::remove_if(active.begin(), active.end(), [&](const Read& r) { return r.end <= lo0; }), active.end());
            for (int c = 0; c < ncol; ++c) {
                if (pr.ents[c].empty()) continue;
                pr.seal_column(c);
                const int64_t pos1 = next_col + c, ri = pos1 - ref_start;
                if (ri < 0 || size_t(ri) >= ref_len) {
                    set_error("cto_pack_from_bam: position %lld outside the supplied reference", (long long)pos1);
                    return CTO_EINVAL;
                }
                const int rc = append_column_packed(pr.p, pr.sc, pos1, ri, ref_seq, ref_len, max_indel_length, pr.ents[c].data(),
                                                    int(pr.ents[c].size()), pr.indels[c].data(), int(pr.indels[c].size()), &pr.err);
                if (rc != CTO_OK) { set_error("%s", pr.err.c_str()); return rc; }
            }
            next_col = last + 1;
        }
        return CTO_OK;
    };
    // Reads are NOT removed from `active` in end order (a deque in file order, popped only from the front): a long read
    // at the front keeps shorter finished ones behind it alive, which only costs the bounds check in the loop above.
    std::vector<uint8_t> rec;
    bool done = false;
    const bool timing = getenv("CTO_PACK_TIMING") != nullptr;
    double t_flush = 0.0, t_read = 0.0;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    for (size_t ci = 0; ci < chunks.size() && !done; ++ci) {
        CTO_REQUIRE(bz.seek(chunks[ci].beg), CTO_EINVAL, "cto_pack_from_bam: seek into BAM failed: %s", bz.err.c_str());
        while (bz.tell() < chunks[ci].end) {
            const double tr0 = timing ? now() : 0.0;
            if (!bz.read(h4, 4)) { CTO_REQUIRE(bz.err.empty(), CTO_EINVAL, "cto_pack_from_bam: %s", bz.err.c_str()); done = true; break; }
            const int bsz = le32(h4);
            CTO_REQUIRE(bsz >= 32 && bsz <= (1 << 28), CTO_EINVAL, "cto_pack_from_bam: bad alignment block size %d", bsz);
            rec.resize(size_t(bsz));
            CTO_REQUIRE(bz.read(rec.data(), rec.size()), CTO_EINVAL, "cto_pack_from_bam: truncated alignment record");
            if (timing) t_read += now() - tr0;
            const uint8_t* b = rec.data();
            const int rtid = le32(b), pos = le32(b + 4);
            const int l_name = b[8], mapq = b[9];
            const int n_cig = b[12] | (b[13] << 8), flag = b[14] | (b[15] << 8);
            const int l_seq = le32(b + 16);
            if (rtid != tid) { if (rtid > tid || rtid < 0) { done = true; break; } continue; }
            if (pos >= end0) { done = true; break; }
            if ((flag & excl_flags) || (flag & 4) || mapq < min_mq || n_cig == 0 || l_seq <= 0 || pos < 0) continue;
            if ((flag & 1) && !(flag & 2)) continue;           // orphan (mpileup without -A)
            const size_t need = 32 + size_t(l_name) + size_t(n_cig) * 4 + size_t((l_seq + 1) / 2) + size_t(l_seq);
            CTO_REQUIRE(need <= rec.size(), CTO_EINVAL, "cto_pack_from_bam: alignment record shorter than its fields");
            const uint8_t* cg = b + 32 + l_name;
            const uint8_t* sq = cg + size_t(n_cig) * 4;
            const uint8_t* ql = sq + (l_seq + 1) / 2;
            Read r;
            r.pos = pos;
            r.mapq = uint8_t(mapq);
            r.rev = (flag & 16) != 0;
            // CIGARs with more than 65535 operations (ultra-long reads) live in the CG:B,I tag; the CIGAR field then holds the
            // placeholder <l_seq>S<ref_len>N (SAM specification, section 4.2.2)
            int n_ops = n_cig;
            const uint8_t* ops = cg;
            resolve_cg_tag(cg, n_cig, l_seq, ql + l_seq, rec.data() + rec.size(), &ops, &n_ops);
            r.cigar.resize(size_t(n_ops));
            int64_t rlen = 0, qlen = 0;                        // 64-bit: a crafted CIGAR must not wrap the sums
            for (int i = 0; i < n_ops; ++i) {
                const uint32_t c = uint32_t(le32(ops + i * 4));
                r.cigar[size_t(i)] = c;
                const int opc = int(c & 15), len = int(c >> 4);
                if (opc == 0 || opc == 2 || opc == 3 || opc == 7 || opc == 8) rlen += len;
                if (opc == 0 || opc == 1 || opc == 4 || opc == 7 || opc == 8) qlen += len;
            }
            if (qlen != l_seq || rlen == 0) continue;          // inconsistent or reference-less record
            CTO_REQUIRE(int64_t(pos) + rlen <= INT32_MAX, CTO_EINVAL, "cto_pack_from_bam: alignment at %d runs past 2^31 - 1", pos);
            r.end = int32_t(pos + rlen);
            if (r.end <= beg0) continue;
            r.op_ref = pos;
            r.l_seq = l_seq;
            r.no_qual = ql[0] == 0xff;                                                             // QUAL absent
            r.seq_off = size_t(sq - b);
            r.qual_off = size_t(ql - b);
            r.raw.swap(rec);                                   // the record's buffer moves into the read (b, sq, ql stay valid: same
                                                               // heap block); the next record gets a fresh one
            // Columns strictly before the PREVIOUS accepted read's start are emitted now; those between the two starts wait for
            // the next record.  That is htslib's order (bam_plp_next hands out column p only once a read starting beyond p has
            // been pushed, so a read is pushed - and its mate's qualities are edited - while the iterator stands at the
            // previous read's start), and it shows in one place: a deletion placeholder of the first mate that lies between
            // the two starts already prints the edited quality of the base after the deletion.
            const double tf0 = timing ? now() : 0.0;
            const int rcf = flush_until(std::min<int64_t>(prev_start, end));     // 1-based columns <= prev_start (0-based) are final
            if (rcf != CTO_OK) return rcf;
            if (timing) t_flush += now() - tf0;
            if (max_depth > 0) {
                int live = 0;
                for (const Read& a : active) live += a.end > pos;
                if (live >= max_depth) { if (cap_hit) *cap_hit = true; continue; }
            }
            if ((flag & 1) && l_name > 1) {                    // paired: the mate may already be in the pileup
                r.mate_key.assign(reinterpret_cast<const char*>(b + 32), size_t(l_name - 1));
                for (Read& a : active)
                    if (a.end > pos && a.mate_key == r.mate_key) { soften_overlap(a, r); break; }
            }
            prev_start = pos;
            active.push_back(std::move(r));
        }
    }
    const double tf1 = timing ? now() : 0.0;
    const int rcf = flush_until(end);
    if (rcf != CTO_OK) return rcf;
    if (timing)
        fprintf(stderr, "cto_pack_from_bam: total %.1f ms: read+inflate %.1f, pileup %.1f (libdeflate %d)\n", now() - t_begin, t_read,
                t_flush + now() - tf1, int(libdeflate().ok()));
    *out = pk.release();
    return CTO_OK;
}

}  // namespace

namespace {

// No C++ exception may cross the C ABI or leave a worker thread (std::terminate would take the host process down with it):
// allocation failures on hostile input (a record claiming 256 MB, a 4 GB index) come back as CTO_ENOMEM.
int pack_from_bam_impl(const char* bam_path, const char* bai_path, const char* ctg_name, int64_t start, int64_t end,
                       const int64_t* bed, int64_t n_bed, const char* ref_seq, int64_t ref_start, size_t ref_len,
                       int excl_flags, int min_mq, int max_depth, int max_indel_length, const PreInflated& pre, cto_pack** out) {
    CTO_REQUIRE(bam_path && ctg_name && ref_seq && out, CTO_EINVAL, "cto_pack_from_bam: null argument");
    CTO_REQUIRE(start >= 1 && end >= start, CTO_EINVAL, "cto_pack_from_bam: bad region %lld-%lld", (long long)start, (long long)end);
    CTO_REQUIRE(n_bed == 0 || bed, CTO_EINVAL, "cto_pack_from_bam: bed intervals missing");
    for (int64_t i = 1; i < n_bed; ++i)
        CTO_REQUIRE(bed[2 * i] >= bed[2 * i - 1], CTO_EINVAL, "cto_pack_from_bam: bed intervals must be sorted and non-overlapping");
    // Position ranges are independent (every range re-queries the index for the reads that overlap it), so a chunk is cut into
    // ranges of equal numbers of requested positions and piled up on several host threads, like the text tokeniser.  The
    // --max-depth cap is order dependent (a read is dropped when max_depth reads are live at its start), and a range does not see
    // the reads that ended before it: as long as NO range drops a read, no read is dropped in the unsplit order either (at a read's
    // start every live read reaches into the range that holds that start, so its count there is exact); as soon as one does, the
    // call is redone unsplit, so the pack never depends on how many threads the host happens to have.
    int64_t want = 0;                                       // requested positions inside [start, end]
    if (bed) {
        for (int64_t i = 0; i < n_bed; ++i) {
            const int64_t lo = std::max<int64_t>(bed[2 * i] + 1, start), hi = std::min<int64_t>(bed[2 * i + 1], end);
            if (hi >= lo) want += hi - lo + 1;
        }
    } else want = end - start + 1;
    unsigned nt = std::thread::hardware_concurrency();
    nt = std::max(1u, std::min(nt, 32u));
    nt = cto::pack_threads_or(nt);
    nt = unsigned(std::max<int64_t>(1, std::min<int64_t>(nt, want / 2000)));      // at least ~2000 positions per thread
    if (nt == 1)
        return pack_from_bam_range(bam_path, bai_path, ctg_name, start, end, bed, n_bed, ref_seq, ref_start, ref_len, excl_flags, min_mq,
                                   max_depth, max_indel_length, pre, out);
    // cut points: the position at which each thread's share of the requested positions begins
    std::vector<int64_t> cut(nt + 1, end + 1);
    cut[0] = start;
    {
        int64_t seen = 0;
        unsigned t = 1;
        auto feed = [&](int64_t lo, int64_t hi) {           // inclusive run of requested positions
            while (t < nt && seen + (hi - lo + 1) > want * t / nt) {
                cut[t] = lo + (want * t / nt - seen);
                ++t;
            }
            seen += hi - lo + 1;
        };
        if (bed) {
            for (int64_t i = 0; i < n_bed; ++i) {
                const int64_t lo = std::max<int64_t>(bed[2 * i] + 1, start), hi = std::min<int64_t>(bed[2 * i + 1], end);
                if (hi >= lo) feed(lo, hi);
            }
        } else feed(start, end);
    }
    std::vector<std::unique_ptr<cto_pack>> parts(nt);
    std::vector<int> rcs(nt, CTO_OK);
    std::vector<std::string> errs(nt);
    std::vector<char> capped(nt, 0);
    auto work = [&](unsigned t) {
        cto_pack* p = nullptr;
        if (cut[t + 1] - 1 < cut[t]) { parts[t].reset(new cto_pack()); pack_begin(parts[t].get(), 16, 16); return; }
        rcs[t] = guarded("cto_pack_from_bam", [&] {
            bool hit = false;
            const int rc = pack_from_bam_range(bam_path, bai_path, ctg_name, cut[t], cut[t + 1] - 1, bed, n_bed, ref_seq, ref_start, ref_len,
                                               excl_flags, min_mq, max_depth, max_indel_length, pre, &p, &hit);
            capped[t] = hit;
            return rc;
        });
        if (rcs[t] != CTO_OK) errs[t] = cto_last_error();
        parts[t].reset(p);
    };
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t) th.emplace_back(work, t);
        for (auto& x : th) x.join();
    }
    for (unsigned t = 0; t < nt; ++t)
        if (capped[t]) {                                    // the cap bit somewhere: the unsplit order decides which reads go
            parts.clear();
            return pack_from_bam_range(bam_path, bai_path, ctg_name, start, end, bed, n_bed, ref_seq, ref_start, ref_len, excl_flags, min_mq,
                                       max_depth, max_indel_length, pre, out);
        }
    for (unsigned t = 0; t < nt; ++t)
        if (rcs[t] != CTO_OK) { set_error("%s", errs[t].c_str()); return rcs[t]; }
    std::string merr;
    std::unique_ptr<cto_pack> p = merge_parts(parts, &merr);
    CTO_REQUIRE(p != nullptr, CTO_EINVAL, "cto_pack_from_bam: %s", merr.c_str());
    *out = p.release();
    return CTO_OK;
}

}  // namespace

extern "C" int cto_pack_from_bam(const char* bam_path, const char* bai_path, const char* ctg_name, int64_t start, int64_t end,
                                  const int64_t* bed, int64_t n_bed, const char* ref_seq, int64_t ref_start, size_t ref_len,
                                  int excl_flags, int min_mq, int max_depth, int max_indel_length, cto_pack** out) {
    return guarded("cto_pack_from_bam", [&] {
        return pack_from_bam_impl(bam_path, bai_path, ctg_name, start, end, bed, n_bed, ref_seq, ref_start, ref_len, excl_flags, min_mq,
                                  max_depth, max_indel_length, PreInflated{}, out);
    });
}

extern "C" int cto_pack_from_bam_inflated(const char* bam_path, const char* bai_path, const char* ctg_name, int64_t start, int64_t end,
                                           const int64_t* bed, int64_t n_bed, const char* ref_seq, int64_t ref_start, size_t ref_len,
                                           int excl_flags, int min_mq, int max_depth, int max_indel_length, const uint8_t* inflated,
                                           size_t inflated_len, const cto_bgzf_block* blocks, int64_t n_blocks, cto_pack** out) {
    return guarded("cto_pack_from_bam_inflated", [&] {
        CTO_REQUIRE(n_blocks >= 0 && (n_blocks == 0 || (inflated && blocks)), CTO_EINVAL, "cto_pack_from_bam_inflated: null argument");
        for (int64_t i = 0; i < n_blocks; ++i) {
            CTO_REQUIRE(i == 0 || blocks[i].file_off > blocks[i - 1].file_off, CTO_EINVAL, "cto_pack_from_bam_inflated: block table not sorted");
            CTO_REQUIRE(blocks[i].isize <= 65536 && blocks[i].out_off <= inflated_len && blocks[i].isize <= inflated_len - blocks[i].out_off,
                        CTO_EINVAL, "cto_pack_from_bam_inflated: block %lld lies outside the inflated buffer", (long long)i);
        }
        PreInflated pre;
        pre.data = inflated;
        pre.blocks = blocks;
        pre.n = n_blocks;
        return pack_from_bam_impl(bam_path, bai_path, ctg_name, start, end, bed, n_bed, ref_seq, ref_start, ref_len, excl_flags, min_mq,
                                  max_depth, max_indel_length, pre, out);
    });
}

// The byte range of the BAM that holds every BGZF block the index names for ctg:start-end (the block the last chunk ends in
// included: a BGZF block is at most 64 KiB long).
extern "C" int cto_bam_chunk_span(const char* bam_path, const char* bai_path, const char* ctg_name, int64_t start, int64_t end,
                                   int64_t* file_begin, int64_t* file_end) {
    return guarded("cto_bam_chunk_span", [&] {
        CTO_REQUIRE(bam_path && ctg_name && file_begin && file_end, CTO_EINVAL, "cto_bam_chunk_span: null argument");
        CTO_REQUIRE(start >= 1 && end >= start, CTO_EINVAL, "cto_bam_chunk_span: bad region %lld-%lld", (long long)start, (long long)end);
        Bgzf bz;
        CTO_REQUIRE(bz.open(bam_path), CTO_EINVAL, "cto_bam_chunk_span: %s", bz.err.c_str());
        int tid = -1;
        const int rch = read_header_tid(bz, bam_path, ctg_name, &tid);
        if (rch != CTO_OK) return rch;
        std::vector<Chunk> chunks;
        std::vector<uint64_t> linear;
        std::string err, idx = bai_path ? std::string(bai_path) : std::string(bam_path) + ".bai";
        CTO_REQUIRE(bai_query(idx.c_str(), tid, start - 1, end, &chunks, &err, &linear), CTO_EINVAL, "cto_bam_chunk_span: %s", err.c_str());
        const int64_t fsize = bz.fsize;
        int64_t lo = fsize, hi = 0;
        for (const Chunk& c : chunks) {
            lo = std::min<int64_t>(lo, int64_t(c.beg >> 16));
            hi = std::max<int64_t>(hi, int64_t(c.end >> 16) + 65536);
        }
        if (chunks.empty()) { lo = 0; hi = 0; }
        // The chunk lists of the coarse bins (a 512 Mb bin holds every read that straddles a finer boundary) end far behind the
        // region - the reader stops at the first alignment that starts after it, this range has to be cut beforehand.  The file is
        // sorted: the linear index names, per 16 kb window, the first alignment that overlaps it, and once THAT alignment starts
        // after the region everything from its block on does.  A few one-block probes find the window.
        if (!chunks.empty()) {
            const int64_t w0 = ((end - 1) >> 14) + 1;
            uint64_t last = 0;
            int probes = 0;
            for (int64_t w = w0; w < int64_t(linear.size()) && probes < 48; ++w) {
                const uint64_t v = linear[size_t(w)];
                if (v == 0 || v == last || int64_t(v >> 16) < lo) continue;
                last = v;
                ++probes;
                uint8_t head[12];
                if (!bz.seek(v) || !bz.read(head, 12)) break;         // damaged index / file: keep the wide range
                const int32_t rid = le32(head + 4), pos0 = le32(head + 8);
                if (rid != tid || int64_t(pos0) >= end) {             // starts after the region (1-based end = 0-based exclusive end)
                    hi = std::min<int64_t>(hi, int64_t(v >> 16) + 65536);
                    break;
                }
            }
        }
        *file_begin = lo;
        *file_end = std::min(hi, fsize);
        return CTO_OK;
    });
}

// `samtools view BAM ctg:start-end [-q min_mq]` without samtools: the alignments overlapping the region as SAM rows (no header),
// QNAME FLAG RNAME POS MAPQ CIGAR RNEXT PNEXT TLEN SEQ QUAL and, when the record has one, its HP:i tag - what
// src/realign_reads.py:255-300 reads of every row.  Returns the number of rows; *need = bytes of text (CTO_ENOMEM when cap is less).
extern "C" int64_t cto_bam_view(const char* bam_path, const char* bai_path, const char* ctg_name, int64_t start, int64_t end, int min_mq,
                                char* buf, size_t cap, size_t* need) {
    int64_t n_rows = 0;
    const int rc = guarded("cto_bam_view", [&] {
        CTO_REQUIRE(bam_path && ctg_name && need && (buf || cap == 0), CTO_EINVAL, "cto_bam_view: null argument");
        CTO_REQUIRE(start >= 1 && end >= start, CTO_EINVAL, "cto_bam_view: bad region");
        Bgzf bz;
        CTO_REQUIRE(bz.open(bam_path), CTO_EINVAL, "cto_bam_view: %s", bz.err.c_str());
        // header: reference names (RNEXT of a mate on another contig)
        uint8_t h4[4];
        CTO_REQUIRE(bz.read(h4, 4) && memcmp(h4, "BAM\1", 4) == 0, CTO_EINVAL, "cto_bam_view: %s is not a BAM file", bam_path);
        CTO_REQUIRE(bz.read(h4, 4) && le32(h4) >= 0 && le32(h4) <= (1 << 28), CTO_EINVAL, "cto_bam_view: bad header");
        { std::vector<uint8_t> text(size_t(le32(h4))); CTO_REQUIRE(text.empty() || bz.read(text.data(), text.size()), CTO_EINVAL, "cto_bam_view: truncated header"); }
        CTO_REQUIRE(bz.read(h4, 4) && le32(h4) >= 0, CTO_EINVAL, "cto_bam_view: truncated header");
        std::vector<std::string> names(size_t(le32(h4)));
        int tid = -1;
        for (size_t r = 0; r < names.size(); ++r) {
            CTO_REQUIRE(bz.read(h4, 4) && le32(h4) > 0 && le32(h4) <= 65536, CTO_EINVAL, "cto_bam_view: bad reference list");
            std::vector<char> nm(size_t(le32(h4)));
            CTO_REQUIRE(bz.read(nm.data(), nm.size()) && bz.read(h4, 4), CTO_EINVAL, "cto_bam_view: truncated reference list");
            nm.back() = 0;
            names[r] = nm.data();
            if (tid < 0 && names[r] == ctg_name) tid = int(r);
        }
        CTO_REQUIRE(tid >= 0, CTO_EINVAL, "cto_bam_view: contig %s not in the BAM header", ctg_name);
        std::vector<Chunk> chunks;
        std::string err, idx = bai_path ? std::string(bai_path) : std::string(bam_path) + ".bai";
        CTO_REQUIRE(bai_query(idx.c_str(), tid, start - 1, end, &chunks, &err), CTO_EINVAL, "cto_bam_view: %s", err.c_str());
        std::string out;
        std::vector<uint8_t> rec;
        bool done = false;
        const int64_t beg0 = start - 1, end0 = end;
        for (size_t ci = 0; ci < chunks.size() && !done; ++ci) {
            CTO_REQUIRE(bz.seek(chunks[ci].beg), CTO_EINVAL, "cto_bam_view: seek into BAM failed: %s", bz.err.c_str());
            while (bz.tell() < chunks[ci].end) {
                if (!bz.read(h4, 4)) { CTO_REQUIRE(bz.err.empty(), CTO_EINVAL, "cto_bam_view: %s", bz.err.c_str()); done = true; break; }
                const int bsz = le32(h4);
                CTO_REQUIRE(bsz >= 32 && bsz <= (1 << 28), CTO_EINVAL, "cto_bam_view: bad alignment block size %d", bsz);
                rec.resize(size_t(bsz));
                CTO_REQUIRE(bz.read(rec.data(), rec.size()), CTO_EINVAL, "cto_bam_view: truncated alignment record");
                const uint8_t* b = rec.data();
                const int rtid = le32(b), pos = le32(b + 4), l_name = b[8], mapq = b[9];
                const int n_cig = b[12] | (b[13] << 8), flag = b[14] | (b[15] << 8), l_seq = le32(b + 16);
                const int nref = le32(b + 20), npos = le32(b + 24), tlen = le32(b + 28);
                if (rtid != tid) { if (rtid > tid || rtid < 0) { done = true; break; } continue; }
                if (pos >= end0) { done = true; break; }
                const size_t need_b = 32 + size_t(l_name) + size_t(n_cig) * 4 + size_t((std::max(l_seq, 0) + 1) / 2) + size_t(std::max(l_seq, 0));
                CTO_REQUIRE(l_seq >= 0 && need_b <= rec.size(), CTO_EINVAL, "cto_bam_view: alignment record shorter than its fields");
                const uint8_t* cg = b + 32 + l_name;
                const uint8_t* sq = cg + size_t(n_cig) * 4;
                const uint8_t* ql = sq + (l_seq + 1) / 2;
                // the real CIGAR of a read with more than 65535 operations is its CG:B,I tag (samtools view prints that one too)
                int n_ops = n_cig;
                const uint8_t* ops = cg;
                resolve_cg_tag(cg, n_cig, l_seq, ql + l_seq, rec.data() + rec.size(), &ops, &n_ops);
                int64_t rlen = 0;
                for (int i = 0; i < n_ops; ++i) {
                    const uint32_t c = uint32_t(le32(ops + i * 4));
                    const int opc = int(c & 15);
                    if (opc == 0 || opc == 2 || opc == 3 || opc == 7 || opc == 8) rlen += int64_t(c >> 4);
                }
                if (int64_t(pos) + std::max<int64_t>(rlen, 1) <= beg0 || mapq < min_mq) continue;
                char num[32];
                out.append(reinterpret_cast<const char*>(b + 32), size_t(std::max(0, l_name - 1)));
                out += '\t'; out += std::to_string(flag); out += '\t'; out += ctg_name; out += '\t'; out += std::to_string(pos + 1);
                out += '\t'; out += std::to_string(mapq); out += '\t';
                if (n_ops == 0) out += '*';
                for (int i = 0; i < n_ops; ++i) {
                    const uint32_t c = uint32_t(le32(ops + i * 4));
                    snprintf(num, sizeof(num), "%u%c", c >> 4, "MIDNSHP=X???????"[c & 15]);
                    out += num;
                }
                out += '\t';
                out += nref < 0 ? "*" : (nref == tid ? "=" : (size_t(nref) < names.size() ? names[size_t(nref)].c_str() : "*"));
                out += '\t'; out += std::to_string(npos + 1); out += '\t'; out += std::to_string(tlen); out += '\t';
                if (l_seq == 0) out += '*';
                for (int i = 0; i < l_seq; ++i) out += kNt16[(sq[i >> 1] >> ((~i & 1) << 2)) & 15];
                out += '\t';
                if (l_seq == 0 || ql[0] == 0xff) out += '*';
                else for (int i = 0; i < l_seq; ++i) out += char(std::min(int(ql[i]), 93) + 33);
                // HP:i of the auxiliary fields
                const uint8_t* aux = ql + l_seq;
                const uint8_t* aend = rec.data() + rec.size();
                while (aux + 3 <= aend) {
                    const char t0 = char(aux[0]), t1 = char(aux[1]), ty = char(aux[2]);
                    aux += 3;
                    size_t skip = 0;
                    long long val = 0;
                    bool is_int = true;
                    if (ty == 'c' && aux + 1 <= aend) { val = int8_t(aux[0]); skip = 1; }
                    else if (ty == 'C' && aux + 1 <= aend) { val = aux[0]; skip = 1; }
                    else if (ty == 's' && aux + 2 <= aend) { val = int16_t(aux[0] | (aux[1] << 8)); skip = 2; }
                    else if (ty == 'S' && aux + 2 <= aend) { val = aux[0] | (aux[1] << 8); skip = 2; }
                    else if (ty == 'i' && aux + 4 <= aend) { val = le32(aux); skip = 4; }
                    else if (ty == 'I' && aux + 4 <= aend) { val = uint32_t(le32(aux)); skip = 4; }
                    else {
                        is_int = false;
                        if (ty == 'A') skip = 1;
                        else if (ty == 'f') skip = 4;
                        else if (ty == 'Z' || ty == 'H') { while (aux + skip < aend && aux[skip]) ++skip; ++skip; }
                        else if (ty == 'B') {
                            if (aux + 5 > aend) break;
                            const char sub = char(aux[0]);
                            const size_t esz = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                            skip = 5 + size_t(uint32_t(le32(aux + 1))) * esz;
                        } else break;
                    }
                    if (is_int && t0 == 'H' && t1 == 'P') { out += "\tHP:i:"; out += std::to_string(val); }
                    aux += skip;
                }
                out += '\n';
                ++n_rows;
            }
        }
        *need = out.size();
        CTO_REQUIRE(out.size() <= cap, CTO_ENOMEM, "cto_bam_view: %zu bytes of text, room for %zu", out.size(), cap);
        if (!out.empty()) memcpy(buf, out.data(), out.size());
        return CTO_OK;
    });
    return rc == CTO_OK ? n_rows : rc;
}

// Record boundaries the index knows inside [file_begin, file_end): the starts of the region's chunks and, per 16 kb window of the
// region, the first alignment that overlaps it (BAI linear index) - virtual offsets, ascending, the first one being where a
// reader of the region starts.  The device pileup (csrc/pileup.hip) walks one chain of records from each of them.
extern "C" int64_t cto_bam_record_starts(const char* bam_path, const char* bai_path, const char* ctg_name, int64_t start, int64_t end,
                                         int64_t file_begin, int64_t file_end, uint64_t* voffs, int64_t cap, int32_t* tid_out) {
    int64_t n_out = 0;
    const int rc = guarded("cto_bam_record_starts", [&] {
        CTO_REQUIRE(bam_path && ctg_name && voffs && cap > 0 && tid_out, CTO_EINVAL, "cto_bam_record_starts: null argument");
        CTO_REQUIRE(start >= 1 && end >= start, CTO_EINVAL, "cto_bam_record_starts: bad region");
        Bgzf bz;
        CTO_REQUIRE(bz.open(bam_path), CTO_EINVAL, "cto_bam_record_starts: %s", bz.err.c_str());
        int tid = -1;
        const int rch = read_header_tid(bz, bam_path, ctg_name, &tid);
        if (rch != CTO_OK) return rch;
        *tid_out = tid;
        std::vector<Chunk> chunks;
        std::vector<uint64_t> linear;
        std::string err, idx = bai_path ? std::string(bai_path) : std::string(bam_path) + ".bai";
        CTO_REQUIRE(bai_query(idx.c_str(), tid, start - 1, end, &chunks, &err, &linear), CTO_EINVAL, "cto_bam_record_starts: %s", err.c_str());
        if (chunks.empty()) return CTO_OK;
        const uint64_t first = chunks.front().beg;
        std::vector<uint64_t> v;
        for (const Chunk& c : chunks) v.push_back(c.beg);
        const int64_t w0 = (start - 1) >> 14, w1 = std::min<int64_t>(int64_t(linear.size()) - 1, (end - 1) >> 14);
        for (int64_t w = w0; w <= w1; ++w)
            if (w >= 0 && linear[size_t(w)] > first) v.push_back(linear[size_t(w)]);
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
        for (uint64_t x : v) {
            const int64_t coff = int64_t(x >> 16);
            if (coff < file_begin || coff >= file_end) continue;
            if (n_out < cap) voffs[n_out] = x;
            ++n_out;
        }
        CTO_REQUIRE(n_out <= cap, CTO_ENOMEM, "cto_bam_record_starts: %lld offsets, room for %lld", (long long)n_out, (long long)cap);
        return CTO_OK;
    });
    return rc == CTO_OK ? n_out : rc;
}

// Block table of a run of whole BGZF blocks (a trailing partial block is left out).
extern "C" int64_t cto_bgzf_scan(const uint8_t* bytes, size_t len, int64_t file_begin, cto_bgzf_block* blocks, int64_t cap, int64_t* out_bytes) {
    if (!bytes || !blocks || !out_bytes) { set_error("cto_bgzf_scan: null argument"); return CTO_EINVAL; }
    size_t o = 0;
    int64_t n = 0, out = 0;
    while (o + 18 <= len) {
        const uint8_t* h = bytes + o;
        if (h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) { set_error("cto_bgzf_scan: not a BGZF block header at byte %zu", o); return CTO_EINVAL; }
        const size_t xlen = size_t(h[10]) | (size_t(h[11]) << 8);
        if (o + 12 + xlen > len) break;
        int64_t bsize = -1;
        for (size_t i = 0; i + 4 <= xlen;) {
            const uint8_t* e = h + 12 + i;
            const size_t slen = size_t(e[2]) | (size_t(e[3]) << 8);
            if (e[0] == 'B' && e[1] == 'C' && slen == 2 && i + 6 <= xlen) bsize = int64_t(e[4] | (e[5] << 8)) + 1;
            i += 4 + slen;
        }
        if (bsize < 0) { set_error("cto_bgzf_scan: BGZF block without BC subfield at byte %zu", o); return CTO_EINVAL; }
        const int64_t cdata = bsize - int64_t(xlen) - 12 - 8;
        if (cdata < 0) { set_error("cto_bgzf_scan: bad BGZF block size at byte %zu", o); return CTO_EINVAL; }
        if (o + size_t(bsize) > len) break;                     // partial block at the end of the range
        const uint8_t* tail = h + bsize - 8;
        const uint32_t isize = uint32_t(tail[4]) | (uint32_t(tail[5]) << 8) | (uint32_t(tail[6]) << 16) | (uint32_t(tail[7]) << 24);
        if (isize > 65536) { set_error("cto_bgzf_scan: BGZF block claims more than 64 KiB of data"); return CTO_EINVAL; }
        if (n >= cap) { set_error("cto_bgzf_scan: more than %lld blocks", (long long)cap); return CTO_ENOMEM; }
        cto_bgzf_block& b = blocks[n++];
        b.file_off = uint64_t(file_begin) + o;
        b.in_off = o + 12 + xlen;
        b.out_off = uint64_t(out);
        b.csize = uint32_t(cdata);
        b.isize = isize;
        b.bsize = uint32_t(bsize);
        b.crc32 = uint32_t(tail[0]) | (uint32_t(tail[1]) << 8) | (uint32_t(tail[2]) << 16) | (uint32_t(tail[3]) << 24);
        out += (int64_t(isize) + CTO_BGZF_SLOT_PAD + 255) / 256 * 256;   // the inflate kernel checks a literal run's output bound once
                                                                         // per 32 input bits: a malformed block may run that far past its isize
        o += size_t(bsize);
    }
    *out_bytes = out;
    return n;
}
