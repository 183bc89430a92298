"""Synthetic RUN directories for end-to-end measurements and tests (there are no BAMs, no samtools and no network on
either box): what `run_clairs_to` leaves in <out>/tmp/ before STEP 2 - a reference FASTA + .fai, candidate BED chunk files
(`<ctg>.<i>_<n>_snv`, extract_candidates_calling.py:450-488) and their list - plus, per chunk, either the text
`samtools mpileup --reverse-del --output-MQ --min-BQ 0` would print (make_text_run) or one coordinate-sorted, indexed BAM for
the whole region (make_bam_run).  Input synthesis only: nothing here is on the measured path.

The BAM writer is vectorised (numpy packs sequences and qualities; zlib level 1 BGZF blocks) so that a 50x megabase region
takes seconds, not minutes.  Formats per the SAM/BAM specification v1 (4.1 BGZF, 4.2 BAM records, 5.2 BAI)."""
import os
import struct
import zlib

import numpy as np

from .synth import SynthChunk, mpileup_text

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _write_fasta(path, name, seq):
    with open(path, "w") as f:
        f.write(">%s\n" % name)
        f.write("\n".join(seq[i:i + 60] for i in range(0, len(seq), 60)))
        f.write("\n")
    with open(path + ".fai", "w") as f:
        f.write("%s\t%d\t%d\t60\t61\n" % (name, len(seq), len(name) + 2))


def make_text_run(d, n_chunks=8, sites_per_chunk=4096, distinct=3, seed=20260928, ctg="chr1", **chunk_kw):
    """`distinct` different synthetic chunks (SynthChunk presets) written out as `n_chunks` chunk files: chunk i
    re-uses the pileup of chunk i % distinct (same positions - the chunk VCFs overlap, which a throughput run does not care
    about; tests use distinct == n_chunks).  Returns dict(ref_fn, chunk_list, chunks, mpileup_dir, n_sites)."""
    os.makedirs(d, exist_ok=True)
    mp_dir = os.path.join(d, "mpileup")
    os.makedirs(mp_dir, exist_ok=True)
    base, texts, span = [], [], 0
    for k in range(distinct):
        ch = SynthChunk(sites_per_chunk, seed=seed + k, start=1000 + k * (sites_per_chunk * 300 + 5000), **chunk_kw)
        base.append(ch)
        texts.append(mpileup_text(ch, 0, ctg=ctg))
        span = max(span, int(ch.col_pos[-1]) + 200)
    ref = np.full(span, ord("A"), dtype=np.uint8)
    for ch in base:
        ref[ch.col_pos.astype(np.int64) - 1] = ch.col_ref_char
    ref_fn = os.path.join(d, "ref.fa")
    _write_fasta(ref_fn, ctg, ref.tobytes().decode())
    names, n_sites = [], 0
    for i in range(n_chunks):
        ch = base[i % distinct]
        fn = os.path.join(d, "%s.%d_%d_snv" % (ctg, i + 1, n_chunks))
        with open(fn, "w") as f:
            f.write("".join("%s\t%d\t%d\n" % (ctg, x - 17, x + 17) for x in ch.site_pos.tolist()))
        with open(os.path.join(mp_dir, os.path.basename(fn) + ".mpileup"), "w") as f:
            f.write(texts[i % distinct])
        names.append(fn)
        n_sites += ch.n_sites
    chunk_list = os.path.join(d, "CANDIDATES_FILES")
    with open(chunk_list, "w") as f:
        f.write("".join(n + "\n" for n in names))
    return dict(ref_fn=ref_fn, chunk_list=chunk_list, chunks=names, mpileup_dir=mp_dir, n_sites=n_sites, base=base)


# ---------------------------------------------------------------------------------------------------------- BAM
def _reg2bin(beg, end):
    end -= 1
    for shift, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return off + (beg >> shift)
    return 0


def _bgzf(data, level=1):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    head = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25)
    return head + comp + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))


def write_simple_bam(path, ctg, ctg_len, reads):
    """reads: iterable of (pos0, is_reverse, mapq, codes uint8 [n] in 0..3, quals uint8 [n], cigar [(op_char, n)]) in coordinate
    order.  Writes path + path.bai."""
    nt16 = np.array([1, 2, 4, 8], dtype=np.uint8)
    ops = {c: i for i, c in enumerate("MIDNSHP=X")}
    stream = [b"BAM\1" + struct.pack("<ii", 0, 1) + struct.pack("<i", len(ctg) + 1) + ctg.encode() + b"\0" + struct.pack("<i", ctg_len)]
    upos = len(stream[0])
    spans = []
    for i, (pos, rev, mapq, codes, quals, cigar) in enumerate(reads):
        n = len(codes)
        rlen = sum(c for o, c in cigar if o in "MDN=X")
        c4 = nt16[codes]
        if n & 1:
            c4 = np.append(c4, np.uint8(0))
        name = b"r%d\0" % i
        body = struct.pack("<iiBBHHHiiii", 0, pos, len(name), mapq, _reg2bin(pos, pos + max(1, rlen)), len(cigar), 16 if rev else 0, n, -1, -1, 0)
        body += name + b"".join(struct.pack("<I", (c << 4) | ops[o]) for o, c in cigar)
        body += ((c4[0::2] << 4) | c4[1::2]).tobytes() + np.asarray(quals, dtype=np.uint8).tobytes()
        rec = struct.pack("<i", len(body)) + body
        spans.append((upos, upos + len(rec), pos, pos + max(1, rlen)))
        upos += len(rec)
        stream.append(rec)
    data = b"".join(stream)
    payload = 65000
    coffs, off = [], 0
    with open(path, "wb") as f:
        for u in range(0, len(data), payload):
            b = _bgzf(data[u:u + payload])
            coffs.append(off)
            off += len(b)
            f.write(b)
        f.write(_bgzf(b""))

    def voff(u):
        return (off << 16) if u == len(data) else ((coffs[u // payload] << 16) | (u % payload))
    bins, lin = {}, {}
    for us, ue, beg, end in spans:
        vb, ve = voff(us), voff(ue)
        ch = bins.setdefault(_reg2bin(beg, end), [])
        if ch and ch[-1][1] == vb:
            ch[-1][1] = ve
        else:
            ch.append([vb, ve])
        for w in range(beg >> 14, ((end - 1) >> 14) + 1):
            lin[w] = min(lin.get(w, vb), vb)
    out = [b"BAI\1" + struct.pack("<ii", 1, len(bins))]
    for b, chunks in sorted(bins.items()):
        out.append(struct.pack("<Ii", b, len(chunks)) + b"".join(struct.pack("<QQ", vb, ve) for vb, ve in chunks))
    n_intv = (max(lin) + 1) if lin else 0
    out.append(struct.pack("<i", n_intv))
    last = 0
    for w in range(n_intv):
        last = lin.get(w, last)
        out.append(struct.pack("<Q", last))
    with open(path + ".bai", "wb") as f:
        f.write(b"".join(out))


def make_bam_run(d, region_kb=1000, n_chunks=4, depth=50, spacing=250, seed=1, ctg="chr1", read_len_mu=9.0, p_mismatch=0.02,
                 p_indel_per_kb=1.5):
    """A `depth`x long-read BAM (log-normal read lengths, substitutions, a few short insertions / deletions, BQ ~ N(28, 8)) over
    one contig of region_kb kilobases, candidates every `spacing` bp cut into n_chunks chunk files.
    Returns dict(ref_fn, bam_fn, chunk_list, chunks, n_sites)."""
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(seed)
    L = region_kb * 1000
    ref = rng.integers(0, 4, size=L).astype(np.uint8)
    starts, lens = [], []
    bases = 0
    while bases < depth * L:
        n = int(np.clip(rng.lognormal(read_len_mu, 0.5), 1000, 30000))
        pos = int(rng.integers(0, max(1, L - 1000)))
        n = min(n, L - pos)
        starts.append(pos)
        lens.append(n)
        bases += n
    order = np.argsort(np.asarray(starts), kind="stable")

    def gen():
        for k in order:
            pos, n = starts[k], lens[k]
            seg = ref[pos:pos + n].copy()
            mm = rng.random(n) < p_mismatch
            seg[mm] = rng.integers(0, 4, size=int(mm.sum()))
            cigar, pieces, at = [], [], 0
            n_ev = rng.poisson(p_indel_per_kb * n / 1000.0)
            cuts = np.sort(rng.integers(50, max(51, n - 50), size=n_ev)) if n > 200 else []
            for c in cuts:
                c = int(c)
                if c - at < 20:
                    continue
                cigar.append(("M", c - at))
                pieces.append(seg[at:c])
                ln = int(min(rng.geometric(0.5), 8))
                if rng.random() < 0.5:
                    cigar.append(("I", ln))
                    pieces.append(rng.integers(0, 4, size=ln).astype(np.uint8))
                    at = c
                else:
                    cigar.append(("D", ln))
                    at = c + ln
            if n - at > 0:
                cigar.append(("M", n - at))
                pieces.append(seg[at:n])
            codes = np.concatenate(pieces) if len(pieces) > 1 else pieces[0]
            q = np.clip(np.rint(rng.normal(28, 8, size=len(codes))), 1, 50).astype(np.uint8)
            yield pos, bool(rng.random() < 0.5), 60 if rng.random() < 0.93 else int(rng.integers(0, 60)), codes, q, cigar
    bam_fn = os.path.join(d, "tumor.bam")
    write_simple_bam(bam_fn, ctg, L, gen())
    ref_fn = os.path.join(d, "ref.fa")
    _write_fasta(ref_fn, ctg, _ACGT[ref].tobytes().decode())
    sites = list(range(1000, L - 1000, spacing))
    per = (len(sites) + n_chunks - 1) // n_chunks
    names = []
    for c in range(n_chunks):
        part = sites[c * per:(c + 1) * per]
        if not part:
            continue
        fn = os.path.join(d, "%s.%d_%d_snv" % (ctg, c + 1, n_chunks))
        with open(fn, "w") as f:
            f.write("".join("%s\t%d\t%d\n" % (ctg, x - 17, x + 17) for x in part))
        names.append(fn)
    chunk_list = os.path.join(d, "CANDIDATES_FILES")
    with open(chunk_list, "w") as f:
        f.write("".join(n + "\n" for n in names))
    return dict(ref_fn=ref_fn, bam_fn=bam_fn, chunk_list=chunk_list, chunks=names, n_sites=len(sites), contig_len=L, ctg=ctg)
