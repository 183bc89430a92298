"""BAM -> pack producer (csrc/bam.cpp) against an independent naive pileup of the same rules, on BAM + BAI
files written by tests/bamutil.py.  PARITY UNPINNED against samtools (absent from both boxes): this pins the reader to
the documented rules and to the text tokeniser, nothing more."""
import numpy as np
import pytest

from bamutil import write_bam, mpileup_rows, ref_len_of


def _random_reads(rng, n, ref_lens, paired_frac=0.0, weird=True):
    reads = []
    for i in range(n):
        ref = int(rng.integers(0, len(ref_lens)))
        L = ref_lens[ref]
        pos = int(rng.integers(0, L - 450))
        cigar = []
        if rng.random() < 0.3:
            cigar.append(("H" if rng.random() < 0.3 else "S", int(rng.integers(1, 12))))
        if weird and rng.random() < 0.05:
            cigar.append(("I", int(rng.integers(1, 4))))              # insertion before any aligned base
        nblk = int(rng.integers(1, 6))
        for b in range(nblk):
            cigar.append((str(rng.choice(list("MMMM=X"))), int(rng.integers(1, 70))))
            if b + 1 < nblk:
                u = rng.random()
                if u < 0.4:
                    cigar.append(("I", int(rng.integers(1, 70 if rng.random() < 0.1 else 6))))
                elif u < 0.8:
                    cigar.append(("D", int(rng.integers(1, 70 if rng.random() < 0.1 else 6))))
                    if weird and rng.random() < 0.1:
                        cigar.append(("I", int(rng.integers(1, 3))))  # insertion right after a deletion
                elif u < 0.9:
                    cigar.append(("N", int(rng.integers(1, 30))))
                elif weird:
                    cigar.append(("P", 1))
                    cigar.append(("I", 2))
                else:
                    cigar.append(("D", 1))
        if rng.random() < 0.3:
            cigar.append(("S", int(rng.integers(1, 12))))
        # merge accidental adjacent equal ops is not needed for validity; query length from the CIGAR
        qlen = sum(n for op, n in cigar if op in "MIS=X")
        seq = "".join(rng.choice(list("ACGTACGTACGTN=RY"), size=qlen))
        qual = None if rng.random() < 0.03 else [int(q) for q in rng.integers(0, 100, size=qlen)]
        flag = 0
        if rng.random() < 0.5:
            flag |= 16
        u = rng.random()
        if u < 0.04:
            flag |= 256
        elif u < 0.08:
            flag |= 2048
        elif u < 0.10:
            flag |= 1024                                                # duplicates are kept by --excl-flags 2316
        if rng.random() < paired_frac:
            flag |= 1 | (2 if rng.random() < 0.8 else 0) | (8 if rng.random() < 0.1 else 0)
        mapq = int(rng.choice([0, 3, 19, 20, 60, 60, 60, 120, 255]))
        reads.append(dict(name="r%d" % i, flag=flag, ref=ref, pos=pos, mapq=mapq, cigar=cigar, seq=seq, qual=qual,
                          cg_tag=bool(rng.random() < 0.1)))               # some CIGARs travel in the CG tag
    # mates: a second read under the same name, a little downstream, same CIGAR, bases partly disagreeing - the overlap rule
    # (agree: sum of the qualities on the first mate; differ: 0.8 x the better one; the other base -> 0) has to kick in
    for r in [r for r in reads if (r["flag"] & 3) == 3 and not (r["flag"] & (8 | 256 | 2048))]:
        if rng.random() < 0.6:
            m = dict(r)
            m["pos"] = min(r["pos"] + int(rng.integers(0, 90)), ref_lens[r["ref"]] - 450)
            m["flag"] = (r["flag"] & ~(16 | 64)) | 128 | (16 if rng.random() < 0.5 else 0)
            seq = list(r["seq"])
            for k in rng.integers(0, len(seq), size=max(1, len(seq) // 6)).tolist():
                seq[k] = str(rng.choice(list("ACGT")))
            m["seq"] = "".join(seq)
            m["qual"] = None if r["qual"] is None or rng.random() < 0.05 else [int(q) for q in rng.integers(0, 100, size=len(seq))]
            reads.append(m)
    reads.sort(key=lambda r: (r["ref"], r["pos"]))
    return reads


def _pack_arrays(pack):
    a = {k: v.copy() for k, v in pack.numpy().items()}
    a["keys"] = [pack.key_string(k) for k in range(pack.n_keys)]
    return a


def _assert_same(a, b):
    for k in ("col_pos", "col_ref", "col_off", "key_off", "entries", "key_meta", "key_group"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert a["keys"] == b["keys"]


@pytest.mark.parametrize("seed,paired", [(1, 0.0), (2, 0.5), (3, 0.0)])
def test_pack_from_bam_matches_naive_pileup(tmp_path, seed, paired):
    from clairs_to_amd.pack import ColumnPack
    rng = np.random.default_rng(seed)
    ref_lens = [40000, 3000]
    refs = [("chrA", ref_lens[0]), ("chrB", ref_lens[1])]
    ref_seqs = ["".join(rng.choice(list("ACGTN"), p=[.24, .24, .24, .24, .04], size=L)) for L in ref_lens]
    reads = _random_reads(rng, 900, ref_lens, paired_frac=paired)
    # a deep spot, to exercise max_depth
    for i in range(40):
        reads.append(dict(name="d%d" % i, flag=16 * (i & 1), ref=0, pos=20000 + (i % 3), mapq=60, cigar=[("M", 30)],
                          seq="ACGT" * 7 + "AC", qual=[30] * 30))
    reads.sort(key=lambda r: (r["ref"], r["pos"]))
    bam = str(tmp_path / "t.bam")
    write_bam(bam, refs, reads, block_payload=1500 if seed != 3 else 60000)
    cases = [(0, 1, ref_lens[0], None, 8000), (0, 5000, 9000, None, 8000), (0, 16380, 16400, None, 8000),
             (0, 19990, 20040, None, 10), (1, 1, ref_lens[1], None, 8000), (1, 700, 2400, [(650, 720), (900, 934), (2000, 2500)], 8000),
             (0, 39000, 40000, [(38990, 39010)], 8000)]
    nonempty = 0
    for ref_i, start, end, bed, max_depth in cases:
        name = refs[ref_i][0]
        text = mpileup_rows(reads, ref_i, name, start, end, bed=bed, max_depth=max_depth, ref_seq=ref_seqs[ref_i], ref_start=1)
        want = ColumnPack.from_mpileup(text, ref_seqs[ref_i], 1)
        got = ColumnPack.from_bam(bam, name, start, end, ref_seqs[ref_i], 1, bed=bed, max_depth=max_depth)
        _assert_same(_pack_arrays(got), _pack_arrays(want))
        nonempty += want.n_cols > 0
    assert nonempty >= 5


def test_pack_from_bam_errors(tmp_path):
    from clairs_to_amd.pack import ColumnPack
    from clairs_to_amd._lib import CtoError
    refs = [("chrA", 1000)]
    reads = [dict(name="a", flag=0, ref=0, pos=10, mapq=60, cigar=[("M", 20)], seq="A" * 20, qual=[20] * 20)]
    bam = str(tmp_path / "e.bam")
    write_bam(bam, refs, reads)
    ref = "A" * 1000
    with pytest.raises(CtoError):
        ColumnPack.from_bam(bam, "chrZ", 1, 100, ref, 1)              # unknown contig
    with pytest.raises(CtoError):
        ColumnPack.from_bam(str(tmp_path / "missing.bam"), "chrA", 1, 100, ref, 1)
    with pytest.raises(CtoError):
        ColumnPack.from_bam(bam, "chrA", 1, 100, ref, 1, bai_fn=str(tmp_path / "nope.bai"))
    (tmp_path / "junk.bam").write_bytes(b"not a bam at all, definitely" * 10)
    with pytest.raises(CtoError):
        ColumnPack.from_bam(str(tmp_path / "junk.bam"), "chrA", 1, 100, ref, 1, bai_fn=bam + ".bai")
    with pytest.raises(CtoError):
        ColumnPack.from_bam(bam, "chrA", 1, 100, ref[:5], 1)          # reference too short for the covered positions
    p = ColumnPack.from_bam(bam, "chrA", 500, 600, ref, 1)            # nothing there: an empty pack, not an error
    assert p.n_cols == 0 and p.n_entries == 0


def test_bgzf_crc_is_checked(tmp_path):
    """One flipped bit inside a BGZF block's payload: either the DEFLATE stream breaks or the inflated bytes fail the gzip trailer's
    CRC-32 (htslib checks it too) - the read never succeeds with different bytes."""
    from clairs_to_amd.pack import ColumnPack
    from clairs_to_amd._lib import CtoError
    rng = np.random.default_rng(5)
    reads = _random_reads(rng, 150, [5000], weird=False)
    bam = str(tmp_path / "ok.bam")
    write_bam(bam, [("chrA", 5000)], reads, block_payload=4000)
    ref = "".join(rng.choice(list("ACGT"), size=5000))
    good = ColumnPack.from_bam(bam, "chrA", 1, 5000, ref, 1).numpy()["entries"].copy()
    raw = open(bam, "rb").read()
    first = 18 + 8                                  # skip the first block's header (the BAM header block may be tiny)
    n_err = n_same = 0
    for off in range(first + 200, len(raw) - 60, max(1, (len(raw) - 300) // 40)):
        b = bytearray(raw)
        b[off] ^= 0x08
        p = tmp_path / "flip.bam"
        p.write_bytes(bytes(b))
        (tmp_path / "flip.bam.bai").write_bytes(open(bam + ".bai", "rb").read())
        try:
            got = ColumnPack.from_bam(str(p), "chrA", 1, 5000, ref, 1).numpy()["entries"]
            assert np.array_equal(got, good)        # only a flip in bytes the reader never looks at may pass
            n_same += 1
        except CtoError:
            n_err += 1
    assert n_err >= 25 and n_same <= 10


def test_producers_survive_corrupt_input(tmp_path):
    """Neither pack producer may crash on damaged input: every call returns a pack or raises CtoError.  200 random byte
    flips / truncations of a valid BAM, of its index and of a valid mpileup text."""
    from clairs_to_amd.pack import ColumnPack
    from clairs_to_amd._lib import CtoError
    rng = np.random.default_rng(99)
    ref_lens = [5000]
    reads = _random_reads(rng, 120, ref_lens)
    bam = str(tmp_path / "ok.bam")
    write_bam(bam, [("chrA", 5000)], reads, block_payload=900)
    ref = "".join(rng.choice(list("ACGT"), size=5000))
    good_bam, good_bai = open(bam, "rb").read(), open(bam + ".bai", "rb").read()
    text = mpileup_rows(reads, 0, "chrA", 1, 5000).encode()

    def damage(b):
        b = bytearray(b)
        mode = rng.integers(0, 3)
        if mode == 0 and len(b) > 10:
            del b[int(rng.integers(1, len(b))):]
        elif mode == 1:
            for _ in range(int(rng.integers(1, 8))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        else:
            i = int(rng.integers(0, len(b)))
            b[i:i] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 40)), dtype=np.uint8))
        return bytes(b)
    outcomes = {"ok": 0, "err": 0}
    for it in range(200):
        which = it % 3
        try:
            if which == 0:
                p = tmp_path / "bad.bam"
                p.write_bytes(damage(good_bam))
                (tmp_path / "bad.bam.bai").write_bytes(good_bai)
                ColumnPack.from_bam(str(p), "chrA", 1, 5000, ref, 1)
            elif which == 1:
                (tmp_path / "bad2.bai").write_bytes(damage(good_bai))
                ColumnPack.from_bam(bam, "chrA", 1, 5000, ref, 1, bai_fn=str(tmp_path / "bad2.bai"))
            else:
                ColumnPack.from_mpileup(damage(text), ref, 1)
            outcomes["ok"] += 1
        except CtoError:
            outcomes["err"] += 1
    assert outcomes["ok"] + outcomes["err"] == 200 and outcomes["err"] > 20


def test_pack_from_bam_is_invariant_to_the_thread_count(tmp_path, monkeypatch):
    """the multi-threaded range path (CTO_PACK_THREADS > 1: position ranges piled up independently, then merged) yields the
    pack the single-threaded path yields"""
    from clairs_to_amd.pack import ColumnPack
    rng = np.random.default_rng(9)
    L = 60000
    ref = "".join(rng.choice(list("ACGT"), size=L))
    reads = _random_reads(rng, 1500, [L], paired_frac=0.0)
    bam = str(tmp_path / "t.bam")
    write_bam(bam, [("chrA", L)], reads, block_payload=4000)
    bed = [(a, a + 40) for a in range(100, L - 100, 97)]
    packs = {}
    for nt in ("1", "3", "8", "13"):
        monkeypatch.setenv("CTO_PACK_THREADS", nt)
        packs[nt] = {bedded: _pack_arrays(ColumnPack.from_bam(bam, "chrA", 1, L, ref, 1, bed=bed if bedded else None)) for bedded in (False, True)}
    for nt in ("3", "8", "13"):
        for bedded in (False, True):
            _assert_same(packs[nt][bedded], packs["1"][bedded])
    assert len(packs["1"][False]["col_pos"]) > 40000 and len(packs["1"][True]["col_pos"]) > 15000


def test_chunk_span_is_tight_and_sufficient(tmp_path):
    """cto_bam_chunk_span (what is sent to the device inflate for a chunk): the byte range must be cut near the first alignment that
    starts after the region - not run to the end of the coarse bins' chunk lists - and still hold every block the pile-up needs:
    the blocks of the span, inflated here with zlib and handed to cto_pack_from_bam_inflated, give the pack cto_pack_from_bam reads."""
    import ctypes as C
    import os
    import zlib
    from clairs_to_amd._lib import check, lib
    from clairs_to_amd.bgzf import BGZF_PAD, scan
    from clairs_to_amd.pack import ColumnPack
    from clairs_to_amd.synth_run import make_bam_run
    run = make_bam_run(str(tmp_path / "run"), region_kb=600, n_chunks=6, depth=12)
    bam = run["bam_fn"]
    fsize = os.path.getsize(bam)
    ref = open(run["ref_fn"]).read().split("\n", 1)[1].replace("\n", "")
    for lo, hi in ((100_001, 200_000), (1, 50_000), (550_001, 600_000), (299_000, 301_000)):
        fb, fe = C.c_int64(0), C.c_int64(0)
        check(lib.cto_bam_chunk_span(bam.encode(), None, b"chr1", lo, hi, C.byref(fb), C.byref(fe)))
        nbytes = fe.value - fb.value
        # reads are <= 30 kb long: the blocks needed are those of alignments starting in [lo - 30 kb, hi]; a block more on each side is allowed
        assert 0 < nbytes <= fsize * ((hi - lo) + 80_000) / 600_000 + 3 * 65536, (lo, hi, nbytes, fsize)
        raw = np.zeros(nbytes + BGZF_PAD, dtype=np.uint8)
        with open(bam, "rb") as f:
            f.seek(fb.value)
            raw[:nbytes] = np.frombuffer(f.read(nbytes), dtype=np.uint8)
        blocks, out_bytes = scan(raw, nbytes, fb.value)
        inflated = np.zeros(max(out_bytes, 256), dtype=np.uint8)
        for b in blocks:
            data = zlib.decompress(raw[int(b["in_off"]):int(b["in_off"]) + int(b["csize"])].tobytes(), -15)
            assert len(data) == int(b["isize"])
            inflated[int(b["out_off"]):int(b["out_off"]) + len(data)] = np.frombuffer(data, dtype=np.uint8)

        class Host:                      # what pack.from_bam expects of the page-locked tensor
            def __init__(self, a): self.a = a
            def data_ptr(self): return self.a.ctypes.data
            def numel(self): return self.a.size
        p_want = ColumnPack.from_bam(bam, "chr1", lo, hi, ref, 1)                 # numpy() gives views: the packs have to stay alive
        p_got = ColumnPack.from_bam(bam, "chr1", lo, hi, ref, 1, inflated=(Host(inflated), blocks))
        want, got = p_want.numpy(), p_got.numpy()
        assert want["entries"].size > 1000
        for k in want:
            np.testing.assert_array_equal(got[k], want[k], err_msg="%s %d-%d" % (k, lo, hi))


def test_max_depth_cap_does_not_depend_on_the_thread_count(tmp_path):
    """--max-depth is order dependent; a call cut over several decoding threads must give the pack of the unsplit order
    (advisor finding, round 2: depths in (max_depth, 32767] used to depend on the host's core count)"""
    from clairs_to_amd._lib import lib
    from clairs_to_amd.pack import ColumnPack
    rng = np.random.default_rng(21)
    L = 30000
    ref = "".join(rng.choice(list("ACGT"), size=L))
    reads = []
    for i in range(2500):                                   # ~80x over the contig, capped at 25
        pos = int(rng.integers(0, L - 1200))
        n = int(rng.integers(300, 1100))
        reads.append(dict(name="r%d" % i, flag=16 * int(rng.random() < 0.5), ref=0, pos=pos, mapq=60, cigar=[("M", n)],
                          seq="".join(rng.choice(list("ACGT"), size=n)), qual=[30] * n))
    reads.sort(key=lambda r: r["pos"])
    bam = str(tmp_path / "deep.bam")
    write_bam(bam, [("chrA", L)], reads, block_payload=20000)
    packs = []
    for threads in (1, 2, 7):
        lib.cto_set_pack_threads(threads)
        try:
            packs.append(_pack_arrays(ColumnPack.from_bam(bam, "chrA", 1, L, ref, 1, max_depth=25)))
        finally:
            lib.cto_set_pack_threads(0)
    _assert_same(packs[0], packs[1])
    _assert_same(packs[0], packs[2])
    depth = np.diff(packs[0]["col_off"])
    assert depth.max() <= 25 + 5 and (depth >= 25).sum() > 1000       # the cap really bit
    text = mpileup_rows(reads, 0, "chrA", 1, L, max_depth=25, ref_seq=ref, ref_start=1)
    _assert_same(packs[0], _pack_arrays(ColumnPack.from_mpileup(text, ref, 1)))


def test_bam_view_prints_the_real_cigar_of_cg_tag_reads(tmp_path):
    """cto_bam_view (the `samtools view` stand-in of realign_reads --bam_reader native): a record whose CIGAR travels in the CG:B,I
    tag (SAM spec 4.2.2: > 65535 operations; here forced by the writer) prints that CIGAR, not the <l_seq>S<ref_len>N placeholder,
    and is selected by its real reference span - the same row as the identical read written the ordinary way."""
    from clairs_to_amd.realign_reads import bam_view
    refs = [("chrA", 5000)]
    cigar = [("S", 3), ("M", 20), ("I", 2), ("M", 10), ("D", 4), ("M", 15)]
    seq = "ACGTTGCA" * 6 + "AC"
    base = dict(flag=0, ref=0, pos=100, mapq=60, cigar=cigar, seq=seq, qual=[30] * len(seq))
    rows = {}
    for tag in (False, True):
        bam = str(tmp_path / ("cg%d.bam" % tag))
        write_bam(bam, refs, [dict(base, name="r", cg_tag=tag)])
        rows[tag] = bam_view(bam, "chrA", 120, 140)
        assert len(rows[tag]) == 1
        assert bam_view(bam, "chrA", 160, 200) == []                  # behind the read's real end (pos 100 + 49 reference bases)
    assert rows[True] == rows[False]
    assert rows[True][0].split("\t")[5] == "3S20M2I10M4D15M"
