"""Hand-derived mpileup rows for the column rules csrc/bam.cpp lists, one small BAM per rule: the expected text below was written
out by hand from samtools-mpileup(1) / the SAM specification as restated at the top of bam.cpp (overlapping mates, an insertion
before the first aligned base, a deletion followed by an insertion, the --max-depth cut, orphan and excluded reads) - NOT
produced by any program.  The native reader must yield the pack the text tokeniser yields from these rows.

PARITY UNPINNED against samtools itself (absent from both boxes): these vectors pin the reader to its documented rules."""
import numpy as np

from bamutil import write_bam

REF = "ACGTACGTACGTACGTACGTACGT"          # chr1, 24 bp;  1-based position p has base REF[p - 1]


def _pack_of_bam(tmp_path, reads, **kw):
    from clairs_to_amd.pack import ColumnPack
    bam = str(tmp_path / "t.bam")
    reads = sorted(reads, key=lambda r: r["pos"])
    write_bam(bam, [("chr1", len(REF))], [dict(ref=0, **r) for r in reads], block_payload=200)
    return ColumnPack.from_bam(bam, "chr1", 1, len(REF), REF, 1, **kw)


def _same_pack(pack, rows):
    from clairs_to_amd.pack import ColumnPack
    want = ColumnPack.from_mpileup("".join(r + "\n" for r in rows), REF, 1)
    a, b = pack.numpy(), want.numpy()
    for k in ("col_pos", "col_ref", "col_off", "key_off", "entries", "key_meta", "key_group"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert [pack.key_string(i) for i in range(pack.n_keys)] == [want.key_string(i) for i in range(want.n_keys)]


def read(name, flag, pos, cigar, seq, qual, mapq=60):
    return dict(name=name, flag=flag, pos=pos, mapq=mapq, cigar=cigar, seq=seq, qual=qual)


def test_overlapping_mates(tmp_path):
    """first mate 1-8 forward (Q30), second mate 5-12 reverse (Q20, Q35 on a mismatch at 8).  Positions 5-7: the mates agree ->
    the first keeps 30 + 20 = 50 ('S'), the second drops to 0 ('!').  Position 8: they differ and the second is better ->
    it keeps int(0.8 * 35) = 28 ('='), the first drops to 0."""
    a = read("p1", 1 | 2 | 64, 0, [("M", 8)], "ACGTACGT", [30] * 8)
    b = read("p1", 1 | 2 | 128 | 16, 4, [("M", 8)], "ACGAACGT", [20, 20, 20, 35, 20, 20, 20, 20])
    rows = ["chr1\t1\tN\t1\tA\t?\t]", "chr1\t2\tN\t1\tC\t?\t]", "chr1\t3\tN\t1\tG\t?\t]", "chr1\t4\tN\t1\tT\t?\t]",
            "chr1\t5\tN\t2\tAa\tS!\t]]", "chr1\t6\tN\t2\tCc\tS!\t]]", "chr1\t7\tN\t2\tGg\tS!\t]]", "chr1\t8\tN\t2\tTa\t!=\t]]",
            "chr1\t9\tN\t1\ta\t5\t]", "chr1\t10\tN\t1\tc\t5\t]", "chr1\t11\tN\t1\tg\t5\t]", "chr1\t12\tN\t1\tt\t5\t]"]
    _same_pack(_pack_of_bam(tmp_path, [a, b]), rows)


def test_overlap_sum_is_capped_and_deletions_are_left_alone(tmp_path):
    """agreeing Q150 + Q90 -> capped at 200, printed as 93 ('~'); where the second mate has a deletion there is no second base:
    the first mate's quality stays as it is"""
    a = read("p2", 1 | 2 | 64, 0, [("M", 6)], "ACGTAC", [150, 150, 40, 40, 40, 40])
    b = read("p2", 1 | 2 | 128 | 16, 0, [("M", 2), ("D", 2), ("M", 2)], "ACAC", [90, 90, 10, 10])
    rows = ["chr1\t1\tN\t2\tAa\t~!\t]]", "chr1\t2\tN\t2\tCc-2nn\t~!\t]]",
            # '#' carries the BQ of the query base after the deletion - which the overlap at position 5 has already set to 0
            # (the adjustment edits the record's qualities when the second mate enters, before any of its columns is emitted)
            "chr1\t3\tN\t2\tG#\tI!\t]]", "chr1\t4\tN\t2\tT#\tI!\t]]",
            "chr1\t5\tN\t2\tAa\tS!\t]]", "chr1\t6\tN\t2\tCc\tS!\t]]"]
    _same_pack(_pack_of_bam(tmp_path, [a, b]), rows)


def test_insertion_before_first_base_and_after_deletion(tmp_path):
    """2I3M2D1I3M at position 3: the leading insertion and the one that follows the deletion have no aligned base of the read in
    front of them and are not reported; the deletion hangs on the last aligned base before it; its placeholders carry the BQ of
    the query base that follows the deletion (here the unreported inserted base, Q13)"""
    c = read("s1", 0, 2, [("I", 2), ("M", 3), ("D", 2), ("I", 1), ("M", 3)], "TTGTACTAC", [40, 40, 40, 40, 40, 13, 40, 40, 40])
    rows = ["chr1\t3\tN\t1\tG\tI\t]", "chr1\t4\tN\t1\tT\tI\t]", "chr1\t5\tN\t1\tA-2NN\tI\t]", "chr1\t6\tN\t1\t*\t.\t]",
            "chr1\t7\tN\t1\t*\t.\t]", "chr1\t8\tN\t1\tT\tI\t]", "chr1\t9\tN\t1\tA\tI\t]", "chr1\t10\tN\t1\tC\tI\t]"]
    _same_pack(_pack_of_bam(tmp_path, [c]), rows)


def test_max_depth_cut(tmp_path):
    """--max-depth 2: the third read starts while two are active and is dropped; the fourth starts after both ended and is kept"""
    d1 = read("d1", 0, 0, [("M", 6)], "ACGTAC", [30] * 6)
    d2 = read("d2", 16, 1, [("M", 6)], "CGTACG", [31] * 6)
    d3 = read("d3", 0, 2, [("M", 6)], "GTACGT", [32] * 6)
    d4 = read("d4", 0, 8, [("M", 4)], "ACGT", [33] * 4)
    rows = ["chr1\t1\tN\t1\tA\t?\t]", "chr1\t2\tN\t2\tCc\t?@\t]]", "chr1\t3\tN\t2\tGg\t?@\t]]", "chr1\t4\tN\t2\tTt\t?@\t]]",
            "chr1\t5\tN\t2\tAa\t?@\t]]", "chr1\t6\tN\t2\tCc\t?@\t]]", "chr1\t7\tN\t1\tg\t@\t]",
            "chr1\t9\tN\t1\tA\tB\t]", "chr1\t10\tN\t1\tC\tB\t]", "chr1\t11\tN\t1\tG\tB\t]", "chr1\t12\tN\t1\tT\tB\t]"]
    _same_pack(_pack_of_bam(tmp_path, [d1, d2, d3, d4], max_depth=2), rows)


def test_orphans_and_excluded_flags(tmp_path):
    """kept: unpaired, proper pairs, duplicates (1024) and QC-fail (512).  dropped: paired without PROPER_PAIR (orphan, mpileup
    without -A), mate unmapped (8), secondary (256), supplementary (2048), unmapped (4)"""
    keep = [read("k1", 0, 0, [("M", 4)], "ACGT", [30] * 4), read("k2", 1 | 2 | 64, 0, [("M", 4)], "ACGT", [31] * 4),
            read("k3", 1024, 0, [("M", 4)], "ACGT", [32] * 4), read("k4", 512 | 16, 0, [("M", 4)], "ACGT", [33] * 4)]
    drop = [read("x1", 1 | 64, 0, [("M", 4)], "ACGT", [40] * 4), read("x2", 1 | 2 | 8, 0, [("M", 4)], "ACGT", [40] * 4),
            read("x3", 256, 0, [("M", 4)], "ACGT", [40] * 4), read("x4", 2048, 0, [("M", 4)], "ACGT", [40] * 4),
            read("x5", 4, 0, [("M", 4)], "ACGT", [40] * 4)]
    rows = ["chr1\t%d\tN\t4\t%s\t?@AB\t]]]]" % (p + 1, b + b + b + b.lower()) for p, b in enumerate("ACGT")]
    _same_pack(_pack_of_bam(tmp_path, [keep[0], drop[0], keep[1], drop[1], keep[2], drop[2], keep[3], drop[3], drop[4]]), rows)
