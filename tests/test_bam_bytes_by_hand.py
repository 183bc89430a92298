"""A BAM the suite's own writer (tests/bamutil.py) never touched: the example alignment of the SAM specification (SAMv1 section 1.1:
r001/1, r002, r003, r004, the supplementary line of r003, r001/2 on a 45-base reference) assembled into BAM + BGZF + BAI bytes HERE,
field by field from the BAM chapter of the specification (section 4.2), with nothing shared with bamutil - a reader bug mirrored in that
writer (nibble order, CIGAR packing, bin numbers, virtual offsets) would pass every other BAM test of the suite and fail this one.  The
bytes of the first record are additionally written out by hand below and compared with what the assembler here produces.

The expected pileup rows were derived by hand from the alignment picture of section 1.1 and the column rules at the top of csrc/bam.cpp
(the insertion of r002 behind a padding operation hangs on position 14; r003's supplementary line is dropped by --excl-flags 2316; the
example has no qualities - each read gets a constant one here so that the columns' quality strings say which read a base came from;
the reference-skip of r004, positions 22-35, is kept out of the requested positions).
PARITY UNPINNED against samtools itself (absent from both boxes)."""
import struct
import zlib

import numpy as np

REF = "AGCATGTTAGATAAGATAGCTGTGCTAGTAGGCAGTCAGCGCCAT"      # section 1.1, 45 bases (the padding columns of the picture removed)

# (name, flag, 1-based pos, mapq, cigar, next pos (1-based, 0 = none), tlen, seq, constant quality)
READS = [("r001", 99, 7, 30, "8M2I4M1D3M", 37, 39, "TTAGATAAAGGATACTG", 40),
         ("r002", 0, 9, 30, "3S6M1P1I4M", 0, 0, "AAAAGATAAGGATA", 35),
         ("r003", 0, 9, 30, "5S6M", 0, 0, "GCCTAAGCTAA", 30),
         ("r004", 0, 16, 30, "6M14N5M", 0, 0, "ATAGCTTCAGC", 25),
         ("r003", 2064, 29, 17, "6H5M", 0, 0, "TAGGC", 22),
         ("r001", 147, 37, 30, "9M", 7, -39, "CAGCGGCAT", 20)]

# record of r001/1 by hand (section 4.2: block_size refID pos l_read_name mapq bin n_cigar_op flag l_seq next_refID next_pos tlen | name |
# cigar | seq | qual), little-endian.  bin = reg2bin(6, 22) = 4681 + (6 >> 14) = 0x1249; 8M = 8 << 4 | 0, 2I = 2 << 4 | 1, 1D = 1 << 4 | 2;
# sequence nibbles "=ACMGRSVTWYHKDBN": A 1, C 2, G 4, T 8, first base in the high nibble
R001_BY_HAND = bytes.fromhex(
    "53000000" "00000000" "06000000" "05" "1e" "4912" "0500" "6300" "11000000" "00000000" "24000000" "27000000"
    "7230303100"
    "80000000" "21000000" "40000000" "12000000" "30000000"
    "881418111441812840"
    + "28" * 17)


def _reg2bin(beg, end):                   # section 5.3 of the specification
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def _record(name, flag, pos1, mapq, cigar, next_pos1, tlen, seq, q):
    ops, num = [], ""
    for ch in cigar:
        if ch.isdigit():
            num += ch
        else:
            ops.append((int(num), "MIDNSHP=X".index(ch)))
            num = ""
    ref_len = sum(n for n, op in ops if op in (0, 2, 3, 7, 8))
    nib = ["=ACMGRSVTWYHKDBN".index(c) for c in seq] + [0]
    packed = bytes((nib[i] << 4) | nib[i + 1] for i in range(0, len(seq), 2))
    body = struct.pack("<iiBBHHHiiii", 0, pos1 - 1, len(name) + 1, mapq, _reg2bin(pos1 - 1, pos1 - 1 + ref_len), len(ops), flag, len(seq),
                       0 if next_pos1 else -1, next_pos1 - 1, tlen)
    body += name.encode() + b"\0" + b"".join(struct.pack("<I", (n << 4) | op) for n, op in ops) + packed + bytes([q]) * len(seq)
    return struct.pack("<i", len(body)) + body


def _bgzf(data):                          # section 4.1: gzip member with the BC extra subfield (BSIZE = block size - 1)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    payload = co.compress(data) + co.flush()
    bsize = len(payload) + 26
    return (bytes.fromhex("1f8b08040000000000ff060042430200") + struct.pack("<H", bsize - 1) + payload
            + struct.pack("<II", zlib.crc32(data), len(data)))


EOF_BLOCK = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _write(tmp_path):
    text = b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:ref\tLN:45\n"
    header = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 1) + struct.pack("<i", 4) + b"ref\0" + struct.pack("<i", 45)
    records = [_record(*r) for r in READS]
    assert records[0] == R001_BY_HAND
    block = _bgzf(header + b"".join(records))
    assert len(block) == struct.unpack("<H", block[16:18])[0] + 1
    bam = tmp_path / "spec.bam"
    bam.write_bytes(block + EOF_BLOCK)
    # .bai (section 5.2): one reference, one bin (every read of the example lies in the first 16 kb window: bin 4681) with one chunk
    # from the first record (block 0, offset = the header's length) to the start of the EOF block, one linear-index window
    first, end = len(header), len(block) << 16
    bai = b"BAI\1" + struct.pack("<i", 1) + struct.pack("<i", 1) + struct.pack("<Ii", 4681, 1) + struct.pack("<QQ", first, end) + \
        struct.pack("<i", 1) + struct.pack("<Q", first)
    (tmp_path / "spec.bam.bai").write_bytes(bai)
    return str(bam)


ROWS = [  # pos, bases, qualities (I r001/1, D r002, ? r003, : r004, 5 r001/2); every mapping quality is 30 = '?'
    (7, "T", "I"), (8, "T", "I"), (9, "AAA", "ID?"), (10, "GGG", "ID?"), (11, "AAC", "ID?"), (12, "TTT", "ID?"), (13, "AAA", "ID?"),
    (14, "A+2AGA+1GA", "ID?"), (15, "GG", "ID"), (16, "AAA", "ID:"), (17, "TTT", "ID:"), (18, "A-1NAA", "ID:"), (19, "*G", "I:"),
    (20, "CC", "I:"), (21, "TT", "I:"),
    (36, "T", ":"), (37, "Cc", ":5"), (38, "Aa", ":5"), (39, "Gg", ":5"), (40, "Cc", ":5"), (41, "g", "5"), (42, "g", "5"), (43, "c", "5"),
    (44, "a", "5"), (45, "t", "5")]


def test_the_specifications_example_assembled_by_hand(tmp_path):
    from clairs_to_amd.pack import ColumnPack
    bam = _write(tmp_path)
    got = ColumnPack.from_bam(bam, "ref", 1, 45, REF, 1, bed=[(6, 21), (35, 45)])
    text = "".join("ref\t%d\tN\t%d\t%s\t%s\t%s\n" % (p, len(q), b, q, "?" * len(q)) for p, b, q in ROWS)
    want = ColumnPack.from_mpileup(text, REF, 1)
    a, b = got.numpy(), want.numpy()
    for k in ("col_pos", "col_ref", "col_off", "key_off", "entries", "key_meta", "key_group"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert [got.key_string(i) for i in range(got.n_keys)] == [want.key_string(i) for i in range(want.n_keys)]
    assert got.n_cols == 25 and got.n_keys == 3
    # the same file through the `samtools view` stand-in: the supplementary line is a record like any other there
    from clairs_to_amd.realign_reads import bam_view
    rows = bam_view(bam, "ref", 1, 45)
    assert [r.split("\t")[0] for r in rows] == ["r001", "r002", "r003", "r004", "r003", "r001"]
    f = rows[0].rstrip("\n").split("\t")
    assert f[1:9] == ["99", "ref", "7", "30", "8M2I4M1D3M", "=", "37", "39"] and f[9] == "TTAGATAAAGGATACTG" and f[10] == "I" * 17
    assert rows[1].split("\t")[5] == "3S6M1P1I4M" and rows[4].split("\t")[1:6] == ["2064", "ref", "29", "17", "6H5M"]
