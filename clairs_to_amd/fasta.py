"""Minimal indexed-FASTA reader (replaces the `samtools faidx` sub-process of shared/utils.py:148-174)."""
import os


def read_region(fasta_fn, ctg, start, end, as_bytes=False):
    """1-based inclusive [start, end] of contig `ctg`, upper-cased like reference_sequence_from(); clipped to the
    contig.  Needs <fasta>.fai (or <fasta without extension>.fai, as file_path_from(..., sep='.') accepts).
    as_bytes: return `bytes` (what the C entry points and numpy take as is) instead of `str`."""
    fai = fasta_fn + ".fai"
    if not os.path.exists(fai):
        alt = ".".join(fasta_fn.split(".")[:-1]) + ".fai"
        if os.path.exists(alt):
            fai = alt
        else:
            raise FileNotFoundError("[ERROR] file %s not found" % fai)
    rec = None
    with open(fai) as f:
        for row in f:
            c = row.rstrip("\n").split("\t")
            if c[0] == ctg:
                rec = (int(c[1]), int(c[2]), int(c[3]), int(c[4]))
                break
    if rec is None:
        raise KeyError("contig %s not in %s" % (ctg, fai))
    length, offset, linebases, linewidth = rec
    start = max(1, int(start))
    end = min(length, int(end))
    if end < start:
        return b"" if as_bytes else ""
    s0, e0 = start - 1, end
    b0 = offset + (s0 // linebases) * linewidth + s0 % linebases
    b1 = offset + ((e0 - 1) // linebases) * linewidth + (e0 - 1) % linebases + 1
    with open(fasta_fn, "rb") as f:
        if f.read(2) == b"\x1f\x8b":
            # `samtools faidx` also serves bgzip-compressed references through their .gzi index; byte offsets of the .fai
            # mean nothing inside a compressed stream, so refuse loudly instead of returning garbage
            raise ValueError("[ERROR] %s is gzip / bgzip compressed: decompress it (and re-run samtools faidx) before use" % fasta_fn)
        f.seek(b0)
        raw = f.read(b1 - b0)
    seq = raw.replace(b"\n", b"").replace(b"\r", b"").upper()
    return seq if as_bytes else seq.decode()
