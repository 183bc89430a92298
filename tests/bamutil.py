"""Test-only BAM tooling (no samtools / htslib on either box): a minimal BAM + BAI writer and an independent, deliberately
naive restatement of the `samtools mpileup --reverse-del --output-MQ --min-BQ 0` column rules that csrc/bam.cpp lists.
Formats per the SAM/BAM specification v1 (sections 4.1 BGZF, 4.2 BAM, 5.1-5.3 indexing)."""
import struct
import zlib

CIGAR_OPS = "MIDNSHP=X"
NT16 = "=ACMGRSVTWYHKDBN"
CONSUMES_REF = set("MDN=X")
CONSUMES_QUERY = set("MIS=X")


def reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def ref_len_of(cigar):
    return sum(n for op, n in cigar if op in CONSUMES_REF)


def _bgzf_block(data):
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25
    head = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, ord("B"), ord("C"), 2, bsize)
    return head + comp + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))


def _record(read, tid):
    name = read["name"].encode() + b"\0"
    cigar = read["cigar"]
    seq, qual = read["seq"], read["qual"]
    pos = read["pos"]
    end = pos + max(1, ref_len_of(cigar))
    body = struct.pack("<iiBBHHHiiii", tid, pos, len(name), read["mapq"], reg2bin(pos, end), 2 if read.get("cg_tag") else len(cigar),
                       read["flag"], len(seq), -1, -1, 0)
    body += name
    real = b"".join(struct.pack("<I", (n << 4) | CIGAR_OPS.index(op)) for op, n in cigar)
    if read.get("cg_tag"):
        # long-CIGAR convention (SAM spec 4.2.2): placeholder <l_seq>S<ref_len>N in the CIGAR field, the real one in CG:B,I
        body_cigar = struct.pack("<II", (len(seq) << 4) | 4, (max(1, ref_len_of(cigar)) << 4) | 3)
    else:
        body_cigar = real
    body += body_cigar
    codes = [NT16.index(c) for c in seq]
    if len(codes) & 1:
        codes.append(0)
    body += bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(codes), 2))
    body += bytes(qual if qual is not None else [0xff] * len(seq))
    # a few auxiliary fields of every kind the reader has to skip over
    body += b"NMi" + struct.pack("<i", 3) + b"XAZ" + b"chr9,+1,5M;" + b"\0" + b"xsA" + b"+" + b"mlBC" + struct.pack("<I", 3) + bytes([1, 2, 3])
    if read.get("cg_tag"):
        body += b"CGBI" + struct.pack("<I", len(cigar)) + real
    body += b"ASs" + struct.pack("<h", -7)
    return struct.pack("<i", len(body)) + body


def write_bam(path, refs, reads, block_payload=3000):
    """refs: [(name, length)]; reads: dicts(name, flag, ref (index), pos 0-based, mapq, cigar [(op, n)], seq, qual list|None),
    already in coordinate order.  Writes path and path + '.bai'.  Small blocks on purpose: records straddle block boundaries."""
    header = b"BAM\1" + struct.pack("<i", 0) + struct.pack("<i", len(refs))
    for name, length in refs:
        header += struct.pack("<i", len(name) + 1) + name.encode() + b"\0" + struct.pack("<i", length)
    stream = bytearray(header)
    spans = []                                     # (ustart, uend, read)
    for r in reads:
        rec = _record(r, r["ref"])
        spans.append((len(stream), len(stream) + len(rec), r))
        stream += rec
    # cut into blocks, remember each block's file offset
    blocks, ustarts, coffs = [], [], []
    off = 0
    for u in range(0, len(stream), block_payload):
        b = _bgzf_block(bytes(stream[u:u + block_payload]))
        ustarts.append(u)
        coffs.append(off)
        blocks.append(b)
        off += len(b)
    eof = _bgzf_block(b"")
    with open(path, "wb") as f:
        for b in blocks:
            f.write(b)
        f.write(eof)

    def voff(u):
        if u == len(stream):                        # end of the last record = start of the EOF block
            return off << 16
        k = u // block_payload
        return (coffs[k] << 16) | (u - ustarts[k])
    # ---- index ----
    bins = [dict() for _ in refs]
    lin = [dict() for _ in refs]
    for ustart, uend, r in spans:
        if r["flag"] & 4:
            continue
        t, beg = r["ref"], r["pos"]
        end = beg + max(1, ref_len_of(r["cigar"]))
        vb, ve = voff(ustart), voff(uend)
        ch = bins[t].setdefault(reg2bin(beg, end), [])
        if ch and ch[-1][1] == vb:
            ch[-1][1] = ve
        else:
            ch.append([vb, ve])
        for w in range(beg >> 14, ((end - 1) >> 14) + 1):
            lin[t][w] = min(lin[t].get(w, vb), vb)
    out = b"BAI\1" + struct.pack("<i", len(refs))
    for t in range(len(refs)):
        out += struct.pack("<i", len(bins[t]))
        for b, chunks in sorted(bins[t].items()):
            out += struct.pack("<Ii", b, len(chunks))
            for vb, ve in chunks:
                out += struct.pack("<QQ", vb, ve)
        n_intv = (max(lin[t]) + 1) if lin[t] else 0
        out += struct.pack("<i", n_intv)
        last = 0
        for w in range(n_intv):
            last = lin[t].get(w, last)
            out += struct.pack("<Q", last)
    with open(path + ".bai", "wb") as f:
        f.write(out)


def _aligned_positions(r):
    """{0-based reference position: query index} of the aligned (M / = / X) bases of a read"""
    out, rp, qp = {}, r["pos"], 0
    for op, n in r["cigar"]:
        if op in "M=X":
            for k in range(n):
                out[rp + k] = qp + k
        if op in CONSUMES_REF:
            rp += n
        if op in CONSUMES_QUERY:
            qp += n
    return out


def _with_overlaps_softened(reads, ref_index, start, end, excl_flags, min_mq, max_depth):
    """mpileup's read-pair overlap handling (on by default), naive form: copies of the reads with the qualities of overlapping
    mates adjusted.  Only reads that make it into the pileup take part (same acceptance rules and max-depth cut as below);
    the mate that entered first keeps min(200, qa + qb) where the bases agree, the better base keeps int(0.8 * q) where they
    differ, the other base drops to 0."""
    out, active_ends, by_name, prev_start = [], [], {}, 0
    for r in reads:
        r = dict(r)
        used = not (r["ref"] != ref_index or (r["flag"] & excl_flags) or (r["flag"] & 4) or r["mapq"] < min_mq)
        used = used and not ((r["flag"] & 1) and not (r["flag"] & 2))
        used = used and bool(r["cigar"]) and bool(r["seq"]) and sum(n for op, n in r["cigar"] if op in CONSUMES_QUERY) == len(r["seq"])
        rl = ref_len_of(r["cigar"]) if r["cigar"] else 0
        used = used and rl > 0 and not (r["pos"] >= end or r["pos"] + rl <= start - 1)
        if used and max_depth > 0 and sum(1 for e in active_ends if e > r["pos"]) >= max_depth:
            used = False
        if used:
            active_ends.append(r["pos"] + rl)
            if (r["flag"] & 1) and r["qual"] is not None:
                r["qual"] = list(r["qual"])
                mate = by_name.get(r["name"])
                if mate is not None and mate["pos"] + ref_len_of(mate["cigar"]) > r["pos"]:
                    # htslib edits the qualities when the second mate is pushed, with the iterator standing at the start of the
                    # read pushed before it: the first mate's columns up to there were printed with the old values (this only
                    # shows on deletion placeholders, which print the quality of the base AFTER the deletion)
                    mate.setdefault("qual_before", list(mate["qual"]))
                    mate.setdefault("adjusted_from", prev_start + 1)
                    pa, pb = _aligned_positions(mate), _aligned_positions(r)
                    for p in sorted(set(pa) & set(pb)):
                        qa, qb = pa[p], pb[p]
                        if mate["seq"][qa] == r["seq"][qb]:
                            mate["qual"][qa], r["qual"][qb] = min(200, mate["qual"][qa] + r["qual"][qb]), 0
                        elif mate["qual"][qa] >= r["qual"][qb]:
                            mate["qual"][qa], r["qual"][qb] = int(0.8 * mate["qual"][qa]), 0
                        else:
                            mate["qual"][qa], r["qual"][qb] = 0, int(0.8 * r["qual"][qb])
                by_name.setdefault(r["name"], r)
            prev_start = r["pos"]
        out.append(r)
    return out


def mpileup_rows(reads, ref_index, ctg, start, end, bed=None, excl_flags=2316, min_mq=0, max_depth=8000, ref_seq=None,
                 ref_start=1):
    """Naive per-position pileup -> mpileup text (7 columns).  start / end 1-based inclusive; bed: 0-based [b, e) intervals."""
    cols = {}                                       # 1-based position -> list of (token, bq, mq)
    active_ends = []
    reads = _with_overlaps_softened(reads, ref_index, start, end, excl_flags, min_mq, max_depth)
    for r in reads:
        if r["ref"] != ref_index or (r["flag"] & excl_flags) or (r["flag"] & 4) or r["mapq"] < min_mq:
            continue
        if (r["flag"] & 1) and not (r["flag"] & 2):
            continue
        cigar, seq = r["cigar"], r["seq"]
        if not cigar or not seq or sum(n for op, n in cigar if op in CONSUMES_QUERY) != len(seq):
            continue
        rl = ref_len_of(cigar)
        if rl == 0 or r["pos"] >= end or r["pos"] + rl <= start - 1:
            continue
        if max_depth > 0 and sum(1 for e in active_ends if e > r["pos"]) >= max_depth:
            continue
        active_ends.append(r["pos"] + rl)
        qual = r["qual"] if r["qual"] is not None else [0] * len(seq)
        rev = bool(r["flag"] & 16)
        mq = min(r["mapq"], 93)
        rp, qp = r["pos"], 0
        ops = [(op, n) for op, n in cigar]
        for i, (op, n) in enumerate(ops):
            if op in "M=X":
                for k in range(n):
                    b = seq[qp + k]
                    if b == "=":
                        ri = rp + k + 1 - ref_start
                        b = ref_seq[ri].upper() if ref_seq is not None and 0 <= ri < len(ref_seq) else "N"
                    if b not in "ACGT":
                        b = "N"
                    tok = b.lower() if rev else b
                    if k == n - 1:
                        j = i + 1
                        while j < len(ops) and ops[j][0] == "P":
                            j += 1
                        if j < len(ops) and ops[j][0] == "I":
                            ins = "".join("N" if c == "=" else c for c in seq[qp + n: qp + n + ops[j][1]])
                            tok += "+%d%s" % (ops[j][1], ins.lower() if rev else ins)
                        elif j < len(ops) and ops[j][0] == "D":
                            tok += "-%d%s" % (ops[j][1], ("n" if rev else "N") * ops[j][1])
                    cols.setdefault(rp + k + 1, []).append((tok, min(qual[qp + k], 93), mq))
                rp += n
                qp += n
            elif op == "D":
                for k in range(n):
                    src = r["qual_before"] if (rp + k + 1) < r.get("adjusted_from", 0) else qual
                    bq = min(src[qp], 93) if qp < len(src) else 0
                    cols.setdefault(rp + k + 1, []).append(("#" if rev else "*", bq, mq))
                rp += n
            elif op == "N":
                rp += n
            elif op in "IS":
                qp += n
    rows = []
    for pos in sorted(cols):
        if pos < start or pos > end:
            continue
        if bed is not None and not any(b <= pos - 1 < e for b, e in bed):
            continue
        toks = cols[pos]
        rows.append("%s\t%d\tN\t%d\t%s\t%s\t%s\n" % (ctg, pos, len(toks), "".join(t[0] for t in toks),
                                                        "".join(chr(t[1] + 33) for t in toks), "".join(chr(t[2] + 33) for t in toks)))
    return "".join(rows)
