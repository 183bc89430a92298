"""Simulated short-read data for the Illumina realignment filter (tests/golden/gen_realign.py `flow`, tests/test_realign_flow.py).

A 9 kb contig, 2 x 150 paired reads at ~45x from two haplotypes.  The alternative haplotype carries indels; reads that END within a
few bases behind an indel are written the way a seed-and-extend aligner leaves them - ungapped, the indel turned into a run of
mismatches, or soft-clipped - which is what creates low-QUAL false SNV calls next to real indels, the case `realign_variants` exists
for.  Clean SNVs (nothing to realign), low-MQ reads, low-BQ bases, N bases and supplementary / QC-fail flags are mixed in.

Also here: the text of a fake `samtools` (faidx / view / mpileup on the simulated "BAM" / mpileup on SAM from stdin), which answers
with tests/bamutil.py's naive pileup - samtools itself exists on neither box, so what is pinned is everything BETWEEN the samtools
calls."""
import os

import numpy as np

CTG = "chr1"
READ_LEN = 150


def _unique(rng, n):
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    while True:
        s = bytes(rng.choice(bases, n)).decode()
        if len({s[i:i + 12] for i in range(n - 11)}) == n - 11:
            return s


def simulate(seed=20260929, length=9000, depth=45):
    rng = np.random.default_rng(seed)
    ref = _unique(rng, length)
    # (0-based position on the reference, kind, payload): the alternative haplotype
    events = [(1500, "del", 4), (2300, "ins", "GATTC"), (3100, "snv", None), (3900, "del", 9), (4700, "ins", "TTTTTTTG"), (5500, "snv", None),
              (6300, "del", 2), (7100, "ins", "C"), (7800, "del", 15)]
    alt, amap, shift = [], [], 0           # alt haplotype and, per alt base, its reference coordinate (insertions: the anchor)
    i = 0
    ev = {e[0]: e for e in events}
    while i < length:
        if i in ev:
            _, kind, pay = ev[i]
            if kind == "snv":
                alt.append("A" if ref[i] != "A" else "C"); amap.append(i); i += 1
            elif kind == "del":
                alt.append(ref[i]); amap.append(i); i += 1 + pay
            else:
                alt.append(ref[i]); amap.append(i)
                for ch in pay:
                    alt.append(ch); amap.append(-1)
                i += 1
        else:
            alt.append(ref[i]); amap.append(i); i += 1
    alt = "".join(alt)

    def align(h_start, n):
        """true alignment of alt[h_start:h_start+n] to the reference: (0-based pos, [(op, len)])"""
        ops, pos, last = [], None, None
        k = h_start
        while k < h_start + n:
            r = amap[k]
            if r < 0:
                ops.append(["I", 1]) if not ops or ops[-1][0] != "I" else ops[-1].__setitem__(1, ops[-1][1] + 1)
            else:
                if pos is None:
                    pos = r
                if last is not None and r > last + 1:
                    ops.append(["D", r - last - 1])
                ops.append(["M", 1]) if not ops or ops[-1][0] != "M" else ops[-1].__setitem__(1, ops[-1][1] + 1)
                last = r
            k += 1
        while ops and ops[0][0] != "M":                 # a read cannot start inside an insertion here
            ops.pop(0)
        return pos, [(o, ln) for o, ln in ops]

    reads = []
    n_pairs = int(length * depth / (2 * READ_LEN))
    for pi in range(n_pairs):
        hap_alt = rng.random() < 0.5
        src = alt if hap_alt else ref
        frag = int(rng.integers(260, 420))
        fs = int(rng.integers(0, len(src) - frag))
        mq = 60 if rng.random() < 0.9 else int(rng.integers(0, 30))
        for mate in (0, 1):
            hs = fs if mate == 0 else fs + frag - READ_LEN
            seq = list(src[hs:hs + READ_LEN])
            qual = [int(q) for q in np.clip(np.round(rng.normal(36, 3, READ_LEN)), 25, 41)]
            for _ in range(int(rng.integers(0, 3))):                      # sequencing errors, some of them low quality
                p = int(rng.integers(0, READ_LEN))
                seq[p] = "ACGT"[int(rng.integers(0, 4))]
                qual[p] = int(rng.choice([8, 12, 30, 35]))
            if rng.random() < 0.01:
                seq[int(rng.integers(0, READ_LEN))] = "N"
            seq = "".join(seq)
            if hap_alt:
                pos, cigar = align(hs, READ_LEN)
                if sum(n for o, n in cigar if o in "MI") != READ_LEN:     # started inside an insertion: trimmed above
                    lead = READ_LEN - sum(n for o, n in cigar if o in "MI")
                    cigar = [("S", lead)] + cigar
            else:
                pos, cigar = hs, [("M", READ_LEN)]
            # the aligner's view of indels near a read end: <= 10 aligned bases behind (or before) the indel -> no gap
            if len(cigar) > 1:
                ops = [(o, n) for o, n in cigar]
                if ops[-1][0] == "M" and ops[-1][1] <= 10 and ops[-2][0] in "ID":
                    tail = ops[-1][1] + (ops[-2][1] if ops[-2][0] == "I" else 0)
                    style = rng.random()
                    ops = ops[:-2]
                    if style < 0.6:
                        ops[-1] = ("M", ops[-1][1] + tail)                 # ungapped: mismatches behind the indel
                    else:
                        ops.append(("S", tail))
                    cigar = ops
                elif ops[0][0] == "M" and ops[0][1] <= 10 and ops[1][0] in "ID":
                    head = ops[0][1] + (ops[1][1] if ops[1][0] == "I" else 0)          # query bases in front of the surviving run
                    after = pos + ops[0][1] + (ops[1][1] if ops[1][0] == "D" else 0)    # where that run starts on the reference
                    ops = ops[2:]
                    if rng.random() < 0.6:
                        ops[0] = ("M", ops[0][1] + head)
                        pos = after - head
                    else:
                        ops = [("S", head)] + ops
                        pos = after
                    cigar = ops
            # merge neighbours of the same op
            merged = []
            for o, n in cigar:
                if merged and merged[-1][0] == o:
                    merged[-1] = (o, merged[-1][1] + n)
                else:
                    merged.append((o, n))
            rev = mate == 1
            flag = 1 | 2 | (16 if rev else 32) | (64 if mate == 0 else 128)
            r = rng.random()
            if r < 0.01:
                flag |= 2048
            elif r < 0.02:
                flag |= 512
            elif r < 0.03:
                flag &= ~2
            reads.append(dict(name="frag%05d" % pi, flag=flag, ref=0, pos=pos, mapq=mq, cigar=merged, seq=seq, qual=qual))
    reads.sort(key=lambda r: r["pos"])
    for i, r in enumerate(reads):          # mate fields: every pair here maps to the same contig
        r["rnext"], r["pnext"], r["tlen"] = "=", 0, 0
    by = {}
    for r in reads:
        by.setdefault(r["name"], []).append(r)
    for a in by.values():
        if len(a) == 2:
            a[0]["pnext"], a[1]["pnext"] = a[1]["pos"] + 1, a[0]["pos"] + 1
            t = max(x["pos"] + sum(n for o, n in x["cigar"] if o in "MD") for x in a) - min(x["pos"] for x in a)
            a[0]["tlen"], a[1]["tlen"] = (t, -t) if a[0]["pos"] <= a[1]["pos"] else (-t, t)
    return dict(ref=ref, alt=alt, events=events, reads=reads)


def sam_text(sim):
    rows = ["@HD\tVN:1.6\tSO:coordinate\n", "@SQ\tSN:%s\tLN:%d\n" % (CTG, len(sim["ref"]))]
    for r in sim["reads"]:
        rows.append("\t".join([r["name"], str(r["flag"]), CTG, str(r["pos"] + 1), str(r["mapq"]), "".join("%d%s" % (n, o) for o, n in r["cigar"]),
                               r["rnext"], str(r["pnext"]), str(r["tlen"]), r["seq"], "".join(chr(q + 33) for q in r["qual"])]) + "\n")
    return "".join(rows)


def parse_sam(text):
    """SAM rows -> the read dicts tests/bamutil.mpileup_rows takes"""
    reads = []
    for row in text.split("\n"):
        if not row or row[0] == "@":
            continue
        c = row.split("\t")
        cigar, n = [], 0
        for ch in c[5]:
            if ch.isdigit():
                n = n * 10 + int(ch)
            else:
                cigar.append((ch, n)); n = 0
        reads.append(dict(name=c[0], flag=int(c[1]), ref=0, pos=int(c[3]) - 1, mapq=int(c[4]), cigar=cigar, seq=c[9],
                          qual=[ord(q) - 33 for q in c[10]]))
    return reads


def calls_vcf(sim):
    """a pileup VCF as the caller would have left it: low-QUAL PASS calls on and around the events, high-QUAL ones, non-PASS rows"""
    ref = sim["ref"]
    head = ("##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n##FILTER=<ID=LowQual,Description=\"Low quality variant\">\n"
            "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n"
            "##FORMAT=<ID=TU,Number=1,Type=Integer,Description=\"Count of T in the tumor BAM\">\n##extra=dropped by the filter\n"
            "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n")
    rows = []

    def snv(p0, qual, flt="PASS", alt=None):
        a = alt or ("A" if ref[p0] != "A" else "C")
        rows.append((p0, "%s\t%d\t.\t%s\t%s\t%s\t%s\t.\tGT:GQ:DP:AF\t0/1:3:40:0.2500\n" % (CTG, p0 + 1, ref[p0], a, qual, flt)))
    alt_h = sim["alt"]
    for p, kind, pay in sim["events"]:
        if kind == "snv":
            snv(p, "4.1200")                                   # true SNV, low QUAL: realignment must not hurt it
            snv(p + 700 if p + 700 < len(ref) else p - 700, "21.5000")      # high QUAL elsewhere: never looked at
        elif kind == "del":
            for k in (1, 2, 3, 5):                             # false SNVs behind the deletion: ref[p+k+len] read as ref[p+k]
                q = p + pay + k
                if ref[q] != ref[q - pay]:
                    snv(q, "3.%d000" % k, alt=ref[q + pay] if q + pay < len(ref) and ref[q + pay] != ref[q] else None)
            rows.append((p, "%s\t%d\t.\t%s\t%s\t5.5000\tPASS\t.\tGT:GQ:DP:AF\t0/1:5:40:0.4000\n" % (CTG, p + 1, ref[p:p + pay + 1], ref[p])))
        else:
            for k in (1, 2, 4):
                q = p + k
                a = (pay + ref[p + 1:])[k - 1]
                if a != ref[q]:
                    snv(q, "2.%d000" % k, alt=a)
            rows.append((p, "%s\t%d\t.\t%s\t%s\t6.2500\tPASS\t.\tGT:GQ:DP:AF\t0/1:6:40:0.4000\n" % (CTG, p + 1, ref[p], ref[p] + pay)))
    snv(200, "1.0000", flt="LowQual")
    snv(8000, "0.0000", flt="RefCall")
    snv(8700, "2.5000")                                        # a quiet place: nothing to realign
    rows.sort()
    return head + "".join(r for _, r in rows)


SHIM = r'''#!/usr/bin/env python3
# fake samtools for the realignment-flow fixtures: faidx / view / mpileup from files next to the "BAM" (tests/golden/realignsim.py)
import os, sys
sys.path.insert(0, os.environ["REALIGNSIM_TESTS"])
sys.path.insert(0, os.path.join(os.environ["REALIGNSIM_TESTS"], "golden"))
import bamutil, realignsim
a = sys.argv[1:]
def region(s):
    ctg, rng = s.split(":")
    lo, hi = rng.split("-")
    return ctg, int(lo), int(hi)
if a[0] == "faidx":
    ref = open(a[1]).read().split("\n", 1)[1].replace("\n", "")
    ctg, lo, hi = region(a[2])
    sub = ref[lo - 1:hi]
    sys.stdout.write(">%s\n" % a[2])
    for i in range(0, len(sub), 60):
        sys.stdout.write(sub[i:i + 60] + "\n")
elif a[0] == "view":
    bam = a[2]
    ctg, lo, hi = region(a[3])
    min_mq = int(a[a.index("-q") + 1]) if "-q" in a else 0
    for row in open(bam + ".sam"):
        if row[0] == "@":
            sys.stdout.write(row)
            continue
        c = row.split("\t")
        reads = realignsim.parse_sam(row)
        end = reads[0]["pos"] + bamutil.ref_len_of(reads[0]["cigar"])
        if int(c[4]) >= min_mq and reads[0]["pos"] < hi and end > lo - 1:
            sys.stdout.write(row)
elif a[0] == "mpileup":
    src = a[1]
    min_mq = int(a[a.index("--min-MQ") + 1])
    min_bq = int(a[a.index("--min-BQ") + 1])
    excl = int(a[a.index("--excl-flags") + 1])
    text = sys.stdin.read() if src == "-" else open(src + ".sam").read()
    reads = realignsim.parse_sam(text)
    if "-r" in a:
        ctg, lo, hi = region(a[a.index("-r") + 1])
    else:
        lo, hi = 1, max([r["pos"] + bamutil.ref_len_of(r["cigar"]) for r in reads] + [1])
    rows = bamutil.mpileup_rows(reads, 0, realignsim.CTG, lo, hi, excl_flags=excl, min_mq=min_mq)
    for row in rows.split("\n"):
        if not row:
            continue
        c = row.split("\t")
        keep = [i for i, q in enumerate(c[5]) if ord(q) - 33 >= min_bq]
        # tokens of the base string, one per read entry
        toks, s, i = [], c[4], 0
        while i < len(s):
            j = i + 1
            if j < len(s) and s[j] in "+-":
                k = j + 1
                n = 0
                while s[k].isdigit():
                    n = n * 10 + int(s[k]); k += 1
                j = k + n
            toks.append(s[i:j]); i = j
        if "--reverse-del" not in a:
            toks = [t.replace("#", "*") for t in toks]
        sys.stdout.write("\t".join([c[0], c[1], c[2], str(len(keep)), "".join(toks[i] for i in keep) or "*", "".join(c[5][i] for i in keep) or "*"]) + "\n")
else:
    sys.exit(1)
'''


def write_inputs(sim, d):
    """ref.fa(+.fai), fake.bam (+ .sam text the shim reads), calls.vcf, samtools shim -> paths"""
    import stat
    ref = sim["ref"]
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "ref.fa"), "w").write(">%s\n%s\n" % (CTG, ref))
    open(os.path.join(d, "ref.fa.fai"), "w").write("%s\t%d\t6\t%d\t%d\n" % (CTG, len(ref), len(ref), len(ref) + 1))
    open(os.path.join(d, "fake.bam"), "w").write("")
    open(os.path.join(d, "fake.bam.sam"), "w").write(sam_text(sim))
    open(os.path.join(d, "calls.vcf"), "w").write(calls_vcf(sim))
    shim = os.path.join(d, "samtools")
    open(shim, "w").write(SHIM)
    os.chmod(shim, os.stat(shim).st_mode | stat.S_IEXEC)
    return dict(ref=os.path.join(d, "ref.fa"), bam=os.path.join(d, "fake.bam"), vcf=os.path.join(d, "calls.vcf"), samtools=shim)
