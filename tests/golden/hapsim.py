"""Synthetic phased long-read pileups for the haplotype-filter fixtures: what
`samtools mpileup --min-MQ q --min-BQ q --excl-flags 2316 -l <bed> -r <region> --output-MQ --output-QNAME --output-extra HP`
prints for a haplotagged tumour BAM - nine columns: chr, pos, ref, depth, bases (with ^<mq> read starts and $ read ends), BQ, MQ,
read names, HP tags ('*' when a read has none) - on a small contig seeded with germline variants on two haplotypes and a
catalogue of somatic / artefact calls designed to trip every filter of src/haplotype_filtering.py at least once.
Input synthesis for gen_golden.py only (build container); the product never imports this."""
import numpy as np

ACGT = "ACGT"


def simulate(seed=7, length=9000, n_reads=420):
    rng = np.random.default_rng(seed)
    ref = "".join(ACGT[i] for i in rng.integers(0, 4, size=length))
    ref = list(ref)
    # a low-complexity stretch (homopolymer + dinucleotide repeat) for the sequence-entropy filter
    for i in range(5000, 5040):
        ref[i] = "A"
    for i in range(5040, 5080):
        ref[i] = "AC"[(i - 5040) & 1]
    ref = "".join(ref)
    other = lambda b, k=1: ACGT[(ACGT.index(b) + k) % 4]

    reads = []
    for i in range(n_reads):
        ln = int(rng.integers(1500, 4000))
        s = int(rng.integers(-1000, length - 500))
        e = min(length, s + ln)
        s = max(0, s)
        if e - s < 300:
            continue
        hap = 1 + int(rng.integers(0, 2))
        reads.append(dict(name="r%d" % i, s=s, e=e, rev=bool(rng.random() < 0.5), hap=hap, tagged=bool(rng.random() < 0.8),
                          mq=60 if rng.random() < 0.9 else int(rng.integers(21, 60)), edits={}, bq={}))
    reads.sort(key=lambda r: r["s"])

    def covering(p, margin=0):
        return [r for r in reads if r["s"] + margin <= p < r["e"] - margin]

    germ, calls = [], []          # (pos1, ref, alt, gt)   /   (pos1, ref, alt)
    # ---- germline: heterozygous SNPs alternating between the haplotypes, a few indels, some homozygous ----
    p = 150
    k = 0
    while p < length - 200:
        if 4950 < p < 5150:
            p += 230
            continue
        kind = ("snp", "snp", "snp", "ins", "snp", "del", "hom", "snp", "homins")[k % 9]
        hap = 1 + (k % 2)
        rb = ref[p]
        if kind in ("snp", "hom"):
            ab = other(rb)
            for r in covering(p):
                if kind == "hom" or r["hap"] == hap:
                    r["edits"][p] = ("X", ab)
            germ.append((p + 1, rb, ab, "1/1" if kind == "hom" else "0/1"))
        elif kind in ("ins", "homins"):
            ins = "GT" if kind == "ins" else "C"
            for r in covering(p, 2):
                if kind == "homins" or r["hap"] == hap:
                    r["edits"][p] = ("I", ins)
            germ.append((p + 1, rb, rb + ins, "1/1" if kind == "homins" else "0/1"))
        else:
            for r in covering(p, 4):
                if r["hap"] == hap:
                    r["edits"][p] = ("D", 2)
            germ.append((p + 1, ref[p:p + 3], rb, "0/1"))
        p += int(rng.integers(170, 330))
        k += 1

    germ_pos = {g[0] - 1 for g in germ}

    def free_site(lo, hi):
        while True:
            q = int(rng.integers(lo, hi))
            if all(abs(q - g) > 6 for g in germ_pos) and all(abs(q - c[0] + 1) > 12 for c in calls) and not (4940 < q < 5100):
                return q

    def put_snv(q, pick, bq=None):
        ab = other(ref[q], 2)
        n = 0
        for r in covering(q):
            if pick(r) and q not in r["edits"]:
                r["edits"][q] = ("X", ab)
                if bq is not None:
                    r["bq"][q] = bq
                n += 1
        calls.append((q + 1, ref[q], ab))
        return n

    # 1. clean somatic SNVs on a fraction of one haplotype's reads
    for h in (1, 2, 1, 2):
        put_snv(free_site(300, length - 300), lambda r, h=h: r["hap"] == h and rng.random() < 0.55)
    # 2. low-AF calls seen on both haplotypes (MultiHap)
    for _ in range(3):
        put_snv(free_site(300, length - 300), lambda r: rng.random() < 0.09)
    # 3. strand bias: alternative allele on forward reads only
    for _ in range(3):
        put_snv(free_site(300, length - 300), lambda r: (not r["rev"]) and rng.random() < 0.5)
    # 4. alternative allele carried by reads that start or end right there (ReadStartEnd)
    for _ in range(3):
        q = free_site(600, length - 600)
        near = [r for r in reads if abs(r["s"] - q) < 400 or abs(r["e"] - q) < 400]
        for r in near[:14]:
            if rng.random() < 0.5:
                r["s"] = q if r["s"] < q + 1 and abs(r["s"] - q) < 400 else r["s"]
            else:
                r["e"] = q + 1 if abs(r["e"] - q) < 400 else r["e"]
        put_snv(q, lambda r, q=q: r["s"] == q or r["e"] == q + 1 or rng.random() < 0.05)
    # 5. low base quality / low mapping quality support
    put_snv(free_site(300, length - 300), lambda r: r["hap"] == 1 and rng.random() < 0.5, bq=12)
    put_snv(free_site(300, length - 300), lambda r: r["hap"] == 2 and rng.random() < 0.5, bq=19)
    q = free_site(300, length - 300)
    for r in covering(q):
        if r["hap"] == 1 and rng.random() < 0.5:
            r["mq"] = 20                   # kept by --min-MQ 20, and an average of 20 fails `> 20`
    put_snv(q, lambda r: r["mq"] == 20)
    q_low_mq = q
    # 6. variant clusters: the supporting reads share further mismatches / a long insertion close by
    for _ in range(3):
        q = free_site(400, length - 400)
        sup = [r for r in covering(q, 40) if r["hap"] == 1 and rng.random() < 0.5]
        ids = {id(r) for r in sup}
        put_snv(q, lambda r: id(r) in ids)
        for d in (-23, -9, 14, 31):
            if (q + d) in germ_pos:
                continue
            ab = other(ref[q + d], 3)
            for r in sup:
                if q + d not in r["edits"]:
                    r["edits"][q + d] = ("X", ab)
    q = free_site(400, length - 400)
    sup = [r for r in covering(q, 60) if r["hap"] == 2 and rng.random() < 0.6]
    ids = {id(r) for r in sup}
    put_snv(q, lambda r: id(r) in ids)
    for r in covering(q + 20, 5):
        if q + 20 not in r["edits"]:
            r["edits"][q + 20] = ("I", "ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT" * 3)
    # 7. no ancestry: a call on hap-1 reads that do NOT carry hap 1's own germline allele next to it
    for g in [g for g in germ if g[3] == "0/1" and len(g[1]) == 1 and len(g[2]) == 1][2:14:4]:
        gp = g[0] - 1
        q = gp + 37
        if q in germ_pos or any(abs(q - c[0] + 1) < 12 for c in calls):
            continue
        carriers = [r for r in covering(q) if r["s"] <= gp < r["e"] and r["edits"].get(gp, (None,))[0] == "X"]
        if not carriers:
            continue
        h = carriers[0]["hap"]
        sup = [r for r in covering(q) if r["hap"] == h and r["s"] <= gp < r["e"] and rng.random() < 0.5]
        for r in sup:                      # strip the germline allele from the supporting reads
            r["edits"].pop(gp, None)
        ids = {id(r) for r in sup}
        put_snv(q, lambda r: id(r) in ids)
    # 8. homozygous germline variant missing on the supporting reads
    for g in [g for g in germ if g[3] == "1/1" and len(g[2]) == 1][:2]:
        gp = g[0] - 1
        q = gp - 41
        if q in germ_pos:
            q += 3
        sup = [r for r in covering(q) if r["s"] <= gp < r["e"] and rng.random() < 0.3]
        for r in sup:
            r["edits"].pop(gp, None)
        ids = {id(r) for r in sup}
        put_snv(q, lambda r: id(r) in ids)
    # 9. a call close to the contig start (window clipped at position 1)
    put_snv(60, lambda r: r["hap"] == 1 and rng.random() < 0.6)
    # 10. indel calls: clean insertion / deletion, one inside the low-complexity stretch, a strand-biased one
    indel_calls = []

    def put_indel(q, kind, pick, seq="TG", dl=3):
        n = 0
        for r in covering(q, dl + 3):
            if pick(r) and q not in r["edits"]:
                r["edits"][q] = ("I", seq) if kind == "I" else ("D", dl)
                n += 1
        indel_calls.append((q + 1, ref[q], ref[q] + seq) if kind == "I" else (q + 1, ref[q:q + dl + 1], ref[q]))
        return n
    put_indel(free_site(300, length - 300), "I", lambda r: r["hap"] == 1 and rng.random() < 0.6)
    put_indel(free_site(300, length - 300), "D", lambda r: r["hap"] == 2 and rng.random() < 0.6)
    put_indel(free_site(300, length - 300), "I", lambda r: rng.random() < 0.25, seq="A")
    put_indel(5020, "D", lambda r: r["hap"] == 1 and rng.random() < 0.5, dl=2)
    put_indel(5062, "I", lambda r: r["hap"] == 2 and rng.random() < 0.5, seq="AC")
    put_indel(free_site(300, length - 300), "D", lambda r: (not r["rev"]) and rng.random() < 0.6, dl=4)
    put_indel(free_site(300, length - 300), "I", lambda r: r["hap"] == 2 and rng.random() < 0.5, seq="GGA")
    # sequencing noise
    for r in reads:
        for q in rng.integers(r["s"], r["e"], size=max(1, (r["e"] - r["s"]) // 150)).tolist():
            if q not in r["edits"] and not any(q - 6 <= k2 <= q for k2 in r["edits"]):
                r["edits"][q] = ("X", other(ref[q], int(rng.integers(1, 4))))
                r["bq"][q] = int(rng.integers(5, 40))
    for r in reads:                        # the low-MQ call keeps only its MQ-20 supporters (no well-mapped noise read on it)
        if r["mq"] != 20:
            r["edits"].pop(q_low_mq, None)
    reads = [r for r in reads if r["e"] - r["s"] >= 50]
    reads.sort(key=lambda r: r["s"])
    return dict(ref=ref, reads=reads, germline=germ, snv_calls=sorted(calls), indel_calls=sorted(indel_calls), rng=rng)


def pileup_rows(sim, positions, ctg="chr1", min_bq=0, min_mq=20):
    """nine-column rows for the given 1-based positions (sorted)"""
    ref, reads = sim["ref"], sim["reads"]
    rows = {}
    want = set(positions)
    for r in reads:
        if r["mq"] < min_mq:
            continue
        lo, hi = r["s"], r["e"]
        p = lo
        skip = 0
        first = True
        while p < hi:
            pos1 = p + 1
            ed = r["edits"].get(p) if skip == 0 else None
            bq = r["bq"].get(p, 30)
            if skip > 0:
                tok = "#" if r["rev"] else "*"
                skip -= 1
            else:
                b = ref[p]
                if ed is not None and ed[0] == "X":
                    b = ed[1]
                tok = b.lower() if r["rev"] else b
                if ed is not None and ed[0] == "I":
                    tok += "+%d%s" % (len(ed[1]), ed[1].lower() if r["rev"] else ed[1])
                elif ed is not None and ed[0] == "D":
                    n = min(ed[1], hi - p - 1)
                    if n > 0:
                        tok += "-%d%s" % (n, ("n" if r["rev"] else "N") * n)
                        skip = n
            if first:
                tok = "^" + chr(min(r["mq"], 93) + 33) + tok
                first = False
            if p == hi - 1:
                tok += "$"
            if pos1 in want and bq >= min_bq:
                rows.setdefault(pos1, []).append((tok, bq, r["mq"], r["name"], str(r["hap"]) if r["tagged"] else "*"))
            p += 1
    out = []
    for pos1 in sorted(rows):
        t = rows[pos1]
        out.append("%s\t%d\tN\t%d\t%s\t%s\t%s\t%s\t%s\n" % (ctg, pos1, len(t), "".join(x[0] for x in t), "".join(chr(x[1] + 33) for x in t),
                                                              "".join(chr(min(x[2], 93) + 33) for x in t), ",".join(x[3] for x in t),
                                                              ",".join(x[4] for x in t)))
    return "".join(out)
