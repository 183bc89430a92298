"""Synthetic realignment windows in the shape `reads_realignment` builds them (src/realign_reads.py:544-591): reference =
prefix + window + suffix, candidate haplotypes (the consensus list: substitutions and indels in the window part), reads drawn from
the haplotypes - exact, <= 2 mismatches (both placed by the k-mer fast pass), more mismatches / indels / garbage ends (the
Smith-Waterman fallback), N bases, tandem repeats and two-letter sequences (equal-score cells).  Used by the tests (whose golden
fixture tests/golden/realign.json.gz pins the random sequence: do not reorder the draws), tools/realign_bench.py and the
`configs3_realign` leg of bench.py."""
import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def _rand_seq(rng, n, style):
    if style == 0:
        return bytes(rng.choice(BASES, n))
    if style == 1:                                   # short tandem repeats and homopolymers: many equal-score alignments
        out = bytearray()
        while len(out) < n:
            unit = bytes(rng.choice(BASES, int(rng.integers(1, 5))))
            out += unit * int(rng.integers(2, 12))
            out += bytes(rng.choice(BASES, int(rng.integers(0, 15))))
        return bytes(out[:n])
    out = bytearray(rng.choice(BASES[:2], n))        # two-letter alphabet
    return bytes(out)


def _mutate(rng, s, n_snv, n_indel, lo, hi, max_indel):
    s = bytearray(s)
    for _ in range(n_indel):
        p = int(rng.integers(lo, max(lo + 1, hi)))
        ln = int(rng.integers(1, max_indel + 1))
        if rng.random() < 0.5:
            s[p:p] = bytes(rng.choice(BASES, ln))
        else:
            del s[p:p + ln]
        hi = min(hi, len(s) - 1)
    for _ in range(n_snv):
        if not s:
            break
        p = min(int(rng.integers(lo, max(lo + 1, min(hi, len(s) - 1)))), len(s) - 1)
        s[p] = int(rng.choice(BASES))
    return bytes(s)


def gen_window(rng, n_reads=None):
    """One synthetic realignment window in the shape `reads_realignment` builds (src/realign_reads.py:544-591)."""
    style = int(rng.choice([0, 0, 0, 1, 1, 2]))
    prefix = int(rng.integers(0, 190))
    suffix = int(rng.integers(0, 190))
    centre = int(rng.integers(160, 420))
    ref = _rand_seq(rng, prefix + centre + suffix, style)
    if rng.random() < 0.1:                           # N runs in the reference
        b = bytearray(ref); p = int(rng.integers(0, len(b) - 5)); b[p:p + int(rng.integers(1, 5))] = b"NNNN"[:int(rng.integers(1, 5))]; ref = bytes(b[:len(ref)])
    n_hap = int(rng.choice([1, 1, 2, 2, 3, 4, 6, 18]))
    haps = []
    for h in range(n_hap):
        if h == 0 and rng.random() < 0.7:
            haps.append(ref)
            continue
        cons = _mutate(rng, ref[prefix:prefix + centre], int(rng.integers(0, 4)), int(rng.integers(0, 3)), 20, centre - 20,
                       int(rng.choice([1, 2, 5, 12, 40])))
        haps.append(ref[:prefix] + cons + ref[prefix + centre:])
    haps = [h for h in haps if len(h) >= 40]
    if not haps:
        haps = [ref]
    if rng.random() < 0.5:
        haps = sorted(set(haps))                     # the consensus list arrives sorted and distinct
    n = int(n_reads if n_reads is not None else rng.choice([1, 3, 8, 20, 40, 60]))
    seqs, positions, cigars = [], [], []
    ref_start = int(rng.integers(0, 100000))
    for _ in range(n):
        src = haps[int(rng.integers(0, len(haps)))] if rng.random() < 0.9 else ref
        rl = int(rng.choice([20, 32, 33, 50, 100, 101, 125, 150, 151, 250]))
        rl = min(rl, len(src))
        st = int(rng.integers(0, len(src) - rl + 1))
        s = src[st:st + rl]
        kind = rng.random()
        if kind < 0.45:
            pass                                     # exact: fast pass
        elif kind < 0.65:
            s = _mutate(rng, s, int(rng.integers(1, 3)), 0, 0, len(s) - 1, 1)          # <= 2 mismatches: fast pass
        elif kind < 0.8:
            s = _mutate(rng, s, int(rng.integers(3, 8)), 0, 0, len(s) - 1, 1)          # SSW fallback, substitutions
        elif kind < 0.93:
            s = _mutate(rng, s, int(rng.integers(0, 4)), int(rng.integers(1, 3)), 3, len(s) - 3, int(rng.choice([1, 2, 6, 15])))
        else:                                        # soft-clip-like garbage at one or both ends
            a = bytes(rng.choice(BASES, int(rng.integers(3, 25))))
            s = (a + s[len(a):]) if rng.random() < 0.5 else (s[:-len(a)] + a)
            if rng.random() < 0.3:
                b = bytes(rng.choice(BASES, int(rng.integers(3, 15))))
                s = b + s[len(b):]
        if rng.random() < 0.05 and len(s) > 4:
            b = bytearray(s); b[int(rng.integers(0, len(b)))] = ord("N"); s = bytes(b)
        if not s:
            s = b"A"
        seqs.append(s.decode())
        positions.append(ref_start + st)
        cigars.append("%dM" % len(s) if rng.random() < 0.8 else "%dS%dM" % (3, len(s) - 3) if len(s) > 3 else "%dM" % len(s))
    return dict(seqs=seqs, positions=positions, cigars=cigars, reference=ref.decode(), haplotypes=[h.decode() for h in haps],
                ref_start=ref_start, ref_prefix=prefix, ref_suffix=suffix)

