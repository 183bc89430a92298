"""Window generator and ctypes drivers for the Illumina realigner tests (tests/test_realign.py, tests/golden/gen_golden.py).

`ref_realign` calls the REFERENCE's native entry point compiled here into oracle/_ref/librealigner_ref.so (`make -C oracle ref`;
src/realign/realigner.cpp:860-865 `realign_reads`), `amd_realign` the build's `cto_realign_reads`.  Test infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "librealigner_ref.so")
MAX_READS = 1000          # struct_str_arr of the reference (src/realign/realigner.h:42-46)


class _RefOut(C.Structure):
    _fields_ = [("position", C.c_int * MAX_READS), ("cigar_string", C.c_char_p * MAX_READS)]


_ref_lib = None


def ref_lib():
    global _ref_lib
    if _ref_lib is None:
        if not os.path.exists(REF_SO):
            return None
        _ref_lib = C.CDLL(REF_SO)
        _ref_lib.realign_reads.restype = C.POINTER(_RefOut)
        _ref_lib.free_memory.restype = None
    return _ref_lib


def ref_realign(w):
    """(positions, cigars) from the reference's realigner for window dict w."""
    lib = ref_lib()
    n = len(w["seqs"])
    assert 0 < n <= MAX_READS
    seqs = (C.c_char_p * n)(*[s.encode() for s in w["seqs"]])
    pos = (C.c_int * n)(*w["positions"])
    cig = (C.c_char_p * n)(*[s.encode() for s in w["cigars"]])
    lib.realign_reads.argtypes = [C.c_char_p * n, C.c_int * n, C.c_char_p * n, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.free_memory.argtypes = [C.POINTER(_RefOut), C.c_int]
    p = lib.realign_reads(seqs, pos, cig, w["reference"].encode(), " ".join(w["haplotypes"]).encode(), w["ref_start"],
                          w["ref_prefix"], w["ref_suffix"], n)
    out = (list(p.contents.position[:n]), [c.decode() for c in p.contents.cigar_string[:n]])
    lib.free_memory(p, n)
    return out


def amd_realign(w):
    from clairs_to_amd._lib import lib, check
    n = len(w["seqs"])
    seqs = (C.c_char_p * n)(*[s.encode() for s in w["seqs"]])
    pos = (C.c_int32 * n)(*w["positions"])
    cig = (C.c_char_p * n)(*[s.encode() for s in w["cigars"]])
    out_pos = (C.c_int32 * n)()
    cap = 64 * n + sum(len(s) for s in w["seqs"]) * 6 + 64
    buf = C.create_string_buffer(cap)
    off = (C.c_int64 * (n + 1))()
    check(lib.cto_realign_reads(n, seqs, pos, cig, w["reference"].encode(), " ".join(w["haplotypes"]).encode(), w["ref_start"],
                                w["ref_prefix"], w["ref_suffix"], out_pos, buf, cap, off))
    raw = buf.raw
    return list(out_pos), [raw[off[i]:off[i + 1] - 1].decode() for i in range(n)]


def amd_ssw(ref, query):
    from clairs_to_amd._lib import lib, check
    score, begin = C.c_int32(), C.c_int32()
    buf = C.create_string_buffer(8 * (len(query) + len(ref)) + 64)
    check(lib.cto_ssw_align(ref.encode(), query.encode(), C.byref(score), C.byref(begin), buf, len(buf)))
    return score.value, begin.value, buf.value.decode()


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


def window_args(w):
    """the argument tuple of clairs_to_amd.realign_reads.realign_window / realign_windows for window dict w"""
    return (w["seqs"], w["positions"], w["cigars"], w["reference"], w["haplotypes"], w["ref_start"], w["ref_prefix"], w["ref_suffix"])


def amd_realign_batch(ws, where, threads=0, stats=None):
    from clairs_to_amd.realign_reads import realign_windows
    return [(p, c) for p, c in realign_windows([window_args(w) for w in ws], where=where, threads=threads, stats=stats)]
