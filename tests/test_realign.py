"""Illumina realigner (SURVEY.md 8f #4b): cto_realign_reads against the reference's own native code.

Pins: (1) tests/golden/realign.json.gz - what the reference's `realign_reads` (src/realign/realigner.cpp:782-869, compiled here by
`make -C oracle ref`) returned on 640 synthetic windows, regenerated inputs checked by SHA-256; (2) when oracle/_ref/ is present,
fresh windows against the compiled reference itself; (3) the reference-ABI shim libraries (clairs_to_amd/realign/*.so), called the
way src/realign_reads.py:532-615 calls the reference's modules; (4) the striped Smith-Waterman pass alone (SSE2 in the product)
against its scalar model in oracle/ssw_model.cpp.  The de Bruijn consensus has no compiled reference here (Boost):
hand-derived vectors and properties only - PARITY UNPINNED."""
import ctypes as C
import gzip
import hashlib
import json
import os

import numpy as np
import pytest

import realignutil as ru

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def load(name):
    with gzip.open(os.path.join(HERE, "golden", name), "rb") as f:
        return json.loads(f.read())


def test_realign_reads_equals_the_reference_on_the_golden_windows():
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from gen_realign import window_digest
    g = load("realign.json.gz")
    rng = np.random.default_rng(g["seed"])
    h = hashlib.sha256()
    realigned = fallback = 0
    for want in g["windows"]:
        w = ru.gen_window(rng)
        window_digest(h, w)
        pos, cig = ru.amd_realign(w)
        got = [[p - w["ref_start"], c] for p, c in zip(pos, cig)]
        assert got == want
        realigned += sum(1 for c, c0 in zip(cig, w["cigars"]) if c != c0)
        fallback += sum(1 for c in cig if "I" in c or "D" in c or "S" in c)
    assert h.hexdigest() == g["inputs_sha256"], "the window generator changed: regenerate tests/golden/realign.json.gz"
    assert len(g["windows"]) >= 500 and realigned > 10000 and fallback > 2000


@pytest.mark.skipif(ru.ref_lib() is None, reason="oracle/_ref/librealigner_ref.so not built (`make -C oracle ref`, needs /root/reference)")
def test_realign_reads_equals_the_compiled_reference_on_fresh_windows():
    seed = int.from_bytes(os.urandom(4), "little")
    rng = np.random.default_rng(seed)
    for i in range(120):
        w = ru.gen_window(rng)
        assert ru.amd_realign(w) == ru.ref_realign(w), "window %d of np.random.default_rng(%d)" % (i, seed)


@pytest.mark.skipif(ru.ref_lib() is None, reason="oracle/_ref/librealigner_ref.so not built")
def test_realign_reads_full_window_of_a_thousand_reads():
    """max_region_reads_num = 1000 (src/realign_reads.py:51): the largest call the reference makes"""
    rng = np.random.default_rng(7)
    w = ru.gen_window(rng, n_reads=1000)
    assert ru.amd_realign(w) == ru.ref_realign(w)


def _random_pair(rng, it):
    R, Q = int(rng.integers(2, 400)), int(rng.integers(2, 260))
    style = it % 4
    if style == 0:                                   # unrelated
        ref, read = rng.integers(0, 4, R), rng.integers(0, 4, Q)
    elif style == 1:                                 # a piece of ref with substitutions, insertions and deletions
        ref = rng.integers(0, 4, R)
        a = int(rng.integers(0, R))
        read = ref[a:min(R, a + Q)].copy()
        for _ in range(int(rng.integers(0, 6))):
            if len(read) < 4:
                break
            k, op = int(rng.integers(0, len(read))), int(rng.integers(0, 3))
            if op == 0:
                read[k] = (read[k] + 1) % 4
            elif op == 1:
                read = np.concatenate([read[:k], rng.integers(0, 4, int(rng.integers(1, 12))), read[k:]])
            else:
                read = np.concatenate([read[:k], read[min(len(read), k + int(rng.integers(1, 12))):]])
        if len(read) == 0:
            read = ref[:1].copy()
    elif style == 2:                                 # tandem repeats and N: many equal-score cells, long F chains
        unit = rng.integers(0, 4, int(rng.integers(1, 4)))
        ref, read = np.tile(unit, R // len(unit) + 1)[:R].copy(), np.tile(unit, Q // len(unit) + 1)[:Q].copy()
        if rng.random() < 0.5:
            read[int(rng.integers(0, Q))] = 4
        if rng.random() < 0.5:
            ref[int(rng.integers(0, R))] = 4
    else:                                            # a long exact match: the 8-bit pass overflows
        ref = rng.integers(0, 4, R)
        a = int(rng.integers(0, max(1, R - 70)))
        read = ref[a:a + min(Q, R - a)].copy()
    return np.ascontiguousarray(ref, dtype=np.int8), np.ascontiguousarray(read, dtype=np.int8)


def test_striped_pass_equals_the_scalar_model():
    """The SSE2 pass of the product (cto_ssw_pass) against oracle/ssw_model.cpp - the lane-by-lane scalar statement of the same
    recurrence that was pinned to the compiled reference - on both widths, both directions, with and without the early stop."""
    import oracle
    from clairs_to_amd._lib import lib, check
    rng = np.random.default_rng(11)
    overflowed = stopped = 0
    for it in range(1500):
        ref, read = _random_pair(rng, it)
        for lanes in (16, 8):
            for reverse in (False, True):
                term = 255 if lanes == 16 else 65535
                if reverse and it % 3 == 0:
                    term = oracle.ssw_pass(ref, read, lanes, False)[0]
                    stopped += 1
                want = oracle.ssw_pass(ref, read, lanes, reverse, term)
                out = np.zeros(4, dtype=np.int32)
                check(lib.cto_ssw_pass(ref.ctypes.data, len(ref), int(reverse), read.ctypes.data, len(read), lanes, term, out.ctypes.data))
                assert (int(out[0]), int(out[1]), int(out[2]), bool(out[3])) == want, (it, lanes, reverse, term)
                overflowed += int(want[3])
    assert overflowed > 300 and stopped > 500


def test_lazy_f_closed_form_equals_the_loops():
    """What the device kernel rests on (realign_batch.hip): the two lazy-F loops of the striped pass (ssw.c:207-241, :446-459) compute,
    exits and all, the max-plus scan of the stripes' outgoing F over the lanes followed by one sweep.  Pinned here on the scalar model
    (itself pinned to the compiled reference): every H column after its lazy-F step (hashed), and the pass's result, on both widths and
    directions - random pairs of every style of this file, queries shorter than the lane count, two-letter alphabets (ties everywhere),
    and hundreds of bases of exact match with one long gap (F chains that cross every stripe)."""
    import oracle
    rng = np.random.default_rng(5)
    cases = [_random_pair(rng, it) for it in range(1200)]
    for it in range(600):                           # short queries, tiny alphabets
        R, Q = int(rng.integers(1, 90)), int(rng.integers(1, 40))
        k = int(rng.integers(1, 3))
        cases.append((rng.integers(0, k + 1, R).astype(np.int8), rng.integers(0, k + 1, Q).astype(np.int8)))
    for it in range(150):                           # long matches around one deletion / insertion of 1..60 bases
        R = int(rng.integers(150, 700))
        ref = rng.integers(0, 4, R).astype(np.int8)
        cut, g = int(rng.integers(20, R - 20)), int(rng.integers(1, 60))
        read = np.concatenate([ref[:cut], ref[min(R, cut + g):]]) if it % 2 else np.concatenate([ref[:cut], rng.integers(0, 4, g).astype(np.int8), ref[cut:]])
        cases.append((ref, read.astype(np.int8)) if it % 4 < 2 else (read.astype(np.int8), ref))
    crossings = 0
    for it, (ref, read) in enumerate(cases):
        for lanes in (16, 8):
            for reverse in (False, True):
                term = None
                if reverse and it % 3 == 0:
                    term = oracle.ssw_pass(ref, read, lanes, False)[0]
                a = oracle.ssw_pass_ex(ref, read, lanes, reverse, term, closed_form=False)
                b = oracle.ssw_pass_ex(ref, read, lanes, reverse, term, closed_form=True)
                assert a == b, (it, lanes, reverse, term, len(ref), len(read))
                assert a[0] == oracle.ssw_pass(ref, read, lanes, reverse, term)
                crossings += int(a[0][0] > 100)
    assert crossings > 1500


def test_realign_reads_does_not_depend_on_the_thread_count():
    from clairs_to_amd._lib import lib, check
    rng = np.random.default_rng(21)
    wins = [ru.gen_window(rng, n_reads=n) for n in (40, 120, 7, 300)]
    try:
        check(lib.cto_set_realign_threads(1))
        want = [ru.amd_realign(w) for w in wins]
        for nt in (2, 5, 16):
            check(lib.cto_set_realign_threads(nt))
            assert [ru.amd_realign(w) for w in wins] == want
    finally:
        check(lib.cto_set_realign_threads(1))
    assert lib.cto_set_realign_threads(0) != 0


def test_ssw_known_answers():
    """hand-derived: score 4 / -6, gap open 8 (a gap of length 1 costs 8), extension 2; soft clips, '=' / 'X' runs"""
    ref = "TTTTACGTACGGATCCAGTTTT"
    assert ru.amd_ssw(ref, "ACGTACGGATCCAG") == (56, 4, "14=")
    assert ru.amd_ssw(ref, "ACGTACGTATCCAG") == (46, 4, "7=1X6=")                # one mismatch: 13 * 4 - 6
    q = "GGGACGTACGGATCCAGCCC"
    s, b, c = ru.amd_ssw(ref, q)
    assert (s, b, c) == (56, 4, "3S14=3S")
    ref2 = "ACGTTGCATGCCGATTACAGGCATCGATCGGACT"
    q2 = ref2[:17] + ref2[19:]                                                    # 2-base deletion: 32 * 4 - 8 - 2
    s, b, c = ru.amd_ssw(ref2, q2)
    assert s == 118 and b == 0 and c.count("D") == 1 and sum(int(x) for x in c.replace("D", "=").split("=")[:-1]) == 34
    assert ru.amd_ssw("ACGT", "NNNN") == (0, 0, "")                              # nothing aligns


def test_reference_abi_shims_export_the_reference_symbols():
    """clairs_to_amd/realign/{realigner,debruijn_graph}.so carry the names src/realign_reads.py binds (:532-536, :582-615)"""
    class Out(C.Structure):
        _fields_ = [("position", C.c_int * 1000), ("cigar_string", C.c_char_p * 1000)]

    class Dbg(C.Structure):
        _fields_ = [("consensus_size", C.c_int), ("consensus", C.c_char_p * 200)]

    import clairs_to_amd  # noqa: F401  (the shims link against the product library)
    real = C.CDLL(os.path.join(ROOT, "clairs_to_amd", "realign", "realigner.so"))
    dbg = C.CDLL(os.path.join(ROOT, "clairs_to_amd", "realign", "debruijn_graph.so"))
    rng = np.random.default_rng(3)
    w = ru.gen_window(rng, n_reads=12)
    n = len(w["seqs"])
    real.realign_reads.restype = C.POINTER(Out)
    real.realign_reads.argtypes = [C.c_char_p * n, C.c_int * n, C.c_char_p * n, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
    p = real.realign_reads((C.c_char_p * n)(*[s.encode() for s in w["seqs"]]), (C.c_int * n)(*w["positions"]),
                           (C.c_char_p * n)(*[s.encode() for s in w["cigars"]]), w["reference"].encode(),
                           " ".join(w["haplotypes"]).encode(), w["ref_start"], w["ref_prefix"], w["ref_suffix"], n)
    got = (list(p.contents.position[:n]), [c.decode() for c in p.contents.cigar_string[:n]])
    real.free_memory.argtypes = [C.POINTER(Out), C.c_int]
    real.free_memory(p, n)
    assert got == ru.amd_realign(w)
    ref = w["reference"][w["ref_prefix"]:len(w["reference"]) - w["ref_suffix"]]
    dbg.get_consensus.restype = C.POINTER(Dbg)
    dbg.get_consensus.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
    reads = [ref[10:110], ref[10:110], ref[40:140]]
    q = dbg.get_consensus(ref.encode(), ",".join(reads).encode(), ",".join(["", "3 4", ""]).encode(), len(reads))
    cons = [c.decode() for c in q.contents.consensus[:q.contents.consensus_size]]
    dbg.free_memory.argtypes = [C.POINTER(Dbg), C.c_int]
    dbg.free_memory(q, q.contents.consensus_size)
    assert cons == ([ref] if len(set(ref[i:i + 10] for i in range(len(ref) - 9))) == len(ref) - 9 or cons else [])


# ---------------------------------------------------------------------------------------------------------------- de Bruijn
def consensus(ref, reads, lowbq=None):
    from clairs_to_amd.realign_reads import dbg_consensus
    return dbg_consensus(ref, reads, lowbq)


def _unique_ref(rng, n):
    while True:
        s = bytes(rng.choice(ru.BASES, n)).decode()
        if len({s[i:i + 10] for i in range(n - 9)}) == n - 9:
            return s


def test_debruijn_hand_derived_vectors():
    """PARITY UNPINNED (no Boost here).  Contract of src/realign/debruijn_graph.cpp:208-232, :387-428 on cases worked by hand."""
    rng = np.random.default_rng(11)
    ref = _unique_ref(rng, 200)
    # reads that agree with the reference: the reference path only
    assert consensus(ref, [ref[20:170], ref[0:150]]) == [ref]
    # a SNV seen once is pruned (non-reference edges need weight >= 2); seen twice it opens a second path
    alt = ref[:100] + ("A" if ref[100] != "A" else "C") + ref[101:]
    assert consensus(ref, [alt[30:180]]) == [ref]
    assert consensus(ref, [alt[30:180], alt[20:170]]) == sorted([ref, alt])
    # an insertion and a deletion 40 bases apart, each supported twice, on separate reads: two independent bubbles = four
    # source-to-sink walks (the graph knows no read phase), sorted bytewise
    ins = ref[:80] + "GATTACA" + ref[80:]
    dele = ref[:120] + ref[126:]
    both = ref[:80] + "GATTACA" + ref[80:120] + ref[126:]
    got = consensus(ref, [ins[20:170], ins[30:180], dele[40:190], dele[50:194]])
    assert got == sorted([ref, ins, dele, both])
    # low-quality positions (BQ < 15) and non-ACGT bases break a read into runs: the variant k-mers vanish
    assert consensus(ref, [alt[30:180], alt[20:170]], lowbq=[[70], [80]]) == [ref]            # position 100 of alt, read-relative
    n_alt = alt[:100] + "N" + alt[101:]
    assert consensus(ref, [n_alt[30:180], n_alt[20:170]]) == [ref]
    # a reference window with a repeated 10-mer needs a larger k; one that repeats at every k <= 101 gives nothing
    rep = ref[:60] + ref[20:60] + ref[60:]
    assert consensus(rep, [rep[10:160]]) == [rep]
    assert consensus("ACGT" * 60, ["ACGT" * 30]) == []
    # reads that close a cycle at the smallest k push k up until the graph is acyclic
    cyc_read = ref[50:90] + ref[60:90] + ref[90:140]                                          # tandem duplication of 30 bases
    got = consensus(ref, [cyc_read, cyc_read])
    assert ref in got and all(len(h) >= len(ref) for h in got)


def test_debruijn_properties():
    """every haplotype starts / ends with the reference's first / last k-mer, the list is sorted and distinct, and a variant
    supported by >= 2 clean reads that span it is among the haplotypes"""
    rng = np.random.default_rng(12)
    for trial in range(40):
        ref = _unique_ref(rng, int(rng.integers(160, 400)))
        p = int(rng.integers(40, len(ref) - 60))
        kind = trial % 3
        alt = ref[:p] + {0: ("A" if ref[p] != "A" else "C") + ref[p + 1:], 1: "TTGACC"[:int(rng.integers(1, 7))] + ref[p:],
                         2: ref[p + int(rng.integers(1, 9)):]}[kind]
        reads = [alt[max(0, p - 90):p + 60], alt[max(0, p - 60):p + 90], ref[max(0, p - 100):p + 50]]
        got = consensus(ref, reads)
        assert got == sorted(set(got))
        assert ref in got and alt in got
        for h in got:
            assert h[:10] == ref[:10] and h[-10:] == ref[-10:]


def test_debruijn_path_limit():
    """more than 256 open + closed paths -> no haplotypes at all (:296-299): nine independent biallelic sites = 512 paths"""
    rng = np.random.default_rng(13)
    ref = _unique_ref(rng, 600)
    sites = list(range(60, 60 + 9 * 50, 50))
    alt = list(ref)
    for s in sites:
        alt[s] = "A" if ref[s] != "A" else "C"
    alt = "".join(alt)
    reads = []
    for s in sites:                                     # each site on its own pair of reads, 30 bases either side
        reads += [ref[:s - 24] and (ref[s - 24:s] + alt[s] + ref[s + 1:s + 25])] * 2
    assert consensus(ref, reads) == []
    assert len(consensus(ref, reads[:16])) == 256       # eight sites: 2^8 paths is still inside the limit


def test_debruijn_against_an_independent_enumerator():
    """tests/consensus_enum.py - the reference's contract restated a second time with other means (recursion instead of the queue, dictionaries
    instead of vertex arrays) - agrees with cto_dbg_consensus on windows with SNVs, indels, low-quality positions, N bases, repeated
    reference k-mers and reads that close cycles.  PARITY STAYS UNPINNED: neither side is the compiled reference (Boost.Graph)."""
    from consensus_enum import enumerate_consensus, consensus_windows
    checked = multi = 0
    for it, (ref, reads, lowbq) in enumerate(consensus_windows(2026, 250)):
        want = enumerate_consensus(ref, reads, lowbq)
        if want is None:                                            # near the 256-path cut-off: order dependent in the reference
            continue
        got = consensus(ref, reads, lowbq)
        assert got == want, (it, ref, reads, lowbq)
        checked += 1
        multi += int(len(want) > 1)
    assert checked > 150 and multi > 50
