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


from clairs_to_amd.synth_realign import BASES, _rand_seq, _mutate, gen_window  # noqa: E402,F401  (the generator lives with the other synthetic inputs)


def window_args(w):
    """the argument tuple of clairs_to_amd.realign_reads.realign_window / realign_windows for window dict w"""
    return (w["seqs"], w["positions"], w["cigars"], w["reference"], w["haplotypes"], w["ref_start"], w["ref_prefix"], w["ref_suffix"])


def amd_realign_batch(ws, where, threads=0, stats=None):
    from clairs_to_amd.realign_reads import realign_windows
    return [(p, c) for p, c in realign_windows([window_args(w) for w in ws], where=where, threads=threads, stats=stats)]


def model_ends(ref, q):
    """ssw_align's two passes (ssw.c:781-830) composed from the scalar model of one pass (oracle/ssw_model.cpp): the row
    cto_sw_ends_batch writes for this pair"""
    import oracle
    if len(ref) == 0 or len(q) == 0:
        return [0, 0, 0, 0, 0, 16]
    fw, lanes = oracle.ssw_pass(ref, q, 16), 16
    if fw[3]:
        fw, lanes = oracle.ssw_pass(ref, q, 8), 8
    if fw[0] <= 0:
        return [0, 0, 0, 0, 0, 16]
    bw = oracle.ssw_pass(ref[:fw[1] + 1], q[:fw[2] + 1][::-1], lanes, True, fw[0])
    return [fw[0], fw[1], fw[2], bw[1], bw[2], lanes]


def adversarial_pairs(rng, n, max_len=700):
    """(reference, query) code pairs that stress the lazy-F step: long matches around one or several long gaps (F chains that cross
    every stripe), tandem repeats and two-letter alphabets (ties), queries shorter than the lane count, N bases, unrelated pairs"""
    pairs = []
    for it in range(n):
        style = it % 6
        R = int(rng.integers(1, max_len))
        ref = rng.integers(0, 4, R).astype(np.int8)
        if style == 0:                                   # a copy with 0..4 gaps of 1..80 bases and a few substitutions
            q = ref.copy()
            for _ in range(int(rng.integers(0, 5))):
                k, g = int(rng.integers(0, max(1, len(q)))), int(rng.integers(1, 80))
                q = np.concatenate([q[:k], q[k + g:]]) if rng.random() < 0.5 else np.concatenate([q[:k], rng.integers(0, 4, g).astype(np.int8), q[k:]])
            for _ in range(int(rng.integers(0, 4))):
                if len(q):
                    k = int(rng.integers(0, len(q)))
                    q[k] = (q[k] + 1) % 4
            if len(q) == 0:
                q = ref[:1].copy()
            if rng.random() < 0.5:
                ref, q = q, ref
        elif style == 1:                                 # tandem repeats
            unit = rng.integers(0, 4, int(rng.integers(1, 5))).astype(np.int8)
            Q = int(rng.integers(1, max_len))
            ref, q = np.tile(unit, R // len(unit) + 1)[:R].copy(), np.tile(unit, Q // len(unit) + 1)[:Q].copy()
            if rng.random() < 0.5:
                q[int(rng.integers(0, Q))] = 4
        elif style == 2:                                 # tiny alphabet
            k = int(rng.integers(1, 3))
            ref = rng.integers(0, k + 1, R).astype(np.int8)
            q = rng.integers(0, k + 1, int(rng.integers(1, max_len))).astype(np.int8)
        elif style == 3:                                 # a query shorter than the lanes
            a = int(rng.integers(0, R))
            q = ref[a:a + int(rng.integers(1, 17))].copy()
        elif style == 4:                                 # a read-length window of the reference
            a = int(rng.integers(0, R))
            q = ref[a:a + int(rng.integers(20, 300))].copy()
            if len(q) > 4 and rng.random() < 0.7:
                k = int(rng.integers(1, len(q) - 1))
                q = np.concatenate([q[:k], q[k + int(rng.integers(1, 30)):]])
        else:                                            # unrelated
            q = rng.integers(0, 4, int(rng.integers(1, max_len))).astype(np.int8)
        pairs.append((np.ascontiguousarray(ref, dtype=np.int8), np.ascontiguousarray(q, dtype=np.int8)))
    return pairs
