"""cto_realign_windows (csrc/realign_batch.hip), host form: every window of a run in one call, dealt to host threads - the same
stages as cto_realign_reads (src/realign/realigner.cpp:782-857 per window), so the same bytes.  The device form is held to the same
fixtures in tests/test_gpu_realign.py."""
import gzip
import json
import os

import numpy as np
import pytest

import realignutil as ru

HERE = os.path.dirname(os.path.abspath(__file__))


def test_batch_host_equals_the_one_window_call_and_the_golden_windows():
    with gzip.open(os.path.join(HERE, "golden", "realign.json.gz"), "rb") as f:
        g = json.loads(f.read())
    rng = np.random.default_rng(g["seed"])
    ws = [ru.gen_window(rng) for _ in g["windows"][:200]]
    for threads in (1, 4):
        got = ru.amd_realign_batch(ws, "host", threads=threads)
        for w, (pos, cig), want in zip(ws, got, g["windows"]):
            assert [[p - w["ref_start"], c] for p, c in zip(pos, cig)] == want
    assert got[:20] == [ru.amd_realign(w) for w in ws[:20]]


def test_batch_reports_the_failing_window_and_finishes_the_others():
    from clairs_to_amd._lib import lib
    rng = np.random.default_rng(3)
    ws = [ru.gen_window(rng, n_reads=5) for _ in range(4)]
    ws[2] = dict(ws[2], haplotypes=["ACGT"])                     # shorter than the 32-mer seed: CTO_EINVAL for that window
    with pytest.raises(RuntimeError) as ei:
        ru.amd_realign_batch(ws, "host")
    assert "window 2" in str(ei.value) and "shorter" in str(ei.value)
    assert ru.amd_realign_batch([], "host") == []
    st = {}
    ru.amd_realign_batch(ws[:2], "host", stats=st)
    assert st["windows"] == 2 and st["host_windows"] == 2 and st["reads"] == 10 and st["sw_pairs"] == 0


def test_window_batcher_serves_many_callers_with_what_one_call_returns():
    """WindowBatcher (clairs_to_amd/realign_reads.py): worker threads park windows, one cto_realign_windows call per batch - every
    caller gets what realign_window gives, a failing window raises in its own caller only."""
    from concurrent.futures import ThreadPoolExecutor
    from clairs_to_amd.realign_reads import WindowBatcher, realign_window
    rng = np.random.default_rng(9)
    ws = [ru.gen_window(rng, n_reads=6) for _ in range(80)]
    want = [realign_window(*ru.window_args(w)) for w in ws]
    bad = dict(ws[5], haplotypes=["ACGT"])
    with WindowBatcher("host", threads=2, max_batch=16) as b:
        def one(w):
            with b.worker():
                try:
                    return b(*ru.window_args(w))
                except RuntimeError as e:
                    return str(e)
        with ThreadPoolExecutor(24) as ex:
            got = list(ex.map(one, ws + [bad]))
        assert b.windows == 81 and 1 < b.batches < 81
    assert got[:80] == [(p, c) for p, c in want]
    assert "shorter" in got[80]
    with pytest.raises(RuntimeError):
        b(*ru.window_args(ws[0]))


def test_a_job_takes_pointer_arrays_or_joined_buffers():
    """cto_realign_job: seqs / cigars as [n_reads] pointers (what the reference's binding builds, realign_reads.py:582-591) or, when
    those are NULL, as seqs_joined / cigars_joined - the same bytes out"""
    import ctypes as C
    from clairs_to_amd._lib import lib, check, RealignJob
    rng = np.random.default_rng(8)
    w = ru.gen_window(rng, n_reads=25)
    want = ru.amd_realign_batch([w], "host", threads=1)[0]
    m = len(w["seqs"])
    jobs = (RealignJob * 1)()
    j = jobs[0]
    a_seq = (C.c_char_p * m)(*[s.encode() for s in w["seqs"]])
    a_cig = (C.c_char_p * m)(*[c.encode() for c in w["cigars"]])
    a_pos = (C.c_int32 * m)(*w["positions"])
    out_pos, off, buf = (C.c_int32 * m)(), (C.c_int64 * (m + 1))(), C.create_string_buffer(1 << 16)
    j.n_reads, j.seqs, j.cigars, j.positions = m, C.cast(a_seq, C.c_void_p), C.cast(a_cig, C.c_void_p), C.cast(a_pos, C.c_void_p)
    j.reference, j.haplotypes = w["reference"].encode(), " ".join(w["haplotypes"]).encode()
    j.ref_start, j.ref_prefix, j.ref_suffix = w["ref_start"], w["ref_prefix"], w["ref_suffix"]
    j.out_positions, j.cigar_buf, j.cigar_cap, j.cigar_off = C.cast(out_pos, C.c_void_p), C.cast(buf, C.c_void_p), 1 << 16, C.cast(off, C.c_void_p)
    check(lib.cto_realign_windows(1, jobs, 0, 1, None, None))
    got = (list(out_pos), [buf.raw[off[i]:off[i + 1] - 1].decode() for i in range(m)])
    assert got == want
    j.seqs, j.cigars = None, None                      # neither form: refused, not read
    assert lib.cto_realign_windows(1, jobs, 0, 1, None, None) != 0


def test_sw_ends_batch_host_equals_the_scalar_model():
    """cto_sw_ends_batch, host form: the two striped passes per alignment as ssw_align composes them (ssw.c:781-830), against the same
    composition of the scalar model's passes - incl. empty operands"""
    from clairs_to_amd.realign_reads import sw_ends_batch
    rng = np.random.default_rng(17)
    pairs = ru.adversarial_pairs(rng, 900, max_len=400)
    pairs += [(np.zeros(0, dtype=np.int8), pairs[0][1]), (pairs[0][0], np.zeros(0, dtype=np.int8))]
    got = sw_ends_batch(pairs, "host", threads=4)
    word = 0
    for (ref, q), o in zip(pairs, got):
        assert o.tolist() == ru.model_ends(ref, q), (len(ref), len(q))
        word += int(o[5] == 8)
    assert word > 100
    assert sw_ends_batch([], "host").shape == (0, 6)


def test_planned_tracebacks_give_the_same_windows(monkeypatch):
    """The device stage's plumbing on the host (CTO_REALIGN_PLAN_HOST=1): the tracebacks finish() will ask for are planned from the end
    points (every haplotype against the reference, per unplaced read the pair it would pick), run, and installed as runs - the windows
    come out as without the plan (golden POS + CIGAR), and nearly every traceback finish() needs was planned."""
    with gzip.open(os.path.join(HERE, "golden", "realign.json.gz"), "rb") as f:
        g = json.loads(f.read())
    rng = np.random.default_rng(g["seed"])
    ws = [ru.gen_window(rng) for _ in g["windows"][:250]]
    monkeypatch.setenv("CTO_REALIGN_PLAN_HOST", "1")
    got = ru.amd_realign_batch(ws, "host", threads=3)
    for w, (pos, cig), want in zip(ws, got, g["windows"]):
        assert [[p - w["ref_start"], c] for p, c in zip(pos, cig)] == want


def test_ssw_align_batch_host_equals_the_one_pair_call():
    """cto_ssw_align_batch, host form, against cto_ssw_align pair by pair: score, reference start, CIGAR - incl. pairs without an
    alignment (nothing matches) and empty operands"""
    import ctypes as C
    from clairs_to_amd.realign_reads import ssw_align_batch
    from clairs_to_amd._lib import lib, check
    rng = np.random.default_rng(41)
    pairs = ru.adversarial_pairs(rng, 500, max_len=300)
    pairs += [(np.zeros(5, dtype=np.int8), np.ones(7, dtype=np.int8)), (np.zeros(0, dtype=np.int8), np.ones(7, dtype=np.int8))]
    sc, rb, cg = ssw_align_batch(pairs, "host", threads=3)
    letters = "ACGTN"
    buf = C.create_string_buffer(1 << 16)
    for (r, q), s, b, c in zip(pairs, sc, rb, cg):
        score, beg = C.c_int32(), C.c_int32()
        check(lib.cto_ssw_align("".join(letters[x] for x in r).encode(), "".join(letters[x] for x in q).encode(), C.byref(score), C.byref(beg), buf, 1 << 16))
        assert (score.value, beg.value, buf.value.decode()) == (int(s), int(b), c)
    assert cg[-1] == "" and cg[-2] == "" and int(sc[-1]) == 0
    assert ssw_align_batch([], "host")[2] == []
