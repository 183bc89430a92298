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
