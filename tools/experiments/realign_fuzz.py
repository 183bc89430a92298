#!/usr/bin/env python3
"""Development fuzz of the device realigner beyond what the suite runs each time: fresh windows against oracle/_ref, adversarial pairs
(tests/realignutil.py) through cto_sw_ends_batch / cto_ssw_align_batch, device against the host form.  python tools/experiments/realign_fuzz.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import realignutil as ru
from clairs_to_amd.realign_reads import sw_ends_batch, ssw_align_batch
torch.cuda.set_device(0)
seed = int.from_bytes(os.urandom(4), "little")
rng = np.random.default_rng(seed)
print("seed", seed)
t = time.time()
bad = 0
for rnd in range(6):
    ws = [ru.gen_window(rng) for _ in range(800)] + [ru.gen_window(rng, n_reads=int(rng.integers(1, 600))) for _ in range(8)]
    got = ru.amd_realign_batch(ws, "device", threads=16)
    for i, (w, g) in enumerate(zip(ws, got)):
        if g != ru.ref_realign(w):
            bad += 1; print("WINDOW DIFF", rnd, i)
print("windows vs oracle/_ref: %d differences of %d (%.0f s)" % (bad, 6 * 808, time.time() - t))
t = time.time()
tot = 0
for rnd in range(8):
    pairs = ru.adversarial_pairs(rng, 24000, max_len=int(rng.choice([64, 300, 700, 1200]))) + ru.adversarial_pairs(rng, 300, max_len=2048)
    d, h = sw_ends_batch(pairs, "device"), sw_ends_batch(pairs, "host", threads=16)
    nb = int((d != h).any(axis=1).sum())
    ds, db, dc = ssw_align_batch(pairs, "device", threads=16)
    hs, hb, hc = ssw_align_batch(pairs, "host", threads=16)
    na = sum(1 for i in range(len(pairs)) if (int(ds[i]), int(db[i]), dc[i]) != (int(hs[i]), int(hb[i]), hc[i]))
    tot += len(pairs); bad += nb + na
    if nb or na: print("PAIR DIFF round", rnd, nb, na)
print("pairs device vs host: %d differences over %d pairs, end points and alignments (%.0f s)" % (bad, tot, time.time() - t))
