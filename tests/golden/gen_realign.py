#!/usr/bin/env python3
"""Golden vectors for the Illumina realigner (SURVEY.md 8f #4b) from the REFERENCE's own native code.

`make -C oracle ref` compiles /root/reference/src/realign/{realigner.cpp,ssw_cpp.cpp,ssw.c} into oracle/_ref/librealigner_ref.so
(plain g++, outputs only); this script drives its C ABI `realign_reads` (src/realign/realigner.cpp:860-865) on synthetic windows
(tests/realignutil.gen_window: fast-pass hits, <= 2 mismatches, Smith-Waterman fallbacks with substitutions / indels / clipped
ends, repeats and two-letter sequences for tie-breaking, N bases, 1..18 haplotypes, reads of 20..250 bases) and stores what it
returned.  Inputs are regenerated from the seed by the test and checked by SHA-256; outputs (positions, CIGARs) are stored.

  realign.json.gz   {"seed", "n_windows", "inputs_sha256", "windows": [[[pos - ref_start, cigar], ...], ...]}

Usage: make -C oracle ref && python tests/golden/gen_realign.py      (from the repo root; build container only)
"""
import gzip
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import realignutil as ru  # noqa: E402

SEED, N_WINDOWS = 20260928, 640


def window_digest(h, w):
    h.update(json.dumps([w["seqs"], w["positions"], w["cigars"], w["reference"], w["haplotypes"], w["ref_start"], w["ref_prefix"],
                         w["ref_suffix"]], separators=(",", ":")).encode())


def main():
    assert ru.ref_lib() is not None, "run `make -C oracle ref` first"
    rng = np.random.default_rng(SEED)
    h = hashlib.sha256()
    out = []
    moved = reads = 0
    for _ in range(N_WINDOWS):
        w = ru.gen_window(rng)
        window_digest(h, w)
        pos, cig = ru.ref_realign(w)
        out.append([[p - w["ref_start"], c] for p, c in zip(pos, cig)])
        reads += len(cig)
        moved += sum(1 for c, c0 in zip(cig, w["cigars"]) if c != c0)
    raw = json.dumps({"seed": SEED, "n_windows": N_WINDOWS, "inputs_sha256": h.hexdigest(), "windows": out}, separators=(",", ":")).encode()
    with open(os.path.join(HERE, "realign.json.gz"), "wb") as f:
        with gzip.GzipFile(fileobj=f, mode="wb", mtime=0) as g:
            g.write(raw)
    print("wrote realign.json.gz: %d windows, %d reads, %d realigned, %d bytes raw" % (N_WINDOWS, reads, moved, len(raw)))


if __name__ == "__main__":
    main()
